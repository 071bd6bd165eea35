#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel of the HIP library, from the code-object
metadata hipcc emits for gfx950 (no GPU needed): compiles every translation unit of csrc/ device-only
to assembly with the product flags and prints one row per kernel.

  python scripts/kernel_regs.py [-D...] [--filter k_frame] [--asm /tmp/crender.s]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cython3dmodelrenderer_amd import _build  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--asm", default="/tmp/crender_device.s")
    args, extra = ap.parse_known_args()
    flags = [f for f in _build.HIPCC_FLAGS if f not in ("-shared", "-fPIC") and not f.startswith("-Wl,")]
    text = ""
    for src in _build.SOURCES:
        asm = args.asm + "." + os.path.splitext(src)[0]
        cmd = [_build._hipcc()] + flags + extra + ["--offload-device-only", "-S", "-o", asm, os.path.join(_build.SRC_DIR, src)]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        text += open(asm).read()
    # amdhsa.kernels metadata: one YAML map per kernel
    rows = []
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)

        def g(key, default="0"):
            mm = re.search(r"\." + key + r":\s+(\S+)", blk)
            return mm.group(1) if mm else default
        name = g("name", "?")
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True,
                                 text=True).stdout.strip()
        except OSError:
            dem = name
        dem = re.sub(r"\(.*", "", dem).replace("(anonymous namespace)::", "").replace("void ", "")
        if args.filter and args.filter not in dem:
            continue
        rows.append((dem, int(g("vgpr_count")), int(g("agpr_count")), int(g("sgpr_count")),
                     int(g("vgpr_spill_count")), int(g("sgpr_spill_count")),
                     int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
    print(f"{'kernel':48s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for r in sorted(rows):
        print(f"{r[0]:48s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:6d} {r[5]:6d} {r[6]:7d} {r[7]:6d}")


if __name__ == "__main__":
    main()
