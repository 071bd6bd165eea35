"""Compiles csrc/*.hip into the in-tree shared library with hipcc.

The library is plain HIP behind a C ABI (include/crender_hip.h): no torch headers, so
``hipcc -c`` per translation unit (in parallel) and one ``hipcc -shared`` are the whole build.
gfx950 only.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libcrender_hip.so")
# binning.hip  K1 + the binning passes      raster.hip  K2 (tile rasterizer, k_frame, atomic path)
# model_ops.hip  rows f1-f4 of SURVEY 8f     abi.hip  plans, frames, swap chain (no kernels)
SOURCES = ["abi.hip", "binning.hip", "raster.hip", "model_ops.hip"]
HEADERS = ["common.h", "binning.h", "plan.h", "raster_math.h", os.path.join("..", "..", "include", "crender_hip.h")]

# Float parity with the reference depends on these (DESIGN.md "Numerics"):
#   -ffp-contract=off                           no FMA contraction (hipcc defaults to fast)
#   -fhip-fp32-correctly-rounded-divide-sqrt    IEEE division / sqrt
#   -fno-gpu-flush-denormals-to-zero            f32 denormals kept
# Speed only:
#   -fno-slp-vectorize   the SLP vectoriser pairs the barycentric arithmetic into v_pk_*_f32
#                        ops, which cost as much as two scalar ops each plus ~20 register moves
#                        per sample to line the operands up, and 10-25 more VGPRs; without it
#                        every workload is 3-10 % faster (r01 A/B, same box)
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-gpu-flush-denormals-to-zero",
    "-fno-slp-vectorize",
    "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra",
]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-soname,libcrender_hip.so"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def source_sha16() -> str:
    """Fingerprint of everything the kernels are built from — the translation units, their headers and
    the compiler flags: profiles/*.json carry it from profiling time, and bench.py quotes a committed
    profile figure only when it matches the sources it runs."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(SRC_DIR, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(SRC_DIR, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def compile_library(out: str, extra_flags=(), sources=None, src_dir: str = SRC_DIR, verbose: bool = False,
                    quiet: bool = False) -> str:
    """hipcc -c every translation unit (side by side), then link them into `out`.  `extra_flags`:
    defines of development / diagnostic builds (-DCRENDER_DEV_KNOBS, -DCRENDER_STAMPS, ...)."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    sources = list(sources or SOURCES)
    err = subprocess.DEVNULL if quiet else None
    with tempfile.TemporaryDirectory(prefix="crender_build_") as tmp:
        objs = [os.path.join(tmp, os.path.splitext(s)[0] + ".o") for s in sources]

        def one(job):
            src, obj = job
            cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-c", "-o", obj, os.path.join(src_dir, src)]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, stderr=err)

        with ThreadPoolExecutor(max_workers=min(4, len(sources))) as pool:
            list(pool.map(one, zip(sources, objs)))
        # link beside the target and rename onto it: ranks started together on a stale library all
        # build, and none may dlopen a half-written file
        part = f"{out}.{os.getpid()}.part"
        cmd = [_hipcc()] + LINK_FLAGS + ["-o", part] + objs
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd, stderr=err)
            os.replace(part, out)
        finally:
            if os.path.exists(part):
                os.remove(part)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Build libcrender_hip.so if missing or stale; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    return compile_library(LIB_PATH, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
