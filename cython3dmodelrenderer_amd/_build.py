"""Compiles csrc/crender_hip.hip into the in-tree shared library with hipcc.

The library is plain HIP behind a C ABI (include/crender_hip.h): no torch headers, so
a bare ``hipcc -shared`` is the whole build.  gfx950 only.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libcrender_hip.so")
SOURCES = ["crender_hip.hip"]
HEADERS = ["raster_math.h", os.path.join("..", "..", "include", "crender_hip.h")]

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
    "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra",
    "-Wl,-soname,libcrender_hip.so",
]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(SRC_DIR, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Build libcrender_hip.so if missing or stale; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(SRC_DIR, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
