#!/bin/bash
# Development build of the HIP library with the measurement knobs compiled in
# (CRENDER_DEBUG bit mask, see csrc/raster.hip) -> /tmp/libcrender_hip_dev.so.
# Use with CRENDER_LIB=/tmp/libcrender_hip_dev.so.  The product library has none of them.
#   scripts/dev_build.sh [-DNAME=VALUE ...] [--out PATH]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
python - "$@" <<'PY'
import sys
from cython3dmodelrenderer_amd import _build
args = sys.argv[1:]
out = "/tmp/libcrender_hip_dev.so"
if "--out" in args:
    i = args.index("--out"); out = args[i + 1]; del args[i:i + 2]
nodev = "--no-dev-knobs" in args
args = [a for a in args if a != "--no-dev-knobs"]
_build.compile_library(out, ([] if nodev else ["-DCRENDER_DEV_KNOBS"]) + args, quiet=True)
print(out)
PY
