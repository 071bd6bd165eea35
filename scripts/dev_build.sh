#!/bin/bash
# Development build of the HIP library with the measurement knobs compiled in
# (CRENDER_DEBUG bit mask, see csrc/crender_hip.hip) -> /tmp/libcrender_hip_dev.so.
# Use with CRENDER_LIB=/tmp/libcrender_hip_dev.so.  The product library has none of them.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
python - "$@" <<'PY'
import subprocess, sys, os
from cython3dmodelrenderer_amd import _build
out = "/tmp/libcrender_hip_dev.so"
subprocess.check_call([_build._hipcc()] + _build.HIPCC_FLAGS + ["-DCRENDER_DEV_KNOBS"] + sys.argv[1:] +
                      ["-o", out, os.path.join(_build.SRC_DIR, "crender_hip.hip")], stderr=subprocess.DEVNULL)
print(out)
PY
