#!/bin/bash
# Device ISA of the HIP library's translation units -> /tmp/cr_<name>.s (product flags; add defines as arguments)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
FLAGS=$(python -c "from cython3dmodelrenderer_amd import _build; print(' '.join(f for f in _build.HIPCC_FLAGS if f != '-fPIC'))")
for src in $(python -c "from cython3dmodelrenderer_amd import _build; print(' '.join(_build.SOURCES))"); do
  /opt/rocm/bin/hipcc $FLAGS "$@" -S --cuda-device-only -o /tmp/cr_$(basename $src .hip).s cython3dmodelrenderer_amd/csrc/$src 2>/dev/null && echo /tmp/cr_$(basename $src .hip).s
done
