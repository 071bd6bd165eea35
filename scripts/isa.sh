#!/bin/bash
# Device ISA of the HIP library's translation unit(s) -> /tmp/cr_<name>.s (product flags; add defines as arguments)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
FLAGS=$(python -c "from cython3dmodelrenderer_amd import _build; print(' '.join(f for f in _build.HIPCC_FLAGS if f not in ('-shared','-fPIC') and not f.startswith('-Wl')))")
for src in $(python -c "from cython3dmodelrenderer_amd import _build; print(' '.join(s for s in _build.SOURCES if s.endswith('.hip')))"); do
  /opt/rocm/bin/hipcc $FLAGS "$@" -S --cuda-device-only -o /tmp/cr_$(basename $src .hip).s cython3dmodelrenderer_amd/csrc/$src 2>/dev/null && echo /tmp/cr_$(basename $src .hip).s
done
