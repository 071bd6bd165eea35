#!/bin/sh
# Builds the one piece of the reference's hot path that compiles and loads here
# without anything the image lacks: math_utils.pyx (the per-pixel barycentric
# function, reference crender/cy/pixel_buffer_filler/math_utils.pyx:8-34).
#
# The source is compiled from where it lies under /root/reference; only the
# generated C and the shared object are written, both into oracle/_ref/ (git-ignored,
# travels to the GPU box with the snapshot).  Nothing is copied into the repository.
#
# The filler itself (advanced_pixel_buffer_filler.pyx) is NOT built: its module-level
# `from crender.py.data_structures import Model` (.pyx:17) imports cv2, which this
# image does not have, and no stand-in is written for it ("Oracle").
set -e
REF=${REFERENCE_ROOT:-/root/reference}
SRC="$REF/crender/cy/pixel_buffer_filler/math_utils.pyx"
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/_ref"
if [ ! -f "$SRC" ]; then
    echo "build_ref.sh: $SRC not present; keeping any prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$OUT"
PYINC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
# legacy_implicit_noexcept: the reference predates Cython 3; without it every cdef
# call would carry an exception check (same values, different ABI of the capsule).
cython -3 -X legacy_implicit_noexcept=True "$SRC" -o "$OUT/math_utils.c" 2>/dev/null
# Same code generation as the reference's default build: -O2, no -march, no fast-math.
gcc -O2 -fPIC -shared -I"$PYINC" "$OUT/math_utils.c" -o "$OUT/math_utils.so"
rm -f "$OUT/math_utils.c"
echo "built $OUT/math_utils.so"
