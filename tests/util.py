"""Shared helpers of the parity tests."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_equal(got, want, what):
    """Bit-exact comparison of float32 / int32 arrays with a useful failure message."""
    g, w = bits(got), bits(want)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    bad = np.argwhere(g != w)
    if len(bad):
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {g.size} elements differ; first at {i}: "
                             f"got {np.asarray(got)[i]!r} want {np.asarray(want)[i]!r}")


def random_soup(rng, T, res, size_px=(1.0, 40.0), z=(0.5, 3.0), frac_backface=0.2, margin=0.3):
    """Random projected-space-friendly triangles in MODEL space for a fov-45 filler:
    centres spread a bit beyond the frustum, sizes in pixels at resolution `res`."""
    f = 2.4142137
    cz = rng.uniform(z[0], z[1], (T, 1)).astype(np.float32)
    half = (1.0 + margin) / f
    cxy = rng.uniform(-half, half, (T, 2)).astype(np.float32) * cz
    centre = np.concatenate([cxy, cz], 1)[:, None, :]
    px = rng.uniform(size_px[0], size_px[1], (T, 1, 1)).astype(np.float32)
    r = px * (2.0 / res) / f * cz[:, None, :]
    tri = (centre + rng.uniform(-1, 1, (T, 3, 3)).astype(np.float32) * r).astype(np.float32)
    nrm = rng.standard_normal((T, 3, 3)).astype(np.float32)
    nrm[..., 2] = -np.abs(nrm[..., 2])
    flip = rng.uniform(size=T) < frac_backface
    nrm[flip, :, 2] *= -1
    col = rng.uniform(0, 255, (T, 3, 3)).astype(np.float32)
    return tri, col, nrm


def numpy_dot3_is_double_accumulated():
    """True if this numpy's np.dot of float32 3-vectors is float32 products in a double accumulator,
    rounded once (scipy-openblas 0.3.29's x86-64 sdot: what row f2's device kernels spell out).  The
    bit-exact device-vs-host-Model checks of the vertex normals hold for such a numpy only; on any
    other BLAS they are 'parity unpinned' and fall back to the bounded check."""
    rng = np.random.default_rng(0)
    n = 4000
    a = rng.standard_normal((n, 3)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + rng.standard_normal((n, 3)).astype(np.float32) * np.float32(1e-4)
    ref = np.array([np.dot(a[i], b[i]) for i in range(n)], dtype=np.float32)
    p = a * b
    mine = (p[:, 0].astype(np.float64) + p[:, 1].astype(np.float64) + p[:, 2].astype(np.float64)).astype(np.float32)
    return bool(np.array_equal(mine.view(np.uint32), ref.view(np.uint32)))
