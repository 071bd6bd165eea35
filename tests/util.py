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
