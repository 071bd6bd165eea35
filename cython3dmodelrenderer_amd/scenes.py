"""Benchmark / test scenes of BASELINE.json's configs as rasterizer input arrays.

Each scene is the triple the reference's filler reads off a Model
(``_vertices_by_triangles, _colors_by_triangles, _normals_by_triangles``, float32
[T,3,3]; reference: advanced_pixel_buffer_filler.pyx:94-96).  The .obj assets live in
the reference checkout, which does not exist on the GPU box, so the arrays produced
from them by ``data_structures.Model`` are committed under tests/golden/ (see
scripts/make_golden.py) and loaded from there.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                          "tests", "golden")

# name -> (fixture file, default (h, w), fov)
CONFIGS = {
    "cube256": ("cube_inputs.npz", (256, 256), 45.0),      # configs[0]
    "trex1024": ("trex_inputs.npz", (1024, 1024), 45.0),   # configs[1], README benchmark
    "bunny4096": ("bunny_inputs.npz", (4096, 4096), 45.0), # configs[2]
    "trex8192": ("trex_inputs.npz", (8192, 8192), 45.0),   # configs[3]
    "synth10m": (None, (4096, 4096), 45.0),                # configs[4]
}


def fit_model(m):
    """The README's fit (reference: run.py:30-33, README.md:62-65)."""
    m.shift(-m.get_mean_vertex())
    m.scale(1 / m.get_max_span())
    m.shift(shift=[0, 0, 1])


def load_fixture(name):
    """(tri, col, nrm) float32 [T,3,3] from tests/golden/<name>."""
    with np.load(os.path.join(GOLDEN_DIR, name)) as z:
        tri = z["tri"]
        nrm = z["nrm"]
        col = z["col"] if "col" in z.files else np.full_like(tri, 255.0)
    return tri, col, nrm


def synthetic_triangles(T, res=4096, seed=12345, chunk=1 << 20):
    """Config 5: T small random triangles (about +-4 px at `res`)
    spread over the fov-45 frustum, nothing back-facing, colours U(0,255)."""
    rng = np.random.default_rng(seed)
    tri = np.empty((T, 3, 3), np.float32)
    col = np.empty((T, 3, 3), np.float32)
    nrm = np.empty((T, 3, 3), np.float32)
    for a in range(0, T, chunk):
        b = min(T, a + chunk)
        n = b - a
        cz = rng.uniform(0.8, 1.5, (n, 1)).astype(np.float32)
        cxy = rng.uniform(-0.40, 0.40, (n, 2)).astype(np.float32) * cz
        centre = np.concatenate([cxy, cz], axis=1)[:, None, :]
        r = (np.float32(4.0 * (2.0 / res) / 2.4142) * cz)[:, None, :]
        tri[a:b] = centre + rng.uniform(-1.0, 1.0, (n, 3, 3)).astype(np.float32) * r
        v = rng.standard_normal((n, 3, 3)).astype(np.float32)
        v /= np.linalg.norm(v, axis=-1, keepdims=True)
        v[..., 2] = -np.abs(v[..., 2])
        nrm[a:b] = v
        col[a:b] = rng.uniform(0.0, 255.0, (n, 3, 3)).astype(np.float32)
    return tri, col, nrm


def scene(name, synth_T=10_000_000):
    """(tri, col, nrm, (h, w), fov) for a CONFIGS entry."""
    fixture, size, fov = CONFIGS[name]
    if fixture is None:
        tri, col, nrm = synthetic_triangles(synth_T, res=size[0])
    else:
        tri, col, nrm = load_fixture(fixture)
    return tri, col, nrm, size, fov
