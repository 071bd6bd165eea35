#!/usr/bin/env python3
"""CPU model of the lane sweep's work on the synthetic small-triangle frame (same density as
config 5, smaller frame): records per tile, sweep-length classes, trips per wavefront with and
without the in-batch sort, candidates per record.  Statistics only (float64 geometry, no parity)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from cython3dmodelrenderer_amd import scenes

RES = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(10_000_000 * (RES / 4096) ** 2)
TS = 32
tri, col, nrm = scenes.synthetic_triangles(T, res=RES)
f = 1.0 / np.tan(np.radians(45.0) / 2)
x = (tri[..., 0] * f / tri[..., 2] + 1) * RES / 2
y = (tri[..., 1] * f / tri[..., 2] + 1) * RES / 2
xl = np.clip(np.ceil(x.min(1)), 0, RES).astype(int); xr = np.clip(np.ceil(x.max(1)), 0, RES).astype(int)
yt = np.clip(np.ceil(y.min(1)), 0, RES).astype(int); yb = np.clip(np.ceil(y.max(1)), 0, RES).astype(int)
ok = (xl < xr) & (yt < yb)
print("triangles", T, "non-empty", ok.sum(), "bbox samples/tri", ((xr - xl) * (yb - yt))[ok].mean())
# records: (tile, clipped box)
recs = []
ntx = RES // TS
for i in np.nonzero(ok)[0][:]:
    pass
tx0, tx1 = xl // TS, (xr - 1) // TS
ty0, ty1 = yt // TS, (yb - 1) // TS
tile_list = {}
tiles_idx = []; bw = []; bh = []
for dx in range(0, 3):
    for dy in range(0, 3):
        sel = ok & (tx0 + dx <= tx1) & (ty0 + dy <= ty1)
        tx = tx0[sel] + dx; ty = ty0[sel] + dy
        cxl = np.maximum(xl[sel], tx * TS); cxr = np.minimum(xr[sel], tx * TS + TS)
        cyt = np.maximum(yt[sel], ty * TS); cyb = np.minimum(yb[sel], ty * TS + TS)
        tiles_idx.append(ty * ntx + tx); bw.append(cxr - cxl); bh.append(cyb - cyt)
tiles_idx = np.concatenate(tiles_idx); bw = np.concatenate(bw); bh = np.concatenate(bh)
print("records", len(bw), "per triangle", len(bw) / T, "samples/record", (bw * bh).mean())
pairs = ((bw + 1) // 2) * bh
print("pairs/record mean", pairs.mean(), "max", pairs.max(), "hist", np.bincount(np.minimum(pairs, 40))[:41])
order = np.argsort(tiles_idx, kind="stable")
tiles_sorted = tiles_idx[order]; pairs_s = pairs[order]
starts = np.searchsorted(tiles_sorted, np.arange(ntx * ntx)); ends = np.searchsorted(tiles_sorted, np.arange(ntx * ntx), side="right")
rng = np.random.default_rng(0)
tot_unsorted = tot_sorted = tot_sorted128 = nwaves = 0; wgmax = 0; wgsum = 0
nb = 0
for a, b in zip(starts, ends):
    p = pairs_s[a:b].copy()
    rng.shuffle(p)     # (list order within a tile is arbitrary)
    for o in range(0, len(p), 256):
        q = np.zeros(256, int); q[:len(p[o:o + 256])] = p[o:o + 256]
        nb += 1
        u = q.reshape(4, 64).max(1); tot_unsorted += u.sum()
        s = np.sort(q).reshape(4, 64).max(1); tot_sorted += s.sum()
        wgmax += s.max() * 4; wgsum += s.sum()
        nwaves += 4
print("batches", nb, "wave passes", nwaves)
print("trips per wave pass: ideal(mean pairs)", pairs.sum() / 64 / nwaves, "unsorted", tot_unsorted / nwaves, "sorted", tot_sorted / nwaves)
print("workgroup-coupled (every wave waits for the slowest): ", wgmax / nwaves)
scale = (4096 / RES) ** 2
print("at 4096^2: wave passes %.0fK, trips sorted %.2fM unsorted %.2fM" % (nwaves * scale / 1e3, tot_sorted * scale / 1e6, tot_unsorted * scale / 1e6))
# ---- flattened items of four x-neighbours (wave_sweep): items per record, trips per sub-batch of 64 records
quads = ((bw + 3) // 4) * bh
print("quads/record mean", quads.mean(), "samples per quad", (bw * bh).sum() / quads.sum())
qs = quads[order]
trips = subb = 0
for a, b in zip(starts, ends):
    n = b - a
    for o in range(0, n, 256):
        q = qs[a + o:a + min(n, o + 256)]
        for w in range(4):
            qq = q[w * 64:(w + 1) * 64]
            if len(qq) and qq.sum():
                subb += 1
                trips += -(-qq.sum() // 64)
print("sub-batches", subb, "trips", trips, "trips/sub-batch", trips / subb, "at 4096^2: sub-batches %.0fK trips %.2fM" % (subb * scale / 1e3, trips * scale / 1e6))
