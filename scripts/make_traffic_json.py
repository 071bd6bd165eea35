#!/usr/bin/env python3
"""profiles/traffic.json from a round report (gpurun_out/report/rocprof_<workload>.txt):
HBM bytes per k_raster launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters), FETCH_SIZE doubled as
MI355X_MICROARCH.md's HBM section prescribes for gfx950 (it counts 64 B per 128-B request of a
wide read); both passes were collected separately (scripts/profile_gpu.sh)."""
import json, os, re, sys
rep = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/report"
out = {}
for wl in ("trex1024", "bunny4096", "trex8192", "synth10m"):
    path = os.path.join(rep, f"rocprof_{wl}.txt")
    if not os.path.exists(path):
        continue
    txt = open(path).read()
    f = re.search(r"k_raster\s+FETCH_SIZE\s+n=\s*\d+ avg=\s*([\d.]+)", txt)
    w = re.search(r"k_raster\s+WRITE_SIZE\s+n=\s*\d+ avg=\s*([\d.]+)", txt)
    if not (f and w):
        continue
    fetch_raw = float(f.group(1)) * 1024
    write = float(w.group(1)) * 1024
    out[wl] = {"raster_fetch_bytes_raw": fetch_raw, "raster_fetch_bytes_x2": 2 * fetch_raw,
               "raster_write_bytes": write,
               "raster_hbm_bytes_per_launch": 2 * fetch_raw + write,
               "source": f"{path} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)"}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cython3dmodelrenderer_amd import _build  # noqa: E402
out["csrc_sha16"] = _build.source_sha16()       # the kernels these figures were measured on (bench.py checks it)
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
