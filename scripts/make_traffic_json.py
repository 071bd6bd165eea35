#!/usr/bin/env python3
"""profiles/traffic.json from a round report (gpurun_out/report/rocprof_<workload>.txt):
HBM bytes per launch of each raster kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters), FETCH_SIZE doubled as
MI355X_MICROARCH.md's HBM section prescribes for gfx950 (it counts 64 B per 128-B request of a
wide read); both passes were collected separately (scripts/profile_gpu.sh)."""
import json, os, re, sys
rep = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/report"
out = {}
def bytes_of(txt, kernel):
    f = re.search(kernel + r"\s+FETCH_SIZE\s+n=\s*\d+ avg=\s*([\d.]+)", txt)
    w = re.search(kernel + r"\s+WRITE_SIZE\s+n=\s*\d+ avg=\s*([\d.]+)", txt)
    if not (f and w):
        return None
    fetch_raw = float(f.group(1)) * 1024
    write = float(w.group(1)) * 1024
    return {"fetch_bytes_raw": fetch_raw, "fetch_bytes_x2": 2 * fetch_raw, "write_bytes": write,
            "hbm_bytes_per_launch": 2 * fetch_raw + write}


for wl in ("trex1024", "bunny4096", "trex8192", "synth10m"):
    e = {}
    # per KERNEL (the line's roofline.kernel picks its own): k_raster / k_raster_owners from the single-stream run,
    # k_frame / k_frame_owners (a frame's raster workgroups + the next frame's binning wavefronts) from the pipelined one
    for suffix, kernels in (("", ("k_raster_owners", "k_raster")), ("_pipelined", ("k_frame_owners", "k_frame"))):
        path = os.path.join(rep, f"rocprof_{wl}{suffix}.txt")
        if not os.path.exists(path):
            continue
        txt = open(path).read()
        for k in kernels:
            b = bytes_of(txt, k + " ")
            if b:
                b["source"] = f"{path} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)"
                e[k] = b
    if e:
        out[wl] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cython3dmodelrenderer_amd import _build  # noqa: E402
out["csrc_sha16"] = _build.source_sha16()       # the kernels these figures were measured on (bench.py checks it)
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
