"""GPU-side timeline of a 20-frame burst (the driver's bench.py --steps 20 --warmup 5, T-Rex 1024^2):
run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python scripts/k20_trace.py`, then
`python scripts/k20_trace.py --read DIR` prints, for the last burst, when each frame's two launches
started and ended (us since the burst's first kernel start) and on which queue."""
import os, sys, glob, csv
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = []
    for p in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(p)))
    rows = [r for r in rows if "k_raster" in r["Kernel_Name"] or "k_setup" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-40:]
    t0 = int(last[0]["Start_Timestamp"])
    qs = sorted({r["Queue_Id"] for r in last})
    print("queues:", qs)
    ends = []
    for r in last:
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        name = "raster" if "k_raster" in r["Kernel_Name"] else "setup "
        print(f"  q{qs.index(r['Queue_Id'])} {name} start {a:8.2f} end {b:8.2f} dur {b-a:6.2f}")
        if name == "raster":
            ends.append(b)
    ends.sort()
    print("frame completion times:", " ".join(f"{e:.1f}" for e in ends))
    print("gaps:", " ".join(f"{b-a:.1f}" for a, b in zip(ends, ends[1:])))
    sys.exit(0)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
tri, col, nrm, (H, W), fov = scenes.scene("trex1024")
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
f.render_frame(); f.synchronize()
for rep in range(3):
    for _ in range(5): f.render_frame()
    f.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f.render_frame()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); te = time.perf_counter()
    print(f"issue {1e6*(t1-t0):.1f} us, total {1e6*(te-t0):.1f} us, per frame {1e6*(te-t0)/20:.2f} us")
    time.sleep(0.05)
