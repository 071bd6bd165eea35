"""How fast the GPU gets through a burst of frames when the host is not in the way: the caller's
stream is kept busy (torch.cuda._sleep) while K frames are submitted behind it, so all of them are
queued before the first one starts.  Compare with scripts/k20_host.py (live submission).
Modes: "events" times with torch events on the caller's stream; "host" with the host clock (a second
thread submits while the main thread waits for the sleep to end)."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cython3dmodelrenderer_amd import scenes
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller
wl = sys.argv[1] if len(sys.argv) > 1 else "trex1024"
tri, col, nrm, (H, W), fov = scenes.scene(wl)
f = AdvancedPixelBufferFiller(H, W, fov=fov, pipeline=True)
f.render_arrays(tri, col, nrm, clear=True); f.synchronize()
f.render_frame(); f.synchronize()
for mode in ("events", "host", "live"):
    for K in (20, 20, 20, 100):
        for _ in range(5): f.render_frame()
        f.synchronize(); torch.cuda.synchronize()
        if mode == "events":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(int(4e6 * K / 20))
            a.record()
            for _ in range(K):
                f.render_frame()
            f.join(); b.record()
            torch.cuda.synchronize()
            us = 1e3 * a.elapsed_time(b)
        elif mode == "host":
            a = torch.cuda.Event()
            torch.cuda._sleep(int(8e6 * K / 20))
            a.record()
            done = []
            def submit():
                torch.cuda.set_device(0)
                for _ in range(K):
                    f.render_frame()
                done.append(time.perf_counter())
            th = threading.Thread(target=submit); th.start()
            a.synchronize(); ta = time.perf_counter()
            th.join()
            f.synchronize(); torch.cuda.synchronize(); te = time.perf_counter()
            us = 1e6 * (te - ta)
            print(f"   submission finished {1e6*(ta-done[0]):.0f} us before the sleep ended")
        else:
            t0 = time.perf_counter()
            for _ in range(K):
                f.render_frame()
            f.synchronize(); torch.cuda.synchronize()
            us = 1e6 * (time.perf_counter() - t0)
        print(f"{wl} {mode} K={K}: burst {us:.1f} us = {us/K:.2f} us per frame")
