"""Diagnostic: bench.py itself with per-call times recorded inside its api_call loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
real = bench._timed
def timed(call, sync, budget):
    call(); call(); sync()
    t0 = time.perf_counter(); call(); sync(); one = time.perf_counter() - t0
    n = int(max(3, min(200, budget / max(one, 1e-5))))
    ts = []
    t0 = time.perf_counter()
    for _ in range(n):
        t1 = time.perf_counter(); call(); ts.append(time.perf_counter() - t1)
    sync()
    total = (time.perf_counter() - t0) / n * 1e3
    ts = np.array(ts) * 1e3
    print(f"    _timed: {total:.3f} ms over {n} calls (first timed call {one*1e3:.3f}); per-call median {np.median(ts):.3f}, p90 {np.percentile(ts,90):.3f}, "
          "slowest " + " ".join(f"{i}:{ts[i]:.2f}" for i in np.argsort(-ts)[:4]), file=sys.stderr)
    return total
bench._timed = timed
bench.main()
