"""Thread sweep of bench.py's `cpu_baseline` forward leg (same weights, clips and oracle call) on this host: which thread count
is the port's best case.  Output committed under profiles/ (rNN_cpu_thread_sweep.log)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

print("logical cores:", os.cpu_count(), flush=True)
for n in (4, 8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 1):
        break
    r = bench.cpu_baseline(passes=3, warmups=1, with_train=False, threads=n)
    print(n, "threads:", r["value"], "audio-s/s (median of 3 passes of 4 clips x 2 s after 1 warm-up)", flush=True)
