import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for n in (8, 16, 32, 64):
    os.environ["GG_CPU_THREADS"] = str(n)
    t = time.time(); r = bench.cpu_baseline(5.0); print(n, r["value"], r["sample"], round(time.time() - t, 1), flush=True)
