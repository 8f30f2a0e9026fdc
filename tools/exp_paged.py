"""Host-side study of the page-at-a-time protocol: Q3 fed as pages of 2^k rows, alone (under rocprofv3 --hip-trace --stats this gives the HIP API time per page).
   python tools/exp_paged.py [--sf 100] [--steps 5] [--log2 20]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=float, default=100.0)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--log2", type=int, default=20)
ap.add_argument("--q1", action="store_true")
a = ap.parse_args()
ns = argparse.Namespace(gpus=1, steps=a.steps, warmup=a.warmup, sf=a.sf, only="paged", no_cpu_baseline=True, cpu_sample_sf=1.0)
b = bench.Bench(ns)
b.ctx.profile_enable(True)
b.ctx.set_device_input_stable(True)
if a.q1:
    b.setup_q1(int(6_000_000 * a.sf))
    r = b.q1_paged(a.steps, a.warmup, 1 << a.log2)
else:
    b.setup_q3(a.sf)
    b.q3_result = None
    r = b.q3_paged(a.steps, a.warmup, 1 << a.log2)
print(json.dumps(r), flush=True)
