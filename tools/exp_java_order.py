"""Times Q1 in the strict Java order (TGPU_SUM_ORDER_JAVA) at a few sizes: python tools/exp_java_order.py [sf ...]"""
import argparse
import json
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    sfs = [float(x) for x in sys.argv[1:]] or [1.0]
    b = bench.Bench(argparse.Namespace())
    p = b.pkg
    for sf in sfs:
        n = int(6_000_379.02 * sf)
        b.ctx.set_double_sum_order(p.SUM_ORDER_JAVA)
        try:
            b.setup_q1(n)
            s, prof = b.timed(b.step_q1, 3, 1, profile_apart=True)
        finally:
            b.ctx.set_double_sum_order(p.SUM_ORDER_EXACT)
        top = {k: round(v["total_ms"] / 3, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:5]}
        print(json.dumps({"sf": sf, "rows": n, "ms_per_step": round(s * 1e3, 3), "rows_per_sec": round(n / s), "kernels_ms": top}), flush=True)
        b.q1_result = None


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("seconds", round(time.time() - t0, 1))
