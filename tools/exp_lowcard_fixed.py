"""Kernel study: per-launch time of the Q1 fused kernels (group probe, project + accumulate) as a function of the page's row count --
the intercept is the launch's fixed cost (DESIGN.md "Page granularity").  Prints one line per page size.

  python tools/exp_lowcard_fixed.py [rows ...]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 16384, 131072, 1 << 20, 1 << 22, 1 << 24]
    args = argparse.Namespace(gpus=1, steps=1, warmup=0, sf=1.0)
    b = bench.Bench(args)
    b.setup_q1(max(sizes) + 1)
    b.ctx.profile_enable(True)
    for n in sizes:
        page = b.q1_page(0, n)
        reps = 30
        aop = b.q1_agg.createOperator()
        for _ in range(3):
            aop.addInput(page)
        b.ctx.profile_reset()
        for _ in range(reps):
            aop.addInput(page)
        prof = b.ctx.profile()
        b.finish(aop)
        aop.close()
        line = {k: round(v["total_ms"] / max(v["count"], 1) * 1e3, 1) for k, v in prof.items() if not k.startswith("__")}
        print(n, line, flush=True)


if __name__ == "__main__":
    main()
