"""Fused filter/project/aggregation with a few hundred to a few thousand groups (the accumulators' ORDERED mode: DOUBLE sums in row order):
the chained kernel (one workgroup per group) against one lane per group (TGPU_DISABLE_ORDERED_CHAIN=1), same process, same table.
python tools/exp_medium_groups.py [rows]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    b = bench.Bench(argparse.Namespace())
    p = b.pkg
    B, D = p.BIGINT, p.DOUBLE
    f, c = p.field, p.constant
    i = torch.arange(n, device=b.dev, dtype=torch.int64)
    price = (90000 + bench.rnd(7, i, 120001)).to(torch.float64) / 100.0
    disc = bench.rnd(8, i, 11).to(torch.float64) / 100.0
    for groups in [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (64, 1000, 4096):
        keys = bench.rnd(3, i, groups)
        if len(sys.argv) > 3 and sys.argv[3] == "skew":      # 30 % of the rows in one group
            keys = torch.where(bench.rnd(4, i, 10) < 3, torch.zeros_like(keys), keys)
        page = p.Page(b.dblock(B, keys), b.dblock(D, price), b.dblock(D, disc))
        fac = p.FilterProjectHashAggregationOperatorFactory(b.ctx, 77, [B, D, D], f(2, D) < 0.095, [f(0, B), f(1, D), f(1, D) * (c(1.0, D) - f(2, D))], [B], [0],
                                                            [(p.SUM_DOUBLE, 1), (p.SUM_DOUBLE, 2), (p.AVG_DOUBLE, 2), (p.COUNT_ALL, -1)], expected_groups=groups)
        res = {}

        def step():
            op = fac.createOperator()
            op.addInput(page)
            outs = b.finish(op)
            res["pages"] = [o.to_host() for o in outs]
            op.close()
        line = {"rows": n, "groups": groups}
        got = {}
        for name, env in (("chained", None), ("lane_per_group", "1")):
            if env:
                os.environ["TGPU_DISABLE_ORDERED_CHAIN"] = env
            else:
                os.environ.pop("TGPU_DISABLE_ORDERED_CHAIN", None)
            s, prof = b.timed(step, 3, 1, profile_apart=True)
            line[name + "_ms"] = round(s * 1e3, 3)
            line[name + "_top"] = {k: round(v["total_ms"] / 3, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:4]}
            got[name] = sorted(r for o in res["pages"] for r in o.rows())
        os.environ.pop("TGPU_DISABLE_ORDERED_CHAIN", None)
        line["same_bits"] = got["chained"] == got["lane_per_group"]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
