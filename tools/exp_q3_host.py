"""Kernel study: where the host time of one Q3 step goes -- every call of bench.py's step_q3 timed on the host with the stream drained
after it (so a call's time = its own host work + the device work it enqueued), next to the device time the library's profile reports.

  python tools/exp_q3_host.py [sf]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    b = bench.Bench(argparse.Namespace())
    b.setup_q3(sf)
    p, ctx, f, pages = b.pkg, b.ctx, b.q3_fac, b.q3_pages
    B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
    acc = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        a = acc.setdefault(name, [0.0, 0.0])
        a[0] += t1 - t0
        a[1] += t2 - t1
        return r

    def step():
        pp = timed("bench_page_processors", lambda: b.entry.bench_page_processors(p))
        cb = timed("cust HashBuilderFactory", lambda: p.HashBuilderOperatorFactory(ctx, 10, [B], [], [0]))
        cbuild = timed("cust build createOperator", cb.createOperator)
        cfp = timed("cust fp createOperator", f["cust_fp"].createOperator)
        timed("cust fp addInput", lambda: cfp.addInput(pages["customer"]))
        o = timed("cust fp getOutput", cfp.getOutput)
        timed("cust build addInput", lambda: cbuild.addInput(o))
        timed("cust release", o.release)
        timed("cust build finish", cbuild.finish)
        oj = timed("orders JoinFactory", lambda: p.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3]))
        ob = timed("orders HashBuilderFactory", lambda: p.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0]))
        obuild = timed("orders build createOperator", ob.createOperator)
        ojoin = timed("orders join createOperator", oj.createOperator)
        timed("orders join addInput", lambda: ojoin.addInput(pages["orders"]))
        j = timed("orders join getOutput", ojoin.getOutput)
        timed("orders build addInput", lambda: obuild.addInput(j))
        timed("orders release", j.release)
        timed("orders build finish", obuild.finish)
        timed("orders join close", ojoin.close)
        lj = timed("lineitem JoinFactory", lambda: p.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, *pp["q3_lineitem"], [0], probe_output_channels=[0, 1]))
        agg = timed("agg Factory", lambda: p.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(p.SUM_DOUBLE, 1)], expected_groups=1 << 20))
        ljoin = timed("lineitem join createOperator", lj.createOperator)
        aop = timed("agg createOperator", agg.createOperator)
        timed("lineitem join addInput", lambda: ljoin.addInput(pages["lineitem"]))
        j = timed("lineitem join getOutput", ljoin.getOutput)
        timed("agg addInput", lambda: aop.addInput(j))
        timed("lineitem release", j.release)
        timed("agg finish", aop.finish)
        outs = []
        while not aop.isFinished():
            o2 = timed("agg getOutput", aop.getOutput)
            if o2 is not None:
                outs.append(o2)
        for o2 in outs:
            timed("agg out release", o2.release)
        timed("closes", lambda: [x.close() for x in (ljoin, cbuild, obuild, aop, cfp)])

    for _ in range(3):
        step()
    acc.clear()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    total = (time.perf_counter() - t0) / reps
    print(f"step with a drain after every call: {total * 1e3:.3f} ms")
    for k, (h, d) in acc.items():
        print(f"{k:34s} call {h / reps * 1e6:9.1f} us   drain after {d / reps * 1e6:9.1f} us")


if __name__ == "__main__":
    main()
