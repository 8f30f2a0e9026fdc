"""Kernel study: host cost of creating the six operator factories + operators of one Q3 step (measured: ~53 us per step)."""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import torch
pkg = importlib.import_module("presto-1_amd")
entry = importlib.import_module("__graft_entry__")
ctx = pkg.Context(0)
B, D, DT, I = pkg.BIGINT, pkg.DOUBLE, pkg.DATE, pkg.INTEGER
def once():
    pp = entry.bench_page_processors(pkg)
    cb = pkg.HashBuilderOperatorFactory(ctx, 10, [B], [], [0])
    f1 = pkg.FilterAndProjectOperatorFactory(ctx, 9, *pp["q3_customer"])
    oj = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
    ob = pkg.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0])
    lj = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, *pp["q3_lineitem"], [0], probe_output_channels=[0, 1])
    agg = pkg.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 2, 3], [(pkg.SUM_DOUBLE, 1)], expected_groups=1 << 20)
    ops = [cb.createOperator(), f1.createOperator(), oj.createOperator(), ob.createOperator(), lj.createOperator(), agg.createOperator()]
    for o in ops: o.close()
    for f in (cb, f1, oj, ob, lj, agg): f.close()
for _ in range(5): once()
t0 = time.perf_counter()
for _ in range(200): once()
print("factories + operators per step: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
