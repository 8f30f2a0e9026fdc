"""Diagnostic: the page-granularity test's Q1 half alone (2.4 M rows as 2^16-row pages, directly and through MergePages), phase markers on stderr."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
b = bench.Bench(argparse.Namespace())
b.setup_q1(2_400_011)
b.step_q1()
single = [r for pg in b.q1_result for r in pg]
for merge in (None, 8):
    print("== merge", merge, file=sys.stderr, flush=True)
    res = b.q1_paged(1, 0, 1 << 16, merge_mb=merge)
    print("== done", merge, res["ok"], [r for pg in b.q1_result for r in pg] == single, file=sys.stderr, flush=True)
