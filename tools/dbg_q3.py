import argparse, importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
b = bench.Bench(argparse.Namespace())
p = b.pkg
seq = sys.argv[1].split(",")
for s in seq:
    if s == "cfg2":
        b.setup_cfg2(1_000_000); b.step_cfg2(); print("cfg2", b.check_cfg2(), flush=True)
    elif s.startswith("q1j"):
        b.ctx.set_double_sum_order(p.SUM_ORDER_JAVA)
        b.setup_q1(int(s[3:])); b.step_q1(); print(s, len(b.q1_result), flush=True)
        b.ctx.set_double_sum_order(p.SUM_ORDER_EXACT)
    elif s.startswith("q1"):
        b.setup_q1(int(s[2:])); b.step_q1(); print(s, b.check_q1()["ok"], flush=True)
    elif s.startswith("q3c"):
        b.setup_q3(float(s[3:])); b.q3_stats = {}; b.capture = {}
        b.step_q3(); b.capture = None
        print(s, b.q3_stats, flush=True)
    elif s.startswith("q3"):
        b.setup_q3(float(s[2:])); b.q3_stats = {}
        b.step_q3()
        print(s, b.q3_stats, flush=True)
