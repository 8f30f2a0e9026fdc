"""Lists who waits for the device in one Q3 step (TGPU_DEBUG_READBACKS=1 python tools/exp_q3_readbacks.py [sf] 2> trace.txt)"""
import argparse
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402

b = bench.Bench(argparse.Namespace())
b.setup_q3(float(sys.argv[1]) if len(sys.argv) > 1 else 10.0)
b.step_q3()
b.step_q3()
print("---- traced step ----", file=sys.stderr, flush=True)
b.ctx.profile_reset()
b.step_q3()
print("---- end ----", file=sys.stderr, flush=True)
print(b.ctx.profile().get("__readbacks"))
