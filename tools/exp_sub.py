"""Kernel study: bench.py's sub-benchmarks alone (open-address join layout, duplicate build keys, BenchmarkGroupByHash shapes), without the headline programs.
   python tools/exp_sub.py [join_hash_layout] [join_duplicate_keys] [group_by_hash] [group_by_hash_10M_3M] [group_by_hash_100M_40M] [--sf 100] [--steps 10]
   (one group-by shape alone: what the PMC passes of tools/final_measure.sh profile, so that a kernel name maps to one shape)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="*", default=["join_hash_layout", "join_duplicate_keys", "group_by_hash"])
ap.add_argument("--sf", type=float, default=100.0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
ns = argparse.Namespace(gpus=1, steps=a.steps, warmup=a.warmup, sf=a.sf, only="sub", no_cpu_baseline=True, cpu_sample_sf=1.0)
b = bench.Bench(ns)
b.ctx.profile_enable(True)
out = {}
if "join_hash_layout" in a.which:
    out["join_hash_layout"] = bench.sub_join_hash_layout(b, a.steps, a.warmup, a.sf)
if "join_duplicate_keys" in a.which:
    out["join_duplicate_keys"] = bench.sub_join_duplicate_keys(b, a.steps, a.warmup, a.sf)
if "group_by_hash" in a.which or "group_by_hash_10M_3M" in a.which:
    out["group_by_hash_10M_3M"] = bench.sub_group_by_hash(b, a.steps, a.warmup, int(100_000 * a.sf), int(30_000 * a.sf))
if "group_by_hash" in a.which or "group_by_hash_100M_40M" in a.which:
    out["group_by_hash_100M_40M"] = bench.sub_group_by_hash(b, a.steps, a.warmup, int(1_000_000 * a.sf), int(400_000 * a.sf))
for k, v in out.items():
    print(k, json.dumps({x: v[x] for x in ("ms_per_step", "kernels_ms_per_step", "ok")}), "frac", v["roofline"]["frac"] if v.get("roofline") else None, flush=True)
