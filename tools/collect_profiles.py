"""Turns rocprofv3 output directories (under gpurun_out/, scratch) into the tracked summaries under profiles/.

    python tools/collect_profiles.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> [<run_dir>]]

  <stats_dir>      rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  <pmc_*_dir>      rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
                   (separate passes; never combined with other trace domains)
  <run_dir>        tools/final_measure.sh's output directory: its pmc_gbh_<shape>_<counter> passes (tools/exp_sub.py group_by_hash_<shape>, one
                   BenchmarkGroupByHash shape per pass) become "sub_group_by_hash_<shape>:gbh_insert"

Writes profiles/<tag>_kernel_stats.csv (rocprofv3's own kernel statistics, this library's kernels only) and
profiles/<tag>_pmc_traffic.json (HBM bytes per launch of the hot kernels, corrected as MI355X_MICROARCH.md prescribes)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("fj_", "fg_", "fa_", "fp_", "fq_", "tgpu::", "void tgpu::")   # rocPRIM kernels (scan / radix sort) are shared with torch: left out
PROFILE_NAME = {"fj_probe_direct": "fused_filter_probe", "fj_emit_direct": "fused_probe_emit", "fg_probe": "fused_filter_group_probe",
                "fa_accumulate_lowcard": "fused_project_accumulate_lowcard", "fq_onepass": "fused_filter_group_accumulate_onepass", "fp_count": "filter_count",
                "fp_emit": "filter_project_emit",
                # sub-benchmarks (bench.py looks them up under these keys: their kernels share profile names with the headline's)
                "fj_probe_bloom": "sub_join_hash_layout:fused_filter_probe",
                "void tgpu::(anonymous namespace)::probe_count_kernel<true>": "sub_join_duplicate_keys:join_probe_count"}


def profile_key(kernel_name):
    for prefix in PROFILE_NAME:
        if kernel_name == prefix or (prefix.startswith("void ") and kernel_name.startswith(prefix)):
            return prefix
    return None


def find(d, suffix):
    m = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not m:
        raise SystemExit(f"no *{suffix} under {d}")
    return m[0]


def kernel_stats(tag, d):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    keep = [r for r in rows if r["Name"].startswith(OURS)]
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    print("wrote", out, len(keep), "kernels")


def counter(d, name):
    per = {}
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if r["Counter_Name"] != name:
            continue
        k = profile_key(r["Kernel_Name"])
        if k is not None:
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    return per


GBH_KERNEL = "gbi_insert_kernel"   # the single-BIGINT-key table's insert kernel (groupby_bigint.hip), profile scope gbh_insert


def gbh_counter(d, name):
    vals = []
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if r["Counter_Name"] == name and GBH_KERNEL in r["Kernel_Name"] and "true>" in r["Kernel_Name"]:   # <key type, INSERT = true>
            vals.append(float(r["Counter_Value"]))
    return vals


def pmc(tag, dfetch, dwrite, run_dir=None):
    fetch, write = counter(dfetch, "FETCH_SIZE"), counter(dwrite, "WRITE_SIZE")
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `bench.py --steps 1 --warmup 1 "
                       "--no-cpu-baseline` on MI355X; warm-up launches included in the per-launch lists.  Counter unit = KiB.  gfx950 correction "
                       "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a coalesced streaming read -> doubled; "
                       "WRITE_SIZE is used as reported.", "kernels": {}}
    for k, name in PROFILE_NAME.items():
        if k not in fetch:
            continue
        f, w = fetch[k], write.get(k, [0.0] * len(fetch[k]))
        n = min(len(f), len(w))
        traffic = sum(2.0 * f[i] * 1024.0 + w[i] * 1024.0 for i in range(n)) / max(n, 1)
        out["kernels"][name] = {"rocprof_kernel_name": k, "launches": n, "fetch_size_kib_per_launch": f[:n], "write_size_kib_per_launch": w[:n],
                                "traffic_bytes_per_launch_avg": traffic}
    for shape in ("10M_3M", "100M_40M"):
        df, dw = (os.path.join(run_dir, f"pmc_gbh_{shape}_{c}") for c in ("FETCH_SIZE", "WRITE_SIZE")) if run_dir else (None, None)
        if not df or not os.path.isdir(df) or not os.path.isdir(dw):
            continue
        f, w = gbh_counter(df, "FETCH_SIZE"), gbh_counter(dw, "WRITE_SIZE")
        n = min(len(f), len(w))
        if n == 0:
            continue
        out["kernels"][f"sub_group_by_hash_{shape}:gbh_insert"] = {
            "rocprof_kernel_name": GBH_KERNEL + "<long long, true>", "command": f"python3 tools/exp_sub.py group_by_hash_{shape} --steps 2 --warmup 1", "launches": n,
            "fetch_size_kib_per_launch": f[:n], "write_size_kib_per_launch": w[:n],
            "traffic_bytes_per_launch_avg": sum(2.0 * f[i] * 1024.0 + w[i] * 1024.0 for i in range(n)) / n}
    p = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
    json.dump(out, open(p, "w"), indent=1)
    print("wrote", p, list(out["kernels"]))


if __name__ == "__main__":
    tag = sys.argv[1]
    kernel_stats(tag, sys.argv[2])
    if len(sys.argv) >= 5:
        pmc(tag, sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) >= 6 else None)
