#!/bin/bash
# kernel study: dump the generated Q1 fused-aggregation source (no-nulls specialisation) and compile it to ISA under /tmp
cd /root/repo
find presto-1_amd/_kcache -name '*.hip' -delete
TGPU_JIT_DUMP=1 python - <<'PY' 2>&1 | tail -3
import importlib, sys
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("presto-1_amd")
e = importlib.import_module("__graft_entry__")
pp = e.bench_page_processors(pkg)
pkg.precompile_fused_aggregation(*pp["q1"], e.q1_aggregates(pkg), [0, 1])
PY
for f in presto-1_amd/_kcache/*.hip; do if head -1 $f | grep -q "NO_NULLS 1"; then (echo '#include <hip/hip_runtime.h>'; cat $f) > /tmp/fa.hip; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S /tmp/fa.hip -o /tmp/fa.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: f[ga]_" | grep -E "Name|VGPRs:|Scratch|Occupancy"
awk '/^fg_probe:/,/s_endpgm/' /tmp/fa.s > /tmp/fg.s
awk '/^fa_accumulate_lowcard:/,/s_endpgm/' /tmp/fa.s > /tmp/fa_lc.s
wc -l /tmp/fg.s /tmp/fa_lc.s
