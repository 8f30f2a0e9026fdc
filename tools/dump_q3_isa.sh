#!/bin/bash
# kernel study: dump the generated fused lineitem probe source and compile it to ISA under /tmp
cd /root/repo
find presto-1_amd/_kcache -name '*.hip' -delete
TGPU_JIT_DUMP=1 python - ${4:-q3_lineitem} <<'PY' 2>&1 | tail -3
import importlib, sys
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("presto-1_amd")
e = importlib.import_module("__graft_entry__")
pp = e.bench_page_processors(pkg)
pkg.precompile_fused_probe(*pp[sys.argv[1] if len(sys.argv) > 1 else "q3_lineitem"], *((1, [0, 2, 3]) if len(sys.argv) > 1 and sys.argv[1] == "q3_orders" else (0, [0, 1])))
PY
for f in presto-1_amd/_kcache/*.hip; do if head -3 $f | tr "\n" " " | grep -q "FJ_PF ${1:-1} #define FJ_NO_NULLS ${2:-1} #define FJ_CARRY ${3:-0}"; then (echo '#include <hip/hip_runtime.h>'; cat $f) > /tmp/fj.hip; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S /tmp/fj.hip -o /tmp/fj.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: fj_" | grep -E "Name|VGPRs:|Scratch|Occupancy"
awk '/^fj_probe[a-z_]*:/,/s_endpgm/' /tmp/fj.s > /tmp/fj_p.s
wc -l /tmp/fj_p.s
