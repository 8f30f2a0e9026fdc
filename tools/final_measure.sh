#!/bin/bash
# the round's final measurement set on the GPU box: default bench line (paged and PCIe-inclusive sub-lines included), rocprofv3 kernel statistics and the two PMC
# passes (separate runs, kernel trace only) of the default command; tools/collect_profiles.py turns the output into profiles/
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo bench done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only q3,q1,cfg2,sub > $O/stats.log 2>&1 || exit 1
echo stats done
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only q3,q1,cfg2,sub > $O/pmc_fetch.log 2>&1 || exit 1
echo fetch done
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only q3,q1,cfg2,sub > $O/pmc_write.log 2>&1 || exit 1
echo write done
# the group-by table's two BenchmarkGroupByHash shapes, one shape per pass (the kernel name is the same for both)
for shape in 10M_3M 100M_40M; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_gbh_${shape}_$c -o gbh -- python3 $R/tools/exp_sub.py group_by_hash_$shape --steps 2 --warmup 1 > $O/pmc_gbh_${shape}_$c.log 2>&1 || exit 1
  done
done
echo gbh done
ls $O
