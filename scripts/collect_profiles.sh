#!/bin/bash
# Collect the judged artifacts of one bench configuration on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py ARGS`          -> kernel stats csv
#   2. one --pmc pass per counter (FETCH_SIZE, WRITE_SIZE) of one timed step of the same
#      configuration, plus the same counters on calibration launches of known size
#      (scripts/micro/gather_bw c) -- counters are collected in their own runs, with
#      --kernel-trace only
#   3. scripts/pmc_to_json.py -> an entry (keyed by configuration, seed and the hash of the
#      kernel sources) appended to profiles/pmc_traffic.json, which bench.py looks up
# usage: bash scripts/collect_profiles.sh TAG [bench.py args...]   (outputs: gpurun_out/TAG_*)
#   SKIP_STATS=1 skips step 1, SKIP_PMC=1 steps 2-3; STEPS/WARMUP set the stats run (default 2 / 0: the kernel's
#   average duration in the stats is then the average of the timed launches)
set -u
TAG=${1:-r02_c4}
shift || true
ARGS="$*"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
[ -x $R/scripts/micro/gather_bw ] || hipcc -O3 --offload-arch=gfx950 -o $R/scripts/micro/gather_bw $R/scripts/micro/gather_bw.hip
cd /tmp && export TMPDIR=/tmp
if [ -z "${SKIP_STATS:-}" ]; then
  export SLIM_GPU_TRACE=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python $R/bench.py --steps ${STEPS:-2} --warmup ${WARMUP:-0} --no-item-space $ARGS > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
  unset SLIM_GPU_TRACE
fi
for c in ${SKIP_PMC:+} $( [ -z "${SKIP_PMC:-}" ] && echo FETCH_SIZE WRITE_SIZE ); do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o run -- python $R/bench.py --warmup 0 --steps 1 --cpu-seconds 0 $ARGS > $O/${TAG}_pmc_$c.json 2> $O/${TAG}_pmc_$c.err
  [ -f $O/cal_$c/cal_counter_collection.csv ] || rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/cal_$c -o cal -- $R/scripts/micro/gather_bw c > /dev/null 2>&1
done
cd $R
[ -n "${SKIP_PMC:-}" ] || python scripts/pmc_to_json.py $O/${TAG}_pmc_FETCH_SIZE/run_counter_collection.csv $O/${TAG}_pmc_WRITE_SIZE/run_counter_collection.csv \
  $O/cal_FETCH_SIZE/cal_counter_collection.csv $O/cal_WRITE_SIZE/cal_counter_collection.csv $O/${TAG}_pmc_FETCH_SIZE.json > $O/${TAG}_pmc_entry.json
# the item_space_step launch of the same passes (default workload only): its own entry
if [ -z "${SKIP_PMC:-}" ] && grep -q '"item_space_step": {"columns"' $O/${TAG}_pmc_FETCH_SIZE.json 2>/dev/null; then
  PMC_ITEM_SPACE=1 python scripts/pmc_to_json.py $O/${TAG}_pmc_FETCH_SIZE/run_counter_collection.csv $O/${TAG}_pmc_WRITE_SIZE/run_counter_collection.csv \
    $O/cal_FETCH_SIZE/cal_counter_collection.csv $O/cal_WRITE_SIZE/cal_counter_collection.csv $O/${TAG}_pmc_FETCH_SIZE.json > $O/${TAG}_pmc_entry_item_space.json
  if grep -q '"item_space_whole_matrix": {"columns"' $O/${TAG}_pmc_FETCH_SIZE.json 2>/dev/null; then
    PMC_ITEM_SPACE=whole python scripts/pmc_to_json.py $O/${TAG}_pmc_FETCH_SIZE/run_counter_collection.csv $O/${TAG}_pmc_WRITE_SIZE/run_counter_collection.csv \
      $O/cal_FETCH_SIZE/cal_counter_collection.csv $O/cal_WRITE_SIZE/cal_counter_collection.csv $O/${TAG}_pmc_FETCH_SIZE.json > $O/${TAG}_pmc_entry_item_space_whole.json
  fi
fi
cp profiles/pmc_traffic.json $O/pmc_traffic.json
grep trace $O/${TAG}_bench.err 2>/dev/null | cut -c1-400
cat $O/${TAG}_bench.json 2>/dev/null | tail -1 | cut -c1-1500
grep -E "cd_tile|cd_wave|cd_gram|Name" $O/${TAG}_stats/run_kernel_stats.csv 2>/dev/null | head -8
cat $O/${TAG}_pmc_entry.json
# keep the merge-back small: the per-dispatch traces are large
find $O -name "*kernel_trace.csv" -size +20M -delete
