#!/bin/bash
# Collect the judged artifacts of the default bench on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py`           -> kernel stats csv
#   2. one --pmc pass per counter (FETCH_SIZE, WRITE_SIZE) of the timed step, plus the same
#      counters on calibration launches of known size (scripts/micro/gather_bw c)
#   3. scripts/pmc_to_json.py -> gpurun_out/pmc_entry.json (copied into profiles/pmc_traffic.json)
# usage: bash scripts/collect_profiles.sh TAG        (outputs under gpurun_out/TAG_*)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
[ -x $R/scripts/micro/gather_bw ] || hipcc -O3 --offload-arch=gfx950 -o $R/scripts/micro/gather_bw $R/scripts/micro/gather_bw.hip
cd /tmp && export TMPDIR=/tmp
export SLIM_GPU_TRACE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o c4 -- python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
unset SLIM_GPU_TRACE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_$c -o c4 -- python $R/bench.py --warmup 0 --steps 1 --cpu-seconds 0 > $O/${TAG}_pmc_$c.json 2> $O/${TAG}_pmc_$c.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_cal_$c -o cal -- $R/scripts/micro/gather_bw c > /dev/null 2>&1
done
cd $R
python scripts/pmc_to_json.py $O/${TAG}_pmc_FETCH_SIZE/c4_counter_collection.csv $O/${TAG}_pmc_WRITE_SIZE/c4_counter_collection.csv \
  $O/${TAG}_cal_FETCH_SIZE/cal_counter_collection.csv $O/${TAG}_cal_WRITE_SIZE/cal_counter_collection.csv $O/${TAG}_pmc_FETCH_SIZE.json > $O/${TAG}_pmc_entry.json
grep trace $O/${TAG}_bench.err | cut -c1-400
python scripts/benchline.py < $O/${TAG}_bench.json
cat $O/${TAG}_bench.json
grep -E "cd_tile|Name" $O/${TAG}_stats/c4_kernel_stats.csv | head -5
cat $O/${TAG}_pmc_entry.json
# keep the merge-back small: the per-dispatch traces are large
find $O -name "*kernel_trace.csv" -size +20M -delete
