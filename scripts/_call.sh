cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
SKIP_STATS=1 timeout 2400 bash scripts/collect_profiles.sh r03_c4whole --scaling strong --batch 0 > gpurun_out/r03/collect_c4whole.log 2>&1
tail -14 gpurun_out/r03/collect_c4whole.log | cut -c1-500
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
