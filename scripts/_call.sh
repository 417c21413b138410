cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r03/call6_tests.txt 2>&1
tail -3 gpurun_out/r03/call6_tests.txt
export SLIM_GPU_TRACE=1
( echo "## c4 ratings"; timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --ratings 2>&1 | grep -E "trace\]|^\{" | cut -c1-1500
  echo "## c4-0.1pct default batch"; timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c4-0.1pct 2>&1 | grep -E "trace\]|^\{" | cut -c1-1500
  echo "## c4-0.1pct whole matrix"; timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c4-0.1pct --scaling strong --batch 0 2>&1 | grep -E "trace\]|^\{" | cut -c1-1500
) > gpurun_out/r03/call6_bench.txt 2>&1
cut -c1-330 gpurun_out/r03/call6_bench.txt
timeout 1200 python scripts/admm_c5.py > gpurun_out/r03/call6_admm_c5.txt 2>&1
tail -25 gpurun_out/r03/call6_admm_c5.txt
