cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export SLIM_GPU_TRACE=1
timeout 3300 python scripts/c5_grid.py --workload c5 --pairs 45 > gpurun_out/r03/c5_grid_45pairs.txt 2> gpurun_out/r03/c5_grid_45pairs.err
grep -E "^\{" gpurun_out/r03/c5_grid_45pairs.txt | tail -5 | cut -c1-300
