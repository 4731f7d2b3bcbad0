# usage: bash tools/run_gpu_subset.sh "<pytest args>"  (runs on the GPU box)
set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest $TESTS -q -m gpu -x > gpurun_out/t_subset.txt 2>&1; echo "rc=$?" >> gpurun_out/t_subset.txt
