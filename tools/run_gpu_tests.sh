# full -m gpu suite + smoke on the GPU box (the .so must have been built in-tree before the snapshot is taken)
set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/t_all.txt 2>&1; echo "rc=$?" >> gpurun_out/t_all.txt
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; echo "rc=$?" >> gpurun_out/smoke.txt
