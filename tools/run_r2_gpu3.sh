set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_comm_gpu.py tests/test_lda_gpu.py tests/test_ctpf_gpu.py -x -q -m gpu > gpurun_out/t_comm.txt 2>&1; echo "rc=$?" >> gpurun_out/t_comm.txt
python -c "
import sys; sys.path.insert(0,'.')
import bench, os; print(bench.usable_cpus()); print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')
" > gpurun_out/cpus.txt 2>&1
nproc >> gpurun_out/cpus.txt; lscpu | head -20 >> gpurun_out/cpus.txt
