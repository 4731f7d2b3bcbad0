set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ctpf_r2 -- python $R/tools/ctpf_probe.py > $R/gpurun_out/prof_ctpf_r2.log 2>&1
cd $R
db=$(find gpurun_out/prof_ctpf_r2 -name "*.db" | head -1)
python tools/prof_summary.py $db > gpurun_out/prof_ctpf_r2_summary.txt 2>&1
python tools/prof_window.py $db ctpf_rates_kernel 3 > gpurun_out/prof_ctpf_r2_timeline.txt 2>&1
