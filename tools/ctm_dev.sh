#!/bin/bash
# developer loop of the batched CTM kernel: ISA statistics + in-flight SMEM check, build, GPU probe + CTM tests
cd /root/repo/topicmodelsvb.jl_amd/csrc || exit 1
mkdir -p /tmp/ctmb
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed -I../../include -save-temps=obj -c tmvb_ctm.hip -o /tmp/ctmb/tmvb_ctm.o 2>&1 | grep -E "error" -A5 | head -20
bash /tmp/ctmb/stats.sh 2>/dev/null
python /root/repo/tools/check_smem_inflight.py /tmp/ctmb/tmvb_ctm-hip-amdgcn-amd-amdhsa-gfx950.s ctm_estep_batch || exit 1
cd /root/repo || exit 1
rm -rf topicmodelsvb.jl_amd/csrc/gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 3000 /usr/local/graft/bin/gpurun --timeout 1800 -- "ITERS=2 $EXTRA_ENV python tools/ctm_probe.py > gpurun_out/ctm_probe_batch.txt 2>&1; TMVB_CTM_PROF=1 ITERS=${ITERS:-3} $EXTRA_ENV python tools/ctm_probe.py >> gpurun_out/ctm_probe_batch.txt 2>&1; timeout 900 python -m pytest tests/test_ctm_gpu.py -q -m gpu -x > gpurun_out/t_ctm_batch.txt 2>&1" > /tmp/gpurun10.log 2>&1
tail -1 /tmp/gpurun10.log; cat gpurun_out/ctm_probe_batch.txt; tail -3 gpurun_out/t_ctm_batch.txt
