#!/bin/bash
# developer loop of the lane-per-document CTM kernel (csrc/tmvb_ctm_batch.h): build (the build runs the ISA check), instruction
# statistics of the KP = 52 instantiation from the saved ISA, then GPU probe (plain and with TMVB_CTM_PROF=1) + the CTM tests.
# Usage (from anywhere): tools/ctm_dev.sh        env: ITERS (probe iterations, default 3), EXTRA_ENV ("NAME=value ..." for the probe)
cd /root/repo || exit 1
python -c "import __graft_entry__ as g; g.build()" > /tmp/ctm_dev_build.log 2>&1 || { grep -A6 "error" /tmp/ctm_dev_build.log | head -40; echo BUILD FAILED; exit 1; }; tail -1 /tmp/ctm_dev_build.log
S=topicmodelsvb.jl_amd/build/tmvb_ctm-hip-amdgcn-amd-amdhsa-gfx950.s
if [ -f $S ]; then
  awk '/^_Z22ctm_estep_batch_kernelILi52ELb0ELb0E/{on=1} on&&/s_endpgm/{on=0} on' $S > /tmp/ctm_b52.s
  for pat in v_readlane v_writelane v_accvgpr_read v_accvgpr_write scratch_ s_load_dwordx16 v_pk_fma_f32 v_fma_f64 ds_read ds_write global_load s_waitcnt s_nop; do
    echo -n "$pat $(grep -c "$pat" /tmp/ctm_b52.s); "
  done; echo
  python tools/check_smem_inflight.py $S ctm_estep_batch | tail -1
fi
timeout 3000 /usr/local/graft/bin/gpurun --timeout 1800 -- "ITERS=${ITERS:-3} $EXTRA_ENV python tools/ctm_probe.py > gpurun_out/ctm_probe_batch.txt 2>&1; TMVB_CTM_PROF=1 ITERS=${ITERS:-3} $EXTRA_ENV python tools/ctm_probe.py >> gpurun_out/ctm_probe_batch.txt 2>&1; timeout 900 python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py -q -m gpu -x > gpurun_out/t_ctm_batch.txt 2>&1" > /tmp/gpurun_ctm_dev.log 2>&1
tail -1 /tmp/gpurun_ctm_dev.log; cat gpurun_out/ctm_probe_batch.txt; tail -3 gpurun_out/t_ctm_batch.txt
