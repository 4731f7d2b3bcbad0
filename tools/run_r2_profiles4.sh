# round-2 measurement pass, part 4 (after the cooperative token gather + lambda in LDS of the lane-per-document CTM kernel):
# model lines, CTM phase cycles, CTM kernel trace and counters (VALU, wave-cycle split, vector L1 / L2)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
timeout 900 python tools/model_bench.py > $O/models_bench.jsonl 2> $O/models_bench.err
TMVB_CTM_PROF=1 ITERS=4 timeout 300 python tools/ctm_probe.py > $O/ctm_phase_cycles.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctm -- python $R/tools/model_bench.py ctm > $O/prof_ctm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_valu_ctm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/pmc_wave_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_wave_ctm.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --kernel-trace -d $O/pmc_tcp1_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_tcp1_ctm.log 2>&1
timeout 600 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum --kernel-trace -d $O/pmc_tcp2_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_tcp2_ctm.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $O/pmc_tcc_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_tcc_ctm.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_fetch_ctm.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_write_ctm.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_ctm -name "*.db" | head -1) > $O/prof_ctm_summary.txt 2>&1
for d in pmc_valu_ctm pmc_wave_ctm pmc_tcp1_ctm pmc_tcp2_ctm pmc_tcc_ctm; do python tools/counter_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}.txt 2>&1; done
python tools/pmc_summary.py $(find $O/pmc_fetch_ctm -name "*.db" | head -1) $(find $O/pmc_write_ctm -name "*.db" | head -1) --iters 8 > $O/ctm_pmc.txt 2>&1
find $O -name "*.db" -delete
du -sh $O
