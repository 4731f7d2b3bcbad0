# round-2 measurement pass, part 2 (after the lane-per-document CTM kernel): headline bench line, model lines, CTM traces + counters
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python tools/model_bench.py > $O/models_bench.jsonl 2> $O/models_bench.err
TMVB_CTM_PROF=1 ITERS=4 timeout 300 python tools/ctm_probe.py > $O/ctm_phase_cycles.txt 2>&1
TMVB_CTM_BATCH=0 ITERS=3 timeout 300 python tools/ctm_probe.py > $O/ctm_wave_kernel.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctm -- python $R/tools/model_bench.py ctm > $O/prof_ctm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_valu_ctm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --kernel-trace -d $O/pmc_mfma_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_mfma_ctm.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_fetch_ctm.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_write_ctm.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_ctm -name "*.db" | head -1) > $O/prof_ctm_summary.txt 2>&1
for d in pmc_valu_ctm pmc_mfma_ctm; do python tools/counter_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}.txt 2>&1; done
python tools/pmc_summary.py $(find $O/pmc_fetch_ctm -name "*.db" | head -1) $(find $O/pmc_write_ctm -name "*.db" | head -1) --iters 8 > $O/ctm_pmc.txt 2>&1
find $O -name "*.db" -size +8M -delete
du -sh $O
