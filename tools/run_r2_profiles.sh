# round-2 measurement pass: new sharded tests, tolerance figures, model lines, kernel traces, counters
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
mkdir -p $O
timeout 900 python -m pytest tests/test_comm_gpu.py -q -m gpu > $O/t_comm.txt 2>&1; echo "rc=$?" >> $O/t_comm.txt
timeout 300 python tests/measure_tolerances.py > $O/tolerances.txt 2>&1
timeout 900 python tools/model_bench.py > $O/models_bench.jsonl 2> $O/models_bench.err
rocprofv3 -L > $O/counters_avail.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lda -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-plateau --no-cold > $O/prof_lda.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctm -- python $R/tools/model_bench.py ctm > $O/prof_ctm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctpf -- python $R/tools/ctpf_probe.py > $O/prof_ctpf.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_flda -- python $R/tools/model_bench.py flda > $O/prof_flda.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_fctm -- python $R/tools/model_bench.py fctm > $O/prof_fctm.log 2>&1
# counters, each in its own pass, kernel trace only
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 1 --burnin 20 --no-cpu-baseline --no-plateau --no-cold > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 1 --burnin 20 --no-cpu-baseline --no-plateau --no-cold > $O/pmc_write.log 2>&1
TMVB_LDA_PIECES=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_lda -- python $R/bench.py --steps 6 --warmup 2 --burnin 40 --no-cpu-baseline --no-plateau --no-cold > $O/pmc_valu_lda.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_valu_ctm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --kernel-trace -d $O/pmc_mfma_ctm -- python $R/tools/model_bench.py ctm > $O/pmc_mfma_ctm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctpf -- python $R/tools/ctpf_probe.py > $O/pmc_valu_ctpf.log 2>&1
cd $R
for d in prof_lda prof_ctm prof_ctpf prof_flda prof_fctm; do
  db=$(find $O/$d -name "*.db" | head -1)
  python tools/prof_summary.py $db > $O/${d}_summary.txt 2>&1
done
python tools/prof_timeline.py $(find $O/prof_lda -name "*.db" | head -1) 3 > $O/prof_lda_timeline.txt 2>&1
python tools/prof_window.py $(find $O/prof_ctpf -name "*.db" | head -1) ctpf_rates_kernel 3 > $O/prof_ctpf_window.txt 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) --iters 26 --json $O/lda_pmc.json > $O/lda_pmc.txt 2>&1
for d in pmc_valu_lda pmc_valu_ctm pmc_mfma_ctm pmc_valu_ctpf; do
  python tools/counter_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}.txt 2>&1
done
find $O -name "*.db" -size +8M -delete
du -sh $O
