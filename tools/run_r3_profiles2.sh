# round-3 final measurement pass (after the CTM / CTPF changes): kernel traces + timelines, HBM / VALU counters, model lines, soak;
# `python bench.py` runs afterwards with the fresh PMC json in profiles/ (tools/run_r3_final.sh)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p2
mkdir -p $O
timeout 300 python bench.py --docs 16100 --steps 50 --no-cpu-baseline --no-plateau --no-other-configs > $O/lda_k50_bench_16100docs.json 2> /dev/null
for m in lda100 ctm ctpf flda fctm ctm100; do timeout 900 python tools/model_bench.py $m; done > $O/models_bench.jsonl 2> $O/models_bench.err     # one process per model (streams and hardware queues)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lda -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-plateau --no-cold --no-other-configs > $O/prof_lda.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lda100 -- python $R/tools/model_bench.py lda100 > $O/prof_lda100.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctm -- python $R/tools/model_bench.py ctm > $O/prof_ctm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctpf -- python $R/tools/ctpf_probe.py > $O/prof_ctpf.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ctm100 -- python $R/tools/model_bench.py ctm100 > $O/prof_ctm100.log 2>&1
# counters, each in its own pass, kernel trace only
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 1 --burnin 20 --no-cpu-baseline --no-plateau --no-cold --no-other-configs > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 1 --burnin 20 --no-cpu-baseline --no-plateau --no-cold --no-other-configs > $O/pmc_write.log 2>&1
TMVB_LDA_PIECES=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_lda -- python $R/bench.py --steps 6 --warmup 2 --burnin 40 --no-cpu-baseline --no-plateau --no-cold --no-other-configs > $O/pmc_valu_lda.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctpf -- python $R/tools/ctpf_probe.py > $O/pmc_valu_ctpf.log 2>&1
K=50 ITERS=12 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctm -- python $R/tools/ctm_probe.py > $O/pmc_valu_ctm.log 2>&1
K=100 ITERS=8 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_ctm100 -- python $R/tools/ctm_probe.py > $O/pmc_valu_ctm100.log 2>&1
cd $R
for d in prof_lda prof_lda100 prof_ctm prof_ctpf prof_ctm100; do
  python tools/prof_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}_summary.txt 2>&1
done
python tools/prof_timeline.py $(find $O/prof_lda -name "*.db" | head -1) 3 > $O/prof_lda_timeline.txt 2>&1
python tools/prof_timeline.py $(find $O/prof_lda100 -name "*.db" | head -1) 3 > $O/prof_lda100_timeline.txt 2>&1
python tools/prof_window.py $(find $O/prof_ctpf -name "*.db" | head -1) ctpf_rates_kernel 3 > $O/prof_ctpf_window.txt 2>&1
python tools/prof_window.py $(find $O/prof_ctm -name "*.db" | head -1) ctm_estep_batch_kernel 2 > $O/prof_ctm_window.txt 2>&1
python tools/prof_window.py $(find $O/prof_ctm100 -name "*.db" | head -1) ctm_estep_generic_kernel 2 > $O/prof_ctm100_window.txt 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) --iters 26 --json $O/lda_pmc.json > $O/lda_pmc.txt 2>&1
for d in pmc_valu_lda pmc_valu_ctpf pmc_valu_ctm pmc_valu_ctm100; do
  python tools/counter_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}.txt 2>&1
done
timeout 900 python tools/soak.py > $O/soak.txt 2>&1
find $O -name "*.db" -delete
du -sh $O
