# round-2 measurement pass, part 3 (final state: after the LDA fork-wait fix and fCTM on the lane-per-document kernel):
# headline bench line, model lines, LDA kernel trace / timeline / FETCH+WRITE / VALU counters, fCTM trace, small shard, world-2 smoke
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python tools/model_bench.py > $O/models_bench.jsonl 2> $O/models_bench.err
timeout 300 python bench.py --docs 16100 --no-cpu-baseline --no-plateau > $O/bench_small.json 2> $O/bench_small.err
TMVB_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --no-cpu-baseline --no-plateau > $O/bench_w2.json 2> $O/bench_w2.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lda -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-plateau --no-cold > $O/prof_lda.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_fctm -- python $R/tools/model_bench.py fctm > $O/prof_fctm.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 1 --burnin 20 --no-cpu-baseline --no-plateau --no-cold > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 1 --burnin 20 --no-cpu-baseline --no-plateau --no-cold > $O/pmc_write.log 2>&1
TMVB_LDA_PIECES=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu_lda -- python $R/bench.py --steps 6 --warmup 2 --burnin 40 --no-cpu-baseline --no-plateau --no-cold > $O/pmc_valu_lda.log 2>&1
cd $R
for d in prof_lda prof_fctm; do python tools/prof_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}_summary.txt 2>&1; done
python tools/prof_timeline.py $(find $O/prof_lda -name "*.db" | head -1) 3 > $O/prof_lda_timeline.txt 2>&1
python tools/prof_window.py $(find $O/prof_lda -name "*.db" | head -1) beta_norm_kernel 3 > $O/prof_lda_window.txt 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) --iters 26 --json $O/lda_pmc.json > $O/lda_pmc.txt 2>&1
python tools/counter_summary.py $(find $O/pmc_valu_lda -name "*.db" | head -1) > $O/pmc_valu_lda.txt 2>&1
find $O -name "*.db" -size +8M -delete
du -sh $O
