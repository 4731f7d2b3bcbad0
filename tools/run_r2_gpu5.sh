# small-shard (one GPU's share of 8) and full-size steady-state timelines
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --docs 16100 --no-cpu-baseline --no-plateau > gpurun_out/small_bench.json 2> gpurun_out/small_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_small -- python $R/bench.py --docs 16100 --steps 10 --warmup 2 --burnin 40 --no-cpu-baseline --no-plateau --no-cold > $R/gpurun_out/prof_small.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full -- python $R/bench.py --steps 10 --warmup 2 --burnin 60 --no-cpu-baseline --no-plateau --no-cold > $R/gpurun_out/prof_full.log 2>&1
cd $R
for d in prof_small prof_full; do
  db=$(find gpurun_out/$d -name "*.db" | head -1)
  python tools/prof_summary.py $db > gpurun_out/${d}_summary.txt 2>&1
  python tools/prof_timeline.py $db 3 > gpurun_out/${d}_timeline.txt 2>&1
done
find gpurun_out/prof_small gpurun_out/prof_full -name "*.db" -size +20M -delete
