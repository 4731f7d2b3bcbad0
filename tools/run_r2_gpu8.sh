set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_small2 -- python $R/bench.py --docs 16100 --steps 10 --warmup 2 --burnin 40 --no-cpu-baseline --no-plateau --no-cold > $R/gpurun_out/prof_small2.log 2>&1
cd $R
db=$(find gpurun_out/prof_small2 -name "*.db" | head -1)
python tools/prof_timeline.py $db 3 > gpurun_out/prof_small2_timeline.txt 2>&1
find gpurun_out/prof_small2 -name "*.db" -size +20M -delete
