set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lda -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-plateau --no-cold > $O/prof_lda.log 2>&1
cd $R
python tools/prof_window.py $(find $O/prof_lda -name "*.db" | head -1) beta_norm_kernel 3 > $O/window.txt 2>&1
find $O -name "*.db" -size +8M -delete
