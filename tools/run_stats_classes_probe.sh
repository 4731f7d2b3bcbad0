# statistics pass with document-id classes (tmvb_build_inv_index): duration of termstats_recompute_kernel on the whole SYN-NSF corpus
# (CTM config) and LDA bench rate for a few class sizes / posting thresholds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cls
mkdir -p $O
cd /tmp
for cfg in "8192 1024" "8192 256" "4096 256" "4096 64" "2048 128" "16384 512"; do
  set -- $cfg
  export TMVB_CLASS_DOCS=$1 TMVB_CLASS_MIN_POSTINGS=$2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$1_$2 -- python $R/tools/model_bench.py ctm > $O/tr_$1_$2.log 2>&1
  echo "== class_docs $1 min_postings $2" >> $O/sweep.txt
  python $R/tools/prof_summary.py $(find $O/tr_$1_$2 -name "*.db" | head -1) 2>&1 | grep -E "termstats" | head -2 | cut -c1-120 >> $O/sweep.txt
  timeout 300 python $R/bench.py --no-cpu-baseline --no-plateau --no-cold 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   lda', o['value'], o['ms_per_step'])" >> $O/sweep.txt
done
find $O -name "*.db" -delete
