#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "2048 0" "400 0" "400 1" "256 1" "192 1"; do
  set -- $cfg
  export TMVB_CTM_BATCH_MAX_LEN=$1 TMVB_CTM_DEBUG_DROP_LONG=$2
  tag=L$1_D$2
  rm -rf /tmp/prof_$tag
  ITERS=14 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -- python $R/tools/ctm_probe.py > $O/probe_$tag.txt 2>&1
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $R/tools/prof_window.py $db ctm_sigma_mu_kernel 1 > $O/window_$tag.txt 2>&1
  grep "^iter 13" $O/probe_$tag.txt; grep -v "^#" $O/window_$tag.txt | sort -k3 -n -r | head -4
done
