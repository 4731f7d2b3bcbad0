#!/bin/bash
# piece-count / cut-point sweep of the pipelined LDA E-step with the grid-tile kernel (steady state, SYN-NSF K=50)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-plateau --no-other-configs --no-cold 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$tag', '%.1f it/s  %.4f ms  estep %.4f' % (r['value'], r['ms_per_step'], r['roofline']['estep_ms']))" >> $O/pieces.txt; }
run default X=1
run p2 TMVB_LDA_PIECES=2
run p3_a TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.45,0.85
run p3_b TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.5,0.88
run p3_c TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.4,0.75
run p4 TMVB_LDA_PIECES=4
run p4_a TMVB_LDA_PIECES=4 TMVB_LDA_PIECE_FRACS=0.35,0.65,0.88
run p4_b TMVB_LDA_PIECES=4 TMVB_LDA_PIECE_FRACS=0.4,0.7,0.92
run p5 TMVB_LDA_PIECES=5 TMVB_LDA_PIECE_FRACS=0.3,0.55,0.78,0.93
run p6 TMVB_LDA_PIECES=6
cat $O/pieces.txt
