cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-plateau --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'steady', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'estep', round(d['roofline']['estep_ms'],4))" >> gpurun_out/pieces2.txt; }
run TMVB_LDA_PIECES=3
run TMVB_LDA_PIECES=2
run TMVB_LDA_PIECES=2 TMVB_LDA_PIECE_FRACS=0.60
run TMVB_LDA_PIECES=2 TMVB_LDA_PIECE_FRACS=0.70
run TMVB_LDA_PIECES=4
run TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.40,0.75
run TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.45,0.80
run TMVB_LDA_PIECES=3
