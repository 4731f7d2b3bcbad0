# steady-state / cold-start sweep over the number of document pieces of the LDA E-step (TMVB_LDA_PIECES, TMVB_LDA_PIECE_FRACS)
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-plateau --no-cpu-baseline $DOCS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$DOCS $*', 'steady', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'estep', round(d['roofline']['estep_ms'],4), 'cold', round(d['cold_start']['value'],1), round(d['cold_start']['estep_ms'],4))" >> gpurun_out/pieces_steady.txt; }
DOCS=""
run TMVB_LDA_PIECES=3
run TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.45,0.80
run TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.50,0.83
run TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.55,0.87
run TMVB_LDA_PIECES=4 TMVB_LDA_PIECE_FRACS=0.40,0.70,0.90
run TMVB_LDA_PIECES=4 TMVB_LDA_PIECE_FRACS=0.45,0.75,0.92
run TMVB_LDA_PIECES=5 TMVB_LDA_PIECE_FRACS=0.40,0.68,0.85,0.95
run TMVB_LDA_PIECES=2 TMVB_LDA_PIECE_FRACS=0.75
