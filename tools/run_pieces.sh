# steady-state / cold-start sweep over the number of document pieces of the LDA E-step (TMVB_LDA_PIECES, TMVB_LDA_PIECE_FRACS)
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-plateau --no-cpu-baseline $DOCS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$DOCS $*', 'steady', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'estep', round(d['roofline']['estep_ms'],4), 'cold', round(d['cold_start']['value'],1), round(d['cold_start']['estep_ms'],4))" >> gpurun_out/pieces_steady.txt; }
DOCS=""
run TMVB_LDA_PIECES=2
run TMVB_LDA_PIECES=3
run TMVB_LDA_PIECES=4
run TMVB_LDA_PIECES=4
run TMVB_LDA_PIECES=3
DOCS="--docs 64400"
run TMVB_LDA_PIECES=1
run TMVB_LDA_PIECES=2
run TMVB_LDA_PIECES=3
DOCS="--docs 32200"
run TMVB_LDA_PIECES=1
run TMVB_LDA_PIECES=2
