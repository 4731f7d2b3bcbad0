cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-plateau --no-cpu-baseline --steps 40 --docs 16100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16100 docs $*', 'steady', round(d['value'],1), round(d['ms_per_step'],4), 'cold', round(d['cold_start']['value'],1))" >> gpurun_out/pieces_small.txt; }
run TMVB_LDA_PIECES=2 TMVB_LDA_PIECE_FRACS=0.6
run TMVB_LDA_PIECES=2 TMVB_LDA_PIECE_FRACS=0.8
run TMVB_LDA_PIECES=3
run TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=0.5,0.8
run TMVB_LDA_PIECES=4
