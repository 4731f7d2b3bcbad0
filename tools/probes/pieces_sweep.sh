#!/bin/bash
# it/s of bench.py for shard sizes x piece counts (tuning of lda_piece_count)
for D in ${DOCS:-32200 64400}; do for P in ${PIECES:-1 2 3 4}; do
  echo -n "docs=$D pieces=$P  "
  if [ "$P" = auto ]; then unset TMVB_LDA_PIECES; else export TMVB_LDA_PIECES=$P; fi
  python bench.py --docs $D --steps 50 --warmup 5 --no-cpu-baseline --no-plateau 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
done; done
