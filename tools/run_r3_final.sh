#!/bin/bash
# full -m gpu suite, smoke() and the default bench.py line (profiles/r3_lda_k50_bench.json)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > $O/t_all.txt 2>&1; echo "rc=$?" >> $O/t_all.txt
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt
( time timeout 1200 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -5 $O/t_all.txt; tail -3 $O/smoke.txt; cat $O/bench.json | cut -c1-1500
