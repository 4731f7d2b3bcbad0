#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm11; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_predict_gpu.py tests/test_comm_gpu.py tests/test_random_shapes_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
python - <<PY > $O/fctm.txt 2>&1
import sys, time
sys.path.insert(0, "$R")
import numpy as np, tmvb_amd
tm = tmvb_amd.pkg
pc = tm.syn_nsf()
for K in (56, 100):
    gm = tm.gpufCTM(pc, K)
    for it in range(6):
        t0 = time.perf_counter(); gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu(); gm.synchronize(); t1 = time.perf_counter()
    print("fCTM K=%d iteration %.2f ms" % (K, 1e3 * (t1 - t0)))
    gm.close()
PY
cat $O/fctm.txt | tail -3
