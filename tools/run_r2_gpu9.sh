set -x
cd $GRAFT_REPO_ROOT
for f in 0 4; do TMVB_DEBUG_FLAGS=$f python bench.py --docs 16100 --no-cpu-baseline --no-plateau --no-cold > gpurun_out/small_prio$f.json 2> gpurun_out/small_prio$f.err; done
for f in 0 4; do TMVB_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --no-plateau --no-cold > gpurun_out/full_prio$f.json 2> gpurun_out/full_prio$f.err; done
for kb in 64 24 16 8; do echo "== TMVB_CTM_MAX_TILE_KB=$kb"; TMVB_CTM_MAX_TILE_KB=$kb M=32000 ITERS=4 python tools/ctm_probe.py; done > gpurun_out/ctm_tilecap.txt 2>&1
