#!/bin/bash
# The five deliberately WRONG libraries of tests/test_mutants_gpu.py (negative controls of the parity suite), each = the shipped objects with one
# translation unit recompiled under a -DTMVB_MUTANT_* flag (csrc/tmvb_internal.h lists them):
#   topicmodelsvb.jl_amd/libtmvb_hip_mut_lda_eps.so     epsilon dropped from LDA's phi / gamma            (src/LDA.jl:152, :145)
#   topicmodelsvb.jl_amd/libtmvb_hip_mut_ctpf_bet.so    log bet for log vav in CTPF's xi                   (src/CTPF.jl:336 vs src/gpuCTPF.jl:624)
#   topicmodelsvb.jl_amd/libtmvb_hip_mut_ctm_mu.so      update_sigma! centred on the new mu                (src/CTM.jl:207-208, quirk Q2)
#   topicmodelsvb.jl_amd/libtmvb_hip_mut_flda_eps.so    the filtered models' log(beta + eps) without epsilon (src/fLDA.jl:184, :191)
#   topicmodelsvb.jl_amd/libtmvb_hip_mut_fctm_order.so  fCTM's sweep in CTM's order, vsq before lambda     (src/fCTM.jl:239-240)
# Needs the shipped build first (python -c "import __graft_entry__ as g; g.build()").  ~4 minutes; in parallel (the two builds of tmvb_ctm.hip one after the other: they share a temporary).
cd "$(dirname "$0")/.." || exit 1
tools/build_variant.sh mut_lda_eps tmvb_lda.hip -DTMVB_MUTANT_LDA_NO_EPS=1 &
tools/build_variant.sh mut_ctpf_bet tmvb_ctpf.hip -DTMVB_MUTANT_CTPF_LOG_BET=1 &
tools/build_variant.sh mut_ctm_mu tmvb_ctm.hip -DTMVB_MUTANT_CTM_SIGMA_NEW_MU=1 &
tools/build_variant.sh mut_flda_eps tmvb_flda.hip -DTMVB_MUTANT_FLDA_NO_EPS=1 &
wait
tools/build_variant.sh mut_fctm_order tmvb_ctm.hip -DTMVB_MUTANT_FCTM_VSQ_FIRST=1 &
wait
ls -la topicmodelsvb.jl_amd/libtmvb_hip_mut_*.so
