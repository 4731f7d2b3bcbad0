# CTM lane-per-document kernel: where do a wave's cycles go?  wave-cycle split (wait / issue-stall / active), instruction-cache
# counters and the per-type instruction counts, each in its own rocprofv3 pass (counters only, --kernel-trace).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ctm_stall
mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQ_WAIT|SQ_INST_CYCLES|SQ_ACTIVE_INST|SQ_WAVE_CYCLES|SQ_INSTS_|SQC_" > $O/counter_list.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/pmc_a -- python $R/tools/model_bench.py ctm > $O/pmc_a.log 2>&1
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --kernel-trace -d $O/pmc_b -- python $R/tools/model_bench.py ctm > $O/pmc_b.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES --kernel-trace -d $O/pmc_c -- python $R/tools/model_bench.py ctm > $O/pmc_c.log 2>&1
cd $R
for d in pmc_a pmc_b pmc_c; do python tools/counter_summary.py $(find $O/$d -name "*.db" | head -1) > $O/${d}.txt 2>&1; done
find $O -name "*.db" -size +8M -delete
du -sh $O
