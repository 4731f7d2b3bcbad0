#!/bin/bash
# ONE parameterised recipe for everything that runs on the GPU box (round 4; replaces the per-experiment run_r3_*.sh notebook).
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_recipe.sh <tag> <step> [<step> ...]'
#
# Results land in gpurun_out/<tag>/ (merged back by gpurun).  Steps:
#   tests            full `pytest -m gpu` suite              -> t_all.txt
#   tests:<expr>     `pytest -m gpu -k <expr>`               -> t_<n>.txt
#   file:<path>      `pytest -m gpu <path>`                  -> t_<n>.txt
#   smoke            __graft_entry__.smoke()                 -> smoke.txt
#   bench            default bench.py line                   -> bench.json / bench.err
#   bench:<args>     bench.py with extra args (use , for spaces)  -> bench_<n>.json
#   model:<names>    tools/model_bench.py <names> (comma separated) -> models.jsonl
#   trace:<name>:<cmd>       rocprofv3 --kernel-trace --stats -- python <cmd>  -> trace_<name>_summary.txt (+ _window.txt with
#                            env:TRACE_MARKER=<kernel substring>, + _timeline.txt with env:TRACE_TIMELINE=1)
#   pmc:<name>:<counters>:<cmd>   rocprofv3 --pmc <counters> --kernel-trace -- python <cmd> (counters in their OWN pass, never with
#                            --stats / other trace domains)  -> pmc_<name>.txt (tools/counter_summary.py); env:PMC_KEEP_DB=1 keeps the db
#   pmcbytes:<model>         HBM bytes per iteration of lda50 | lda100 | ctm | ctpf (two --pmc passes) -> <model>_pmc.json / .txt
#   py:<script>[,args]       python <script> args            -> py_<n>.txt
#   exe:<binary>[,args]      a prebuilt probe binary (tools/probes/...)  -> exe_<n>.txt
#   env:<VAR=VALUE>  export for the steps that follow
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" || exit 1
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
n=0
for step in "$@"; do
  n=$((n + 1))
  kind=${step%%:*}; arg=${step#*:}; [ "$kind" = "$step" ] && arg=""
  echo "=== [$n] $step" | tee -a "$O/recipe.log"
  case $kind in
    env)    export "$arg" ;;
    tests)  if [ -z "$arg" ]; then ( time timeout 2400 python -m pytest tests -q -m gpu -x ) > "$O/t_all.txt" 2>&1; echo "rc=$?" >> "$O/t_all.txt"; tail -4 "$O/t_all.txt"
            else ( time timeout 2400 python -m pytest tests -q -m gpu -x -s -k "$arg" ) > "$O/t_$n.txt" 2>&1; echo "rc=$?" >> "$O/t_$n.txt"; tail -25 "$O/t_$n.txt"; fi ;;
    file)   ( time timeout 2400 python -m pytest -q -m gpu -x -s ${arg//,/ } ) > "$O/t_$n.txt" 2>&1; echo "rc=$?" >> "$O/t_$n.txt"; tail -40 "$O/t_$n.txt" ;;
    smoke)  timeout 600 python __graft_entry__.py smoke > "$O/smoke.txt" 2>&1; echo "rc=$?" >> "$O/smoke.txt"; tail -4 "$O/smoke.txt" ;;
    bench)  if [ -z "$arg" ]; then f=bench; else f=bench_$n; fi
            ( time timeout 1500 python bench.py ${arg//,/ } ) > "$O/$f.json" 2> "$O/$f.err"; echo "rc=$?" >> "$O/$f.err"; tail -12 "$O/$f.err"; cut -c1-1800 "$O/$f.json" ;;
    model)  ( time timeout 1500 python tools/model_bench.py ${arg//,/ } ) >> "$O/models.jsonl" 2> "$O/models_$n.err"; echo "rc=$?" >> "$O/models_$n.err"; tail -8 "$O/models_$n.err"; tail -n 3 "$O/models.jsonl" | cut -c1-900 ;;
    trace)  name=${arg%%:*}; cmd=${arg#*:}                       # trace:<name>:<python command, commas for spaces>
            ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$O/trace_$name" -- python $R/${cmd//,/ } ) > "$O/trace_$name.log" 2>&1; echo "rc=$?" >> "$O/trace_$name.log"
            db=$(find "$O/trace_$name" -name "*.db" | head -1)
            python tools/prof_summary.py "$db" > "$O/trace_${name}_summary.txt" 2>&1; head -30 "$O/trace_${name}_summary.txt"
            [ -n "$TRACE_MARKER" ] && python tools/prof_window.py "$db" "$TRACE_MARKER" ${TRACE_BACK:-3} > "$O/trace_${name}_window.txt" 2>&1
            [ -n "$TRACE_TIMELINE" ] && python tools/prof_timeline.py "$db" 3 > "$O/trace_${name}_timeline.txt" 2>&1
            [ -z "$KEEP_DB" ] && find "$O/trace_$name" -name "*.db" -delete ;;
    pmc)    name=${arg%%:*}; rest=${arg#*:}; ctr=${rest%%:*}; cmd=${rest#*:}    # pmc:<name>:<counters,>:<python command,>
            ( cd /tmp && timeout 900 rocprofv3 --pmc ${ctr//,/ } --kernel-trace -d "$O/pmc_$name" -- python $R/${cmd//,/ } ) > "$O/pmc_$name.log" 2>&1; echo "rc=$?" >> "$O/pmc_$name.log"
            db=$(find "$O/pmc_$name" -name "*.db" | head -1)
            python tools/counter_summary.py "$db" > "$O/pmc_$name.txt" 2>&1; head -30 "$O/pmc_$name.txt"
            [ -z "$KEEP_DB" ] && [ -z "$PMC_KEEP_DB" ] && find "$O/pmc_$name" -name "*.db" -delete ;;
    pmcbytes) # pmcbytes:<lda50|lda100|ctm|ctpf>: FETCH_SIZE and WRITE_SIZE in separate passes over tools/pmc_window.py <model> 20 6,
            # summarised per kernel and per iteration with the kernel-source hash stamped in -> <model>_pmc.json / .txt (copy to profiles/r5_<model>_pmc.*)
            for c in FETCH_SIZE WRITE_SIZE; do
              ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -d "$O/pmcb_${arg}_$c" -- python $R/tools/pmc_window.py $arg 20 6 ) > "$O/pmcb_${arg}_$c.log" 2>&1; echo "rc=$?" >> "$O/pmcb_${arg}_$c.log"
            done
            ( cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d "$O/pmcb_${arg}_VALU" -- python $R/tools/pmc_window.py $arg 20 6 ) > "$O/pmcb_${arg}_VALU.log" 2>&1; echo "rc=$?" >> "$O/pmcb_${arg}_VALU.log"
            python tools/pmc_summary.py $(find "$O/pmcb_${arg}_FETCH_SIZE" -name "*.db" | head -1) $(find "$O/pmcb_${arg}_WRITE_SIZE" -name "*.db" | head -1) --valu $(find "$O/pmcb_${arg}_VALU" -name "*.db" | head -1) --iters 26 --model $(echo $arg | sed "s/lda.*/lda/") --json "$O/${arg}_pmc.json" > "$O/${arg}_pmc.txt" 2>&1
            head -12 "$O/${arg}_pmc.txt"; grep -A8 "VALU pass" "$O/${arg}_pmc.txt"; find "$O" -path "*pmcb_${arg}_*" -name "*.db" -delete ;;
    exe)    ( time timeout 900 ${arg//,/ } ) > "$O/exe_$n.txt" 2>&1; echo "rc=$?" >> "$O/exe_$n.txt"; tail -40 "$O/exe_$n.txt" ;;
    py)     ( time timeout 1500 python ${arg//,/ } ) > "$O/py_$n.txt" 2>&1; echo "rc=$?" >> "$O/py_$n.txt"; tail -40 "$O/py_$n.txt" ;;
    *)      echo "unknown step $step" ;;
  esac
done
