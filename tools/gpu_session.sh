#!/bin/bash
# One GPU-box session through gpurun, by recipe (replaces the per-session r2_*.sh / r3_*.sh one-offs of earlier rounds; what they
# ran is in the git history).  Run from the repo root:   gpurun --timeout S -- 'bash tools/gpu_session.sh <recipe> [args]'
# Everything lands under gpurun_out/ (merged back by gpurun); copy what is to be judged into profiles/.
#
#   suite [pytest args]            the GPU suite + smoke, slowest tests listed           -> gpurun_out/<tag>_suite.log
#   bench [bench args]             the driver's command, stdout line + detail record      -> gpurun_out/<tag>_bench_default.json / _detail.json
#   profiles <wl> [<wl> ...]       rocprofv3 kernel stats + PMC passes per workload (tools/gpu_prof.sh), stamped with the library hash
#   wave_states <probe> [args]     SQ wave-state counters of a probe binary (where do the waves of a kernel spend their cycles)
#   line_search_ab <wl> ...        bench --no-extras under line_search exact / exact-y / linear, one line each
#   two_ranks [bench args]         the driver's --gpus 2 command rehearsed on ONE GPU (two ranks share GPU 0 over gloo)
#   n_ranks <N> <head> [args]     the driver's --gpus N command on ONE GPU at a reduced shard (N ranks share GPU 0 over gloo)
#   n_ranks_launched <N> <head>   the same through `python -m torch.distributed.run ... bench.py --gpus N` (the driver's form: supervised workers)
#   env_ab <wl> <VAR> <a> <b> [n]  bench --no-extras under VAR=a / VAR=b alternating on one box   -> gpurun_out/<tag>_env_ab_<wl>_<VAR>.txt
#   env_list <wl> <VAR> <n> <v>... the same for any number of values ("-" = unset), exchange sites printed; bench args after "--"
#   kernel_rows <wl> <VAR> <val>   rocprofv3 kernel rows + launch gaps of bench --no-extras under VAR=val
#   probe <name> [args]            build tools/<name>.hip and run it                        -> gpurun_out/<tag>_<name>_<args>.txt
# TAG (environment, default r06) prefixes the outputs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r06}
cd $R; mkdir -p gpurun_out
recipe=$1; shift
case "$recipe" in
  suite)
    T0=$(date +%s)
    python -m pytest tests -q -m gpu --durations=20 "$@" > gpurun_out/${TAG}_suite.log 2>&1
    echo "suite rc=$? in $(( $(date +%s) - T0 )) s" | tee gpurun_out/${TAG}_suite_summary.txt
    tail -30 gpurun_out/${TAG}_suite.log
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_suite_summary.txt ;;
  bench)
    T0=$(date +%s)
    python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/${TAG}_bench_detail.json "$@" 2>gpurun_out/${TAG}_bench_default.err > gpurun_out/${TAG}_bench_default.json
    echo "bench rc=$? wall $(( $(date +%s) - T0 )) s, line $(wc -c < gpurun_out/${TAG}_bench_default.json) bytes"
    cat gpurun_out/${TAG}_bench_default.json ;;
  profiles)
    python3 __graft_entry__.py || exit 1
    for WL in "$@"; do
      bash tools/gpu_prof.sh $TAG $WL 2>&1 | tail -30
      mkdir -p gpurun_out/${TAG}_profiles
      cp gpurun_out/pmc_traffic_$WL.json gpurun_out/${TAG}_rocprof_kernel_stats_$WL.csv gpurun_out/${TAG}_rocprof_kernel_stats_$WL.meta.json \
         gpurun_out/${TAG}_trace_gaps_$WL.txt gpurun_out/${TAG}_profiles/ 2>/dev/null
    done ;;
  wave_states)
    P=$1; shift
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/$P tools/$P.hip 2>/dev/null
    OUT=$R/gpurun_out/${TAG}_pmc_wave_states_${P}_$(echo "$*" | tr ' ' '_').txt
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
      --output-format csv -d $R/gpurun_out/pmc_ws_$P -o ws -- $R/tools/$P "$@" > $R/gpurun_out/pmc_ws_$P.log 2>&1
    tail -3 $R/gpurun_out/pmc_ws_$P.log
    python3 $R/tools/wave_states.py $R/gpurun_out/pmc_ws_$P | tee $OUT
    find $R/gpurun_out/pmc_ws_$P -name "*.csv" -size +1M -delete ;;
  line_search_ab)
    for wl in "$@"; do for ls in exact exact-y linear; do
      python bench.py --workload $wl --no-extras --steps 20 --warmup 5 --repeats 1 --line-search $ls 2>gpurun_out/${TAG}_ab_${wl}_$ls.err > gpurun_out/${TAG}_ab_${wl}_$ls.json
      python -c "
import json; d=json.load(open('gpurun_out/${TAG}_ab_${wl}_$ls.json')); c=d['config']
print('$wl $ls', round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms; X passes', round(c['x_passes_per_iteration'],3), 'trials', round(c['line_search_trials_per_iteration'],3))"
    done; done ;;
  two_ranks)
    T0=$(date +%s)
    LCX_BENCH_DEVICE=0 LCX_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 --detail-out gpurun_out/${TAG}_two_ranks_detail.json "$@" \
      2>gpurun_out/${TAG}_two_ranks.err > gpurun_out/${TAG}_two_ranks.json
    echo "rc=$? wall $(( $(date +%s) - T0 )) s, line $(wc -c < gpurun_out/${TAG}_two_ranks.json) bytes"; cat gpurun_out/${TAG}_two_ranks.json ;;
  n_ranks)
    # the driver's --gpus N command rehearsed on ONE GPU: N ranks share GPU 0, the library's exchange goes through its hook (gloo).
    #   n_ranks <N> <head workload | auto> [bench args]     e.g.  n_ranks 8 50000x16000x128:f32 --steps 5 --warmup 2
    N=$1; HEAD=$2; shift 2
    T0=$(date +%s)
    if [ "$HEAD" != auto ]; then export LCX_BENCH_HEAD=$HEAD; fi
    LCX_BENCH_DEVICE=0 LCX_BENCH_BACKEND=gloo timeout 1700 python bench.py --gpus $N --detail-out gpurun_out/${TAG}_${N}_ranks_detail.json "$@" \
      2>gpurun_out/${TAG}_${N}_ranks.err > gpurun_out/${TAG}_${N}_ranks.json
    echo "rc=$? wall $(( $(date +%s) - T0 )) s, line $(wc -c < gpurun_out/${TAG}_${N}_ranks.json) bytes" | tee gpurun_out/${TAG}_${N}_ranks_summary.txt
    cat gpurun_out/${TAG}_${N}_ranks.json; grep -v BENCH_DETAIL gpurun_out/${TAG}_${N}_ranks.err | tail -40 ;;
  n_ranks_launched)
    # the same rehearsal started THE WAY THE DRIVER STARTS a multi-GPU run: torch.distributed.run launches the N ranks, each supervises its worker
    #   n_ranks_launched <N> <head workload | auto> [bench args]
    N=$1; HEAD=$2; shift 2
    T0=$(date +%s)
    if [ "$HEAD" != auto ]; then export LCX_BENCH_HEAD=$HEAD; fi
    LCX_BENCH_DEVICE=0 LCX_BENCH_BACKEND=gloo timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29655 \
      bench.py --gpus $N --detail-out gpurun_out/${TAG}_${N}_ranks_launched_detail.json "$@" 2>gpurun_out/${TAG}_${N}_ranks_launched.err > gpurun_out/${TAG}_${N}_ranks_launched.json
    echo "rc=$? wall $(( $(date +%s) - T0 )) s, line $(wc -c < gpurun_out/${TAG}_${N}_ranks_launched.json) bytes" | tee gpurun_out/${TAG}_${N}_ranks_launched_summary.txt
    cat gpurun_out/${TAG}_${N}_ranks_launched.json; grep -v BENCH_DETAIL gpurun_out/${TAG}_${N}_ranks_launched.err | grep -v "^\[Gloo\]" | tail -30 ;;
  env_ab)
    # bench --workload <wl> --no-extras under VAR=a and VAR=b, alternating on ONE box:  env_ab <wl> <VAR> <a> <b> [rounds [bench args]]
    WL=$1; VAR=$2; A=$3; B=$4; ROUNDS=${5:-3}; shift 5 2>/dev/null || shift $#      # (anything after the round count goes to bench.py)
    python3 __graft_entry__.py || exit 1
    OUT=gpurun_out/${TAG}_env_ab_${WL}_${VAR}.txt; : > $OUT
    for r in $(seq $ROUNDS); do for val in $A $B; do
      env $VAR=$val python bench.py --workload $WL --no-extras --steps 30 --warmup 5 "$@" 2>gpurun_out/${TAG}_ab.err > gpurun_out/${TAG}_ab.json || { tail -5 gpurun_out/${TAG}_ab.err; exit 1; }
      python -c "
import json; d=json.load(open('gpurun_out/${TAG}_ab.json')); c=d['config']
print('$WL $VAR=$val', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms; X passes', round(c['x_passes_per_iteration'],3), 'trials', round(c['line_search_trials_per_iteration'],4), 'walks', c['windows']['ms_per_step_walk_min_median_max'], 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT
      grep -o '"final_TC": [-0-9.e+]*' gpurun_out/bench_detail.json | head -1 | tee -a $OUT
    done; done ;;
  env_list)
    # the same, any number of values, alternating on ONE box, with the exchange sites of the line (config.exchange_ms_per_iteration):
    #   env_list <wl> <VAR> <rounds> <val> [<val> ...] [-- bench args]   ("-" = VAR unset)  -> gpurun_out/<tag>_env_list_<wl>_<VAR>.txt
    WL=$1; VAR=$2; ROUNDS=$3; shift 3
    VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" = "--" ] && shift
    python3 __graft_entry__.py || exit 1
    OUT=gpurun_out/${TAG}_env_list_${WL}_${VAR}.txt; : > $OUT
    for r in $(seq $ROUNDS); do for val in "${VALS[@]}"; do
      if [ "$val" = "-" ]; then unset $VAR; else export $VAR=$val; fi
      timeout 300 python bench.py --workload $WL --no-extras --steps 30 --warmup 5 "$@" 2>gpurun_out/${TAG}_ab.err > gpurun_out/${TAG}_ab.json || { tail -5 gpurun_out/${TAG}_ab.err; exit 1; }
      python -c "
import json; d=json.load(open('gpurun_out/${TAG}_ab.json')); c=d['config']
x=c.get('exchange_ms_per_iteration') or {}
print('$WL $VAR=$val', ' '.join('$*'.split()), '|', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; X passes', round(c['x_passes_per_iteration'],3), 'trials', round(c['line_search_trials_per_iteration'],4), 'frac', round(d['roofline']['frac'],4), '| all-reduces/it', c.get('allreduces_per_iteration'), 'exchange ms/it', x)" | tee -a $OUT
      grep -o '"final_TC": [-0-9.e+]*' gpurun_out/bench_detail.json | head -1 | tee -a $OUT
    done; done ;;
  kernel_rows)
    # rocprofv3 --kernel-trace --stats of bench --workload <wl> --no-extras under VAR=val, the rows + the dependent-launch gaps:
    #   kernel_rows <wl> <VAR> <val> [bench args]      -> gpurun_out/<tag>_kernel_rows_<wl>_<VAR>_<val>.txt
    WL=$1; VAR=$2; VAL=$3; shift 3
    python3 __graft_entry__.py || exit 1
    export $VAR=$VAL
    OUT=$R/gpurun_out/${TAG}_kernel_rows_${WL}_${VAR}_${VAL}.txt
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rows_$WL -o rows -- python3 $R/bench.py --workload $WL --no-extras "$@" > $R/gpurun_out/prof_rows_$WL.log 2>&1
    echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --no-extras $* ($VAR=$VAL); lib $(python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; print(g._src_hash()[:16])")" > $OUT
    tail -1 $R/gpurun_out/prof_rows_$WL.log | cut -c1-160 >> $OUT
    python3 $R/tools/trace_gaps.py "$(find $R/gpurun_out/prof_rows_$WL -name '*kernel_trace.csv' | head -1)" >> $OUT 2>&1
    cat $OUT; find $R/gpurun_out/prof_rows_$WL -name "*.csv" -size +2M -delete ;;
  probe)
    P=$1; shift
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/$P tools/$P.hip 2>/dev/null || exit 1
    timeout 900 tools/$P "$@" 2>&1 | tee gpurun_out/${TAG}_${P}_$(echo "$*" | tr ' ' '_').txt ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
