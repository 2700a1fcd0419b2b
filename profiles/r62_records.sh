# Round 6, VERDICT r5 items 6 and 7: the records that were missing or did not match what ships.
#   (a) S-Reddit-114M on the round-5 plan: trace + FETCH / WRITE / TCC passes (roofline.traffic was null)
#   (b) config 5, block 3 of the LOAD-balanced sharding (the one that ships; r43_* was the nnz-balanced block)
#   (c) the headline at round 4's split threshold and clock (T = 4 x mean degree = 397, 197 ns: profiles/r28_*) next to
#       the current default on the SAME box: where the +12 % fetch per launch since r28 comes from
# Every pass is its own rocprofv3 run (profiles/profile_cs.sh); the pace is fixed to the autotuner's choice.
set -u
mkdir -p gpurun_out/r62
pace_of() { python - "$1" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = r['config']['cs_autotune_ms_pace']
print(t['fwd_pace'] if 'fwd_pace' in t else t['fwd'][1])
PY
}
keep() {   # tag: copy what summarize.py wrote under profiles/ on this box to gpurun_out/ (the part that is merged back)
    for f in profiles/$1_rocprof_summary.txt profiles/$1_traffic.json profiles/$1_counters.json; do [ -f $f ] && cp $f gpurun_out/r62/; done
    cp gpurun_out/prof_$1/bench_trace.json gpurun_out/r62/$1_bench_under_rocprof.json 2>/dev/null
    rm -rf gpurun_out/prof_$1/*/    # the databases stay on the box
}
want() { [ -z "${WHICH:-}" ] || [[ " $WHICH " == *" $1 "* ]]; }
if want a; then
    ARGS="--workload reddit-114m"
    timeout 600 python bench.py --no-epoch --no-cpu-baseline --steps 5 --warmup 2 $ARGS > gpurun_out/r62/r62_reddit114m_bench.json 2> gpurun_out/r62/a.err
    PACE=$(pace_of gpurun_out/r62/r62_reddit114m_bench.json); echo "114M pace $PACE"
    ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r62_reddit114m $ARGS --tune cs_pace=$PACE 2>&1 | tail -12
    keep r62_reddit114m
fi
if want b; then
    ARGS="--workload rmat-10m --d 256 --shard resident --emulate-shard 3/8"
    timeout 900 python bench.py --no-epoch --no-cpu-baseline --steps 5 --warmup 2 $ARGS > gpurun_out/r62/r62_rmat10m_d256_shard3of8_bench.json 2> gpurun_out/r62/b.err
    PACE=$(pace_of gpurun_out/r62/r62_rmat10m_d256_shard3of8_bench.json); echo "rmat block 3 pace $PACE"
    ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r62_rmat10m_shard3 $ARGS --tune cs_pace=$PACE 2>&1 | tail -12
    keep r62_rmat10m_shard3
fi
if want c; then
    ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r62_headline_T397 --cs-t 397 --tune cs_pace=197 2>&1 | tail -12
    keep r62_headline_T397
    ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r62_headline_T397_p195 --cs-t 397 --tune cs_pace=195 2>&1 | tail -12
    keep r62_headline_T397_p195
    ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r62_headline --tune cs_pace=195 2>&1 | tail -12
    keep r62_headline
    ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r62_headline_p197 --tune cs_pace=197 2>&1 | tail -12
    keep r62_headline_p197
fi
