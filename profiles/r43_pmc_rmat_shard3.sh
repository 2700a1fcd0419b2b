# PMC passes for block 3 of the 8-way NNZ-balanced S-RMAT 10 M / 197 M sharding (the rows of profiles/r26_rmat10m_shard3_*: --shard-row-weight 0)
# with the round-5 plan (four lane groups, clock in work coordinates); pace fixed to the autotuner's choice.
set -u
mkdir -p gpurun_out/r5c
ARGS="--workload rmat-10m --d 256 --shard resident --emulate-shard 3/8 --shard-row-weight 0"
timeout 900 python bench.py --no-epoch --no-cpu-baseline --steps 5 --warmup 2 $ARGS > gpurun_out/r5c/bench.json 2> gpurun_out/r5c/bench.err
PACE=$(python -c "
import json; r=json.loads(open('gpurun_out/r5c/bench.json').read().strip().splitlines()[-1]); print(r['config']['cs_autotune_ms_pace']['fwd_pace'])")
echo "pace $PACE"
ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r43_rmat10m_shard3 $ARGS --tune cs_pace=$PACE 2>&1 | tail -30
