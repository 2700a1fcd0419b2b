# PMC passes for rank 3's block of the 8-way sharded S-RMAT 10 M / 197 M product (BASELINE config 5), pace fixed to the
# autotuner's choice so that no tuning launch pollutes the per-kernel means.
set -u
mkdir -p gpurun_out/r4c
ARGS="--workload rmat-10m --d 256 --shard resident --emulate-shard 3/8"
timeout 900 python bench.py --no-epoch --no-cpu-baseline --steps 5 --warmup 2 $ARGS > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
PACE=$(python -c "
import json; r=json.loads(open('gpurun_out/r4c/bench.json').read().strip().splitlines()[-1]); print(r['config']['cs_autotune_ms_pace']['fwd_pace'])")
echo "pace $PACE"
ONLY="trace pmc_fetch pmc_write pmc_l2" bash profiles/profile_cs.sh r26_rmat10m_shard3 $ARGS --tune cs_pace=$PACE 2>&1 | tail -30
