set -u
mkdir -p gpurun_out/r4b
B="python bench.py --no-epoch --no-cpu-baseline --steps 10 --warmup 2"
for r in 3 0 7; do
  timeout 900 $B --workload rmat-10m --d 256 --shard resident --emulate-shard $r/8 > gpurun_out/r4b/rmat10m_d256_shard${r}of8_bench.json 2> gpurun_out/r4b/rmat_$r.err
  tail -c 600 gpurun_out/r4b/rmat_$r.err | tail -2
done
PACE=$(python -c "
import json; r=json.loads(open('gpurun_out/r4b/rmat10m_d256_shard3of8_bench.json').read().strip().splitlines()[-1]); print(r['config']['cs_autotune_ms_pace'])" 2>/dev/null)
echo "tuned: $PACE"
timeout 900 $B --workload reddit-114m > gpurun_out/r4b/reddit114m_bench.json 2> gpurun_out/r4b/r114.err
for f in gpurun_out/r4b/*_bench.json; do python - "$f" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], r["roofline"]["ms_per_launch"], r["roofline"]["frac"], r["roofline"]["edges_per_s_fwd"], r["config"].get("cs_autotune_ms_pace"), r["roofline"]["kernel"][:80])
PY
done
