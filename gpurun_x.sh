for r in 3 7; do
python bench.py --workload rmat-10m --d 256 --shard resident --emulate-shard $r/8 --steps 10 --no-cpu-baseline > gpurun_out/rmat10m_shard${r}of8.json 2>> gpurun_out/rmat10m.err
cut -c1-1500 gpurun_out/rmat10m_shard${r}of8.json
done
( time python bench.py --workload reddit-114m --steps 10 --no-cpu-baseline > gpurun_out/reddit114m.json 2> gpurun_out/reddit114m.err ) 2>&1 | tail -4
cut -c1-1800 gpurun_out/reddit114m.json; tail -3 gpurun_out/reddit114m.err
python bench.py --workload reddit-114m --kernel rows --steps 10 --no-cpu-baseline 2>/dev/null | cut -c1-400
