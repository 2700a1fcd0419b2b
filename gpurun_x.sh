python -m pytest tests/test_parallel_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -5
for v in resident allgather; do
SGCN_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --shard $v 2>gpurun_out/shard_$v.err | cut -c1-900
done
