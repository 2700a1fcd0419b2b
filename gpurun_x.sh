set -x
python -m pytest tests -m gpu -x -q -k "scatter or train" 2>&1 | tail -3
SGCN_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench2.json 2> gpurun_out/bench2.err
tail -5 gpurun_out/bench2.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/bench2.json').read().strip().splitlines()[-1])
print(j['value'], j['config']['grad_allreduce_ms'], j.get('train_epoch'))
PY
python bench.py --steps 5 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['train_epoch'])"
