python -m pytest tests -m gpu -x -q 2>&1 | tail -2
COMMON="--dataset reddit --normalization graphsage --weight_decay 0 --dropout 0.2 --layer_norm --hidden1 128 --num_fc_layers 2 --epochs 3 --early_stopping 30 --batch_size=512 --test_batch_size=512 --cv --cvd --test_cv --degree=1 --test_degree=1"
timeout 600 python -m stochastic_gcn_amd.train $COMMON 2>&1 | grep -E "sgcn\] epoch|Epoch|Test set" | sed -E 's/mi F1.*time=/time=/' | cut -c1-160 | tail -8
SGCN_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 2>gpurun_out/b2.err | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j.get('train_epoch'))" | cut -c1-400
