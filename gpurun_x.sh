for i in 1 2 3; do python bench.py --steps 3 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['train_epoch']['epoch_times_s'], j['train_epoch']['sch_wait_s'])"; done
lscpu | grep -E "Model name|MHz" | head -3
