python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['train_epoch']['epoch_times_s'])"
