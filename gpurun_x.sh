python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 3 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['train_epoch']['epoch_times_s'])"
