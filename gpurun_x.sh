python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['train_epoch'])"
