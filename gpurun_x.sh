python bench.py --steps 5 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['train_epoch'])"
python profiles/epoch_cprofile.py 2>&1 | grep -v amdgpu.ids | cut -c1-150 | sed -n 3,24p
