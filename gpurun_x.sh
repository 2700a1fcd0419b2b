python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "column_sweep" 2>&1 | tail -2
for pf in 0 512 1024 2048 4096; do
for pace in 240 260 280; do
echo -n "pf=$pf pace=$pace: "; timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-epoch --no-backward --tune cs_pace=$pace --tune cs_pf=$pf 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['roofline']['ms_per_launch'],3))"
done; done
