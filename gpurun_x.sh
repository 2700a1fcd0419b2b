python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; python -c "
import json; j=json.load(open('gpurun_out/bench_default.json')); print(j['value'], j['roofline']['frac'], j['cpu_baseline'], j['train_epoch']['epoch_time_s'])"
