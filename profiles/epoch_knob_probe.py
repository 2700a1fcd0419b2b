"""Epoch time of bench.py's train_epoch leg under library knobs:  python profiles/epoch_knob_probe.py key=value ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, knobs = [sys.argv[0]], sys.argv[1:]
import torch, bench
from stochastic_gcn_amd import synthetic, _ffi
for kv in knobs:
    k, v = kv.split("=")
    _ffi.tune(k, int(v))
data = synthetic.reddit_like(seed=1, with_features=False)
te = bench.train_epoch_leg(data, torch.device("cuda:0"), epochs=8)
print("knobs", knobs, "epoch", round(te["epoch_time_s"], 5), "ms/step", round(te["ms_per_step"], 4), "sch_wait", round(te["sch_wait_s"], 4))
