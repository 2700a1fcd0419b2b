"""--det_dropout training steps: the eager layer-by-layer path against the step program (round 6), on the golden det cases
(tests/model_cases.py DET_CASES): ms per step over 200 steps after 20 warm-up steps."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import model_cases as mc                                     # noqa: E402
import test_step_program_gpu as tp                           # noqa: E402
from stochastic_gcn_amd.flags import FLAGS                   # noqa: E402


def run(case, native, steps=200, warm=20):
    params = mc.make_oracle_model(case, seed=3).params
    m = tp._model(case, {k: v.copy() for k, v in params.items()}, native)
    sch = mc.make_scheduler(case, 1)
    t0 = None
    for step in range(steps + warm):
        if step == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if sch.start >= sch.data.shape[0]:
            sch.start = 0
        pb = sch.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, None)
        pb.dropout = case['flags']['dropout']
        m.run_one_step(None, pb, sync=False)
    torch.cuda.synchronize()
    prog = [p for p in getattr(m, '_programs', {}).values() if p is not None]
    return (time.perf_counter() - t0) / steps * 1e3, (prog[0].n_all if prog else None)


for name in sorted(mc.DET_CASES):
    case = mc.build_case(name)
    e, _ = run(case, False)
    p, nops = run(case, True)
    print(json.dumps(dict(case=name, eager_ms_per_step=round(e, 4), program_ms_per_step=round(p, 4), ops_per_step=nops)), flush=True)
