"""The data-parallel step's dependent chain with a forced ONE-rank RCCL job (SGCN_FORCE_PG=1), per placement of the history
exchange: behind the optimizer on the step's stream (SGCN_EXCHANGE_OVERLAP=0), or on the library's exchange stream right behind
the aggregator with a communicator of its own (=1, round 6) -- with default events (SGCN_XCHG_SYSFENCE=1) or fence-free ones.

    for cfg in "0" "1" "1 SGCN_XCHG_SYSFENCE=1"; do env SGCN_FORCE_PG=1 SGCN_EXCHANGE_OVERLAP=$cfg python profiles/exchange_chain_probe.py; done
    (the cost of each dependency: SGCN_XCHG_DROPS_A_DEPENDENCY=1 SGCN_XCHG_SKIP=fork | join | forkjoin -- results undefined)
"""
import contextlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import step_chain_probe                            # noqa: E402
from stochastic_gcn_amd import synthetic                     # noqa: E402
from stochastic_gcn_amd.flags import FLAGS                   # noqa: E402
from stochastic_gcn_amd.train import Trainer                 # noqa: E402


def main():
    dev = torch.device("cuda:0")
    data = synthetic.reddit_like(with_features=False)
    n, train_adj, full_adj, _, _, _, labels, tr, va, te = data
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2,
                 layer_norm=True, hidden1=128, num_fc_layers=2, batch_size=512, test_batch_size=512,
                 cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, seed=1)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    feats = torch.randn((n, 602), device=dev, generator=g)
    # SGCN_PROBE_SIDE_STREAM=1: the whole job on a NON-default (non-blocking) torch stream -- is an event record cheaper there
    # than on the legacy default stream (8 us of the step's chain)?
    side = torch.cuda.Stream() if os.environ.get("SGCN_PROBE_SIDE_STREAM") else None
    ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
    with ctx:
        return _body(dev, n, train_adj, full_adj, feats, labels, tr, va, te, side is not None)


def _body(dev, n, train_adj, full_adj, feats, labels, tr, va, te, side):
    with contextlib.redirect_stdout(sys.stderr):
        trn = Trainer(data=(n, train_adj, full_adj, feats, None, None, labels, tr, va, te), verbose=False)
    walls = []
    for _ in range(6):
        trn.train_epoch()
        walls.append(trn.last_epoch['train_wall_s'])
    rec = dict(side_stream=side, overlap=os.environ.get("SGCN_EXCHANGE_OVERLAP", "1"), sysfence=os.environ.get("SGCN_XCHG_SYSFENCE"),
               forced_pg=os.environ.get("SGCN_FORCE_PG"), exchange_overlap=bool(trn.par.exchange_overlap),
               native=bool(trn.par.native), epoch_ms=round(min(walls) * 1e3, 3), steps=trn.last_epoch['steps'],
               ms_per_step=round(min(walls) * 1e3 / trn.last_epoch['steps'], 5))
    chains = [step_chain_probe(trn) for _ in range(3)]
    rec["gpu_chain_us"] = [round(c["gpu_chain_us"], 2) for c in chains]
    rec["host_launch_us"] = [round(c["host_launch_us"], 2) for c in chains]
    rec["ops_per_step"] = chains[0]["ops_per_step"]
    print(json.dumps(rec), flush=True)
    trn.par.shutdown()


if __name__ == "__main__":
    main()
