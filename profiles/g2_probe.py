"""Two-lane-group column sweep (ColumnSweepCSR(G=2)): forward SpMM time against the sweep clock, per
alignment bound of the two bins (sgcn_csplan2 `align`).  python profiles/g2_probe.py [align ...]"""
import sys, json, os
sys.path.insert(0, '.')
import torch
from stochastic_gcn_amd import ops, synthetic

def main(aligns, G=2):
    dev = torch.device('cuda:0')
    data = synthetic.reddit_like(seed=1, with_features=False)
    a = data[2].tocsr()
    B = torch.randn(a.shape[1], 608, device=dev)[:, :602]          # the bench's layout: 19 whole cache lines per row
    out = {}
    for al in aligns:
        A = ops.ColumnSweepCSR(a, dev, G=G, align=al) if al >= 0 else ops.ColumnSweepCSR(a, dev)
        C = torch.empty((a.shape[0], 608), device=dev)[:, :602]
        row = {}
        for p in (-1, 150, 170, 180, 190, 200, 210, 220, 230, 240, 250, 260, 270, 280, 300, 320, 380):
            A.pace[602] = p
            ops.spmm_cs(A, B, out=C)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.spmm_cs(A, B, out=C)
            e1.record(); e1.synchronize()
            row[p] = round(e0.elapsed_time(e1) / 3, 3)
        out[al] = dict(pad_fraction=round(getattr(A, 'pad_fraction', 0.0), 4), ms_by_pace=row)
        print(al, json.dumps(out[al]), flush=True)
    return out

if __name__ == '__main__':
    G = int(os.environ.get('G', '2'))
    main([int(x) for x in sys.argv[1:]] or [-1, 0, 1024, 2048], G)
