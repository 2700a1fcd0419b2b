"""``VRGCN`` -- the control-variate model (mirror of gcn/vrgcn.py:15-108)."""
import torch

from .flags import FLAGS
from .layers import VRAggregator
from .models import GCN


class VRGCN(GCN):
    def __init__(self, L, preprocess, placeholders, features, nbr_features, adj, cvd, **kwargs):
        super(VRGCN, self).__init__(L, preprocess, placeholders, features, nbr_features, adj, cvd,
                                    **kwargs)

    def _build_history(self):
        """One zero-initialised, non-trainable N x dims fp32 history per aggregation layer,
        resident in HBM (gcn/vrgcn.py:23-36).  Train and test models own separate histories."""
        self.history = []
        for i in range(self.L):
            dims = self.agg0_dim if i == 0 else FLAGS.hidden1
            n_history = 2 if FLAGS.det_dropout else 1        # (mean, variance) under det-dropout, gcn/vrgcn.py:28
            self.history.append([torch.zeros((self.num_data, dims), dtype=torch.float32, device=self.device)
                                 for _ in range(n_history)])
            print('History size = {} GB'.format(self.num_data * dims * 4 * n_history / 1024.0 / 1024.0 / 1024.0))

    def _build_aggregators(self):
        for l in range(self.L):
            self.aggregators.append(VRAggregator(self, l, self.cvd, name='agg%d' % l))

    def _count(self, feed_dict):
        """FLOP / size counters printed per epoch (gcn/vrgcn.py:50-69)."""
        ph = self.placeholders
        for l in range(self.L):
            adj = feed_dict[ph['adj'][l]][0]
            fadj = feed_dict[ph['fadj'][l]][0]
            dim = self.agg0_dim if l == 0 else FLAGS.hidden1
            g_ops = (fadj.shape[0] + adj.shape[0]) * dim * 4
            if self.cvd:
                g_ops *= 2
            self.g_ops += g_ops
            self.adj_sizes[l] += adj.shape[0]
            self.fadj_sizes[l] += fadj.shape[0]
            self.amt_data += adj.shape[0]
        for l in range(self.L + 1):
            self.field_sizes[l] += feed_dict[ph['fields'][l]].size
        for c, l in self.layer_comp:
            nn_ops = c * feed_dict[ph['fields'][l]].size * 4
            if self.cvd:
                nn_ops *= 2
            self.nn_ops += nn_ops
