"""``PlainGCN`` -- exact / neighbour-sampling model without history (mirror of gcn/plaingcn.py)."""
from .flags import FLAGS
from .layers import PlainAggregator
from .models import GCN


class PlainGCN(GCN):
    def __init__(self, L, preprocess, placeholders, features, nbr_features, adj, cvd, **kwargs):
        super(PlainGCN, self).__init__(L, preprocess, placeholders, features, nbr_features, adj, cvd,
                                       **kwargs)

    def _build_history(self):
        self.history = []

    def _build_aggregators(self):
        for l in range(self.L):
            self.aggregators.append(PlainAggregator(self, l, name='agg%d' % l))

    def _count(self, feed_dict):
        """gcn/plaingcn.py:41-50."""
        ph = self.placeholders
        for l in range(self.L):
            dim = self.agg0_dim if l == 0 else FLAGS.hidden1
            adj = feed_dict[ph['adj'][l]][0]
            self.g_ops += adj.shape[0] * dim * 4
            self.adj_sizes[l] += adj.shape[0]
            self.amt_data += adj.shape[0]
        for l in range(self.L + 1):
            self.field_sizes[l] += feed_dict[ph['fields'][l]].size
        for c, l in self.layer_comp:
            self.nn_ops += c * feed_dict[ph['fields'][l]].size * 4
