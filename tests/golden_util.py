"""Helpers that turn a tests/golden/*.npz fixture (produced by oracle/make_golden.py from the
reference) into inputs for the oracle and for the HIP path."""
import json
import os

import numpy as np

import gmeta_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ['g0_disjoint_h1', 'g1_sampled_h2', 'g1_h3', 'g2_shared', 'g3_linkpred', 'g5_in_gt_out', 'g6_nan_skip', 'g7_wide_h2', 'g8_wide_scales', 'g9_wide_nan']
NAN_CASES = ('g6_nan_skip', 'g9_wide_nan')          # inf features: NaN query loss, no optimiser step (meta.py:163-169)
WIDE_CASES = ('g7_wide_h2', 'g8_wide_scales', 'g9_wide_nan')      # hidden 128: the split-MFMA update kernels can be forced onto them (gm_set_tuning)


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), allow_pickle=False)
        self.z = z
        self.name = name
        self.T = int(z['T'])
        self.args = json.loads(str(z['args']))
        self.config = [(n, p) for n, p in json.loads(str(z['config']))]
        self.link = self.args['link_pred_mode'] == 'True'
        self.edges = [(int(z['g%d_n' % k]), z['g%d_src' % k], z['g%d_dst' % k]) for k in range(int(z['n_graphs']))]
        self.feats = [z['g%d_feat' % k] for k in range(int(z['n_graphs']))]
        self.vars0 = [z['vars0_%d' % k] for k in range(int(z['n_vars']))]
        self.vars1 = [z['vars1_%d' % k] for k in range(int(z['n_vars']))]
        self.grad = [z['grad_%d' % k] for k in range(int(z['n_vars']))] if int(z['stepped']) else None
        self.K = self.args['update_step']
        self.K_test = self.args['update_step_test']

    def graphs(self):
        return [orc.Graph(n, s, d) for n, s, d in self.edges]

    def ref_nodes(self, tag, t, s):
        S = self.z[tag + '_seeds'].shape[1]
        off = self.z[tag + '_nodes_off']
        k = t * S + s
        return self.z[tag + '_nodes_flat'][off[k]:off[k + 1]]

    def ref_edges(self, tag, t, s):
        S = self.z[tag + '_seeds'].shape[1]
        off = self.z[tag + '_edges_off']
        k = t * S + s
        return self.z[tag + '_edges_flat'][off[k]:off[k + 1]].reshape(-1, 2)

    def replay_lists(self, tag, t):
        S = self.z[tag + '_seeds'].shape[1]
        return [self.ref_nodes(tag, t, s) for s in range(S)]

    def logits_sequence(self, flat_key, sizes):
        """Split the recorded flat logits by the call sizes [(rows, C), ...]."""
        flat = self.z[flat_key]
        out, p = [], 0
        for r, c in sizes:
            out.append(flat[p:p + r * c].reshape(r, c)); p += r * c
        assert p == len(flat), (p, len(flat))
        return out


def call_sizes(S_s, S_q, C, K):
    """Order of net() calls for one task (meta.py:122,131,138,145,152)."""
    sizes = [(S_s, C), (S_q, C), (S_q, C)]
    for _ in range(1, K):
        sizes += [(S_s, C), (S_q, C)]
    return sizes
