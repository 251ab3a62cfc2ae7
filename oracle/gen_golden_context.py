#!/usr/bin/env python3
"""tests/golden/ctx_*.npz: the REAL reference's context-biased decode
(ContextGraph of wenet/utils/context_graph.py fed to ctc_prefix_beam_search /
attention_rescoring, search.py:127-249,374-458) on seeded synthetic models.
Runs only where /root/reference exists.

The biasing phrases are cut from the unbiased n-best lists so that the graph is
actually walked: full matches, suffix phrases (output arcs), a phrase that is a
prefix of an earlier one (the reference's is_end-at-creation quirk), partial
matches that must be backed off through fail arcs and at finalize().

Two more fixtures need no model at all:
  ctxsearch_*.npz  reference ctc_prefix_beam_search on seeded peaky log-probs
                   over a small vocabulary (phrases hit in almost every frame).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden, _ref_harness  # noqa: E402

CASES = [
    dict(case='ctx_tiny_causal', config='tiny_causal', wseed=0, batch=3,
         frames=(90, 140), fseed=21, beam=4, context_score=3.0, ctc_weight=0.5,
         reverse_weight=0.3, blank_penalty=4.0),
    dict(case='ctx_aishell', config='aishell_u2pp', wseed=0, batch=2,
         frames=(200, 260), fseed=22, beam=10, context_score=6.0, ctc_weight=0.5,
         reverse_weight=0.3, blank_penalty=0.0),
]

SEARCH_CASES = [
    dict(case='ctxsearch_v12_b4', vocab=12, batch=4, frames=(30, 80), seed=31,
         beam=4, context_score=2.0, n_phrases=6, peak=3.0),
    dict(case='ctxsearch_v40_b10', vocab=40, batch=3, frames=(50, 120), seed=32,
         beam=10, context_score=6.0, n_phrases=25, peak=4.0),
    dict(case='ctxsearch_v20_b16', vocab=20, batch=2, frames=(20, 60), seed=33,
         beam=16, context_score=1.5, n_phrases=8, peak=2.0),
]


def reference_graph(phrases, context_score):
    """A reference ContextGraph over token-id lists (its constructor wants a
    text file + symbol table; build_graph takes the id lists directly)."""
    _ref_harness.install()
    from wenet.utils.context_graph import ContextGraph, ContextState
    g = ContextGraph.__new__(ContextGraph)
    g.context_score = context_score
    g.context_list = phrases
    g.num_nodes = 0
    g.root = ContextState(id=0, token=-1, token_score=0, node_score=0,
                          output_score=0, is_end=False)
    g.root.fail = g.root
    g.build_graph(phrases)
    return g


def phrases_from_nbest(results, vocab, seed):
    rng = np.random.RandomState(seed)
    ph = []
    for r in results:
        for h in r.nbest[:3]:
            h = list(h)
            if len(h) < 4:
                continue
            a = int(rng.randint(0, len(h) - 3))
            ph.append(h[a:a + 3])          # full match
            ph.append(h[a + 1:a + 3])      # suffix of the previous -> output arc
            ph.append(h[a:a + 2])          # prefix of an earlier phrase (never is_end)
            wrong = int((h[a + 3] + 1 + rng.randint(0, vocab - 3)) % vocab)
            wrong = max(wrong, 1)
            ph.append(h[a + 1:a + 3] + [wrong, int(rng.randint(1, vocab - 1))])  # partial
            if len(h) >= 3:
                ph.append(h[-3:] + [int(rng.randint(1, vocab - 1))])  # unfinished at the end
    return [list(map(int, p)) for p in ph if len(p) > 0]


def pack(results):
    return dict(
        nbest=[[list(map(int, h)) for h in r.nbest] for r in results],
        nbest_scores=[[float(s) for s in r.nbest_scores] for r in results],
        nbest_times=[[list(map(int, t)) for t in r.nbest_times] for r in results],
    )


def save(path, meta):
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(),
                                                 dtype=np.uint8))


def main():
    from wenet_amd import synthetic as S
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    outdir = os.path.join(ROOT, 'tests', 'golden')
    _ref_harness.install()
    from wenet.models.transformer.search import ctc_prefix_beam_search
    for c in SEARCH_CASES:
        logp, lens = S.peaky_logprobs(c['batch'], c['frames'], c['vocab'], c['peak'], c['seed'])
        rng = np.random.RandomState(c['seed'])
        phrases = [[int(t) for t in rng.randint(1, c['vocab'], rng.randint(1, 5))]
                   for _ in range(c['n_phrases'])]
        g = reference_graph(phrases, c['context_score'])
        res = ctc_prefix_beam_search(logp, lens, c['beam'], g, 0)
        meta = dict(c)
        meta['phrases'] = phrases
        meta['prefix'] = pack(res)
        save(os.path.join(outdir, c['case'] + '.npz'), meta)
        print(c['case'], [len(r.tokens) for r in res], [round(r.score, 3) for r in res])
    for c in CASES:
        configs = S.make_configs(c['config'])
        sd = S.make_state_dict(configs, c['wseed'])
        model = gen_golden.build_reference_model(configs, sd)
        feats, lens = S.make_features(c['batch'], c['frames'], seed=c['fseed'])
        kw = dict(beam_size=c['beam'], ctc_weight=c['ctc_weight'],
                  reverse_weight=c['reverse_weight'],
                  blank_penalty=c['blank_penalty'])
        with torch.no_grad():
            plain = model.decode(['ctc_prefix_beam_search'], feats, lens, **kw)
            phrases = phrases_from_nbest(plain['ctc_prefix_beam_search'],
                                         configs['output_dim'], c['fseed'])
            g = reference_graph(phrases, c['context_score'])
            res = model.decode(['ctc_prefix_beam_search', 'attention_rescoring'],
                               feats, lens, context_graph=g, **kw)
        meta = dict(c)
        meta['phrases'] = phrases
        meta['prefix'] = pack(res['ctc_prefix_beam_search'])
        meta['rescoring_tokens'] = [list(map(int, r.tokens))
                                    for r in res['attention_rescoring']]
        meta['rescoring_scores'] = [float(r.score) for r in res['attention_rescoring']]
        changed = sum(a.tokens != b.tokens for a, b in
                      zip(plain['ctc_prefix_beam_search'], res['ctc_prefix_beam_search']))
        save(os.path.join(outdir, c['case'] + '.npz'), meta)
        print(c['case'], 'phrases', len(phrases), 'nodes', g.num_nodes,
              'utterances changed by biasing', changed)


if __name__ == '__main__':
    main()
