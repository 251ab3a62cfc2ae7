"""Output check of bench.py: the tokens its timed steps produced against what the
real reference produced on the same batch (tests/golden/bench_*.npz, written by
oracle/gen_golden_bench.py from /root/reference; only the committed fixtures are
read here -- no oracle code, no reference tree)."""
import json
import os
from typing import List, Sequence, Tuple

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                          'tests', 'golden')
NBEST_TIE = 4e-3     # 2 x the fp64 prefix-score tolerance of the parity tests
RESCORE_TIE = 2e-3   # 2 x the 1e-3 rescoring tolerance


def _load_meta(name: str):
    path = os.path.join(GOLDEN_DIR, name + '.npz')
    if not os.path.exists(path):
        return None
    z = np.load(path)
    return json.loads(bytes(z['meta']).decode('utf8'))


def _expected(meta, method: str, world: int):
    """Per global utterance index: (best tokens, runner-up tokens or None, gap, tie)."""
    out = []
    groups = meta['groups'] if 'groups' in meta else [meta]
    for g in groups[:world]:
        for b in range(len(g['prefix'] if 'prefix' in g else g['greedy'])):
            if method == 'attention_rescoring':
                r = g['rescoring'][b]
                order = np.argsort(r['all_scores'])[::-1]
                best = list(g['prefix'][b]['nbest'][order[0]])
                second = (list(g['prefix'][b]['nbest'][order[1]])
                          if len(order) > 1 else None)
                gap = (r['all_scores'][order[0]] - r['all_scores'][order[1]]
                       if len(order) > 1 else float('inf'))
                out.append((best, second, gap, RESCORE_TIE))
            elif method == 'ctc_greedy_search':
                out.append((list(g['greedy'][b]), None, float('inf'), 0.0))
            else:
                p = g['prefix'][b]
                second = list(p['nbest'][1]) if len(p['nbest']) > 1 else None
                gap = (p['nbest_scores'][0] - p['nbest_scores'][1]
                       if len(p['nbest']) > 1 else float('inf'))
                out.append((list(p['nbest'][0]), second, gap, NBEST_TIE))
    return out


def verify_bench_output(workload: str, world: int,
                        results: Sequence[Tuple[int, List[int], float]],
                        method: str = None) -> dict:
    """`results`: (global utterance index, tokens, score) of one bench step.
    verified = every utterance's tokens are the reference's 1-best (or its
    runner-up when the reference's own top-2 gap is inside the parity tolerance)."""
    meta = _load_meta(f'bench_{workload}' if world == 1 else f'bench_{workload}_w8')
    if meta is None and world > 1 and workload == 'config2':
        meta = _load_meta('bench_config2')
    if meta is None:
        return {'verified': None, 'reason': f'no committed golden for {workload}'}
    if method is None:
        from wenet_amd.synthetic import BENCH_WORKLOADS
        method = BENCH_WORKLOADS[workload]['method']
    n_groups = len(meta['groups']) if 'groups' in meta else 1
    if world > n_groups:
        return {'verified': None, 'reason': f'golden covers {n_groups} rank(s)'}
    exp = _expected(meta, method, world)
    if len(results) != len(exp):
        return {'verified': False, 'reason': f'{len(results)} results for {len(exp)} '
                'utterances'}
    identical = near = 0
    wrong = []
    for gi, toks, _score in results:
        best, second, gap, tie = exp[gi]
        if list(toks) == best:
            identical += 1
        elif second is not None and gap < tie and list(toks) == second:
            near += 1
        else:
            wrong.append(int(gi))
    return {
        'verified': not wrong,
        'utterances': len(exp),
        'identical': identical,
        'near_tie': near,
        'mismatched': wrong[:8],
        'against': ('tests/golden/bench_%s%s.npz (real reference, %s)' %
                    (workload, '' if world == 1 else '_w8', method)),
    }
