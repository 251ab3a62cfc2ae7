"""Shared helpers of the GPU parity tests."""
import numpy as np
import torch

from golden_util import build_inputs, load_case


def make_model(configs, sd):
    from wenet_amd.model import ASRModel
    return ASRModel(configs, sd, device='cuda')


_MODEL_CACHE = {}


def cached_model(config_name, wseed):
    from wenet_amd import synthetic as S
    key = (config_name, wseed)
    if key not in _MODEL_CACHE:
        configs = S.make_configs(config_name)
        sd = S.make_state_dict(configs, wseed)
        _MODEL_CACHE[key] = (configs, sd, make_model(configs, sd))
    return _MODEL_CACHE[key]


def compare_nbest(got, ref_nbest, ref_scores, ref_times, score_atol=2e-3,
                  what=''):
    """n-best lists produced from slightly different log-probs: every reference
    hypothesis that is clear of the pruning boundary must be present with a
    matching score and identical time stamps; order may differ only between
    hypotheses whose scores are closer than the tolerance."""
    g_nbest = [list(x) for x in got.nbest]
    r_nbest = [list(x) for x in ref_nbest]
    assert len(g_nbest) == len(r_nbest), what
    cutoff = ref_scores[-1] + 4 * score_atol
    for i, h in enumerate(r_nbest):
        if ref_scores[i] < cutoff and i > 0:
            continue  # too close to the beam edge to be stable
        assert h in g_nbest, f'{what}: reference hyp #{i} {h} missing'
        j = g_nbest.index(h)
        assert abs(got.nbest_scores[j] - ref_scores[i]) < score_atol, \
            (what, i, got.nbest_scores[j], ref_scores[i])
        assert list(got.nbest_times[j]) == list(ref_times[i]), (what, i)
        if j != i:
            assert abs(ref_scores[i] - ref_scores[j]) < 2 * score_atol, \
                f'{what}: order differs beyond tolerance ({i} vs {j})'
    if len(ref_scores) < 2 or ref_scores[0] - ref_scores[1] > 2 * score_atol:
        assert list(got.tokens) == r_nbest[0], what


def frame_margins(logp: torch.Tensor):
    top2 = logp.topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1])
