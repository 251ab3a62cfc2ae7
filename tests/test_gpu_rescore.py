"""wn_rescore: attention_rescoring's score arithmetic, arg-max and confidences on the device
(wenet/models/transformer/search.py:424-457).  The reduce kernel is held BIT FOR BIT to a numpy
replay of the reference's scalar arithmetic on the per-token log-probs the same decoder pass
produced (wn_attention_rescoring exports them); the decoder itself is held to the oracle and
the real reference's goldens in test_gpu_parity.py / test_gpu_bench_parity.py."""
import numpy as np
import pytest
import torch

from gpu_util import cached_model, rescore_replay

pytestmark = pytest.mark.gpu


def _logps(model, raw, reverse_weight):
    """per-token log-probs of every hypothesis through the diagnostic entry point"""
    from wenet_amd import _lib
    from wenet_amd.search import _stream_ptr
    B, beam = raw['hyp_lens'].shape
    max_len = raw['hyp_tokens'].shape[2]
    l2r = np.zeros((B, beam, max_len + 1), dtype=np.float32)
    r2l = np.zeros((B, beam, max_len + 1), dtype=np.float32)
    _lib.check(_lib.lib().wn_attention_rescoring(
        model._h, beam, _lib.i32p(raw['n_hyps']), _lib.i32p(raw['hyp_lens']),
        _lib.i32p(raw['hyp_tokens']), max_len, float(reverse_weight), _lib.f32p(l2r),
        _lib.f32p(r2l), _stream_ptr(model.device)), 'wn_attention_rescoring')
    return l2r, r2l


@pytest.mark.parametrize('config,beam,cw,rw', [
    ('tiny_causal', 4, 0.5, 0.3), ('tiny_causal', 10, 0.3, 0.0), ('tiny_sym', 6, 0.0, 0.0),
    ('aishell_u2pp', 10, 0.5, 0.4), ('librispeech_bidecoder_large', 10, 0.5, 0.3),
    ('tiny_causal', 16, 0.7, 1.0)])
def test_rescore_bit_exact_vs_replay(config, beam, cw, rw):
    from wenet_amd import synthetic as S
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(5, (90, 330), seed=23)
    got = model.decode(['ctc_prefix_beam_search', 'attention_rescoring'], feats.cuda(), lens,
                       beam_size=beam, ctc_weight=cw, reverse_weight=rw)
    raw = model._last_prefix_raw
    pre, res = got['ctc_prefix_beam_search'], got['attention_rescoring']
    l2r, r2l = _logps(model, raw, rw)
    use_r2l = bool(configs['decoder'] == 'bitransformer'
                   and configs['decoder_conf'].get('r_num_blocks', 0) > 0)
    rep = rescore_replay([r.nbest for r in pre], [r.nbest_scores for r in pre], l2r, r2l, cw, rw,
                         use_r2l)
    for b, (bi, scores, confs, tcs) in enumerate(rep):
        r = res[b]
        assert np.array_equal(np.asarray(r.all_scores, dtype=np.float32),
                              np.asarray(scores, dtype=np.float32)), (config, b)
        assert tuple(r.tokens) == tuple(pre[b].nbest[bi])
        assert r.score == float(scores[bi])
        assert list(r.times) == list(pre[b].nbest_times[bi])
        np.testing.assert_allclose(r.confidence, confs[bi], rtol=1e-13)
        np.testing.assert_allclose(r.tokens_confidence, tcs[bi], rtol=1e-13, atol=0)
        assert isinstance(r.score, float) and isinstance(r.tokens, tuple)
        assert isinstance(r.tokens_confidence, list)


def test_rescore_uploaded_nbest_equals_device_nbest():
    """the same n-best handed over as Python lists (search.attention_rescoring on results of any
    origin) and taken from the prefix beam search's device block give identical records"""
    from wenet_amd import search as SR, synthetic as S
    configs, sd, model = cached_model('tiny_causal', 0)
    feats, lens = S.make_features(4, (80, 260), seed=5)
    enc, mask = model._forward_encoder(feats.cuda(), lens)
    enc_lens = mask.squeeze(1).sum(1)
    got = model.decode(['ctc_prefix_beam_search', 'attention_rescoring'], feats.cuda(), lens,
                       beam_size=5, ctc_weight=0.4, reverse_weight=0.25)
    pre, dev = got['ctc_prefix_beam_search'], got['attention_rescoring']
    for r in pre:
        r.nbest  # materialise the lists: the free function must not find raw arrays
    via_lists = SR.attention_rescoring(model, pre, enc, enc_lens, 0.4, 0.25)
    plain = [SR.DecodeResult(r.tokens, nbest=r.nbest, nbest_scores=r.nbest_scores,
                             nbest_times=r.nbest_times) for r in pre]
    via_plain = SR.attention_rescoring(model, plain, enc, enc_lens, 0.4, 0.25)
    for a, b, c in zip(dev, via_lists, via_plain):
        for o in (b, c):
            assert a.tokens == o.tokens and a.score == o.score and a.times == o.times
            assert a.all_scores == o.all_scores
            assert a.confidence == o.confidence and a.tokens_confidence == o.tokens_confidence


def test_rescore_needs_a_prefix_beam_result():
    import ctypes
    from wenet_amd import _lib, synthetic as S
    from wenet_amd.search import _stream_ptr
    configs, sd, model = cached_model('tiny_causal', 0)
    feats, lens = S.make_features(2, (80, 120), seed=1)
    model._forward_encoder(feats.cuda(), lens)      # new batch: the old n-best is void
    best = np.zeros((2, ), dtype=np.int32)
    score = np.zeros((2, ), dtype=np.float32)
    ni, nd = ctypes.POINTER(ctypes.c_int32)(), ctypes.POINTER(ctypes.c_double)()
    nf = ctypes.POINTER(ctypes.c_float)()
    rc = _lib.lib().wn_rescore(model._h, 4, ni, ni, ni, nd, 30, 0.5, 0.0, _lib.i32p(best),
                               _lib.f32p(score), nd, nd, nf, _stream_ptr(model.device))
    assert rc != 0
    assert b'prefix beam' in _lib.lib().wn_last_error()


@pytest.mark.parametrize('config,B,frames', [('tiny_causal', 4, (90, 330)),
                                             ('librispeech_bidecoder_large', 12, (500, 900))])
def test_rescore_prefetch_on_the_side_stream_changes_nothing(config, B, frames):
    """wn_rescore_prefetch (cross-attention K / V of every decoder layer projected on a second
    stream while the prefix beam search runs) against the projections inside the rescoring pass:
    the same GEMMs on the same operands -- identical records, also when a prefetch is abandoned
    (a decode without rescoring in between) and over repeated decodes (ordering of the two
    streams)."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=31)
    kw = dict(beam_size=6, ctc_weight=0.5, reverse_weight=0.3)
    M = ['attention_rescoring']
    try:
        _lib.check(L.wn_tune_set(b'rescore_prefetch', 0), 'tune')
        ref = model.decode(M, feats.cuda(), lens, **kw)[M[0]]
        _lib.check(L.wn_tune_set(b'rescore_prefetch', 1), 'tune')
        runs = [model.decode(M, feats.cuda(), lens, **kw)[M[0]] for _ in range(3)]
        # an abandoned prefetch: prefetch by hand, then a new batch without rescoring
        model._forward_encoder(feats.cuda(), lens)
        _lib.check(L.wn_rescore_prefetch(model._h, 1, torch.cuda.current_stream().cuda_stream),
                   'prefetch')
        model.decode(['ctc_greedy_search'], feats[:2].cuda(), lens[:2])
        runs.append(model.decode(M, feats.cuda(), lens, **kw)[M[0]])
    finally:
        L.wn_tune_set(b'rescore_prefetch', 1)
    for got in runs:
        for a, b in zip(ref, got):
            assert a.tokens == b.tokens and a.score == b.score
            assert a.all_scores == b.all_scores and a.tokens_confidence == b.tokens_confidence


@pytest.mark.parametrize('config,B,frames', [('tiny_causal', 5, (90, 330)),
                                             ('librispeech_bidecoder_large', 12, (500, 900))])
def test_cross_attention_per_utterance_group_changes_nothing(config, B, frames):
    """The rescoring decoder's cross attention with all hypothesis rows of an utterance as ONE
    attention sequence (they share the utterance's encoder frames) against one sequence per
    hypothesis: per query row the same keys in the same tile order -- identical records."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model(config, 0)
    feats, lens = S.make_features(B, frames, seed=37)
    kw = dict(beam_size=7, ctc_weight=0.5, reverse_weight=0.3)
    M = ['attention_rescoring']
    try:
        _lib.check(L.wn_tune_set(b'rescore_groups', 0), 'tune')
        ref = model.decode(M, feats.cuda(), lens, **kw)[M[0]]
        _lib.check(L.wn_tune_set(b'rescore_groups', 1), 'tune')
        got = model.decode(M, feats.cuda(), lens, **kw)[M[0]]
    finally:
        L.wn_tune_set(b'rescore_groups', 1)
    for a, b in zip(ref, got):
        assert a.tokens == b.tokens and a.score == b.score
        assert a.all_scores == b.all_scores and a.tokens_confidence == b.tokens_confidence


@pytest.mark.parametrize('beam', [1, 3, 64])
def test_rescore_edge_shapes_vs_oracle(beam):
    """Rescoring at the edges of its shape range against the oracle's attention_rescoring
    (search.py:374-458 restated): beam 1 (one hypothesis: the arg-max is trivial, the score is
    not), the widest beam the search supports (64: one lane per hypothesis in the reduce kernel),
    utterances of a single encoder frame next to long ones (their n-best holds the EMPTY
    hypothesis: only <eos> is scored), results of the free search function (another handle's
    prefix beam: the lists are uploaded, not read from this model's device block)."""
    from wenet_amd import search as SR, synthetic as S
    from oracle import wenet_oracle as O
    configs, sd, model = cached_model('tiny_causal', 0)
    feats, _ = S.make_features(4, (300, 300), seed=53)
    lens = torch.tensor([300, 7, 11, 163], dtype=torch.int32)     # 7 frames -> T' = 1
    for b in range(4):
        feats[b, int(lens[b]):] = 0.0
    kw = dict(beam_size=beam, ctc_weight=0.3, reverse_weight=0.5)
    M = ['ctc_prefix_beam_search', 'attention_rescoring']
    got = model.decode(M, feats.cuda(), lens, **kw)
    ref = O.decode(configs, sd, M, feats, lens, **kw)
    assert any(len(h) == 0 for r in got[M[0]] for h in r.nbest)
    for b in range(4):
        g, r = got[M[1]][b], ref[M[1]][b]
        assert [list(x) for x in got[M[0]][b].nbest] == [list(x) for x in ref[M[0]][b].nbest], b
        np.testing.assert_allclose(g.all_scores, r.all_scores, rtol=0, atol=1e-3)
        assert list(g.tokens) == list(r.tokens) and abs(g.score - r.score) < 1e-3
        assert abs(g.confidence - r.confidence) < 1e-4
        np.testing.assert_allclose(g.tokens_confidence, r.tokens_confidence, atol=1e-4)
        assert list(g.times) == list(r.times)
    # the free functions: prefix beam on another (workspace) handle, then rescoring by the model
    enc, mask = model._forward_encoder(feats.cuda(), lens)
    enc_lens = mask.squeeze(1).sum(1)
    logp = model.ctc_logprobs(enc, encoder_lens=enc_lens.cpu())
    pre = SR.ctc_prefix_beam_search(logp, enc_lens, beam)
    free = SR.attention_rescoring(model, pre, enc, enc_lens, 0.3, 0.5)
    for b in range(4):
        assert free[b].tokens == got[M[1]][b].tokens and free[b].score == got[M[1]][b].score
        assert free[b].all_scores == got[M[1]][b].all_scores


def test_rescore_rejects_a_result_list_of_another_batch_size():
    """wn_rescore writes one record per utterance of the HANDLE's batch (wn_batch_size): a
    prefix-result list of another length is refused in Python instead of being read / written
    past its arrays (round-4 advice)."""
    from wenet_amd import _lib, search as SR, synthetic as S
    configs, sd, model = cached_model('tiny_causal', 0)
    feats, lens = S.make_features(4, (80, 260), seed=5)
    enc, mask = model._forward_encoder(feats.cuda(), lens)
    enc_lens = mask.squeeze(1).sum(1)
    assert _lib.lib().wn_batch_size(model._h) == 4
    pre = model.decode(['ctc_prefix_beam_search'], feats.cuda(), lens,
                       beam_size=5)['ctc_prefix_beam_search']
    for r in pre:
        r.nbest
    with pytest.raises(ValueError, match='3 prefix beam results for a batch of 4'):
        SR.attention_rescoring(model, pre[:3], enc, enc_lens, 0.4, 0.0)
    ok = SR.attention_rescoring(model, pre, enc, enc_lens, 0.4, 0.0)
    assert len(ok) == 4


def test_prefetch_without_the_right_decoder_then_rescoring_with_it():
    """A C-API caller that prefetches the cross-attention K / V WITHOUT the right-to-left
    decoder and then rescoring with reverse_weight > 0: the prefetch is unusable, the pass
    projects in place -- ordered behind the side stream (it still reads the encoder output) --
    and gives the records of a run without any prefetch."""
    from wenet_amd import _lib, synthetic as S
    L = _lib.lib()
    configs, sd, model = cached_model('librispeech_bidecoder_large', 0)
    feats, lens = S.make_features(12, (500, 900), seed=31)
    kw = dict(beam_size=6, ctc_weight=0.5, reverse_weight=0.3)
    try:
        _lib.check(L.wn_tune_set(b'rescore_prefetch', 0), 'tune')
        ref = model.decode(['attention_rescoring'], feats.cuda(), lens, **kw)['attention_rescoring']
        _lib.check(L.wn_tune_set(b'rescore_prefetch', 1), 'tune')
        pre = model.decode(['ctc_prefix_beam_search'], feats.cuda(), lens,
                           beam_size=6)['ctc_prefix_beam_search']
        _lib.check(L.wn_rescore_prefetch(model._h, 0, torch.cuda.current_stream().cuda_stream),
                   'prefetch')
        got = model._rescore(pre, 0.5, 0.3, raw=model._last_prefix_raw)
        again = model._rescore(pre, 0.5, 0.3, raw=model._last_prefix_raw)
    finally:
        L.wn_tune_set(b'rescore_prefetch', 1)
    for a, b, c in zip(ref, got, again):
        assert a.tokens == b.tokens == c.tokens and a.score == b.score == c.score
        assert a.all_scores == b.all_scores == c.all_scores
