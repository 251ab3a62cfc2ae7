"""CPU tests that pin the oracle (oracle/wenet_oracle.py):
  * against the reference's own known-answer table for CTC prefix beam search
    (runtime/core/test/ctc_prefix_beam_search_test.cc:29-72),
  * against the mask docstring examples (wenet/utils/mask.py:109-113,212-216),
  * against the committed outputs of the real reference (tests/golden/),
  * and, when /root/reference is present, against the live reference.
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import needs_reference
from golden_util import (attention_case_names, build_inputs, case_names,
                         chunk_case_names, chunk_windows,
                         context_case_names, context_search_case_names, load_case,
                         stream_case_names,
                         whisper_case_names)
from oracle import wenet_oracle as O


def _fbank_cases():
    import glob
    import json
    import os
    out = []
    for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden',
                                           'fbank_*.npz'))):
        z = np.load(p)
        out.append((json.loads(bytes(z['meta']).decode()), z['feats']))
    return out


def test_fbank_matches_reference_cpp_golden():
    """oracle fbank vs the committed outputs of the reference's own C++ fbank
    (runtime/core/frontend/fbank.h built by oracle/Makefile,
    oracle/gen_golden_fbank.py)."""
    from wenet_amd import synthetic as S
    cases = _fbank_cases()
    assert len(cases) >= 4
    for meta, ref in cases:
        w = S.make_audio(meta['samples'], seed=meta['seed'])
        assert abs(float(np.sum(w.astype(np.float64))) - meta['wave_sum']) < 1e-6
        got = O.fbank(w)
        assert got.shape == ref.shape == (meta['frames'], 80)
        if ref.size:
            # fp32 radix-2 FFT (reference) vs fp64 rfft (oracle), in log-mel
            assert np.abs(got - ref).max() < 5e-4


def test_fbank_matches_live_reference_cpp():
    from oracle import ref_fbank
    if not ref_fbank.available():
        pytest.skip('oracle/_ref/libref_fbank.so not built (make -C oracle)')
    from wenet_amd import synthetic as S
    for n, seed in [(8000, 21), (401, 22), (33333, 23)]:
        w = S.make_audio(n, seed=seed)
        ref = ref_fbank.ref_fbank(w)
        got = O.fbank(w)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 5e-4


def test_slaney_mel_filter_properties():
    """The librosa mel matrix is third party and absent (parity unpinned): check
    its defining properties -- triangles on Slaney's scale, unit area (slaney
    norm), and the peak value openai-whisper's published 80-bin matrix has."""
    for n_mels in (80, 128):
        w = O.slaney_mel_filters(16000, 400, n_mels)
        assert w.shape == (n_mels, 201) and (w >= 0).all()
        # each triangle is unimodal with contiguous support
        for i in range(n_mels):
            nz = np.flatnonzero(w[i])
            assert nz.size >= 1 and (np.diff(nz) == 1).all()
            pk = int(np.argmax(w[i]))
            assert (np.diff(w[i, nz[0]:pk + 1]) >= 0).all()
            assert (np.diff(w[i, pk:nz[-1] + 1]) <= 0).all()
        # peaks move upwards in frequency
        assert (np.diff([int(np.argmax(r)) for r in w]) >= 0).all()
        # slaney norm: unit area in Hz where the 40 Hz bin grid resolves a triangle
        area = w.sum(1) * 40.0
        assert np.abs(area[n_mels // 2:] - 1.0).max() < 0.12
    assert abs(O.slaney_mel_filters(16000, 400, 80).max() - 0.025880) < 1e-5


def test_log_mel_shape_and_normalisation():
    from wenet_amd import synthetic as S
    x = S.make_audio(16000 * 2 + 77, seed=3)
    f = O.log_mel_spectrogram(x, 80)
    assert f.shape == ((16000 * 2 + 77) // 160, 80)
    # (log10 + 4) / 4 with an 8-decade floor: range is exactly 2.0 wide at most
    assert f.max() - f.min() <= 2.0 + 1e-6
    g = O.log_mel_spectrogram(x, 128, pad_or_trim=True)
    assert g.shape == (3000, 128)


def test_prefix_beam_known_answer():
    data = torch.tensor([[0.25, 0.40, 0.35], [0.40, 0.35, 0.25],
                         [0.10, 0.50, 0.40]]).log().unsqueeze(0)
    r = O.ctc_prefix_beam_search(data, torch.tensor([3]), 3)[0]
    assert [list(x) for x in r.nbest] == [[2, 1], [1, 2], [1]]
    np.testing.assert_allclose(np.exp(r.nbest_scores),
                               [0.2185, 0.1550, 0.1525], rtol=1e-5)
    assert r.nbest_times == [[0, 2], [0, 2], [2]]


def test_mask_examples():
    m = O.subsequent_chunk_mask(4, 2).int().tolist()
    assert m == [[1, 1, 0, 0], [1, 1, 0, 0], [1, 1, 1, 1], [1, 1, 1, 1]]
    p = O.make_pad_mask(torch.tensor([5, 3, 2])).int().tolist()
    assert p == [[0, 0, 0, 0, 0], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1]]


def test_greedy_collapse():
    assert O.remove_duplicates_and_blank([0, 1, 1, 0, 1, 2, 2, 0]) == [1, 1, 2]


def _check_against_meta(meta, arrays, res, enc, enc_lens, logp):
    np.testing.assert_array_equal(enc_lens.numpy(), arrays['enc_lens'])
    for b, n in enumerate(arrays['enc_lens']):
        np.testing.assert_allclose(enc[b, :n].numpy(), arrays['enc_out'][b, :n],
                                   rtol=0, atol=2e-5)
    k = arrays['ctc_topk_idx'].shape[-1]
    topv, _ = logp.topk(k, dim=-1)
    for b, n in enumerate(arrays['enc_lens']):
        np.testing.assert_allclose(topv[b, :n].numpy(),
                                   arrays['ctc_topk_val'][b, :n], atol=2e-4)
    for b in range(meta['batch']):
        assert res['ctc_greedy_search'][b].tokens == meta['greedy'][b]
        p = res['ctc_prefix_beam_search'][b]
        g = meta['prefix'][b]
        assert [list(x) for x in p.nbest] == g['nbest']
        assert [list(x) for x in p.nbest_times] == g['nbest_times']
        np.testing.assert_allclose(p.nbest_scores, g['nbest_scores'],
                                   rtol=0, atol=1e-3)
        r = res['attention_rescoring'][b]
        gr = meta['rescoring'][b]
        assert list(r.tokens) == gr['tokens']
        assert abs(r.score - gr['score']) < 1e-3
        assert abs(r.confidence - gr['confidence']) < 1e-4
        np.testing.assert_allclose(r.tokens_confidence,
                                   gr['tokens_confidence'], atol=1e-4)
        assert list(r.times) == gr['times']


@pytest.mark.parametrize('name', case_names())
def test_oracle_matches_committed_reference_outputs(name):
    meta, arrays = load_case(name)
    configs, sd, feats, lens = build_inputs(meta)
    torch.set_num_threads(8)
    with torch.no_grad():
        enc, mask = O.encoder_forward(configs, sd, feats, lens, meta['chunk'],
                                      meta['left'])
        enc_lens = mask.squeeze(1).sum(1)
        logp = O.ctc_logprobs(sd, enc)
    res = O.decode(configs, sd, ['ctc_greedy_search', 'ctc_prefix_beam_search',
                                 'attention_rescoring'], feats, lens,
                   beam_size=meta['beam'], decoding_chunk_size=meta['chunk'],
                   num_decoding_left_chunks=meta['left'],
                   ctc_weight=meta['ctc_weight'],
                   reverse_weight=meta['reverse_weight'])
    _check_against_meta(meta, arrays, res, enc, enc_lens, logp)


@pytest.mark.parametrize('name', whisper_case_names())
def test_oracle_whisper_encoder_matches_committed_reference_outputs(name):
    """TransformerEncoder (conv1d2 + abs_pos_whisper + gelu, key_bias=False)
    + CTC head + searches against the real reference's outputs
    (oracle/gen_golden_whisper.py)."""
    meta, arrays = load_case(name)
    configs, sd, feats, lens = build_inputs(meta)
    torch.set_num_threads(8)
    with torch.no_grad():
        enc, mask = O.encoder_forward(configs, sd, feats, lens)
        enc_lens = mask.squeeze(1).sum(1)
        logp = O.ctc_logprobs(sd, enc)
    np.testing.assert_array_equal(enc_lens.numpy(), arrays['enc_lens'])
    for b, n in enumerate(arrays['enc_lens']):
        np.testing.assert_allclose(enc[b, :n].numpy(), arrays['enc_out'][b, :n],
                                   rtol=0, atol=5e-5)
    greedy = O.ctc_greedy_search(logp, enc_lens)
    prefix = O.ctc_prefix_beam_search(logp, enc_lens, meta['beam'])
    for b in range(meta['batch']):
        assert greedy[b].tokens == meta['greedy'][b]
        g = meta['prefix'][b]
        assert [list(x) for x in prefix[b].nbest] == g['nbest']
        np.testing.assert_allclose(prefix[b].nbest_scores, g['nbest_scores'],
                                   rtol=0, atol=1e-3)


@pytest.mark.parametrize('name', stream_case_names())
def test_chunk_mask_path_equals_reference_cache_streaming(name):
    """The reference's forward_chunk_by_chunk (attention + conv caches,
    encoder.py:287-362; committed outputs) equals ONE pass under the chunk mask
    (the oracle's encoder_forward) -- the identity the accelerated
    `simulate_streaming=True` relies on."""
    from wenet_amd import synthetic as S
    meta, arrays = load_case(name)
    configs = S.make_configs(meta['config'])
    sd = S.make_state_dict(configs, meta['wseed'])
    feats, lens = S.make_features(1, (meta['frames'], meta['frames']),
                                  seed=meta['fseed'])
    torch.set_num_threads(8)
    with torch.no_grad():
        enc, mask = O.encoder_forward(configs, sd, feats, lens, meta['chunk'],
                                      meta['left'])
    assert enc.shape[1] == arrays['enc_out'].shape[0]
    np.testing.assert_allclose(enc[0].numpy(), arrays['enc_out'], rtol=0, atol=5e-5)
    res = O.decode(configs, sd, ['ctc_greedy_search', 'ctc_prefix_beam_search',
                                 'attention_rescoring'], feats, lens,
                   beam_size=meta['beam'], decoding_chunk_size=meta['chunk'],
                   num_decoding_left_chunks=meta['left'],
                   ctc_weight=meta['ctc_weight'],
                   reverse_weight=meta['reverse_weight'])
    assert res['ctc_greedy_search'][0].tokens == meta['greedy']
    assert [list(x) for x in res['ctc_prefix_beam_search'][0].nbest] == \
        meta['prefix']['nbest']
    assert list(res['attention_rescoring'][0].tokens) == meta['rescoring']['tokens']
    assert abs(res['attention_rescoring'][0].score - meta['rescoring']['score']) < 1e-3


@pytest.mark.parametrize('name', [n for n in attention_case_names() if 'aishell' not in n])
def test_oracle_attention_mode_matches_committed_reference_outputs(name):
    """attention_beam_search (search.py:252-371): the oracle's cache-free
    restatement against the reference's cached decoder steps."""
    meta, _ = load_case(name)
    configs, sd, feats, lens = build_inputs(meta)
    torch.set_num_threads(8)
    got = O.decode(configs, sd, ['attention'], feats, lens, beam_size=meta['beam'],
                   length_penalty=meta['length_penalty'])['attention']
    assert [list(r.tokens) for r in got] == meta['tokens']


@needs_reference
def test_oracle_whisper_encoder_matches_live_reference():
    from oracle import gen_golden_whisper
    from wenet_amd import synthetic as S
    configs = S.make_configs('whisper_tiny_like')
    sd = S.make_state_dict(configs, 5)
    enc, _ = gen_golden_whisper.build_reference_encoder(configs, sd)
    for B, fr, seed in [(3, (30, 71), 1), (2, (40, 41), 2), (4, (9, 33), 3)]:
        f, l = S.make_features(B, fr, seed=seed, feat_dim=configs['input_dim'])
        with torch.no_grad():
            ref, rmask = enc(f, l)
            got, gmask = O.encoder_forward(configs, sd, f, l)
        n = rmask.squeeze(1).sum(1)
        assert n.tolist() == gmask.squeeze(1).sum(1).tolist()
        for b in range(B):
            assert (ref[b, :n[b]] - got[b, :n[b]]).abs().max() < 1e-5


@needs_reference
@pytest.mark.parametrize('config', ['tiny_causal', 'tiny_sym', 'aishell_u2pp'])
def test_synthetic_state_dict_matches_reference_names(config):
    import argparse
    import copy
    from oracle import _ref_harness
    from wenet_amd import synthetic as S
    _ref_harness.install()
    from wenet.utils.init_model import init_model
    configs = S.make_configs(config)
    sd = S.make_state_dict(configs, 0)
    rc = copy.deepcopy(configs)
    rc['cmvn'] = None
    model, _ = init_model(argparse.Namespace(), rc)
    ref = model.state_dict()
    extra = {'encoder.global_cmvn.mean', 'encoder.global_cmvn.istd'}
    assert set(ref.keys()) | extra == set(sd.keys())
    for k, v in ref.items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    np.testing.assert_array_equal(ref['encoder.embed.pos_enc.pe'].numpy(),
                                  sd['encoder.embed.pos_enc.pe'].numpy())


@needs_reference
@pytest.mark.parametrize('config,chunk,left,beam', [
    ('tiny_causal', -1, -1, 7), ('tiny_causal', 3, 1, 2), ('tiny_sym', -1, -1, 4),
    ('tiny_sym', 5, -1, 3)])
def test_oracle_matches_live_reference(config, chunk, left, beam):
    from oracle import _ref_harness, gen_golden
    from wenet_amd import synthetic as S
    _ref_harness.install()
    configs = S.make_configs(config)
    sd = S.make_state_dict(configs, 3)
    model = gen_golden.build_reference_model(configs, sd)
    feats, lens = S.make_features(4, (50, 220), seed=21)
    methods = ['ctc_greedy_search', 'ctc_prefix_beam_search',
               'attention_rescoring']
    rw = 0.4 if configs['decoder'] == 'bitransformer' else 0.0
    with torch.no_grad():
        ref = model.decode(methods, feats, lens, beam_size=beam,
                           decoding_chunk_size=chunk,
                           num_decoding_left_chunks=left, ctc_weight=0.4,
                           reverse_weight=rw)
    got = O.decode(configs, sd, methods, feats, lens, beam_size=beam,
                   decoding_chunk_size=chunk, num_decoding_left_chunks=left,
                   ctc_weight=0.4, reverse_weight=rw)
    for b in range(4):
        assert ref['ctc_greedy_search'][b].tokens == \
            got['ctc_greedy_search'][b].tokens
        rp, gp = ref['ctc_prefix_beam_search'][b], \
            got['ctc_prefix_beam_search'][b]
        assert [list(x) for x in rp.nbest] == [list(x) for x in gp.nbest]
        assert rp.nbest_scores == gp.nbest_scores  # same fp64 arithmetic
        assert rp.nbest_times == gp.nbest_times
        rr, gr = ref['attention_rescoring'][b], got['attention_rescoring'][b]
        assert list(rr.tokens) == list(gr.tokens)
        assert abs(rr.score - gr.score) < 1e-5
        assert abs(rr.confidence - gr.confidence) < 1e-6


# --------------------------------------------------------------------------
# context biasing (ContextGraph + ctc_prefix_beam_search, search.py:127-249)


def _check_prefix_results(got, want):
    for b, r in enumerate(got):
        assert [list(h) for h in r.nbest] == want['nbest'][b], b
        assert [list(t) for t in r.nbest_times] == want['nbest_times'][b], b
        np.testing.assert_allclose(r.nbest_scores, want['nbest_scores'][b],
                                   rtol=0, atol=1e-9)


def test_context_graph_known_answer():
    """Hand-checked Aho-Corasick bonuses: phrases [1,2,3] and [2,3] at 2.0."""
    g = O.ContextGraph([[1, 2, 3], [2, 3]], 2.0)
    assert g.num_nodes == 5
    s1, n1 = g.forward_one_step(0, 1)
    s2, n2 = g.forward_one_step(n1, 2)
    s3, n3 = g.forward_one_step(n2, 3)
    # the third token completes [1,2,3] (6.0) AND the suffix phrase [2,3] (4.0)
    assert (s1, s2, s3) == (2.0, 2.0, 2.0 + 6.0 + 4.0)
    # a mismatch after [1,2] falls back to the root and takes the bonus back
    s, n = g.forward_one_step(n2, 7)
    assert (s, n) == (-4.0, 0)
    # ... unless a suffix still matches: [1,2] + 2 -> root -> [2]
    s, n = g.forward_one_step(n2, 2)
    assert s == 2.0 - 4.0 and g.node_score[n] == 2.0
    assert g.finalize(n2) == (-4.0, 0)


def test_context_graph_is_end_only_at_creation():
    """context_graph.py:160-171: [1,2] added after [1,2,3] never ends a phrase."""
    g = O.ContextGraph([[1, 2, 3], [1, 2]], 1.0)
    n = g.edges[g.edges[0][1]][2]
    assert not g.is_end[n] and g.output_score[n] == 0
    g2 = O.ContextGraph([[1, 2], [1, 2, 3]], 1.0)
    n = g2.edges[g2.edges[0][1]][2]
    assert g2.is_end[n] and g2.output_score[n] == 2.0


@pytest.mark.parametrize('name', context_search_case_names())
def test_oracle_context_search_matches_committed_reference_outputs(name):
    from wenet_amd import synthetic as S
    meta, _ = load_case(name)
    logp, lens = S.peaky_logprobs(meta['batch'], meta['frames'], meta['vocab'],
                                  meta['peak'], meta['seed'])
    g = O.ContextGraph(meta['phrases'], meta['context_score'])
    got = O.ctc_prefix_beam_search(logp, lens, meta['beam'], 0, g)
    _check_prefix_results(got, meta['prefix'])


@pytest.mark.parametrize('name', context_case_names())
def test_oracle_context_decode_matches_committed_reference_outputs(name):
    meta, _ = load_case(name)
    configs, sd, feats, lens = build_inputs(meta)
    torch.set_num_threads(8)
    g = O.ContextGraph(meta['phrases'], meta['context_score'])
    res = O.decode(configs, sd, ['ctc_prefix_beam_search', 'attention_rescoring'],
                   feats, lens, beam_size=meta['beam'], ctc_weight=meta['ctc_weight'],
                   reverse_weight=meta['reverse_weight'],
                   blank_penalty=meta['blank_penalty'], context_graph=g)
    for b, r in enumerate(res['ctc_prefix_beam_search']):
        assert [list(h) for h in r.nbest] == meta['prefix']['nbest'][b], b
        np.testing.assert_allclose(r.nbest_scores, meta['prefix']['nbest_scores'][b],
                                   rtol=0, atol=2e-3)
    assert [list(r.tokens) for r in res['attention_rescoring']] == meta['rescoring_tokens']
    np.testing.assert_allclose([r.score for r in res['attention_rescoring']],
                               meta['rescoring_scores'], rtol=0, atol=2e-3)


@needs_reference
def test_context_graph_matches_live_reference(tmp_path):
    """Graph arrays, every transition and the char tokenizer against
    wenet/utils/context_graph.py on random phrase lists."""
    from oracle import _ref_harness, gen_golden_context
    _ref_harness.install()
    from wenet.utils.context_graph import ContextGraph as RefGraph
    rng = np.random.RandomState(0)
    for trial in range(20):
        vocab = int(rng.randint(3, 9))
        phrases = [[int(t) for t in rng.randint(1, vocab, rng.randint(1, 6))]
                   for _ in range(rng.randint(1, 12))]
        ref = gen_golden_context.reference_graph(phrases, 1.5)
        g = O.ContextGraph(phrases, 1.5)
        assert g.num_nodes == ref.num_nodes
        # walk both automata in lock step over random token streams
        for _ in range(30):
            rs, s = ref.root, 0
            for tok in rng.randint(0, vocab + 1, 25):
                r_score, rs = ref.forward_one_step(rs, int(tok))
                o_score, s = g.forward_one_step(s, int(tok))
                assert r_score == o_score and rs.id == s
                assert ref.finalize(rs)[0] == g.finalize(s)[0]
    # tokenizer (char units): spaces, unknown symbols, <unk>
    table = {'<blank>': 0, '<unk>': 1, 'a': 2, 'b': 3, '\u2581': 4, '\u4f60': 5}
    lines = ['ab a\n', ' b\u4f60z \n', '\n', 'zzz\n']
    path = tmp_path / 'ctx.txt'
    path.write_text(''.join(lines), encoding='utf8')
    ref = RefGraph(str(path), table, None, 2.0)
    assert O.tokenize_context(lines, table) == ref.context_list
    table.pop('<unk>')
    ref = RefGraph(str(path), table, None, 2.0)
    assert O.tokenize_context(lines, table) == ref.context_list


@needs_reference
def test_oracle_context_search_matches_live_reference():
    from oracle import _ref_harness, gen_golden_context
    from wenet_amd import synthetic as S
    _ref_harness.install()
    from wenet.models.transformer.search import ctc_prefix_beam_search as ref_search
    rng = np.random.RandomState(7)
    for trial in range(6):
        vocab = int(rng.randint(5, 30))
        beam = int(rng.randint(1, min(vocab, 12)))
        logp, lens = S.peaky_logprobs(3, (10, 60), vocab, float(rng.uniform(1, 5)),
                                      100 + trial)
        phrases = [[int(t) for t in rng.randint(1, vocab, rng.randint(1, 5))]
                   for _ in range(rng.randint(1, 15))]
        cs = float(rng.choice([0.5, 2.0, 6.0]))
        ref = ref_search(logp, lens, beam, gen_golden_context.reference_graph(phrases, cs), 0)
        got = O.ctc_prefix_beam_search(logp, lens, beam, 0, O.ContextGraph(phrases, cs))
        for r, o in zip(ref, got):
            assert [tuple(h) for h in r.nbest] == [tuple(h) for h in o.nbest]
            assert r.nbest_scores == o.nbest_scores
            assert r.nbest_times == o.nbest_times


# --------------------------------------------------------------------------
# forward_chunk (attention / convolution caches), encoder.py:204-285


@pytest.mark.parametrize('name', chunk_case_names())
def test_oracle_forward_chunk_matches_committed_reference_outputs(name):
    from wenet_amd import synthetic as S
    meta, arrays = load_case(name)
    configs = S.make_configs(meta['config'])
    sd = S.make_state_dict(configs, meta['wseed'])
    feats, _ = S.make_features(1, (meta['frames'], meta['frames']), seed=meta['fseed'])
    torch.set_num_threads(8)
    att = cnn = None
    outs, offset = [], 0
    required = meta['chunk'] * meta['left']
    for i, (a, b) in enumerate(chunk_windows(meta['frames'], meta['chunk'])):
        y, att, cnn = O.forward_chunk(configs, sd, feats[:, a:b], offset, required,
                                      att, cnn)
        outs.append(y)
        offset += y.size(1)
        if i == meta['probe']:
            assert tuple(att.shape) == arrays['att_probe'].shape
            assert tuple(cnn.shape) == arrays['cnn_probe'].shape
            if att.numel():
                assert np.abs(att.numpy() - arrays['att_probe']).max() < 1e-4
            if cnn.numel():
                assert np.abs(cnn.numpy() - arrays['cnn_probe']).max() < 1e-4
    assert [int(y.size(1)) for y in outs] == meta['chunk_sizes']
    ys = torch.cat(outs, 1)[0].numpy()
    assert np.abs(ys - arrays['enc_out']).max() < 1e-4
    assert tuple(att.shape) == arrays['att_last'].shape
    if att.numel():
        assert np.abs(att.numpy() - arrays['att_last']).max() < 1e-4
    assert tuple(cnn.shape) == arrays['cnn_last'].shape
    if cnn.numel():
        assert np.abs(cnn.numpy() - arrays['cnn_last']).max() < 1e-4
    if meta['config'] != 'tiny_sym':
        by, _ = O.forward_chunk_by_chunk(configs, sd, feats, meta['chunk'], meta['left'])
        assert torch.equal(by[0], torch.cat(outs, 1)[0])


# --------------------------------------------------------------------------
# resample (processor.py:177-196; torchaudio absent -> properties only)


def test_resample_properties():
    """PARITY UNPINNED (see O.resample): the restated torchaudio algorithm is
    checked through what any correct band-limited resampler must satisfy."""
    from scipy.signal import resample_poly
    x = np.random.RandomState(0).randn(5000).astype(np.float32)
    assert O.resample(x, 16000, 16000) is not None
    np.testing.assert_array_equal(O.resample(x, 16000, 16000), x)
    for orig, new in [(8000, 16000), (44100, 16000), (48000, 16000), (22050, 16000),
                      (16000, 8000), (11025, 16000)]:
        n = 3 * orig // 4 + 13
        t = np.arange(n) / orig
        f0 = 0.1 * min(orig, new)          # well inside both pass bands
        tone = (0.5 * np.sin(2 * np.pi * f0 * t)).astype(np.float32)
        y = O.resample(tone, orig, new)
        assert y.dtype == np.float32
        assert y.shape[0] == -(-new * n // orig)          # ceil(new * n / orig)
        want = 0.5 * np.sin(2 * np.pi * f0 * np.arange(y.shape[0]) / new)
        edge = 200
        assert np.abs(y[edge:-edge] - want[edge:-edge]).max() < 2e-3, (orig, new)
        dc = O.resample(np.ones(n, np.float32), orig, new)
        assert np.abs(dc[edge:-edge] - 1.0).max() < 2e-3
        g = np.gcd(orig, new)
        poly = resample_poly(tone.astype(np.float64), new // g, orig // g)
        m = min(len(poly), len(y))
        assert np.abs(y[edge:m - edge] - poly[edge:m - edge]).max() < 5e-3
    # linear: resample(a x + b z) == a resample(x) + b resample(z)
    z = np.random.RandomState(1).randn(5000).astype(np.float32)
    lhs = O.resample(2 * x - 3 * z, 44100, 16000)
    rhs = 2 * O.resample(x, 44100, 16000) - 3 * O.resample(z, 44100, 16000)
    assert np.abs(lhs - rhs).max() < 1e-4


@pytest.mark.parametrize('name', ['tiny_lite_one', 'tiny_lite_equal', 'aishell_lite_one'])
def test_filter_blank_embedding_vs_reference_output(name):
    """O.filter_blank_embedding (asr_model.py:153-180) against what the REAL reference's method
    returned on its own encoder output / CTC posteriors (golden `nonblank_out`,
    `nonblank_kept`): same rows, same zero padding; `valid_lens` is a no-op when no frame is
    padding."""
    meta, arrays = load_case(name)
    enc = torch.from_numpy(arrays['enc_out'])
    configs, sd, feats, lens = build_inputs(meta)
    logp = O.ctc_logprobs(sd, enc)
    sel, mask = O.filter_blank_embedding(logp, enc)
    assert mask.squeeze(1).sum(1).tolist() == meta['nonblank_kept']
    np.testing.assert_array_equal(sel.numpy(), arrays['nonblank_out'])
    sel2, _ = O.filter_blank_embedding(logp, enc, torch.from_numpy(arrays['enc_lens']))
    assert torch.equal(sel, sel2)
    assert 0 < min(meta['nonblank_kept']) and max(meta['nonblank_kept']) < enc.shape[1]


@pytest.mark.parametrize('name', ['raggedlite_tiny', 'raggedlite_tiny_padded'])
def test_oracle_reproduces_the_reference_on_ragged_non_blank_embedding_batches(name):
    """apply_non_blank_embedding on a RAGGED batch, real reference (oracle/gen_golden.py): the
    reference's filter takes the arg-max over all maxlen frames -- in `raggedlite_tiny_padded`
    the shorter utterances keep 4 .. 23 rows that are padding -- and rescoring then slices with
    the unfiltered lengths.  The oracle's default mode reproduces that exactly (selected rows
    bit for bit, scores within 1e-3); `nonblank_valid_only=True` (what the accelerated path
    computes: it has no padded frames) differs by up to 0.1 in score on that case and not at all
    where no padded frame is selected."""
    meta, arrays = load_case(name)
    configs, sd, feats, lens = build_inputs(meta)
    torch.set_num_threads(8)
    with torch.no_grad():
        enc, mask = O.encoder_forward(configs, sd, feats, lens)
        logp = O.ctc_logprobs(sd, enc)
        sel, smask = O.filter_blank_embedding(logp, enc)
        _, vmask = O.filter_blank_embedding(logp, enc, mask.squeeze(1).sum(1))
    assert smask.squeeze(1).sum(1).tolist() == meta['nonblank_kept']
    np.testing.assert_allclose(sel.numpy(), arrays['nonblank_out'], rtol=0, atol=2e-5)
    padded_rows = [a - b for a, b in zip(meta['nonblank_kept'], vmask.squeeze(1).sum(1).tolist())]
    kw = dict(beam_size=meta['beam'], ctc_weight=meta['ctc_weight'],
              reverse_weight=meta['reverse_weight'])
    res = O.decode(configs, sd, ['attention_rescoring', 'ctc_prefix_beam_search'], feats, lens, **kw)
    val = O.decode(configs, sd, ['attention_rescoring', 'ctc_prefix_beam_search'], feats, lens,
                   nonblank_valid_only=True, **kw)
    dev = 0.0
    for b in range(meta['batch']):
        r, gr = res['attention_rescoring'][b], meta['rescoring'][b]
        assert list(r.tokens) == gr['tokens']
        assert abs(r.score - gr['score']) < 1e-3
        dev = max(dev, abs(val['attention_rescoring'][b].score - gr['score']))
    if name == 'raggedlite_tiny_padded':
        assert max(padded_rows) >= 20 and 1e-2 < dev < 0.15, (padded_rows, dev)
    else:
        assert max(padded_rows) == 0 and dev < 1e-3, (padded_rows, dev)


@pytest.mark.parametrize('orig,new', [(44100, 16000), (48000, 16000), (8000, 16000),
                                      (22050, 16000), (16000, 8000), (11025, 16000),
                                      (32000, 16000)])
def test_resample_vs_the_independent_fp64_definition(orig, new):
    """The pin of O.resample while torchaudio is unavailable: tests/golden/resample_*.npz is
    the sample-by-sample fp64 evaluation of the definition the published torchaudio kernel
    implements (oracle/gen_golden_resample.py: no polyphase table, no padding arithmetic, no
    fp32) -- a second implementation written from the same published source, so a slip in
    either one (gcd reduction, the (width, width + orig) padding, window clamp, scale, the
    ceil output length) shows up as a difference.  They agree to fp32 rounding (< 1e-6 on
    signals of peak 0.3; measured 3e-8 .. 8e-8)."""
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'resample_{orig}_{new}.npz'))
    assert int(z['orig']) == orig and int(z['new']) == new
    y = O.resample(z['x'], orig, new)
    assert y.shape == z['y'].shape
    assert np.abs(y.astype(np.float64) - z['y']).max() < 1e-6


@needs_reference
@pytest.mark.parametrize('config,rw', [('tiny_causal', 0.0), ('tiny_causal', 0.4),
                                       ('tiny_sym', 0.0)])
def test_forward_attention_decoder_matches_live_reference(config, rw):
    """All positions of the padded (N, L, V) outputs, padded rows included."""
    from oracle import gen_golden
    from wenet_amd import synthetic as S
    configs = S.make_configs(config)
    sd = S.make_state_dict(configs, 4)
    model = gen_golden.build_reference_model(configs, sd)
    sos, eos = O.special_symbols(configs)
    g = torch.Generator().manual_seed(9)
    V = configs['output_dim']
    lens = torch.tensor([7, 3, 1, 5])
    hyps = torch.full((4, 7), eos, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        hyps[i, 0] = sos
        hyps[i, 1:n] = torch.randint(1, V - 1, (n - 1, ), generator=g)
    enc = torch.randn(1, 23, configs['encoder_conf']['output_size'], generator=g)
    with torch.no_grad():
        r_l, r_r = model.forward_attention_decoder(hyps, lens, enc, rw)
    o_l, o_r = O.forward_attention_decoder(configs, sd, hyps, lens, enc, rw, sos, eos)
    assert r_l.shape == o_l.shape and (r_l - o_l).abs().max() < 1e-5
    assert r_r.shape == o_r.shape and (r_r - o_r).abs().max() < 1e-5


# --------------------------------------------------------------------------
# bf16-operand emulation (the product's WN_PREC_BF16 mode; tests/test_gpu_bf16.py
# compares the GPU against it)


def test_bf16_operand_emulation_is_scoped_and_selective():
    """bf16_operands() changes exactly the GEMM-kernel contractions and the
    attention products, leaves the module as it found it, and is idempotent on
    operands that are already bf16 numbers."""
    from wenet_amd import synthetic as S
    configs = S.make_configs('tiny_causal')
    sd = S.make_state_dict(configs, 0)
    feats, lens = S.make_features(3, (40, 120), seed=5)
    with torch.no_grad():
        a, _ = O.encoder_forward(configs, sd, feats, lens)
        with O.bf16_operands(sd):
            b, _ = O.encoder_forward(configs, sd, feats, lens)
            assert O._MM_ROUND and not isinstance(O.F, type(torch.nn.functional))
        c, _ = O.encoder_forward(configs, sd, feats, lens)
        with O.bf16_operands(sd, attention=False):
            d, _ = O.encoder_forward(configs, sd, feats, lens)
    assert torch.equal(a, c) and O.F is torch.nn.functional and not O._MM_ROUND
    assert not torch.equal(a, b) and not torch.equal(b, d)
    rel = (a - b).abs().max() / a.abs().max()
    assert 1e-4 < rel < 5e-2, rel          # bf16 operands: ~3 significant digits
    # selective: depthwise conv, the 1 -> d Conv2d and linear_pos stay fp32
    x = torch.randn(2, 8, 20)
    w = torch.randn(8, 1, 3)
    with O.bf16_operands(sd):
        assert torch.equal(O.F.conv1d(x, w, None, groups=8),
                           torch.nn.functional.conv1d(x, w, None, groups=8))
        img, k1 = torch.randn(1, 1, 9, 9), torch.randn(4, 1, 3, 3)
        assert torch.equal(O.F.conv2d(img, k1, None, stride=2),
                           torch.nn.functional.conv2d(img, k1, None, stride=2))
        wp = sd['encoder.encoders.0.self_attn.linear_pos.weight']
        pe = torch.randn(5, wp.shape[1])
        assert torch.equal(O.F.linear(pe, wp), torch.nn.functional.linear(pe, wp))
        # rounding is idempotent: bf16-valued operands give the exact fp32 product
        xb = torch.randn(7, 16).to(torch.bfloat16).float()
        wb = torch.randn(5, 16).to(torch.bfloat16).float()
        assert torch.equal(O.F.linear(xb, wb), torch.nn.functional.linear(xb, wb))
        assert torch.equal(O._mm(xb, wb.T), torch.matmul(xb, wb.T))


@needs_reference
@pytest.mark.parametrize('config', ['tiny_causal', 'tiny_sym'])
def test_bf16_operand_mode_is_at_least_as_close_to_fp32_as_reference_autocast(config):
    """The reference's `--dtype bf16` is torch autocast around decode()
    (wenet/bin/recognize.py:250-255,278-280), which rounds operands AND results of
    every linear / conv / matmul to bf16.  The product's bf16 mode rounds operands
    only; its emulation must sit closer to the fp32 encoder output than the real
    reference under CPU autocast does, and the two must agree to bf16 accuracy."""
    from oracle import _ref_harness, gen_golden
    from wenet_amd import synthetic as S
    _ref_harness.install()
    configs = S.make_configs(config)
    sd = S.make_state_dict(configs, 3)
    model = gen_golden.build_reference_model(configs, sd)
    feats, lens = S.make_features(4, (60, 200), seed=8)
    with torch.no_grad():
        ref32, mask = model._forward_encoder(feats, lens)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            ref_ac, _ = model._forward_encoder(feats, lens)
        with O.bf16_operands(sd):
            emu, _ = O.encoder_forward(configs, sd, feats, lens)
    n = mask.squeeze(1).sum(1)
    d_ac = max((ref_ac[b, :n[b]].float() - ref32[b, :n[b]]).abs().max().item()
               for b in range(4))
    d_emu = max((emu[b, :n[b]] - ref32[b, :n[b]]).abs().max().item() for b in range(4))
    d_x = max((emu[b, :n[b]] - ref_ac[b, :n[b]].float()).abs().max().item()
              for b in range(4))
    scale = ref32.abs().max().item()
    assert d_emu <= d_ac, (d_emu, d_ac)
    assert d_emu < 4e-2 * scale and d_x < 8e-2 * scale, (d_emu, d_x, scale)


def test_reverse_hyps_worked_example_of_the_reference():
    """The worked example in the reference's own comments
    (wenet/models/transformer/asr_model.py:487-536): hyps
    [[sos,1,2,3],[sos,9,8,4],[sos,2,eos,eos]] with lens [4,4,2] ->
    [[sos,3,2,1],[sos,4,8,9],[sos,2,eos,eos]] -- oracle and product host code."""
    from wenet_amd.model import reverse_hyps as product_reverse
    sos = eos = 11
    hyps = torch.tensor([[sos, 1, 2, 3], [sos, 9, 8, 4], [sos, 2, eos, eos]])
    lens = torch.tensor([4, 4, 2])
    want = [[sos, 3, 2, 1], [sos, 4, 8, 9], [sos, 2, eos, eos]]
    assert O.reverse_hyps(hyps, lens, eos).tolist() == want
    assert product_reverse(hyps, lens, eos).tolist() == want
    # reversing twice restores every hypothesis (padding stays eos)
    assert O.reverse_hyps(O.reverse_hyps(hyps, lens, eos), lens, eos).tolist() == hyps.tolist()
    g = torch.Generator().manual_seed(4)
    for _ in range(20):
        n, L = int(torch.randint(1, 6, (1, ), generator=g)), int(torch.randint(2, 9, (1, ), generator=g))
        lens = torch.randint(2, L + 1, (n, ), generator=g)
        lens[0] = L
        h = torch.randint(20, 90, (n, L), generator=g)
        h[:, 0] = sos
        for i in range(n):
            h[i, int(lens[i]):] = eos
        a, b = O.reverse_hyps(h, lens, eos), product_reverse(h, lens, eos)
        assert a.tolist() == b.tolist()
        for i in range(n):
            k = int(lens[i])
            assert a[i, 1:k].tolist() == h[i, 1:k].flip(0).tolist()
            assert a[i, k:].tolist() == [eos] * (L - k)


def test_mx_quantize_rule_known_answers():
    """MXFP8 block rule of the fp8 mode (csrc/mxfp8.h, restated in the oracle): the
    smallest power of two 2^e with amax <= 448 * 2^e; elements RNE to e4m3."""
    x = torch.zeros(4, 32)
    x[0, 0] = 448.0            # exactly representable with e = 0
    x[1, 0] = 449.0            # just above: e = 1
    x[2, 0] = 56.0             # 1.75 * 2^5 -> e = -3 (boundary stays inside)
    x[2, 1] = 1.0
    x[3, :] = 0.0              # all-zero block: E = 0, elements 0
    q, E = O.mx_quantize(x)
    assert E[:, 0].tolist() == [127, 128, 124, 0]
    d = O.mx_dequantize(q, E)
    assert d[0, 0] == 448.0 and d[1, 0] == 448.0   # 224.5 -> RNE 224 -> x 2
    assert d[2, 0] == 56.0 and d[2, 1] == 1.0
    assert float(d[3].abs().max()) == 0.0
    # round trip is idempotent and within half an e4m3 ulp of the block maximum's scale
    g = torch.Generator().manual_seed(3)
    y = torch.randn(7, 96, generator=g) * 5
    r = O.mx_round(y)
    assert torch.equal(O.mx_round(r), r)
    assert ((r - y).abs() <= 0.0625 * y.abs() + 2.0 ** -9 * y.abs().amax()).all()


def test_slaney_filters_window_and_whisper_norm_pinned_by_the_reference_cpp_frontend():
    """SURVEY 8a3 / VERDICT item 9: the pieces of compute_log_mel_spectrogram the
    reference tree CAN pin.  tests/golden/whisperfb_*.npz hold the output of the
    reference's own C++ frontend in its Whisper configuration
    (frontend/feature_pipeline.h:64-72, fbank.h:100-134,157-162,236-247), built by
    oracle/Makefile, and its Slaney filter bank / periodic Hanning window.  The
    oracle's slaney_mel_filters (evaluated on that 512-point grid), its window and
    its log10 / floor / max-8 / (x+4)/4 normalisation reproduce them; what stays
    unpinned is only the 400-point STFT framing of the Python path."""
    from golden_util import GOLDEN_DIR
    from wenet_amd import synthetic as S
    z = np.load(os.path.join(GOLDEN_DIR, 'whisperfb_filters.npz'))
    for bins, key in ((80, 'w80'), (128, 'w128')):
        mine = O.slaney_mel_filters(16000, 512, bins)[:, :256]
        assert np.abs(mine - z[key]).max() < 1e-6 * z[key].max() * 10
        assert (np.abs(z[key]).sum(1) > 0).all()
    assert np.abs(torch.hann_window(400).numpy() - z['window']).max() < 1e-6
    for name in ('a', 'b'):
        g = np.load(os.path.join(GOLDEN_DIR, f'whisperfb_{name}.npz'))
        wave = np.asarray(S.make_audio(int(g['n']), seed=int(g['seed'])), dtype=np.float32)
        mine = O.whisper_frontend_512(wave, int(g['bins']))
        assert mine.shape == g['feat'].shape
        assert np.abs(mine - g['feat']).max() < 2e-4, name
    # the Python-path function is built from the same pieces
    x = torch.rand(80, 50) * 3
    assert torch.equal(O.whisper_log_norm(x),
                       (torch.maximum(x.clamp(min=1e-10).log10(),
                                      x.clamp(min=1e-10).log10().max() - 8.0) + 4.0) / 4.0)


def test_whisper_frontend_pins_live_against_oracle_ref():
    from oracle import ref_fbank
    if not ref_fbank.has_whisper_frontend():
        pytest.skip('oracle/_ref not built with the Whisper frontend (make -C oracle)')
    from wenet_amd import synthetic as S
    wave = np.asarray(S.make_audio(16000 * 2 + 123, seed=9), dtype=np.float32)
    ref = ref_fbank.ref_whisper_fbank(wave, 80)
    assert np.abs(O.whisper_frontend_512(wave, 80) - ref).max() < 2e-4


def test_x6_plane_split_is_exact_and_the_dropped_products_are_below_fp32_rounding():
    """The arithmetic claim of csrc/gemm_x6.hip (restated in the oracle): the three bf16
    planes of an fp32 value sum to it exactly, and the three plane products the GEMM drops
    (a1b2 + a2b1 + a2b2) are below 2^-26 of |a b| -- a quarter of the half-ulp one fp32
    multiply-add rounds away -- so against fp64 the six-product GEMM is not worse than an
    fp32 GEMM."""
    g = torch.Generator().manual_seed(11)
    # (exact while the third plane stays a NORMAL bf16 number, |x| >= ~2^-108; below that
    # the split is off by less than 2^-133 absolute, the fp32 subnormal grid itself)
    x = torch.randn(50000, generator=g) * torch.logspace(-28, 30, 50000)
    x = x[x.abs() > 2.0 ** -100]
    x = torch.cat([x, torch.tensor([0.0, 1.0, -1.0, 3.0e38, 1.0 + 2.0 ** -23])])
    t0, t1, t2 = O.x6_planes(torch.tensor([1.0e-37, -3.0e-36]))
    assert ((t0.double() + t1.double() + t2.double()) -
            torch.tensor([1.0e-37, -3.0e-36]).double()).abs().max() < 2.0 ** -133
    x0, x1, x2 = O.x6_planes(x)
    for p in (x0, x1, x2):                       # every plane is a bf16 value
        assert torch.equal(p.to(torch.bfloat16).to(torch.float32), p)
    assert torch.equal((x0.double() + x1.double() + x2.double()).float(), x)
    assert torch.equal(x0.double() + x1.double() + x2.double(), x.double())   # exactly
    assert (x1.abs() <= 2.0 ** -8 * x.abs()).all() and (x2.abs() <= 2.0 ** -16 * x.abs()).all()
    a = torch.randn(64, 512, generator=g)
    w = torch.randn(48, 512, generator=g)
    ap, wp = O.x6_planes(a), O.x6_planes(w)
    dropped = (ap[1].double().abs() @ wp[2].double().abs().T +
               ap[2].double().abs() @ wp[1].double().abs().T +
               ap[2].double().abs() @ wp[2].double().abs().T)
    bound = 2.0 ** -26 * (a.double().abs() @ w.double().abs().T)
    assert (dropped <= bound).all()
    ref = a.double() @ w.double().T
    e6 = (O.x6_matmul(a, w) - ref).abs().max().item()
    e32 = ((a @ w.T).double() - ref).abs().max().item()
    assert e6 <= 2.0 ** -26 * float((a.abs() @ w.abs().T).max()) and e6 < e32


def test_relpos_fold_identity_of_the_attention_kernel():
    """The algebra csrc/encoder_kernels.hip (attention_kernel FOLD) relies on, checked against
    the oracle's own rel-pos scores (attention.py:410-428 restated, no rel_shift):
    (q + u).k_j + (q + v).p_j == q.(k_j + p_j) + (u.k_j + v.p_j) for every query / key / head."""
    g = torch.Generator().manual_seed(23)
    B, H, T, D = 2, 4, 37, 64
    q = torch.randn(B, H, T, D, generator=g, dtype=torch.float64)
    k = torch.randn(B, H, T, D, generator=g, dtype=torch.float64)
    p = torch.randn(1, H, T, D, generator=g, dtype=torch.float64)
    u = torch.randn(H, D, generator=g, dtype=torch.float64)
    v = torch.randn(H, D, generator=g, dtype=torch.float64)
    # the reference's two contractions
    ac = torch.matmul(q + u[None, :, None, :], k.transpose(-2, -1))
    bd = torch.matmul(q + v[None, :, None, :], p.transpose(-2, -1))
    ref = ac + bd
    # the folded form: one contraction against k + p, plus a scalar per key and head
    kp = k + p
    c = (k * u[None, :, None, :]).sum(-1) + (p * v[None, :, None, :]).sum(-1)   # (B, H, T)
    got = torch.matmul(q, kp.transpose(-2, -1)) + c[:, :, None, :]
    assert (got - ref).abs().max().item() < 1e-12
    # and in fp32 the two forms differ by reassociation only
    got32 = (torch.matmul(q.float(), kp.float().transpose(-2, -1)) + c.float()[:, :, None, :])
    assert (got32.double() - ref).abs().max().item() < 2e-4 * ref.abs().max().item()
