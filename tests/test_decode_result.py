"""DecodeResult (wenet/models/transformer/search.py:30-61): the n-best fields of a prefix-beam
result are built from the batch's raw arrays on first access -- what is read must be exactly what
the eager construction gave."""
import numpy as np

from wenet_amd.search import DecodeResult, _NBestBatch


def _batch(B=5, beam=4, T=9, seed=3):
    rng = np.random.default_rng(seed)
    n_hyps = rng.integers(1, beam + 1, (B, )).astype(np.int32)
    hyp_lens = rng.integers(0, T + 1, (B, beam)).astype(np.int32)
    hyp_tlens = hyp_lens.copy()
    hyp_tokens = rng.integers(1, 100, (B, beam, T)).astype(np.int32)
    hyp_times = rng.integers(0, 50, (B, beam, T)).astype(np.int32)
    hyp_scores = -rng.random((B, beam))
    return n_hyps, hyp_lens, hyp_tlens, hyp_tokens, hyp_times, hyp_scores


def test_lazy_nbest_equals_eager_lists():
    arrs = _batch()
    n_hyps, hyp_lens, hyp_tlens, hyp_tokens, hyp_times, hyp_scores = arrs
    batch = _NBestBatch(*arrs)
    for b in range(len(n_hyps)):
        r = DecodeResult(tokens=tuple(hyp_tokens[b, 0, :hyp_lens[b, 0]].tolist()),
                         score=float(hyp_scores[b, 0]),
                         times=hyp_times[b, 0, :hyp_tlens[b, 0]].tolist())
        r._lazy, r._b = batch, b
        n = int(n_hyps[b])
        want = [tuple(hyp_tokens[b, i, :hyp_lens[b, i]].tolist()) for i in range(n)]
        want_t = [hyp_times[b, i, :hyp_tlens[b, i]].tolist() for i in range(n)]
        assert r.nbest == want and isinstance(r.nbest[0], tuple)
        assert r.nbest_scores == hyp_scores[b, :n].tolist()
        assert r.nbest_times == want_t
        assert r.nbest[0] == r.tokens and r.nbest_times[0] == r.times
        assert r.nbest_scores[0] == r.score
        assert r.nbest is r.nbest            # built once


def test_plain_construction_and_assignment_keep_working():
    r = DecodeResult([1, 2, 3], 0.5, nbest=[[1, 2, 3]], nbest_scores=[0.5], nbest_times=[[4, 5, 6]])
    assert r.nbest == [[1, 2, 3]] and r.nbest_scores == [0.5] and r.nbest_times == [[4, 5, 6]]
    assert DecodeResult([7]).nbest is None
    arrs = _batch(B=2)
    r = DecodeResult((1, ))
    r._lazy, r._b = _NBestBatch(*arrs), 1
    r.nbest = 'mine'                        # assignment wins over the pending arrays
    assert r.nbest == 'mine' and r.nbest_scores == arrs[5][1, :int(arrs[0][1])].tolist()
    r.extra = 3                             # plain attributes stay assignable
    assert r.extra == 3


def test_as_dict_and_pickle_carry_the_reference_field_names():
    """vars() of the reference's DecodeResult lists its nine fields; here the n-best fields are
    properties, so `as_dict()` / pickling / copy.deepcopy give the same names with the lists
    materialised -- and never the batch-wide raw arrays."""
    import copy
    import pickle
    arrs = _batch(B=3)
    r = DecodeResult(tokens=(5, 6), score=-1.5, times=[1, 2])
    r._lazy, r._b = _NBestBatch(*arrs), 2
    r.all_scores = [0.25]
    d = r.as_dict()
    assert list(d)[:9] == ['tokens', 'score', 'confidence', 'tokens_confidence', 'times',
                           'nbest', 'nbest_scores', 'nbest_times', 'text']
    assert d['all_scores'] == [0.25] and not any(k.startswith('_') for k in d)
    n = int(arrs[0][2])
    assert d['nbest_scores'] == arrs[5][2, :n].tolist() and len(d['nbest']) == n
    blob = pickle.dumps(r)
    assert len(blob) < 4096                  # the (B, beam, T) arrays did not travel
    for q in (pickle.loads(blob), copy.deepcopy(r)):
        assert q.as_dict() == d and q._lazy is None
        assert q.nbest == r.nbest and q.tokens == (5, 6) and q.all_scores == [0.25]


def _naive(arrs):
    n_hyps, hyp_lens, hyp_tlens, hyp_tokens, hyp_times, hyp_scores = arrs
    out = []
    for b in range(len(n_hyps)):
        n = int(n_hyps[b])
        out.append(([tuple(hyp_tokens[b, i, :hyp_lens[b, i]].tolist()) for i in range(n)],
                    hyp_scores[b, :n].tolist(),
                    [hyp_times[b, i, :hyp_tlens[b, i]].tolist() for i in range(n)]))
    return out


def test_whole_batch_builders_agree_c_helper_numpy_pass_and_per_hypothesis_slicing():
    """Round 6: the three n-best list families of a WHOLE batch in one pass -- the C helper
    (wenet_amd/cext/nbest_lists.c) and the numpy pass it falls back to both give exactly what
    slicing every hypothesis out of the arrays gives (types included: token tuples, float
    scores, lists of frame indices), for ragged n_hyps, empty hypotheses, token lengths that
    differ from the time lengths, values beyond the helper's small-int table, non-contiguous
    views."""
    import pytest
    from wenet_amd import search
    from wenet_amd import build
    try:
        build.build_host_ext()
    except Exception as e:  # noqa: BLE001 -- no C compiler / Python headers on this host
        pytest.skip(f'host helper cannot be built here: {e}')
    import importlib
    importlib.reload(search) if search._nbest_lists is None else None
    assert search._nbest_lists is not None, 'host helper built but not importable'
    for seed, (B, beam, T) in enumerate([(5, 4, 9), (32, 10, 60), (1, 1, 1), (3, 64, 7)]):
        arrs = list(_batch(B, beam, T, seed))
        rng = np.random.default_rng(100 + seed)
        arrs[0] = rng.integers(0, beam + 1, (B, )).astype(np.int32)     # n_hyps incl. 0
        arrs[2] = rng.integers(0, T + 1, (B, beam)).astype(np.int32)    # tlens != lens
        arrs[3] = arrs[3] * 977                                          # ids above 8192
        want = _naive(arrs)
        got_c = search._NBestBatch(*arrs).all_utterances()
        got_np = search._NBestBatch(*arrs)._build_numpy()
        assert [tuple(x) for x in got_c] == [tuple(x) for x in want]
        assert [tuple(x) for x in got_np] == [tuple(x) for x in want]
        for rec in (got_c, got_np):
            for nb, ns, nt in rec:
                assert all(type(h) is tuple for h in nb) and all(type(t) is list for t in nt)
                assert all(type(s) is float for s in ns)
                assert all(type(v) is int for h in nb for v in h)
        # a sliced (non-contiguous) token array, as a caller's view may be
        wide = np.zeros((B, beam, T + 3), dtype=np.int32)
        wide[:, :, :T] = arrs[3]
        view = list(arrs)
        view[3] = wide[:, :, :T]
        assert [tuple(x) for x in search._NBestBatch(*view).all_utterances()] == \
            [tuple(x) for x in want]


def test_c_helper_rejects_short_buffers():
    import pytest
    from wenet_amd import search
    if search._nbest_lists is None:
        pytest.skip('helper not built')
    a = _batch(2, 3, 4)
    with pytest.raises(ValueError):
        search._nbest_lists.build(a[0], a[1], a[2], a[3], a[4], a[5], 2, 3, 5)
