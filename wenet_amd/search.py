"""Search functions with the reference's names and signatures
(wenet/models/transformer/search.py:30-61,109,127,374), backed by the HIP
kernels of libwenet_amd (csrc/ctc.hip, csrc/model.hip).

`ctc_greedy_search` / `ctc_prefix_beam_search` take a (B, T, V) log-prob tensor
resident in HBM and return `DecodeResult`s; the per-frame top-k, the blank /
repeat collapse and the whole prefix-beam bookkeeping run on the GPU (fp64 like
the reference's Python floats).  `attention_rescoring` runs every hypothesis of
every utterance through the attention decoder(s) in ONE batched pass; the
reference's scalar fp32 score arithmetic, the arg-max and the confidences are one
more kernel on the gathered log-probs (wn_rescore) -- the host builds B records.
"""
import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from wenet_amd import _lib


class DecodeResult:
    """wenet/models/transformer/search.py:30-61 (same fields).

    The n-best fields of a prefix-beam result may be backed by the raw arrays of the
    whole batch (`_NBestBatch`) and turned into the reference's Python lists on first
    access: building 3 x B x beam lists per batch costs more host time than the GPU
    search of that batch takes, and a caller that reads only the 1-best (or hands the
    result to attention_rescoring later) should not wait for it before the next batch
    can be queued.  What is read is the same objects as before."""

    def __init__(self,
                 tokens: List[int],
                 score: float = 0.0,
                 confidence: float = 0.0,
                 tokens_confidence: List[float] = None,
                 times: List[int] = None,
                 nbest: List[List[int]] = None,
                 nbest_scores: List[float] = None,
                 nbest_times: List[List[int]] = None,
                 text: str = ''):
        self.tokens = tokens
        self.score = score
        self.confidence = confidence
        self.tokens_confidence = tokens_confidence
        self.times = times
        self._nbest = nbest
        self._nbest_scores = nbest_scores
        self._nbest_times = nbest_times
        self.text = text
        self._lazy = None
        self._b = 0

    def _fill(self):
        lazy = self._lazy
        if lazy is not None:
            self._lazy = None
            self._nbest, self._nbest_scores, self._nbest_times = lazy.utterance(self._b)

    _FIELDS = ('tokens', 'score', 'confidence', 'tokens_confidence', 'times', 'nbest',
               'nbest_scores', 'nbest_times', 'text')

    def as_dict(self):
        """The reference's attribute names -> values (what vars() of the reference's
        DecodeResult gives), the n-best lists materialised; extra attributes a search attached
        (e.g. `all_scores`) included."""
        d = {k: getattr(self, k) for k in self._FIELDS}
        d.update({k: v for k, v in self.__dict__.items()
                  if not k.startswith('_') and k not in d})
        return d

    # pickling / copy: by the reference's field names, never the batch-wide raw arrays
    def __getstate__(self):
        return self.as_dict()

    def __setstate__(self, state):
        self._lazy, self._b = None, 0
        self._nbest = state.pop('nbest', None)
        self._nbest_scores = state.pop('nbest_scores', None)
        self._nbest_times = state.pop('nbest_times', None)
        self.__dict__.update(state)

    @property
    def nbest(self):
        self._fill()
        return self._nbest

    @nbest.setter
    def nbest(self, v):
        self._fill()
        self._nbest = v

    @property
    def nbest_scores(self):
        self._fill()
        return self._nbest_scores

    @nbest_scores.setter
    def nbest_scores(self, v):
        self._fill()
        self._nbest_scores = v

    @property
    def nbest_times(self):
        self._fill()
        return self._nbest_times

    @nbest_times.setter
    def nbest_times(self, v):
        self._fill()
        self._nbest_times = v


try:                      # cext/nbest_lists.c, built by wenet_amd.build
    from wenet_amd import _nbest_lists
except ImportError:       # host-side list building only: the numpy pass below does the same
    _nbest_lists = None


class _NBestBatch:
    """The n-best arrays wn_ctc_prefix_beam_search filled for one batch; `utterance(b)`
    gives (nbest, nbest_scores, nbest_times) of utterance b as the reference's lists
    (search.py:30-61: token tuples, floats, lists of frame indices).  The three list
    families of the WHOLE batch are built once, on the first request, in one pass over the
    used elements: `_nbest_lists.build` (C, ~0.3 ms for 32 x 10 hypotheses), or -- when that
    helper has not been built -- one masked gather + one tolist() per family and slicing by
    offsets (a per-hypothesis ndarray.tolist() costs more than the GPU search)."""

    __slots__ = ('n_hyps', 'hyp_lens', 'hyp_tlens', 'hyp_tokens', 'hyp_times',
                 'hyp_scores', '_lists')

    def __init__(self, n_hyps, hyp_lens, hyp_tlens, hyp_tokens, hyp_times, hyp_scores):
        self.n_hyps, self.hyp_lens, self.hyp_tlens = n_hyps, hyp_lens, hyp_tlens
        self.hyp_tokens, self.hyp_times, self.hyp_scores = hyp_tokens, hyp_times, hyp_scores
        self._lists = None

    def _build_numpy(self):
        B, beam = self.hyp_lens.shape
        valid = np.arange(beam)[None, :] < self.n_hyps[:, None]
        lens = np.where(valid, self.hyp_lens, 0)
        tlens = np.where(valid, self.hyp_tlens, 0)

        def flat(arr, ln):
            L = max(int(ln.max(initial=0)), 1)
            mask = np.arange(L)[None, None, :] < ln[:, :, None]
            offs = np.zeros((B * beam + 1, ), dtype=np.int64)
            np.cumsum(ln.ravel(), out=offs[1:])
            return arr[:, :, :L][mask].tolist(), offs.tolist()

        ft, ot = flat(self.hyp_tokens, lens)
        fm, om = flat(self.hyp_times, tlens)
        sc_l, n_l = self.hyp_scores.tolist(), self.n_hyps.tolist()
        out = []
        for b in range(B):
            n, k0 = min(max(n_l[b], 0), beam), b * beam
            out.append(([tuple(ft[ot[k0 + i]:ot[k0 + i + 1]]) for i in range(n)], sc_l[b][:n],
                        [fm[om[k0 + i]:om[k0 + i + 1]] for i in range(n)]))
        return out

    def all_utterances(self):
        if self._lists is None:
            B, beam = self.hyp_lens.shape
            if _nbest_lists is not None and B > 0:
                c = np.ascontiguousarray
                self._lists = _nbest_lists.build(
                    c(self.n_hyps, np.int32), c(self.hyp_lens, np.int32),
                    c(self.hyp_tlens, np.int32), c(self.hyp_tokens, np.int32),
                    c(self.hyp_times, np.int32), c(self.hyp_scores, np.float64), B, beam,
                    self.hyp_tokens.shape[2])
            else:
                self._lists = self._build_numpy()
        return self._lists

    def utterance(self, b: int):
        return self.all_utterances()[b]


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _Workspace:
    """Weight-less wn_model handle per device for the free functions."""
    _by_device: Dict[int, int] = {}

    @classmethod
    def handle(cls, device: torch.device) -> int:
        idx = device.index if device.index is not None else \
            torch.cuda.current_device()
        if idx not in cls._by_device:
            import ctypes
            h = ctypes.c_void_p()
            _lib.check(_lib.lib().wn_workspace_create(idx, ctypes.byref(h)),
                       'wn_workspace_create')
            cls._by_device[idx] = h.value
        return cls._by_device[idx]


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f'{what}: wenet_amd runs on the GPU only (got a {t.device} tensor); '
            'there is no CPU fallback')


def _set_probs(ctc_probs: torch.Tensor, ctc_lens: torch.Tensor, topk: int):
    _require_cuda(ctc_probs, 'ctc search')
    assert ctc_probs.dim() == 3
    probs = ctc_probs.detach().to(torch.float32).contiguous()
    B, T, V = probs.shape
    lens = ctc_lens.detach().cpu().numpy().astype(np.int32)
    h = _Workspace.handle(probs.device)
    _lib.check(
        _lib.lib().wn_set_ctc_probs(h, probs.data_ptr(), _lib.i32p(lens), B, T,
                                    V, topk, _stream_ptr(probs.device)),
        'wn_set_ctc_probs')
    return h, B, T, probs


def _greedy(handle: int, B: int, max_len: int, blank_id: int,
            device) -> List[DecodeResult]:
    max_len = max(max_len, 1)
    tokens = np.zeros((B, max_len), dtype=np.int32)
    lens = np.zeros((B, ), dtype=np.int32)
    _lib.check(
        _lib.lib().wn_ctc_greedy_search(handle, blank_id, _lib.i32p(tokens),
                                        _lib.i32p(lens), max_len,
                                        _stream_ptr(device)),
        'wn_ctc_greedy_search')
    return [DecodeResult(tokens[b, :lens[b]].tolist()) for b in range(B)]


def _prefix_beam(handle: int, B: int, max_len: int, beam_size: int,
                 blank_id: int, device):
    max_len = max(max_len, 1)
    n_hyps = np.zeros((B, ), dtype=np.int32)
    hyp_lens = np.zeros((B, beam_size), dtype=np.int32)
    hyp_tlens = np.zeros((B, beam_size), dtype=np.int32)
    hyp_tokens = np.zeros((B, beam_size, max_len), dtype=np.int32)
    hyp_times = np.zeros((B, beam_size, max_len), dtype=np.int32)
    hyp_scores = np.zeros((B, beam_size), dtype=np.float64)
    _lib.check(
        _lib.lib().wn_ctc_prefix_beam_search(
            handle, beam_size, blank_id, _lib.i32p(n_hyps),
            _lib.i32p(hyp_lens), _lib.i32p(hyp_tlens), _lib.i32p(hyp_tokens),
            _lib.i32p(hyp_times), _lib.f64p(hyp_scores), max_len,
            _stream_ptr(device)), 'wn_ctc_prefix_beam_search')
    # eagerly only the 1-best of every utterance (one bulk conversion of the used corner of
    # the first hypotheses); the n-best lists are built on first access (_NBestBatch)
    batch = _NBestBatch(n_hyps, hyp_lens, hyp_tlens, hyp_tokens, hyp_times, hyp_scores)
    len0, tlen0 = hyp_lens[:, 0].tolist(), hyp_tlens[:, 0].tolist()
    tok0 = hyp_tokens[:, 0, :max(max(len0, default=0), 1)].tolist()
    tim0 = hyp_times[:, 0, :max(max(tlen0, default=0), 1)].tolist()
    sc0 = hyp_scores[:, 0].tolist()
    results = []
    for b in range(B):
        r = DecodeResult(tokens=tuple(tok0[b][:len0[b]]), score=sc0[b],
                         times=tim0[b][:tlen0[b]])
        r._lazy, r._b = batch, b
        results.append(r)
    raw = dict(n_hyps=n_hyps, hyp_lens=hyp_lens, hyp_tlens=hyp_tlens, hyp_tokens=hyp_tokens,
               hyp_times=hyp_times, hyp_scores=hyp_scores, max_len=max_len)
    return results, raw


def ctc_greedy_search(ctc_probs: torch.Tensor,
                      ctc_lens: torch.Tensor,
                      blank_id: int = 0) -> List[DecodeResult]:
    """search.py:109-124 on a (B, T, V) log-prob tensor in HBM."""
    h, B, T, _keep = _set_probs(ctc_probs, ctc_lens, 1)
    return _greedy(h, B, T, blank_id, ctc_probs.device)


def ctc_prefix_beam_search(ctc_probs: torch.Tensor,
                           ctc_lens: torch.Tensor,
                           beam_size: int,
                           context_graph=None,
                           blank_id: int = 0) -> List[DecodeResult]:
    """search.py:127-249 on a (B, T, V) log-prob tensor in HBM."""
    h, B, T, _keep = _set_probs(ctc_probs, ctc_lens, beam_size)
    if context_graph is None:
        return _prefix_beam(h, B, T, beam_size, blank_id, ctc_probs.device)[0]
    from wenet_amd import context_graph as cg
    sp = _stream_ptr(ctc_probs.device)
    cg.install(_lib.lib(), h, context_graph, sp)
    try:
        return _prefix_beam(h, B, T, beam_size, blank_id, ctc_probs.device)[0]
    finally:
        cg.install(_lib.lib(), h, None, sp)


def rescore_nbest(model, n_hyps, hyp_lens, hyp_tokens, hyp_scores, times_of, ctc_weight: float,
                  reverse_weight: float, from_beam: bool) -> List[DecodeResult]:
    """attention_rescoring (search.py:374-458) for the model's current batch through
    wn_rescore: decoder passes, the reference's fp32 score arithmetic, arg-max and
    confidences all run on the device; this function only shapes the B result records.
    `from_beam`: the n-best is the one the handle's last prefix beam search left in HBM (the
    arrays passed here are the host copies of the same search, used for the records only);
    otherwise the arrays are uploaded.  `times_of(b, i)` -> nbest_times[i] of utterance b."""
    B, beam = hyp_lens.shape
    max_len = hyp_tokens.shape[2]
    # wn_rescore writes one entry per utterance of the HANDLE's batch: a prefix-result list of
    # another length would be read / written past these arrays
    hb = int(_lib.lib().wn_batch_size(model._h))
    if hb != B:
        raise ValueError(f'attention_rescoring: {B} prefix beam results for a batch of {hb} '
                         'utterances')
    best = np.zeros((B, ), dtype=np.int32)
    score = np.zeros((B, ), dtype=np.float32)
    conf = np.zeros((B, ), dtype=np.float64)
    tok_conf = np.zeros((B, max_len), dtype=np.float64)
    all_scores = np.zeros((B, beam), dtype=np.float32)
    null_i, null_d = ctypes.POINTER(ctypes.c_int32)(), ctypes.POINTER(ctypes.c_double)()
    _lib.check(
        _lib.lib().wn_rescore(
            model._h, beam, null_i if from_beam else _lib.i32p(n_hyps),
            null_i if from_beam else _lib.i32p(hyp_lens),
            null_i if from_beam else _lib.i32p(hyp_tokens),
            null_d if from_beam else _lib.f64p(hyp_scores), max_len, float(ctc_weight),
            float(reverse_weight), _lib.i32p(best), _lib.f32p(score), _lib.f64p(conf),
            _lib.f64p(tok_conf), _lib.f32p(all_scores), _stream_ptr(model.device)),
        'wn_rescore')
    best_l, score_l, conf_l = best.tolist(), score.tolist(), conf.tolist()
    n_l = n_hyps.tolist()
    len_best = hyp_lens[np.arange(B), best].tolist() if B else []
    results = []
    for b in range(B):
        i, L = best_l[b], len_best[b]
        r = DecodeResult(tuple(hyp_tokens[b, i, :L].tolist()), score_l[b],
                         confidence=conf_l[b], times=times_of(b, i),
                         tokens_confidence=tok_conf[b, :L].tolist())
        r.all_scores = all_scores[b, :n_l[b]].tolist()  # extra: score of every hypothesis
        results.append(r)
    return results


def _nbest_arrays(ctc_prefix_results: List[DecodeResult]):
    """(n_hyps, hyp_lens, hyp_tokens, hyp_scores, times_of) of a list of prefix beam results:
    the raw arrays of the search when the records still carry them, else built from the
    records' Python lists (results of any other origin)."""
    lazies = [getattr(r, '_lazy', None) for r in ctc_prefix_results]
    nb = lazies[0] if lazies else None
    if nb is not None and all(z is nb for z in lazies) and \
            [r._b for r in ctc_prefix_results] == list(range(len(nb.n_hyps))):
        def times_of(b, i):
            return nb.hyp_times[b, i, :nb.hyp_tlens[b, i]].tolist()
        return nb.n_hyps, nb.hyp_lens, nb.hyp_tokens, nb.hyp_scores, times_of
    B = len(ctc_prefix_results)
    beam = max(max((len(r.nbest) for r in ctc_prefix_results), default=0), 1)
    max_len = max(max((len(h) for r in ctc_prefix_results for h in r.nbest), default=0), 1)
    n_hyps = np.zeros((B, ), dtype=np.int32)
    hyp_lens = np.zeros((B, beam), dtype=np.int32)
    hyp_tokens = np.zeros((B, beam, max_len), dtype=np.int32)
    hyp_scores = np.zeros((B, beam), dtype=np.float64)
    for b, r in enumerate(ctc_prefix_results):
        n_hyps[b] = len(r.nbest)
        for i, h in enumerate(r.nbest):
            hyp_lens[b, i] = len(h)
            hyp_tokens[b, i, :len(h)] = np.asarray(h, dtype=np.int32)
            hyp_scores[b, i] = r.nbest_scores[i]
    return (n_hyps, hyp_lens, hyp_tokens, hyp_scores,
            lambda b, i: ctc_prefix_results[b].nbest_times[i])


def attention_beam_search(model, batch_size: int, maxlen: int, beam_size: int = 10,
                          length_penalty: float = 0.0) -> List[DecodeResult]:
    """attention_beam_search (wenet/models/transformer/search.py:252-371, the
    non-Whisper branch) over the model's CURRENT batch (the encoder output of the last
    `_decode_begin` / `_forward_encoder` stays on the device).  The whole search --
    one decoder step per token with a self-attention cache, the finished-hypothesis
    masks, the beam x beam re-ranking and the length-penalised arg-max -- runs on the
    device (wn_attention_beam_search); this function only shapes the result."""
    assert maxlen >= 1
    tokens = np.zeros((batch_size, maxlen), dtype=np.int32)
    lens = np.zeros((batch_size, ), dtype=np.int32)
    _lib.check(
        _lib.lib().wn_attention_beam_search(model._h, beam_size, maxlen,
                                            float(length_penalty), _lib.i32p(tokens),
                                            _lib.i32p(lens), _stream_ptr(model.device)),
        'wn_attention_beam_search')
    return [DecodeResult(tokens[b, :lens[b]].tolist()) for b in range(batch_size)]


def attention_rescoring(model,
                        ctc_prefix_results: List[DecodeResult],
                        encoder_outs: torch.Tensor,
                        encoder_lens: torch.Tensor,
                        ctc_weight: float = 0.0,
                        reverse_weight: float = 0.0,
                        infos: Optional[dict] = None) -> List[DecodeResult]:
    """search.py:374-458.  `model` is a wenet_amd ASRModel; `encoder_outs` is
    the padded (B, T', d) encoder output in HBM."""
    _require_cuda(encoder_outs, 'attention_rescoring')
    model._set_encoder_out(encoder_outs, encoder_lens)
    return model._rescore(ctc_prefix_results, ctc_weight, reverse_weight)
