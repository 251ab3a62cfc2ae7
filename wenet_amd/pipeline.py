"""Keep several batches in flight on one GPU.

`ASRModel.decode()` is synchronous like the reference's
(asr_model.py:267-343): it returns Python results, so it ends with a
device-to-host copy and a stream sync.  Within ONE batch the CTC prefix beam
search is T' dependent steps on 32 workgroups while the other ~220 CUs idle,
and the host cannot queue the next batch's ~200 launches until the results are
back.  `DecodePipeline` runs `n_streams` decodes concurrently, each on its own
HIP stream and its own workspace handle (`wn_model_clone`: the weights are
shared), so the search / result copy of batch i overlaps the encoder of batch
i+1.  With one or two streams the encoders are chained with HIP events (encoder
i+1 starts on the GPU when encoder i has finished): the MFMA-bound GEMMs of two
batches started together only slow each other, while the latency-bound search
always has a full encoder to hide under.  With more streams (round 6: four to
eight) the encoders run free: enough decodes are in flight that their phases
drift apart and the chip always holds a mix of matrix-bound, HBM-bound and
latency-bound kernels -- +7.7 % at six streams over the chained pair (see
`__init__`).  Results are identical to `ASRModel.decode()` either way: every
batch runs the same kernels in the same order on its stream.

One host thread per stream drives the C ABI (ctypes drops the GIL inside the
calls); the reference's own multi-process sharding (tools/decode.sh:65-83) is
the precedent for running independent decodes side by side.

A wn_model handle belongs to one host thread at a time (its workspace, descriptor
staging and current batch are per-handle state), so the pipeline works on its OWN
clones only and never on the caller's handle: the caller keeps using `model`
(compute_fbank, resample, a direct decode()) from its thread while batches are
in flight.  The library enforces this: a second thread entering a busy handle
gets status -4 instead of corrupting it.
"""
import concurrent.futures
import gc
import os
import queue
import threading
from typing import List

import torch

from wenet_amd import _lib
from wenet_amd.model import ASRModel


def freeze_host_heap() -> int:
    """Call once a serving process is up (model loaded, first batches decoded): moves every
    object alive so far -- the modules, state dicts and tables of a loaded model are a large,
    static heap -- to the cyclic collector's permanent generation (`gc.freeze`), so that full
    collections stop re-walking it while the decode threads wait.  The collector stays
    enabled for everything allocated afterwards.  Measured with two decodes in flight at
    BASELINE configs[1]: without it about one round in seven of 20 decodes runs 25 % slower
    and the median is 2-4 % lower (docs/LOG_rounds1-3.md section 4).  Returns the number of frozen
    objects; `gc.unfreeze()` undoes it."""
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


class DecodePipeline:

    def __init__(self, model: ASRModel, n_streams: int = 2):
        assert n_streams >= 1
        self.device = model.device
        self._owner = model
        self.models: List[ASRModel] = [model.clone() for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=self.device)
                        for _ in range(n_streams)]
        self._free = queue.SimpleQueue()
        for i in range(n_streams):
            self._free.put(i)
        self._pool = concurrent.futures.ThreadPoolExecutor(
            max_workers=n_streams, thread_name_prefix='wn-decode')
        # encoder chain: launch order == execution order of the encoders
        self._enc_lock = threading.Lock()
        self._enc_done = None
        # The wait for the previous decode's encoder is handed to the library
        # (wn_model_set_encode_gate), which places it BEHIND its own descriptor uploads (five
        # small host -> device copies that otherwise stand in the encoder chain) and in front of
        # conv1 (tune enc_gate_pos = 0).  WN_PIPE_GATE=0: plain stream wait in front of the whole
        # call; WN_PIPE_GATE=2: behind CMVN + conv1 (+0.7 % while conv2 ran as 256-row tiles,
        # r10b; -0.6 % with conv2 as one launch of 128-row tiles: conv1 + conv2 = 0.97 ms then fit
        # under the previous decode's prefix beam search, 0.95 ms, which is over when the
        # single-round kernels behind conv2 start, r12p).
        # Chained or free-running encoders (round 6, r17b-r17d).  With TWO decodes in flight the
        # event chain wins (+2.4 %: two identical kernel sequences started together only fight
        # for the same CUs in the same phases).  From FOUR on the free-running form wins and
        # keeps winning up to ~6 (config 2, same box: 64.3 k chained at any depth; unchained
        # 66.7 k at 4, 67.8 k at 5, 69.3 k at 6, 69.1 k at 8): the decodes drift apart, and at
        # any moment the chip holds a mix of phases -- the power-limited matrix kernels of one
        # decode beside the HBM- and latency-bound phases of the others (reduce / LayerNorm
        # passes, row-block prologues and epilogues, the search) -- instead of every CU being in
        # the same phase.  Default: chained up to two streams, free-running above;
        # WN_PIPE_CHAIN=0 / 1 forces either.
        chain_env = os.environ.get('WN_PIPE_CHAIN')
        self.chain = (n_streams <= 2) if chain_env is None else chain_env != '0'
        gate_mode = os.environ.get('WN_PIPE_GATE', '1')
        self.gate_front_end = gate_mode != '0'
        if gate_mode == '2':
            for m in self.models:
                m.tune('enc_gate_pos', 1)

    def _run(self, ready: torch.cuda.Event, methods, speech, speech_lengths, kw):
        i = self._free.get()
        try:
            torch.cuda.set_device(self.device)
            stream = self.streams[i]
            stream.wait_event(ready)  # inputs produced on the caller's stream
            kw = dict(kw)
            ctc_weight = kw.pop('ctc_weight', 0.0)
            reverse_weight = kw.pop('reverse_weight', 0.0)
            length_penalty = kw.pop('length_penalty', 0.0)
            kw.pop('infos', None)
            with torch.cuda.stream(stream):
                with self._enc_lock:
                    gated = False
                    if self._enc_done is not None and self.chain:
                        if self.gate_front_end:
                            gated = True
                            # the library places the wait behind wn_encode's descriptor
                            # uploads and in front of conv1 (tune enc_gate_pos = 0, the
                            # default; 1 = behind CMVN + conv1).  The gate only ORDERS work
                            # for performance: every handle has its own workspace, so no
                            # result depends on it
                            _lib.check(_lib.lib().wn_model_set_encode_gate(
                                self.models[i]._h, self._enc_done.cuda_event), 'encode gate')
                        else:
                            stream.wait_event(self._enc_done)
                    try:
                        st = self.models[i]._decode_begin(methods, speech,
                                                          speech_lengths, **kw)
                    except BaseException:
                        # wn_encode may not have been reached: a gate left on the handle would
                        # make its NEXT encode wait for an event that no longer exists
                        if gated:
                            _lib.lib().wn_model_set_encode_gate(self.models[i]._h, None)
                        raise
                    done = torch.cuda.Event()
                    done.record(stream)
                    self._enc_done = done
                res = self.models[i]._decode_end(st, ctc_weight, reverse_weight,
                                                 length_penalty)
                # per-decode status the caller reads on ITS model (the clones are private)
                self._owner.last_non_blank_filter_empty = \
                    self.models[i].last_non_blank_filter_empty
                return res
        finally:
            self._free.put(i)

    def submit(self, methods, speech: torch.Tensor, speech_lengths: torch.Tensor,
               **kw) -> concurrent.futures.Future:
        """Queue one `decode()`; returns a Future of its result dict."""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        return self._pool.submit(self._run, ready, methods, speech,
                                 speech_lengths, kw)

    def decode_many(self, methods, batches, **kw):
        """decode() every (speech, speech_lengths) of `batches`, results in
        order."""
        futs = [self.submit(methods, s, l, **kw) for s, l in batches]
        return [f.result() for f in futs]

    def close(self):
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
