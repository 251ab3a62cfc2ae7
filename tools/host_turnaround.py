#!/usr/bin/env python3
"""Where the host time of a plain ASRModel.decode() goes (the gap between the end of one decode's
GPU work and the start of the next one's -- what two decodes in flight hide and a caller that keeps
the reference's loop, wenet/bin/recognize.py:289, pays).

Wraps every C-ABI call of the decode path with a wall-clock timer and prints, per call, the time
spent inside the library (for the synchronous searches that includes waiting for the GPU) and, for
the whole decode, the Python time outside the library.

    python tools/host_turnaround.py [config2|config3|config4] [n_decodes] [knob=value ...]
"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import _lib, synthetic as S  # noqa: E402
from wenet_amd.model import ASRModel  # noqa: E402

wlname = sys.argv[1] if len(sys.argv) > 1 else 'config2'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = S.BENCH_WORKLOADS[wlname]
for kv in sys.argv[3:]:
    k, v = kv.split('=')
    _lib.check(_lib.lib().wn_tune_set(k.encode(), int(v)), 'wn_tune_set ' + kv)
    print('knob', kv)
configs = S.make_configs(wl['config'])
model = ASRModel(configs, S.make_state_dict(configs, 0), device='cuda:0')
feats, lens = S.make_bench_batch(wlname, 1)
fd = feats.cuda()
kw = dict(beam_size=S.BENCH_BEAM, **wl['kw'])


class Timed:
    """ctypes function proxy that accumulates the time spent in the call"""
    acc = collections.OrderedDict()

    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        Timed.acc[self.name] = Timed.acc.get(self.name, 0.0) + time.perf_counter() - t0
        return r


class LibProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k.startswith('wn_') and callable(f) and k not in ('wn_last_error', ):
            return Timed(k, f)
        return f


for _ in range(5):
    model.decode([wl['method']], fd, lens, **kw)
torch.cuda.synchronize()
proxy = LibProxy(model._L)
model._L = proxy
import wenet_amd.search as search  # noqa: E402
real_lib = _lib.lib
_lib.lib = lambda: proxy          # the free functions of search.py fetch the library per call
try:
    t0 = time.perf_counter()
    for _ in range(n):
        res = model.decode([wl['method']], fd, lens, **kw)[wl['method']]
        toks = [r.tokens for r in res]        # what a caller reads
    total = time.perf_counter() - t0
finally:
    _lib.lib = real_lib
    model._L = real_lib()
inlib = sum(Timed.acc.values())
print(f'{wlname}: {n} plain decodes, {total / n * 1e3:.3f} ms each; inside the library '
      f'{inlib / n * 1e3:.3f} ms, Python outside it {(total - inlib) / n * 1e3:.3f} ms')
for k, v in Timed.acc.items():
    print(f'  {k:32s} {v / n * 1e6:9.1f} us per decode')
# the same loop with the searches replaced by a bare stream sync: GPU time of encoder + CTC head
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    st = model._decode_begin([wl['method']], fd, lens, beam_size=S.BENCH_BEAM,
                             **{k: v for k, v in wl['kw'].items()
                                if k in ('decoding_chunk_size', 'num_decoding_left_chunks')})
    torch.cuda.synchronize()
enc = time.perf_counter() - t0
print(f'  encoder + CTC head alone (queue + sync): {enc / n * 1e3:.3f} ms per batch')
