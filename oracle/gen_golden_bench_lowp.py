#!/usr/bin/env python3
"""Token-level goldens for BASELINE.json configs[4] in the dtypes it NAMES (bf16, MXFP8 FFN):
the oracle (oracle/wenet_oracle.py -- pinned bit for bit to the real reference's
TransformerEncoder in fp32, tests/golden/whisperenc_*.npz) run under the product's operand
rounding (`bf16_operands()`, `bf16_operands(fp8_ffn=True)`) on the exact bench batch
(Whisper-large-v3 encoder, 32 blocks, B = 16 x 3000 frames, 128 mel bins) + the CTC head.

    python oracle/gen_golden_bench_lowp.py [bf16] [fp8]     # build container, ~10 min of CPU each

-> tests/golden/bench_config5_{bf16,fp8}.npz: per valid frame the top-2 CTC log-probs and ids
(packed rows), the greedy token lists.  NOT reference outputs (the reference has no such mode:
autocast rounds every matmul RESULT to bf16 as well): they pin the accelerated path's reduced
modes to the oracle's restatement of the SAME arithmetic at the configured shape, frame by frame
-- a frame may differ from the golden only inside its own top-1 margin
(tests/test_gpu_bench_parity.py).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wenet_oracle as O  # noqa: E402


def main():
    from wenet_amd import synthetic as S
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    want = sys.argv[1:] or ['bf16', 'fp8']
    wl = S.BENCH_WORKLOADS['config5']
    configs = S.make_configs(wl['config'])
    sd = S.make_state_dict(configs, 0)
    feats, lens = S.make_bench_batch('config5', 1)
    B = feats.size(0)
    for mode in want:
        t0 = time.time()
        with torch.no_grad(), O.bf16_operands(sd, fp8_ffn=(mode == 'fp8')):
            enc, mask = O.encoder_forward(configs, sd, feats, lens, -1, -1)
            enc_lens = mask.squeeze(1).sum(1)
            logp = O.ctc_logprobs(sd, enc)
        greedy = O.ctc_greedy_search(logp, enc_lens)
        el = enc_lens.tolist()
        topv, topi = logp.topk(2, dim=-1)
        arrays = dict(
            enc_lens=enc_lens.numpy().astype(np.int32),
            row_off=np.concatenate([[0], np.cumsum(el)[:-1]]).astype(np.int32),
            ctc_top2_val=np.concatenate([topv[b, :el[b]].numpy() for b in range(B)]).astype(np.float32),
            ctc_top2_idx=np.concatenate([topi[b, :el[b]].numpy() for b in range(B)]).astype(np.int16))
        meta = dict(workload='config5', mode=mode, config=wl['config'], wseed=0, batch=wl['batch'],
                    lens=lens.tolist(), greedy=[list(map(int, r.tokens)) for r in greedy],
                    source='oracle/wenet_oracle.py under bf16_operands(fp8_ffn=%s)' % (mode == 'fp8'))
        arrays['meta'] = np.frombuffer(json.dumps(meta).encode('utf8'), dtype=np.uint8)
        path = os.path.join(ROOT, 'tests', 'golden', f'bench_config5_{mode}.npz')
        np.savez_compressed(path, **arrays)
        print(path, os.path.getsize(path) // 1024, 'KiB', f'{time.time() - t0:.0f} s',
              'greedy lens', [len(g) for g in meta['greedy']], flush=True)


if __name__ == '__main__':
    main()
