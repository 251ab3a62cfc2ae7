#!/usr/bin/env python3
"""Stage-by-stage error report of the HIP path against the oracle, written to
gpurun_out/diag.txt (debugging aid for the GPU box)."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
OUT = open(os.path.join(ROOT, 'gpurun_out', 'diag.txt'), 'w')


def P(*a):
    s = ' '.join(str(x) for x in a)
    print(s)
    OUT.write(s + '\n')
    OUT.flush()


def main():
    from oracle import wenet_oracle as O
    from wenet_amd import _lib, synthetic as S
    from wenet_amd.model import ASRModel
    P('device', torch.cuda.get_device_name(0), 'lib', _lib.lib().wn_version())
    for config, B, frames, chunk in [('tiny_sym', 3, (50, 200), -1),
                                     ('tiny_causal', 3, (50, 200), 4),
                                     ('aishell_u2pp', 2, (300, 420), -1)]:
        try:
            configs = S.make_configs(config)
            sd = S.make_state_dict(configs, 0)
            model = ASRModel(configs, sd)
            feats, lens = S.make_features(B, frames, seed=5)
            with torch.no_grad():
                ref, mask, layers = O.encoder_forward(configs, sd, feats, lens,
                                                      chunk, -1, return_layers=True)
            ref_lens = mask.squeeze(1).sum(1).numpy()
            L = _lib.lib()
            for n in range(len(layers)):
                L.wn_debug_set(model._h, b'n_layers', n)
                L.wn_debug_set(model._h, b'skip_after_norm', 1)
                enc, m = model._forward_encoder(feats.cuda(), lens, chunk, -1)
                enc = enc.cpu()
                errs = []
                for b in range(B):
                    nb = int(ref_lens[b])
                    errs.append((enc[b, :nb] - layers[n][b, :nb]).abs().max().item())
                P(config, 'chunk', chunk, 'layer', n, 'max|err| per utt',
                  ['%.2e' % e for e in errs], 'ref absmax %.2f' % layers[n].abs().max().item())
            L.wn_debug_set(model._h, b'n_layers', -1)
            L.wn_debug_set(model._h, b'skip_after_norm', 0)
            enc, m = model._forward_encoder(feats.cuda(), lens, chunk, -1)
            P(config, 'final enc err', ['%.2e' % (enc[b, :int(ref_lens[b])].cpu() - ref[b, :int(ref_lens[b])]).abs().max().item() for b in range(B)])
            t = time.time()
            got = model.decode(['ctc_greedy_search', 'ctc_prefix_beam_search', 'attention_rescoring'],
                               feats.cuda(), lens, beam_size=5, decoding_chunk_size=chunk,
                               ctc_weight=0.5, reverse_weight=0.3 if configs['decoder'] == 'bitransformer' else 0.0)
            P(config, 'decode wall %.3fs' % (time.time() - t))
            refd = O.decode(configs, sd, ['ctc_greedy_search', 'ctc_prefix_beam_search', 'attention_rescoring'],
                            feats, lens, beam_size=5, decoding_chunk_size=chunk, ctc_weight=0.5,
                            reverse_weight=0.3 if configs['decoder'] == 'bitransformer' else 0.0)
            for b in range(B):
                P(config, b, 'greedy eq', got['ctc_greedy_search'][b].tokens == refd['ctc_greedy_search'][b].tokens,
                  'prefix eq', [list(x) for x in got['ctc_prefix_beam_search'][b].nbest] == [list(x) for x in refd['ctc_prefix_beam_search'][b].nbest],
                  'score', got['ctc_prefix_beam_search'][b].score, refd['ctc_prefix_beam_search'][b].score,
                  'resc', got['attention_rescoring'][b].score, refd['attention_rescoring'][b].score)
        except Exception:
            P(config, 'FAILED'); P(traceback.format_exc())


if __name__ == '__main__':
    main()
