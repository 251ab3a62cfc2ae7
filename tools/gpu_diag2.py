import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from golden_util import build_inputs, load_case
from oracle import wenet_oracle as O
from wenet_amd import search as S
from gpu_util import cached_model
for name in ['aishell_full', 'aishell_chunk16']:
    meta, arrays = load_case(name)
    configs, sd, feats, lens = build_inputs(meta)
    with torch.no_grad():
        enc, mask = O.encoder_forward(configs, sd, feats, lens, meta['chunk'], meta['left'])
        enc_lens = mask.squeeze(1).sum(1)
        logp = O.ctc_logprobs(sd, enc)
    ref = O.ctc_prefix_beam_search(logp, enc_lens, meta['beam'])
    got = S.ctc_prefix_beam_search(logp.cuda(), enc_lens, meta['beam'])
    for b in range(meta['batch']):
        same = [list(x) for x in got[b].nbest] == [list(x) for x in ref[b].nbest]
        print(name, b, 'same-input nbest eq', same, 'max score diff', np.abs(np.array(got[b].nbest_scores) - np.array(ref[b].nbest_scores)).max())
        print('   golden scores', meta['prefix'][b]['nbest_scores'][:3], 'oracle', ref[b].nbest_scores[:3])
    # now the GPU model's own logp through the python oracle search
    _, _, model = cached_model(meta['config'], meta['wseed'])
    e2, m2 = model._forward_encoder(feats.cuda(), lens, meta['chunk'], meta['left'])
    lp2 = model.ctc_logprobs(e2, encoder_lens=enc_lens)
    print(name, 'logp max abs diff (valid frames)', max((lp2[b, :enc_lens[b]].cpu() - logp[b, :enc_lens[b]]).abs().max().item() for b in range(meta['batch'])))
    ref2 = O.ctc_prefix_beam_search(lp2.cpu(), enc_lens, meta['beam'])
    got2 = model.decode(['ctc_prefix_beam_search'], feats.cuda(), lens, beam_size=meta['beam'], decoding_chunk_size=meta['chunk'], num_decoding_left_chunks=meta['left'])['ctc_prefix_beam_search']
    for b in range(meta['batch']):
        print(name, b, 'gpu-logp: python-search vs gpu-search nbest eq', [list(x) for x in got2[b].nbest] == [list(x) for x in ref2[b].nbest],
              'scores', got2[b].nbest_scores[:3], ref2[b].nbest_scores[:3], 'golden', meta['prefix'][b]['nbest_scores'][:3])
        # topk comparison between lp2 full and decode path
