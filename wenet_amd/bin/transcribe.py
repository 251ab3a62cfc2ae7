#!/usr/bin/env python3
"""Single-file transcription: the `wenet` command of the reference
(wenet/cli/transcribe.py:20-73) on the MI355X path.

    python -m wenet_amd.bin.transcribe audio.wav -m /path/to/model_dir

`model_dir` holds train.yaml, final.pt, units.txt (+ global_cmvn), like
`wenet.load_model` expects (cli/model.py:71-110); there is no model download
(no network) and no CPU device.  Like the reference's `main`, the text of
`model.transcribe(audio_file)` is printed; `--beam`, `--context_path` /
`--context_score` are honoured here (the reference parses and ignores them).
"""
import argparse
import sys


def get_args(argv=None):
    p = argparse.ArgumentParser(description='transcribe one wav file on MI355X')
    p.add_argument('audio_file', help='audio file to transcribe (PCM16 wav)')
    p.add_argument('-m', '--model', required=True, help='local model dir')
    p.add_argument('--device', default='cuda', choices=['cuda'],
                   help='only the MI355X path exists')
    p.add_argument('-t', '--show_tokens_info', action='store_true',
                   help='also print tokens, time stamps and confidences')
    p.add_argument('--beam', type=int, default=None,
                   help='beam size (default: the decode() default of transcribe)')
    p.add_argument('--context_path', type=str, default=None, help='context list file')
    p.add_argument('--context_score', type=float, default=6.0, help='context score')
    return p.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    import torch
    import wenet_amd
    model = wenet_amd.load_model(args.model, device=args.device)
    if args.beam is None and args.context_path is None:
        result = model.transcribe(args.audio_file)      # asr_model.py:345-358
    else:
        graph = None
        if args.context_path is not None:
            from wenet_amd.context_graph import ContextGraph
            bpe = getattr(model.tokenizer, 'bpe_path', None)
            graph = ContextGraph(args.context_path, model.tokenizer.symbol_table, bpe,
                                 args.context_score)
        speech = model.compute_feature(args.audio_file)
        method = model.default_decode_method
        result = model.decode([method], speech.unsqueeze(0),
                              torch.tensor([speech.size(0)]),
                              beam_size=args.beam or 10, context_graph=graph)[method][0]
        result.text = model.tokenizer.detokenize(result.tokens)[0]
    print(result.text)
    if args.show_tokens_info:
        print('tokens', list(result.tokens))
        print('times', result.times)
        print('tokens_confidence', result.tokens_confidence)
    return 0


if __name__ == '__main__':
    sys.exit(main())
