#!/usr/bin/env python3
"""Batch recognition on MI355X: the data side of `wenet/bin/recognize.py`.

Same command line (the subset that applies to the accelerated path), same
result files: for every mode a `<result_dir>/<mode>/text` with one
`<key> <text>` line per utterance, batches in list order and -- inside a batch
-- longest utterance first, exactly the order the reference's `padding`
produces (recognize.py:193-311, processor.py:526-577).

What differs is where the work runs:
  * wav files (PCM16) are read by host threads ahead of the GPU, rates other
    than 16 kHz are resampled on the device (`wn_resample`), fbank is computed
    on the device (`wn_fbank`) straight into the padded batch tensor;
  * two batches are in flight on the GPU (`wenet_amd.pipeline.DecodePipeline`):
    the CTC search of batch i runs under the encoder of batch i+1;
  * under `torchrun` every rank takes every world_size-th batch (the
    reference's `tools/decode.sh` split, without separate processes per list
    file) and rank 0 merges the per-rank parts into the final files.

    python -m wenet_amd.bin.recognize --config train.yaml --checkpoint final.pt \\
        --test_data data.list --result_dir out --modes ctc_prefix_beam_search \\
        attention_rescoring --batch_size 32 --beam_size 10 --ctc_weight 0.5
"""
import argparse
import concurrent.futures
import json
import logging
import os
import sys
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

MODES = ('attention', 'ctc_greedy_search', 'ctc_prefix_beam_search',
         'attention_rescoring')


def get_args(argv=None):
    p = argparse.ArgumentParser(description='recognize with a wenet_amd model')
    p.add_argument('--config', required=True, help='config file (train.yaml)')
    p.add_argument('--test_data', required=True, help='test data list (jsonl)')
    p.add_argument('--data_type', default='raw', choices=['raw', 'shard'])
    p.add_argument('--gpu', type=int, default=-1,
                   help='device index (default: LOCAL_RANK, else 0)')
    p.add_argument('--device', default='cuda', choices=['cuda'],
                   help='only the MI355X path exists; there is no CPU fallback')
    p.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                   help="model's compute dtype (recognize.py:52-56): bf16 = bf16 "
                   'operands, fp32 accumulate, on the bf16 matrix cores')
    p.add_argument('--num_workers', type=int, default=4,
                   help='host threads reading wav files ahead of the GPU')
    p.add_argument('--checkpoint', required=True, help='checkpoint model (.pt)')
    p.add_argument('--beam_size', type=int, default=10)
    p.add_argument('--length_penalty', type=float, default=0.0)
    p.add_argument('--blank_penalty', type=float, default=0.0)
    p.add_argument('--result_dir', required=True)
    p.add_argument('--batch_size', type=int, default=16)
    p.add_argument('--modes', nargs='+', required=True,
                   help='decoding modes: ' + ' '.join(MODES))
    p.add_argument('--ctc_weight', type=float, default=0.0)
    p.add_argument('--decoding_chunk_size', type=int, default=-1)
    p.add_argument('--num_decoding_left_chunks', type=int, default=-1)
    p.add_argument('--simulate_streaming', action='store_true')
    p.add_argument('--reverse_weight', type=float, default=0.0)
    p.add_argument('--override_config', action='append', default=[])
    p.add_argument('--context_bias_mode', type=str, default='',
                   help="'decoding-graph' enables context biasing")
    p.add_argument('--context_list_path', type=str, default='')
    p.add_argument('--context_graph_score', type=float, default=0.0)
    p.add_argument('--streams', type=int, default=2,
                   help='decode() calls in flight on the GPU')
    p.add_argument('--report_rtf', default='',
                   help='write a JSON report of the run to this file: audio seconds decoded, '
                   'wall time of wav -> text, audio-seconds / second, and where the main '
                   'thread waited (wav readers, resample, features, submit, results) -- the '
                   "throughput counterpart of the reference's RTF print "
                   '(runtime/core/bin/decoder_main.cc:52,64-70)')
    # Options of the reference's command line (recognize.py:86-190) that only its
    # transducer / HLG / LoRA modes read.  They are accepted so that existing decode
    # scripts run unchanged; the four modes on this path never look at them, exactly
    # as in the reference.  LoRA would change the weights and is refused.
    for name in ('--search_ctc_weight', '--search_transducer_weight',
                 '--transducer_weight', '--attn_weight', '--lm_scale', '--decoder_scale',
                 '--r_decoder_scale'):
        p.add_argument(name, type=float, default=1.0 if name.startswith('--search_ctc')
                       else 0.0, help='(transducer / HLG modes only: unused here)')
    p.add_argument('--word', default='', help='(HLG modes only: unused here)')
    p.add_argument('--hlg', default='', help='(HLG modes only: unused here)')
    p.add_argument('--use_lora', type=lambda v: str(v).lower() in ('1', 'true', 'yes'),
                   default=False)
    p.add_argument('--lora_ckpt_path', default=None)
    args = p.parse_args(argv)
    if args.use_lora:
        p.error('--use_lora: LoRA checkpoints are not supported on the accelerated path')
    for m in args.modes:
        if m not in MODES:
            p.error(f'mode {m!r} is not on the accelerated path (have: {MODES})')
    return args


def override_config(configs: dict, items: Sequence[str]) -> dict:
    """wenet/utils/config.py override_config: `a.b.c value` strings."""
    import yaml
    for item in items:
        arr = item.split()
        if len(arr) != 2:
            raise ValueError(f"the format of override_config is 'key value': {item!r}")
        keys, d = arr[0].split('.'), configs
        for i, k in enumerate(keys):
            if k not in d:
                raise KeyError(f'override_config: unknown key {arr[0]}')
            if i == len(keys) - 1:
                d[k] = yaml.load(arr[1], Loader=yaml.FullLoader)
            else:
                d = d[k]
    return configs


AUDIO_FORMAT_SETS = {'flac', 'mp3', 'm4a', 'ogg', 'opus', 'wav', 'wma'}  # processor.py:39


class TarMember:
    """One audio file inside a shard: read lazily by the reader threads."""

    __slots__ = ('tar', 'name', 'offset', 'size')

    def __init__(self, tar, name, offset, size):
        self.tar, self.name, self.offset, self.size = tar, name, offset, size

    def read(self) -> bytes:
        if self.offset is not None:        # plain tar: the member's bytes lie as-is
            with open(self.tar, 'rb') as f:
                f.seek(self.offset)
                return f.read(self.size)
        import tarfile
        with tarfile.open(self.tar, 'r:*') as t:   # compressed tar
            return t.extractfile(self.name).read()


def read_data_list(path: str, data_type: str = 'raw') -> List[Tuple[str, object]]:
    """(key, audio source) of every utterance in list order.

    `raw` lists (processor.parse_json + decode_wav, processor.py:66-70,125-153):
    one JSON object per line with `key`, `wav` and optionally `start` / `end`
    (seconds; a segment of the file); `txt` is unused here.  The source is the
    path, or (path, start, end).

    `shard` lists (datapipes.py:365-427 tar_file_and_group): one LOCAL tar path
    (or file:// URL) per line; inside, consecutive members `<key>.<audio ext>` and
    `<key>.txt` form one utterance (an utterance without audio is dropped, like
    the reference's `valid` flag).  The source is a TarMember.  Remote URLs
    (the reference pipes them through wget) are refused: no network on this path."""
    out = []
    with open(path, 'r', encoding='utf8') as f:
        lines = [ln.strip() for ln in f]
    if data_type == 'raw':
        for ln, line in enumerate(lines, 1):
            if not line:
                continue
            obj = json.loads(line)
            if 'key' not in obj or 'wav' not in obj:
                raise ValueError(f'{path}:{ln}: need "key" and "wav"')
            if 'start' in obj:
                if 'end' not in obj:
                    raise ValueError(f'{path}:{ln}: "start" without "end"')
                out.append((obj['key'], (obj['wav'], float(obj['start']),
                                         float(obj['end']))))
            else:
                out.append((obj['key'], obj['wav']))
        return out
    import tarfile
    from urllib.parse import urlparse
    for ln, line in enumerate(lines, 1):
        if not line:
            continue
        pr = urlparse(line)
        if pr.scheme not in ('', 'file'):
            raise ValueError(f'{path}:{ln}: only local shards are supported, got {line}')
        tar_path = pr.path if pr.scheme == 'file' else line
        with open(tar_path, 'rb') as fh:
            plain = fh.read(262)[257:262] == b'ustar'  # else: compressed (or v7) tar
        with tarfile.open(tar_path, 'r:*') as t:
            prev, member = None, None
            for ti in t:
                if not ti.isfile():
                    continue
                pos = ti.name.rfind('.')
                assert pos > 0, f'{tar_path}: member {ti.name!r} has no extension'
                prefix, postfix = ti.name[:pos], ti.name[pos + 1:]
                if prev is not None and prefix != prev:
                    if member is not None:
                        out.append((prev, member))
                    member = None
                if postfix in AUDIO_FORMAT_SETS:
                    member = TarMember(tar_path, ti.name,
                                       ti.offset_data if plain else None, ti.size)
                prev = prefix
            if prev is not None and member is not None:
                out.append((prev, member))
    return out


def static_batches(entries: Sequence, batch_size: int) -> List[List]:
    """processor.static_batch (processor.py:470-486): consecutive groups."""
    assert batch_size > 0
    return [list(entries[i:i + batch_size]) for i in range(0, len(entries), batch_size)]


def padding_order(n_frames: Sequence[int]) -> List[int]:
    """Order of the utterances inside a batch: feature length descending
    (processor.padding, processor.py:539-541), ties in list order."""
    return sorted(range(len(n_frames)), key=lambda i: (-int(n_frames[i]), i))


def check_feature_conf(configs: dict):
    """The feature recipes this path computes on the device: Kaldi fbank (25 ms /
    10 ms, input_dim bins) and the Whisper log-mel (n_fft 400, hop 160)."""
    dc = configs.get('dataset_conf', {})
    feats_type = dc.get('feats_type', 'fbank')
    if feats_type == 'log_mel_spectrogram':
        lc = dc.get('log_mel_spectrogram_conf', {})
        if (lc.get('n_fft', 400) != 400 or lc.get('hop_length', 160) != 160
                or lc.get('num_mel_bins', 80) != configs.get('input_dim', 80)):
            raise NotImplementedError('recognize: log_mel_spectrogram must be n_fft 400 / '
                                      'hop_length 160 with input_dim mel bins')
    elif feats_type != 'fbank':
        raise NotImplementedError('recognize: fbank or log_mel_spectrogram features only')
    else:
        fc = dc.get('fbank_conf', {})
        if (fc.get('num_mel_bins', 80) != configs.get('input_dim', 80)
                or fc.get('frame_length', 25) != 25 or fc.get('frame_shift', 10) != 10):
            raise NotImplementedError('recognize: fbank must be 25 ms / 10 ms frames with '
                                      'input_dim mel bins')
    if dc.get('resample_conf', {}).get('resample_rate', 16000) != 16000:
        raise NotImplementedError('recognize: 16 kHz models only')


def feature_function(model, configs: dict):
    """waveform list -> (padded features in HBM, lengths): dataset_conf.feats_type
    with its conf, like the reference's dataset pipeline (dataset.py:100-118)."""
    dc = configs.get('dataset_conf', {})
    if dc.get('feats_type', 'fbank') == 'log_mel_spectrogram':
        lc = dc.get('log_mel_spectrogram_conf', {})
        kw = dict(num_mel_bins=lc.get('num_mel_bins', 80), padding=lc.get('padding', 0),
                  pad_or_trim=lc.get('pad_or_trim', False),
                  max_duration=lc.get('max_duration', 30))
        return lambda waves: model.compute_log_mel_spectrogram(waves, **kw), None
    # Kaldi fbank, snip_edges: frames of a waveform are known from its length, so a batch can
    # be put into `padding`'s order BEFORE the features are computed (no device-side reorder)
    return model.compute_fbank, lambda n: 0 if n < 400 else 1 + (n - 400) // 160


def format_line(key: str, text: str) -> str:
    return '{} {}'.format(key, text)  # recognize.py:304-306


def merge_parts(result_dir: str, modes: Iterable[str], world_size: int, n_batches: int):
    """Rank files `<mode>/text.part<r>` hold `<batch index>\\t<line>` records;
    write `<mode>/text` in batch order."""
    for mode in modes:
        d = os.path.join(result_dir, mode)
        per_batch: Dict[int, List[str]] = {}
        for r in range(world_size):
            part = os.path.join(d, f'text.part{r}')
            with open(part, 'r', encoding='utf8') as f:
                for rec in f:
                    bi, line = rec.rstrip('\n').split('\t', 1)
                    per_batch.setdefault(int(bi), []).append(line)
            os.remove(part)
        with open(os.path.join(d, 'text'), 'w', encoding='utf8') as f:
            for bi in range(n_batches):
                for line in per_batch.get(bi, []):
                    f.write(line + '\n')


def load_state(configs: dict, checkpoint: str):
    """Checkpoint tensors (+ the global CMVN the reference builds from
    cmvn_conf, init_model.py:100-111)."""
    import torch
    from wenet_amd.model import load_cmvn
    sd = dict(torch.load(checkpoint, map_location='cpu', mmap=True))
    if configs.get('cmvn') == 'global_cmvn' and 'encoder.global_cmvn.mean' not in sd:
        cc = configs['cmvn_conf']
        mean, istd = load_cmvn(cc['cmvn_file'], cc['is_json_cmvn'])
        sd['encoder.global_cmvn.mean'] = torch.from_numpy(mean).float()
        sd['encoder.global_cmvn.istd'] = torch.from_numpy(istd).float()
    return sd


def recognize(model, tokenizer, batches: List[List[Tuple[str, str]]], my_batches,
              args, blank_id: int, context_graph, emit):
    """Decode `my_batches` (indices into `batches`); `emit(batch_index, mode,
    line)` receives the result lines in order."""
    import torch
    from wenet_amd.model import read_wav

    def read16k(src):  # host threads read; rates other than 16 kHz go through
        if isinstance(src, TarMember):          # the device resampler below
            return read_wav(src.read(), return_rate=True)
        if isinstance(src, tuple):              # (path, start, end) segment
            return read_wav(src[0], return_rate=True, start=src[1], end=src[2])
        return read_wav(src, return_rate=True)
    from wenet_amd.pipeline import DecodePipeline, freeze_host_heap
    kw = dict(beam_size=args.beam_size,
              decoding_chunk_size=args.decoding_chunk_size,
              num_decoding_left_chunks=args.num_decoding_left_chunks,
              ctc_weight=args.ctc_weight,
              simulate_streaming=args.simulate_streaming,
              reverse_weight=args.reverse_weight, context_graph=context_graph,
              blank_id=blank_id, blank_penalty=args.blank_penalty,
              length_penalty=args.length_penalty)
    max_fmt = max(len(m) for m in args.modes)
    compute_features, frames_of = feature_function(model, model.configs)
    depth = max(2, 2 * args.streams)  # batches of wav data read ahead
    readers = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, args.num_workers))

    model.stage_pool = readers      # the PCM staging copies of a batch run on the reader pool too

    def load(bi):
        return [readers.submit(read16k, wav) for _, wav in batches[bi]]

    import time
    clock = time.perf_counter
    stat = dict(wav_wait=0.0, resample=0.0, features=0.0, pad_sort=0.0, submit=0.0,
                result_wait=0.0, detokenize=0.0, audio_s=0.0, utts=0, batches=0)
    t_begin = clock()
    pending_wavs = {}
    order = list(my_batches)
    for bi in order[:depth]:
        pending_wavs[bi] = load(bi)
    inflight = []  # (batch index, keys in output order, future)

    def drain(n_keep):
        while len(inflight) > n_keep:
            bi, keys, fut = inflight.pop(0)
            t0 = clock()
            results = fut.result()
            stat['result_wait'] += clock() - t0
            t0 = clock()
            for i, key in enumerate(keys):
                for mode in args.modes:
                    text = tokenizer.detokenize(results[mode][i].tokens)[0]
                    line = format_line(key, text)
                    logging.info('%s %s', mode.ljust(max_fmt), line)
                    emit(bi, mode, line)
            stat['detokenize'] += clock() - t0

    # Two host threads (round 6): the FEEDER waits for the wav readers, stages the PCM, launches
    # the feature kernel and submits the batch (numpy's bulk copies and the C calls drop the
    # GIL); the calling thread takes the results in order, detokenizes and emits.  One thread
    # doing both spent 2.5-2.9 ms per batch in `features` and ~0.9 ms in `detokenize` back to
    # back -- more than the ~4.7 ms the GPU needs for a batch once its waits are added (r15e /
    # r18d: wav -> text 52-54 k against 66-69 k with resident features).
    import queue
    import threading
    handoff = queue.Queue(maxsize=max(1, args.streams))   # bounds the batches in flight
    feeder_err = []

    def feed(pipe):
        try:
            torch.cuda.set_device(model.device)
            for pos, bi in enumerate(order):
                t0 = clock()
                raw = [f.result() for f in pending_wavs.pop(bi)]
                t1 = clock()
                stat['wav_wait'] += t1 - t0
                stat['audio_s'] += sum(len(w) / float(sr) for w, sr in raw)
                stat['utts'] += len(raw)
                stat['batches'] += 1
                waves = [w if sr == 16000 else model.resample(w, sr, 16000) for w, sr in raw]
                t2 = clock()
                stat['resample'] += t2 - t1
                if pos + depth < len(order):
                    nxt = order[pos + depth]
                    pending_wavs[nxt] = load(nxt)
                if frames_of is not None:
                    # longest first (processor.padding) on the host, then ONE feature launch
                    # straight into the padded batch tensor
                    perm = padding_order([frames_of(len(w)) for w in waves])
                    waves = [waves[i] for i in perm]
                    t3 = clock()
                    stat['pad_sort'] += t3 - t2
                    feats, lens = compute_features(waves)
                    t4 = clock()
                    stat['features'] += t4 - t3
                else:
                    feats, n_frames = compute_features(waves)
                    t3 = clock()
                    stat['features'] += t3 - t2
                    perm = padding_order(n_frames.tolist())
                    idx = torch.as_tensor(perm, dtype=torch.long)
                    feats = feats.index_select(0, idx.to(feats.device))
                    lens = n_frames.index_select(0, idx)
                    tmax = int(lens.max()) if len(perm) else 0
                    feats = feats[:, :tmax].contiguous()
                    t4 = clock()
                    stat['pad_sort'] += t4 - t3
                keys = [batches[bi][i][0] for i in perm]
                fut = pipe.submit(args.modes, feats, lens, **kw)
                t5 = clock()
                stat['submit'] += t5 - t4
                handoff.put((bi, keys, fut))          # blocks while `streams` batches wait
                stat['feeder_blocked'] = stat.get('feeder_blocked', 0.0) + clock() - t5
        except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
            feeder_err.append(e)
        finally:
            handoff.put(None)

    with DecodePipeline(model, n_streams=args.streams) as pipe:
        feeder = threading.Thread(target=feed, args=(pipe, ), name='wn-feed', daemon=True)
        feeder.start()
        n_done = 0
        while True:
            t0 = clock()
            item = handoff.get()
            if item is None:
                break
            inflight.append(item)
            stat['result_wait'] += clock() - t0      # (waiting for the feeder)
            drain(0)
            n_done += 1
            if n_done == 3:
                # the process is up: stop the cyclic collector from re-walking the model's
                # static heap on every full collection (pipeline.freeze_host_heap)
                freeze_host_heap()
        feeder.join()
        if feeder_err:
            raise feeder_err[0]
    model.stage_pool = None
    readers.shutdown(wait=True)
    stat['wall_s'] = clock() - t_begin
    return stat


def main(argv=None):
    args = get_args(argv)
    logging.basicConfig(level=logging.INFO,
                        format='%(asctime)s %(levelname)s %(message)s')
    # dmabuf IPC only on these hosts: RCCL needs this before the runtime starts
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import yaml
    from wenet_amd.model import ASRModel
    from wenet_amd.tokenizer import get_blank_id, init_tokenizer

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = args.gpu if args.gpu >= 0 else local
    device = torch.device('cuda', dev_index)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        # WN_BENCH_SHARE_GPU=1: every rank on one GPU (self-test of the N > 1 path on a
        # 1-GPU box; RCCL refuses two ranks on one device, so the barriers run on gloo)
        if os.environ.get('WN_BENCH_SHARE_GPU') == '1':
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    with open(args.config, 'r') as fin:
        configs = yaml.load(fin, Loader=yaml.FullLoader)
    if args.override_config:
        configs = override_config(configs, args.override_config)
    check_feature_conf(configs)
    tokenizer = init_tokenizer(configs)
    blank_id = get_blank_id(configs, tokenizer.symbol_table)
    logging.info('blank_id is %d', blank_id)
    model = ASRModel(configs, load_state(configs, args.checkpoint), device)
    model.set_compute_dtype(args.dtype)

    context_graph = None
    if 'decoding-graph' in args.context_bias_mode:
        from wenet_amd.context_graph import ContextGraph
        context_graph = ContextGraph(args.context_list_path, tokenizer.symbol_table,
                                     configs['tokenizer_conf'].get('bpe_path'),
                                     args.context_graph_score)

    batches = static_batches(read_data_list(args.test_data, args.data_type),
                             args.batch_size)
    mine = list(range(rank, len(batches), world))
    files = {}
    for mode in args.modes:
        d = os.path.join(args.result_dir, mode)
        os.makedirs(d, exist_ok=True)
        files[mode] = open(os.path.join(d, f'text.part{rank}'), 'w', encoding='utf8')

    def emit(bi, mode, line):
        files[mode].write(f'{bi}\t{line}\n')

    stat = recognize(model, tokenizer, batches, mine, args, blank_id, context_graph, emit)
    for f in files.values():
        f.close()
    if args.report_rtf:
        rep = dict(rank=rank, world=world, modes=list(args.modes), batch_size=args.batch_size,
                   num_workers=args.num_workers, streams=args.streams,
                   data_type=args.data_type, dtype=args.dtype,
                   audio_seconds=round(stat['audio_s'], 2), utterances=stat['utts'],
                   batches=stat['batches'], wall_seconds=round(stat['wall_s'], 4),
                   audio_seconds_per_second=round(stat['audio_s'] / max(stat['wall_s'], 1e-9), 1),
                   main_thread_seconds={k: round(stat[k], 4) for k in
                                        ('wav_wait', 'resample', 'features', 'pad_sort', 'submit',
                                         'result_wait', 'detokenize')},
                   note='wall = first wav read submitted .. last result line written (model '
                        'load excluded); main_thread_seconds = where the feeding thread spent '
                        'that time: wav_wait = blocked on the reader threads, features = '
                        'H2D copy + fbank launch (returns device tensors), result_wait = '
                        'blocked on the oldest decode in flight')
        path = args.report_rtf if world == 1 else f'{args.report_rtf}.rank{rank}'
        with open(path, 'w') as f:
            json.dump(rep, f, indent=1)
        logging.info('wav -> text: %.1f audio-s / s (%s)', rep['audio_seconds_per_second'], path)
    if world > 1:
        dist.barrier()
    if rank == 0:
        merge_parts(args.result_dir, args.modes, world, len(batches))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
