#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second (RTF^-1) of the 12-layer
Conformer decode path on MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one `ASRModel.decode()` pass over one batch of synthetic fbank
features that are already resident in HBM: BASELINE.json configs[1]
(AIShell u2++ Conformer 12L/4h/256d, batch 32 x ~10 s, ctc_prefix_beam_search,
beam 10) on every GPU (weak scaling: the global batch is 32 x N utterances,
sorted by length and dealt round-robin; results are gathered with one RCCL
all_gather inside the timed region).  Weights are random-init (sharpened CTC
head, wenet_amd/synthetic.py), inputs synthetic.

`--workload configN` / `--dtype bf16` measure the other BASELINE configs and the
opt-in bf16-operand mode (recognize.py --dtype bf16) as extra data points; the
default line (config2, fp32 = the reference's dtype) is the headline.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the FFN w_1 GEMM (fp32 MFMA), achieved = algorithmic FLOP of
                 its launches / their HIP-event durations inside the timed
                 region, peak 157.3 TF (MI355X_MICROARCH.md);
  cpu_baseline : the oracle (torch-CPU restatement of the reference decode,
                 kind "port") timed on this box's host cores on a bounded
                 sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
# dmabuf IPC only on these hosts: RCCL needs this before the runtime starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_lib_mod = None
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md:41
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md:42
FRAMES = (800, 1200)  # 8..12 s of 10 ms frames, mean ~10 s
BEAM = 10
# BASELINE.json `configs`; the default (and the only one the driver runs) is
# configs[1], the configuration the metric is quoted on.
WORKLOADS = {
    'config2': dict(config='aishell_u2pp', batch=32, method='ctc_prefix_beam_search',
                    kw={}, text='BASELINE.json configs[1]: AIShell u2++ conformer '
                    '12L/4head/256d fbank80, batch 32 x ~10 s per GPU (8-12 s '
                    'ragged), ctc_prefix_beam_search beam 10'),
    'config3': dict(config='librispeech_bidecoder_large', batch=64,
                    method='attention_rescoring',
                    kw=dict(ctc_weight=0.5, reverse_weight=0.3),
                    text='BASELINE.json configs[2]: LibriSpeech conformer '
                    'bidecoder-large 12L/8head/512d fbank80, batch 64 x ~10 s per GPU, '
                    'attention_rescoring beam 10, ctc_weight 0.5, reverse_weight 0.3'),
    'config4': dict(config='wenetspeech_u2pp', batch=32,
                    method='ctc_prefix_beam_search',
                    kw=dict(decoding_chunk_size=16, num_decoding_left_chunks=-1),
                    text='BASELINE.json configs[3]: WenetSpeech u2++ conformer '
                    '12L/8head/512d, decoding_chunk_size 16 (chunk-mask streaming), '
                    'batch 32 x ~10 s per GPU, ctc_prefix_beam_search beam 10'),
    'config5': dict(config='whisper_largev3', batch=16, method='ctc_greedy_search',
                    kw={}, frames=(3000, 3000), feat_dim=128,
                    text='BASELINE.json configs[4]: Whisper-large-v3 encoder '
                    '32L/20head/1280d, 128 mel bins, 30 s windows, batch 16 per GPU, '
                    'fp32 (not the bf16 / fp8 the config names), + a 307-way CTC head '
                    'and greedy search'),
}
CONFIG = 'aishell_u2pp'
BATCH_PER_GPU = 32
METHOD = 'ctc_prefix_beam_search'
DECODE_KW = {}


def audio_seconds(n_frames) -> float:
    # snip_edges framing: samples = (T - 1) * 160 + 400 at 16 kHz
    return float(sum(((int(t) - 1) * 160 + 400) / 16000.0 for t in n_frames))


def cpu_baseline(configs, sd, feats, lens):
    """Oracle decode (same method / beam) on a bounded sample of the batch on
    the host cores.  torch-CPU with one thread per hardware thread collapses on
    these small GEMMs (256 threads: <1 audio-s/s), so a short calibration picks
    the best of a few thread counts first; `cores` is what was used."""
    from oracle import wenet_oracle as O
    ncores = os.cpu_count() or 1
    n = min(8, feats.shape[0])
    f = feats[:n, :int(lens[:n].max())].contiguous()
    l = lens[:n]
    best_t, best_dt = 1, float('inf')
    for nt in sorted({min(ncores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            O.encoder_forward(configs, sd, f[:2], l[:2])  # warm-up
            t0 = time.time()
            O.encoder_forward(configs, sd, f[:2], l[:2])
            dt = time.time() - t0
        if dt < best_dt:
            best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    O.decode(configs, sd, [METHOD], f[:1], l[:1], beam_size=BEAM, **DECODE_KW)  # warm-up
    reps, t0 = 0, time.time()
    while True:
        O.decode(configs, sd, [METHOD], f, l, beam_size=BEAM, **DECODE_KW)
        reps += 1
        if time.time() - t0 > 12.0 or reps >= 3:
            break
    dt = (time.time() - t0) / reps
    return {
        'value': round(audio_seconds(l.tolist()) / dt, 2),
        'unit': 'audio_s/s',
        'cores': torch.get_num_threads(),
        'kind': 'port',
        'sample': f'{n} utterances of the same batch ({audio_seconds(l.tolist()):.0f} s '
                  f'audio), {METHOD} beam {BEAM}, oracle/wenet_oracle.py '
                  f'(torch-CPU fp32 + Python prefix beam), mean of {reps} runs, '
                  f'best of 8/16/32/64 threads on {ncores} hardware threads',
    }


def end_to_end_leg(model, lens, device, total_audio, ms_per_step):
    """PCM -> tokens variant: the same utterance lengths
    as 16 kHz waveforms resident in HBM go through wn_fbank (Kaldi fbank on the
    GPU) before the decode.  The fbank pass is timed on its own (it is a separate
    C-ABI call on the same stream) and added to the measured decode step."""
    n_samp = [(int(t) - 1) * 160 + 400 for t in lens.tolist()]
    g = torch.Generator().manual_seed(99)
    waves = [(torch.rand(n, generator=g) * 0.6 - 0.3).numpy() for n in n_samp]
    model.compute_fbank(waves)  # warm-up (+ host->device copy of the PCM)
    offs = np.zeros((len(waves) + 1, ), dtype=np.int64)
    offs[1:] = np.cumsum(n_samp)
    pcm = torch.from_numpy(np.concatenate(waves)).to(device)
    tmax = int(max(lens.tolist()))
    feats = torch.empty((len(waves), tmax, 80), dtype=torch.float32, device=device)
    nfr = np.zeros((len(waves), ), dtype=np.int32)
    L = _lib_mod.lib()
    stream = torch.cuda.current_stream(device).cuda_stream
    reps = 10
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib_mod.check(L.wn_fbank(model._h, pcm.data_ptr(), _lib_mod.i64p(offs),
                                  len(waves), feats.data_ptr(), tmax,
                                  _lib_mod.i32p(nfr), stream), 'wn_fbank')
    e1.record()
    torch.cuda.synchronize()
    fb_ms = e0.elapsed_time(e1) / reps
    return {
        'value': round(total_audio / ((ms_per_step + fb_ms) * 1e-3), 1),
        'unit': 'audio_s/s',
        'fbank_ms_per_batch': round(fb_ms, 3),
        'note': 'PCM resident in HBM -> wn_fbank -> decode; fbank timed separately '
                f'({reps} reps, HIP events) and added to ms_per_step',
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--streams', type=int, default=2,
                    help='decodes kept in flight per GPU (wenet_amd/pipeline.py); '
                         '1 = plain back-to-back ASRModel.decode() calls')
    ap.add_argument('--workload', default='config2', choices=sorted(WORKLOADS),
                    help='config2 = BASELINE.json configs[1] (the metric\'s '
                         'configuration, default); the others are extra data points')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help='fp32 (default: the reference\'s dtype, the headline) or '
                         'bf16 operands / fp32 accumulate (recognize.py --dtype bf16; '
                         'the dtype BASELINE.json configs[4] names) -- an extra data '
                         'point, never the headline line')
    ap.add_argument('--tune', default='',
                    help='experiments: comma list of key=value for wn_tune_set')
    args = ap.parse_args()
    global CONFIG, BATCH_PER_GPU, METHOD, DECODE_KW
    wl = WORKLOADS[args.workload]
    CONFIG, BATCH_PER_GPU, METHOD, DECODE_KW = (wl['config'], wl['batch'],
                                                wl['method'], wl['kw'])

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with '
                         f'--nproc-per-node {args.gpus} (WORLD_SIZE={world})')
    # WN_BENCH_SHARE_GPU=1: self-test of the N > 1 code path on a 1-GPU box
    # (every rank on cuda:0, results gathered over gloo); never a measurement.
    share_gpu = os.environ.get('WN_BENCH_SHARE_GPU') == '1'
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=device)

    from wenet_amd import _lib, dist as wdist, synthetic as S
    global _lib_mod
    _lib_mod = _lib
    for kv in filter(None, args.tune.split(',')):
        k, v = kv.split('=')
        _lib.check(_lib.lib().wn_tune_set(k.encode(), int(v)), 'tune')
    from wenet_amd.model import ASRModel
    configs = S.make_configs(CONFIG)
    sd = S.make_state_dict(configs, 0)
    model = ASRModel(configs, sd, device=device)
    model.set_compute_dtype(args.dtype)  # before the pipeline clones the handle
    bf16 = args.dtype == 'bf16'

    # global batch, sharded by length (weak scaling: 32 utterances per GPU)
    gfeats, glens = S.make_features(BATCH_PER_GPU * world, wl.get('frames', FRAMES),
                                    seed=1234, feat_dim=wl.get('feat_dim', 80))
    mine = wdist.shard_indices(glens.tolist(), world, rank)
    lens = glens[mine]
    feats = gfeats[mine, :int(lens.max())].contiguous()
    feats_dev = feats.to(device)
    whisper = configs.get('encoder') == 'transformer'
    total_audio = (float(sum(glens.tolist())) * 0.01 if whisper  # hop 160 @ 16 kHz
                   else audio_seconds(glens.tolist()))
    max_tok = 256

    from wenet_amd.pipeline import DecodePipeline
    pipe = DecodePipeline(model, n_streams=max(1, args.streams))

    def finish(res):
        rec = wdist.pack_results(mine, [r.tokens for r in res],
                                 [r.score for r in res], BATCH_PER_GPU, max_tok,
                                 'cpu' if share_gpu else device)
        return wdist.gather_results(rec, world)

    def run_steps(n):
        """n decode passes over the batch, `--streams` of them in flight; the
        per-step result gather (one all_gather) stays on the main thread, in
        step order."""
        futs = [pipe.submit([METHOD], feats_dev, lens, beam_size=BEAM, **DECODE_KW)
                for _ in range(n)]
        out = None
        for f in futs:
            out = finish(f.result()[METHOD])
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    out = run_steps(args.warmup)
    L = _lib.lib()
    for mdl in pipe.models:
        _lib.check(L.wn_profile_enable(mdl._h, 1), 'profile')
    barrier()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    n_launch, ms, flops = ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
    tot_launch, tot_ms, tot_flops = 0, 0.0, 0.0
    for mdl in pipe.models:
        _lib.check(L.wn_profile_collect(mdl._h, ctypes.byref(n_launch),
                                        ctypes.byref(ms), ctypes.byref(flops)),
                   'profile')
        _lib.check(L.wn_profile_enable(mdl._h, 0), 'profile')
        tot_launch += n_launch.value
        tot_ms += ms.value
        tot_flops += flops.value
    n_launch.value, ms.value, flops.value = tot_launch, tot_ms, tot_flops
    pipe.close()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64,
                         device='cpu' if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert len(out) == BATCH_PER_GPU * world, 'result gather lost utterances'

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_audio * args.steps / dt
        achieved = (flops.value / (ms.value * 1e-3)) / 1e12 if ms.value > 0 else 0.0
        d_model = configs['encoder_conf']['output_size']
        enc_rows = (int(sum((int(t) + 1) // 2 for t in lens.tolist())) if whisper
                    else int(sum(max(0, (int(t) - 7) // 4 + 1) for t in lens.tolist())))
        ffn = configs['encoder_conf']['linear_units']
        line = {
            'metric': ('audio-seconds/sec (RTF^-1), Whisper-large-v3 encoder, '
                       if whisper else
                       'audio-seconds/sec (RTF^-1), 12L Conformer fbank80, ') + METHOD,
            'value': round(value, 1),
            'unit': 'audio_s/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'bf16 operands, f32 accumulate / activations' if bf16 else 'f32',
            'data': 'synthetic',
            'config': {
                'workload': (wl['text'].replace('fp32 (not the bf16 / fp8 the config '
                                                'names)', 'bf16 GEMM operands (no fp8)')
                             if bf16 else wl['text'])
                            + ', features resident in HBM, random-init weights',
                'global_batch': BATCH_PER_GPU * world,
                'audio_seconds_per_step': round(total_audio, 1),
                'encoder_frames_per_gpu': enc_rows,
                'parallelism': f'utterance-sharded x{world}, one all_gather of results',
                'decodes_in_flight_per_gpu': max(1, args.streams),
            },
            'roofline': {
                'bound': 'mfma',
                'kernel': ('gemm_bf16_kernel' if bf16 else 'gemm_f32_kernel') +
                          '<128,128,2x4 waves> (FFN w_1, '
                          f'M={enc_rows} N={ffn} K={d_model})',
                'achieved': round(achieved, 2),
                'peak': BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': round(achieved / (BF16_MFMA_PEAK_TFLOPS if bf16
                                          else FP32_MFMA_PEAK_TFLOPS), 4),
                'launches': n_launch.value,
                'avg_launch_us': round(ms.value * 1e3 / max(n_launch.value, 1), 2),
                'traffic': None,
            },
        }
        # HBM traffic of that kernel from the PMC passes (tools/gpu_pmc.sh; cannot
        # be collected inside a timed run): bytes per launch, committed summary
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                           'pmc_roofline_kernel.json')
        if wl is WORKLOADS.get('config2') and not bf16 and os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f)
            line['roofline']['traffic'] = rec['hbm_bytes_per_launch']
            line['roofline']['traffic_unit'] = 'bytes/launch (HBM read + write, PMC)'
            line['roofline']['traffic_source'] = 'profiles/pmc_roofline_kernel.json'
            line['roofline']['algorithmic_bytes'] = int(
                4 * (enc_rows * d_model + ffn * d_model + enc_rows * ffn))
        if world == 1 and not whisper:
            line['end_to_end'] = end_to_end_leg(model, lens, device, total_audio,
                                                ms_per_step)
        if not args.no_cpu_baseline and world == 1 and not whisper:
            line['cpu_baseline'] = cpu_baseline(configs, sd, feats, lens)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
