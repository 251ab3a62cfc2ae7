#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second (RTF^-1) of the 12-layer
Conformer decode path on MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]        # N > 1: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   # or under a launcher

One step = one `ASRModel.decode()` pass over one batch of synthetic fbank
features that are already resident in HBM: BASELINE.json configs[1]
(AIShell u2++ Conformer 12L/4h/256d, batch 32 x ~10 s, ctc_prefix_beam_search,
beam 10) on every GPU (weak scaling: 32 utterances per GPU -- group g of the
global batch is `synthetic.make_bench_group(workload, g)` -- sorted by length and
dealt in snake order; results are gathered with one RCCL all_gather per step on a
worker thread + side stream, drained inside the timed region:
wenet_amd/dist.py ResultGatherer).  Weights are random-init (sharpened CTC head,
wenet_amd/synthetic.py), inputs synthetic.

Timing: W warm-up steps, then ROUNDS of exactly K steps, each round bracketed by a
barrier + torch.cuda.synchronize() on both sides and max-reduced over the ranks;
rounds repeat until they cover >= 2 s, and `value` / `ms_per_step` are the MEDIAN
round (`rounds` carries min / max).  After the timed rounds the tokens the last
timed step produced are compared with what the REAL reference produced on the same
batch (tests/golden/bench_*.npz): `verified`.

`--workload configN` / `--dtype bf16|fp8` measure the other BASELINE configs and
the opt-in reduced-precision modes as extra data points; the default line
(config2, fp32 = the reference's dtype) is the headline.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the feed-forward kernel (d_model 256: both contractions fused in
                 ffn_x6f_kernel; 512: the w_1 GEMM), achieved = algorithmic FLOP of its
                 launches / their HIP-event durations inside the timed rounds, peak =
                 the six-product ceiling (dense bf16 peak / 6, MI355X_MICROARCH.md);
                 `whole_decode_frac` = all encoder + CTC-head contraction FLOPs of the
                 batch / step time / the same ceiling;
  plain_decode, f32_mfma_only, nbest_materialised, end_to_end : transparency legs
                 (one extra round each, never `value`);
  cpu_baseline : the oracle (torch-CPU restatement of the reference decode,
                 kind "port") timed on this box's host cores on the whole batch;
                 profiles/cpu_port_vs_reference.json calibrates the port against
                 the real reference's ASRModel.decode.
"""
import argparse
import ctypes
import gc
import json
import os
import statistics
import sys
import time

import numpy as np
# dmabuf IPC only on these hosts: RCCL needs this before the runtime starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_lib_mod = None
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md:41-43
PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0, 'fp8': 5000.0}
NOMINAL_GHZ = 2.4      # peak engine clock the dense peaks are quoted at (MI355X_MICROARCH.md)
MIN_TIMED_SECONDS = 2.0
MAX_ROUNDS = 400


def audio_seconds(n_frames) -> float:
    # snip_edges framing: samples = (T - 1) * 160 + 400 at 16 kHz
    return float(sum(((int(t) - 1) * 160 + 400) / 16000.0 for t in n_frames))


def contraction_flops(configs, feat_lens) -> float:
    """Algorithmic FLOPs (2 x MAC) of every contraction of the encoder + CTC head
    for utterances of `feat_lens` frames -- SURVEY.md section 8(d)'s per-unit figure
    (11.776 GMAC for one 998-frame AIShell utterance), evaluated per utterance."""
    ec = configs['encoder_conf']
    d, F, L = ec['output_size'], ec['linear_units'], ec['num_blocks']
    V = configs['output_dim']
    mac = 0.0
    if configs.get('encoder') == 'transformer':   # Whisper-style encoder
        fd = configs['input_dim']
        for t in feat_lens:
            t = int(t)
            tp = (t - 1) // 2 + 1
            mac += t * 3 * fd * d + tp * 3 * d * d            # Conv1dSubsampling2
            mac += L * (4 * tp * d * d + 2 * tp * d * F + 2 * tp * tp * d)
            mac += tp * d * V
        return 2.0 * mac
    K = ec['cnn_module_kernel']
    fd = configs['input_dim']
    f1 = (fd - 1) // 2
    f2 = (f1 - 1) // 2
    for t in feat_lens:
        t = int(t)
        t1 = (t - 1) // 2
        tp = (t1 - 1) // 2
        mac += t1 * f1 * d * 9 + tp * f2 * d * 9 * d + tp * f2 * d * d   # subsampling
        per_layer = (2 * 2 * tp * d * F            # two FFNs
                     + 4 * tp * d * d              # q, k, v, out
                     + tp * d * d                  # linear_pos
                     + 3 * tp * tp * d             # ac + bd + pv
                     + tp * d * 2 * d + tp * d * K + tp * d * d)  # conv module
        mac += L * per_layer + tp * d * V
    return 2.0 * mac


def cpu_baseline(configs, sd, feats, lens, method, kw, beam):
    """Oracle decode (same method / beam) of the WHOLE batch on the host cores.
    torch-CPU with one thread per hardware thread collapses on these small GEMMs
    (256 threads: < 1 audio-s/s), so a short calibration picks the best of a few
    thread counts first; `cores` is what was used."""
    from oracle import wenet_oracle as O
    ncores = os.cpu_count() or 1
    best_t, best_dt = 1, float('inf')
    for nt in sorted({min(ncores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            O.encoder_forward(configs, sd, feats[:2], lens[:2])  # warm-up
            t0 = time.time()
            O.encoder_forward(configs, sd, feats[:2], lens[:2])
            dt = time.time() - t0
        if dt < best_dt:
            best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    O.decode(configs, sd, [method], feats[:1], lens[:1], beam_size=beam, **kw)  # warm-up
    ts = []
    t_start = time.time()
    while len(ts) < 3 and (not ts or time.time() - t_start < 20.0):
        t0 = time.time()
        O.decode(configs, sd, [method], feats, lens, beam_size=beam, **kw)
        ts.append(time.time() - t0)
    dt = statistics.median(ts)
    out = {
        'value': round(audio_seconds(lens.tolist()) / dt, 2),
        'unit': 'audio_s/s',
        'cores': torch.get_num_threads(),
        'kind': 'port',
        'sample': f'the whole batch: {feats.shape[0]} utterances '
                  f'({audio_seconds(lens.tolist()):.0f} s audio), {method} beam {beam}, '
                  f'oracle/wenet_oracle.py (torch-CPU fp32 + Python prefix beam), median '
                  f'of {len(ts)} runs, best of 8/16/32/64 threads on {ncores} hardware '
                  f'threads',
    }
    cal = os.path.join(ROOT, 'profiles', 'cpu_port_vs_reference.json')
    if os.path.exists(cal):
        with open(cal) as f:
            c = json.load(f)
        out['port_over_reference_speed'] = c.get('port_over_reference_speed')
        out['calibration'] = ('profiles/cpu_port_vs_reference.json: port vs the real '
                              'reference ASRModel.decode on the same batch, '
                              f"{c.get('threads')} threads, build container")
    return out


def end_to_end_leg(model, pipe, finish, drain, barrier, lens, device, total_audio, steps,
                   method, beam, decode_kw, depth):
    """PCM -> tokens in ONE timed region (the reference's
    runtime/core/bin/decoder_main.cc:52,64-70 times wav -> text the same way): the same
    utterance lengths as 16 kHz waveforms resident in HBM; every step runs wn_fbank (Kaldi
    fbank on the GPU, the caller's stream) into a feature buffer and hands that buffer to the
    same two-decodes-in-flight pipeline as the headline; the round is bracketed by barrier +
    synchronize like the headline's.  At most `depth` steps are submitted ahead and the
    feature buffers form a ring of depth + 1, so a buffer is only rewritten after the decode
    that read it has returned."""
    import collections
    n_samp = [(int(t) - 1) * 160 + 400 for t in lens.tolist()]
    g = torch.Generator().manual_seed(99)
    waves = [(torch.rand(n, generator=g) * 0.6 - 0.3).numpy() for n in n_samp]
    offs = np.zeros((len(waves) + 1, ), dtype=np.int64)
    offs[1:] = np.cumsum(n_samp)
    pcm = torch.from_numpy(np.concatenate(waves)).to(device)
    tmax = int(max(lens.tolist()))
    ring = [torch.empty((len(waves), tmax, 80), dtype=torch.float32, device=device)
            for _ in range(depth + 1)]
    nfr = np.zeros((len(waves), ), dtype=np.int32)
    L = _lib_mod.lib()
    stream = torch.cuda.current_stream(device).cuda_stream

    def fbank_into(buf):
        _lib_mod.check(L.wn_fbank(model._h, pcm.data_ptr(), _lib_mod.i64p(offs), len(waves),
                                  buf.data_ptr(), tmax, _lib_mod.i32p(nfr), stream), 'wn_fbank')

    def steps_pcm(n):
        pending, last = collections.deque(), None
        for i in range(n):
            if len(pending) >= depth:
                last = pending.popleft().result()[method]
                finish(last)
            buf = ring[i % len(ring)]
            fbank_into(buf)
            pending.append(pipe.submit([method], buf, lens, beam_size=beam, **decode_kw))
        while pending:
            last = pending.popleft().result()[method]
            finish(last)
        drain()
        return last

    steps_pcm(max(2, depth))       # warm-up
    assert nfr.tolist() == [int(t) for t in lens.tolist()], 'fbank frame counts'
    # median of three rounds (one round of a single-round leg once caught a stall of the box:
    # 48.7 k beside 65.9 k, r18a)
    dts = []
    for _ in range(3):
        barrier()
        t0 = time.perf_counter()
        last = steps_pcm(steps)
        barrier()
        dts.append(time.perf_counter() - t0)
    dt = statistics.median(dts)
    # the fbank pass alone, for the record (median of five event-bracketed groups of four)
    groups = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fbank_into(ring[0])
        e1.record()
        torch.cuda.synchronize()
        groups.append(e0.elapsed_time(e1) / 4)
    return {
        'value': round(total_audio * steps / dt, 1),
        'unit': 'audio_s/s',
        'ms_per_step': round(dt / steps * 1e3, 3),
        'fbank_ms_per_batch_alone': round(statistics.median(groups), 3),
        'tokens_last_step': int(sum(len(r.tokens) for r in last)),
        'rounds_ms_per_step': [round(x / steps * 1e3, 3) for x in dts],
        'note': 'ONE timed region per round of --steps steps (median of three rounds): 16-kHz PCM resident in HBM -> '
                'wn_fbank -> decode (the headline\'s pipeline) -> token lists on the host, '
                'barrier + synchronize brackets.  No resampling in this leg -- wn_resample '
                '(other sample rates) is pinned to an fp64 evaluation of torchaudio\'s '
                'published definition only: PARITY UNPINNED (torchaudio absent from the image, '
                'oracle/gen_golden_resample.py)',
    }


def main():
    from wenet_amd import synthetic as S
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--min-seconds', type=float, default=MIN_TIMED_SECONDS,
                    help='rounds of --steps steps are repeated until they cover this')
    ap.add_argument('--streams', type=int, default=6,
                    help='decodes kept in flight per GPU (wenet_amd/pipeline.py: up to two '
                         'the encoders are chained by events, above they run free -- r17c: '
                         '64.3 k at 2, 66.7 k at 4, 69.3 k at 6, 69.1 k at 8 on one box); '
                         '1 = plain back-to-back ASRModel.decode() calls')
    ap.add_argument('--workload', default='config2', choices=sorted(S.BENCH_WORKLOADS),
                    help='config2 = BASELINE.json configs[1] (the metric\'s '
                         'configuration, default); the others are extra data points')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16', 'fp8'],
                    help='fp32 (default: the reference\'s dtype, the headline); bf16 = '
                         'bf16 operands / fp32 accumulate (recognize.py --dtype bf16); '
                         'fp8 = bf16 mode with e4m3 FFN GEMMs (BASELINE.json '
                         'configs[4]) -- extra data points, never the headline line')
    ap.add_argument('--no-f32-mfma-leg', action='store_true',
                    help='skip the extra round that runs every GEMM on v_mfma_f32')
    ap.add_argument('--no-clock-sample', action='store_true',
                    help='skip the two untimed steps that sample the shader clock of the '
                         'roofline kernel')
    ap.add_argument('--no-nbest-leg', action='store_true',
                    help='skip the extra round that materialises every n-best list')
    ap.add_argument('--no-e2e-leg', action='store_true',
                    help='skip the extra round that starts from PCM (wn_fbank inside the round)')
    ap.add_argument('--no-two-stream-leg', action='store_true',
                    help='skip the extra round with two chained decodes in flight (the headline '
                         'form of rounds 2-5)')
    ap.add_argument('--no-plain-leg', action='store_true',
                    help='skip the extra round of plain back-to-back decode() calls')
    ap.add_argument('--tune', default='',
                    help='experiments: comma list of key=value for wn_tune_set')
    args = ap.parse_args()
    wl = S.BENCH_WORKLOADS[args.workload]
    config, batch_per_gpu, method, decode_kw = (wl['config'], wl['batch'],
                                                wl['method'], wl['kw'])
    beam = S.BENCH_BEAM

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # WN_BENCH_SHARE_GPU=1: self-test of the N > 1 code path on a 1-GPU box
    # (every rank on cuda:0, results gathered over gloo); never a measurement.
    share_gpu = os.environ.get('WN_BENCH_SHARE_GPU') == '1'
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks, one per GPU (the
        # reference's tools/decode.sh:65-83 loop over `nj` jobs, wenet/bin/recognize.py:43-46);
        # under torch.distributed.run the variables are there and this is skipped
        from wenet_amd.dist import launch_local_ranks
        if not share_gpu and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f'--gpus {args.gpus}: this node shows '
                             f'{torch.cuda.device_count()} GPU(s)')
        raise SystemExit(launch_local_ranks([os.path.abspath(__file__)] + sys.argv[1:],
                                            args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} under a launcher that started {world} rank(s) '
                         f'(WORLD_SIZE={world}): start it with --nproc-per-node {args.gpus}, '
                         'or without a launcher (bench.py starts its own ranks)')
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    # each rank's host threads (decode thread, the pipeline's workers, the result gatherer)
    # on its own slice of the cores next to its GPU; WN_BENCH_PIN=0: leave the scheduler alone
    pinned = []
    if world > 1 and os.environ.get('WN_BENCH_PIN', '1') != '0':
        from wenet_amd.dist import pin_rank_to_local_cores
        pinned = pin_rank_to_local_cores(local_rank, world,
                                         [0] * world if share_gpu else None)
    # WN_BENCH_FORCE_DIST=1: a process group (RCCL) even for one rank, so that the backend,
    # the device-tensor all_gather / all_reduce / barrier of the N > 1 path execute on a
    # 1-GPU box (tests/test_gpu_dist.py); the collectives of one rank are no-ops in time
    dist_on = world > 1 or os.environ.get('WN_BENCH_FORCE_DIST') == '1'
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if share_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=device)

    from wenet_amd import _lib, dist as wdist
    import bench_verify as verify
    global _lib_mod
    _lib_mod = _lib
    for kv in filter(None, args.tune.split(',')):
        k, v = kv.split('=')
        _lib.check(_lib.lib().wn_tune_set(k.encode(), int(v)), 'tune')
    from wenet_amd.model import ASRModel
    configs = S.make_configs(config)
    sd = S.make_state_dict(configs, 0)
    model = ASRModel(configs, sd, device=device)
    model.set_compute_dtype(args.dtype)  # before the pipeline clones the handle
    reduced = args.dtype != 'fp32'

    # global batch, sharded by length (weak scaling: `batch` utterances per GPU)
    gfeats, glens = S.make_bench_batch(args.workload, world)
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

    # the per-step result gather (one all_gather) runs on a worker thread + side stream, in step
    # order; every timed region drains it before its closing barrier (wenet_amd/dist.py)
    gatherer = wdist.ResultGatherer(world, batch_per_gpu, max_tok,
                                    'cpu' if share_gpu else device)

    sync_gather = os.environ.get('WN_BENCH_SYNC_GATHER') == '1'   # A/B: gather on the decode thread

    def finish(res):
        if sync_gather:
            rec = wdist.pack_results(mine, [r.tokens for r in res], [r.score for r in res],
                                     batch_per_gpu, max_tok, 'cpu' if share_gpu else device)
            gatherer.last = wdist.gather_results(rec, world)
            return
        gatherer.submit(mine, [r.tokens for r in res], [r.score for r in res])

    def finish_nbest(res):
        # the reference's DecodeResult carries the n-best lists as plain attributes
        # (search.py:30-61); here they are built on first access -- this leg builds them all
        for r in res:
            assert r.nbest is not None and r.nbest_scores is not None
            assert r.nbest_times is not None
        finish(res)

    t_decoded = [0.0]

    def run_steps(n, fin=None):
        """n decode passes over the batch, `--streams` of them in flight; the
        per-step result gather (one all_gather) stays on the main thread, in
        step order."""
        fin = fin or finish
        futs = [pipe.submit([method], feats_dev, lens, beam_size=beam, **decode_kw)
                for _ in range(n)]
        for f in futs:
            fin(f.result()[method])
        t_decoded[0] = time.perf_counter()   # this rank's last result is on the host
        return gatherer.drain()

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device='cpu' if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = run_steps(args.warmup)
    L = _lib.lib()
    # (one bracketed launch of the roofline kernel per decode inside the timed rounds: every event
    # pair idles the GPU for ~10 us around the launch -- four per decode were 0.8 % of a step in
    # the two-stream timeline, r13b; the roofline figures come from the single-stream pass below,
    # which brackets every 6th launch)
    ffn_launches_per_decode = 2 * configs['encoder_conf']['num_blocks']
    for mdl in pipe.models:
        _lib.check(L.wn_profile_enable(
            mdl._h, int(os.environ.get('WN_BENCH_PROF_STRIDE', max(6, ffn_launches_per_decode)))),
            'profile')
    # The host's cyclic GC walks every tracked object (torch modules, state dicts, the
    # goldens: a large, static heap) on each full collection -- measured 25 % slower rounds
    # about once a second and 4 % on the median (r02ax).  After warm-up the objects alive so
    # far are moved to the permanent generation (what a long-running server does once it is
    # up: wenet_amd.pipeline.freeze_host_heap); the collector stays ON for everything
    # allocated afterwards.  WN_BENCH_GC=1: skip this, =2: disable the collector instead (A/B).
    gc_mode = os.environ.get('WN_BENCH_GC', '0')
    gc_off = gc_mode != '1'
    if gc_off:
        gc.collect()
        if gc_mode == '2':
            gc.disable()
        else:
            from wenet_amd.pipeline import freeze_host_heap
            freeze_host_heap()
    round_s, local_decode_s, local_drain_s = [], [], []
    while True:
        barrier()
        t0 = time.perf_counter()
        out = run_steps(args.steps)   # EXACTLY --steps steps per timed round
        t_drained = time.perf_counter()
        barrier()
        round_s.append(max_over_ranks(time.perf_counter() - t0))
        # this rank's own share of the round (diagnosis of an N-GPU run, never `value`):
        # until its last result was on the host, and what it then waited for the gathers
        local_decode_s.append(t_decoded[0] - t0)
        local_drain_s.append(t_drained - t_decoded[0])
        # every rank sees the same max-reduced times, so they stop together
        if sum(round_s) >= args.min_seconds or len(round_s) >= MAX_ROUNDS:
            break
    if gc_off:
        gc.enable()
        gc.unfreeze()
    prof_name = L.wn_profile_kernel_name(pipe.models[0]._h).decode()
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
    # The roofline kernel's launch duration, ONE decode in flight: under the headline's two
    # overlapping decodes the event pair around a launch also times whatever the other stream
    # runs beside it (config 3: 243 us under two streams, 176 us alone) -- contention, not
    # the kernel.  `roofline.achieved / frac / avg_launch_us` come from this pass (plain
    # back-to-back decode() calls on the caller's handle, every 6th launch bracketed), which is
    # what `rocprofv3 --kernel-trace --stats ... --streams 1` (profiles/) reproduces; the
    # figures of the timed rounds stay in `roofline.timed_rounds`.
    timed_launch, timed_ms, timed_flops = tot_launch, tot_ms, tot_flops
    _lib.check(L.wn_profile_enable(model._h, 1), 'profile')
    roof_steps = max(4, min(args.steps, 12))
    for _ in range(2):
        model.decode([method], feats_dev, lens, beam_size=beam, **decode_kw)
    _lib.check(L.wn_profile_collect(model._h, ctypes.byref(n_launch), ctypes.byref(ms),
                                    ctypes.byref(flops)), 'profile')   # drop the warm-up's
    for _ in range(roof_steps):
        model.decode([method], feats_dev, lens, beam_size=beam, **decode_kw)
    torch.cuda.synchronize()
    _lib.check(L.wn_profile_collect(model._h, ctypes.byref(n_launch), ctypes.byref(ms),
                                    ctypes.byref(flops)), 'profile')
    _lib.check(L.wn_profile_enable(model._h, 0), 'profile')
    if n_launch.value > 0:
        tot_launch, tot_ms, tot_flops = n_launch.value, ms.value, flops.value
        prof_name = L.wn_profile_kernel_name(model._h).decode()
    # transparency leg: the same decode with every GEMM on v_mfma_f32 (gemm_x6 = 0), one
    # round of --steps steps; reported beside the headline, never as `value`
    f32_only = None
    if args.dtype == 'fp32' and 'x6' in prof_name and not args.no_f32_mfma_leg:
        _lib.check(L.wn_tune_set(b'gemm_x6', 0), 'tune')
        try:
            run_steps(max(2, args.warmup // 2))
            barrier()
            t0 = time.perf_counter()
            run_steps(args.steps)
            barrier()
            f32_only = max_over_ranks(time.perf_counter() - t0)
        finally:
            _lib.check(L.wn_tune_set(b'gemm_x6', 1), 'tune')
    # transparency leg: plain back-to-back ASRModel.decode() calls -- what a caller written for
    # the reference (wenet/bin/recognize.py:289) gets without the two-decodes-in-flight
    # pipeline -- one round of --steps steps; reported beside the headline, never as `value`
    plain = None
    if args.streams != 1 and not args.no_plain_leg:
        def plain_steps(n):
            for _ in range(n):
                finish(model.decode([method], feats_dev, lens, beam_size=beam,
                                    **decode_kw)[method])
            return gatherer.drain()
        plain_steps(max(2, args.warmup // 2))
        barrier()
        t0 = time.perf_counter()
        plain_steps(args.steps)
        barrier()
        plain = max_over_ranks(time.perf_counter() - t0)
    # transparency leg: the headline of rounds 2-5 -- TWO decodes in flight, their encoders chained
    # by events -- for continuity with the earlier rounds' numbers; one round of --steps steps
    two_chained = None
    if args.streams > 2 and not args.no_two_stream_leg:
        pipe2 = DecodePipeline(model, n_streams=2)
        try:
            def two_steps(n):
                futs = [pipe2.submit([method], feats_dev, lens, beam_size=beam, **decode_kw)
                        for _ in range(n)]
                for f in futs:
                    finish(f.result()[method])
                return gatherer.drain()
            two_steps(max(2, args.warmup // 2))
            barrier()
            t0 = time.perf_counter()
            two_steps(args.steps)
            barrier()
            two_chained = max_over_ranks(time.perf_counter() - t0)
        finally:
            pipe2.close()
    # transparency leg: the same pipelined steps with the n-best lists of EVERY result
    # materialised inside the timed round (Python lists of the reference's DecodeResult fields;
    # the headline reads tokens / score only and leaves them lazy)
    nbest_leg = None
    if method == 'ctc_prefix_beam_search' and not args.no_nbest_leg:
        run_steps(max(2, args.warmup // 2), finish_nbest)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps, finish_nbest)
        barrier()
        nbest_leg = max_over_ranks(time.perf_counter() - t0)
    ffn_split = int(L.wn_profile_ffn_split(pipe.models[0]._h))
    # the shader clock the roofline kernel actually ran at, sampled in the same pipeline: two
    # more steps with the clock-stamp variant of the kernel (csrc/ffn_x6f.hip VAR & 8192: cycle
    # counter and 100-MHz real-time counter at its entry and end) -- outside every timed region
    clock_ghz = None
    if 'ffn_x6f' in prof_name and not args.no_clock_sample:
        _lib.check(L.wn_tune_set(b'ffn_x6f_var', 25088), 'tune')
        try:
            run_steps(2)
            torch.cuda.synchronize()
            import numpy as _np
            ck = _np.zeros((4, 24), dtype=_np.uint64)
            _lib.check(L.wn_profile_ffn_clocks(
                ck.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))), 'clocks')
            ck = ck.astype(_np.int64)
            cyc, ns = ck[:, 20] - ck[:, 9], (ck[:, 13] - ck[:, 12]) * 10.0
            if (ns > 0).all() and (cyc > 0).all():
                clock_ghz = float((cyc / ns).mean())
        finally:
            _lib.check(L.wn_tune_set(b'ffn_x6f_var', 0), 'tune')
    e2e = None
    if world == 1 and not whisper and not args.no_e2e_leg:
        e2e = end_to_end_leg(model, pipe, finish, gatherer.drain, barrier, lens, device,
                             total_audio, args.steps, method, beam, decode_kw,
                             max(1, args.streams))
    pipe.close()
    gather_ms = list(gatherer.latencies_ms)
    gatherer.close()
    assert len(out) == batch_per_gpu * world, 'result gather lost utterances'

    # per-rank view of the median round (every rank sees the same max-reduced round times, so
    # they pick the same round): own decode time per step, wait for the gathers at the round's
    # end, and the latency of ONE result gather (pack + all_gather + unpack on the worker
    # thread, including the wait for the slowest peer) -- what makes the first real 8-GPU run
    # diagnosable from this one line
    med_i = sorted(range(len(round_s)), key=lambda i: round_s[i])[len(round_s) // 2]
    mine_diag = [local_decode_s[med_i] / args.steps * 1e3, local_drain_s[med_i] * 1e3,
                 statistics.median(gather_ms) if gather_ms else 0.0,
                 max(gather_ms) if gather_ms else 0.0,
                 float(len(pinned)), float(min(pinned)) if pinned else -1.0,
                 float(max(pinned)) if pinned else -1.0]
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor(mine_diag, dtype=torch.float64, device='cpu' if share_gpu else device)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_diag = [x.cpu().tolist() for x in allt]
    else:
        rank_diag = [mine_diag]

    if rank == 0:
        dt = statistics.median(round_s)
        ms_per_step = dt / args.steps * 1e3
        value = total_audio * args.steps / dt
        achieved = (tot_flops / (tot_ms * 1e-3)) / 1e12 if tot_ms > 0 else 0.0
        d_model = configs['encoder_conf']['output_size']
        enc_rows = (int(sum((int(t) + 1) // 2 for t in lens.tolist())) if whisper
                    else int(sum(max(0, (int(t) - 7) // 4 + 1) for t in lens.tolist())))
        ffn = configs['encoder_conf']['linear_units']
        peak = PEAK_TFLOPS[args.dtype]
        x6 = 'x6' in prof_name
        if x6:
            # six bf16 plane products per fp32 multiply-add: the matrix-pipe ceiling of
            # this algorithm is the dense bf16 peak / 6 (in algorithmic fp32 FLOP/s)
            peak = round(PEAK_TFLOPS['bf16'] / 6.0, 1)
        whole_flops = contraction_flops(configs, glens.tolist())
        # the FFN GEMMs run in e4m3 in the fp8 mode, everything else in bf16: the
        # whole-decode fraction is priced against the bf16 peak there
        whole_peak = (PEAK_TFLOPS['bf16'] if args.dtype == 'fp8'
                      else PEAK_TFLOPS[args.dtype])
        # fp32 mode with the six-product GEMMs: ~80 % of the contraction FLOPs (FFNs, conv2,
        # sub_out, QKV / out / pointwise row-block GEMMs) run as six bf16 plane products, so
        # the ceiling of the whole decode is the x6 ceiling (bf16 peak / 6), NOT the
        # v_mfma_f32 peak (round-3 VERDICT: 0.84 against 157.3 was mis-priced; it is 0.32)
        if x6:
            whole_peak = peak
        dtype_txt = {'fp32': ('f32 (feed-forward and subsampling-conv2 GEMMs: fp32 operands as three exact bf16 '
                              'planes, six plane products on the bf16 matrix cores, f32 '
                              'accumulate -- error <= the v_mfma_f32 kernel\'s, '
                              'tests/test_gpu_x6.py; everything else v_mfma_f32)'
                              if x6 else 'f32'),
                     'bf16': 'bf16 operands, f32 accumulate / activations',
                     'fp8': 'OCP MXFP8 FFN GEMM operands (e4m3 elements, one E8M0 scale per 32 k '
                            'of a row, activations and weights), bf16 elsewhere, f32 '
                            'accumulate'}[args.dtype]
        line = {
            'metric': ('audio-seconds/sec (RTF^-1), Whisper-large-v3 encoder, '
                       if whisper else
                       'audio-seconds/sec (RTF^-1), 12L Conformer fbank80, ') + method,
            'value': round(value, 1),
            'unit': 'audio_s/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': dtype_txt,
            'data': 'synthetic',
            'config': {
                'workload': wl['text'] + ', features resident in HBM, random-init weights',
                'global_batch': batch_per_gpu * world,
                'audio_seconds_per_step': round(total_audio, 1),
                'encoder_frames_per_gpu': enc_rows,
                'parallelism': f'utterance-sharded x{world}, one all_gather of results',
                'process_group': (('gloo (ranks sharing one GPU: self-test)' if share_gpu
                                   else 'nccl (RCCL)') if dist_on else None),
                'decodes_in_flight_per_gpu': max(1, args.streams),
            },
            'rounds': {
                'n': len(round_s),
                'steps_per_round': args.steps,
                'timed_seconds': round(sum(round_s), 3),
                'ms_per_step_median': round(ms_per_step, 3),
                'ms_per_step_min': round(min(round_s) / args.steps * 1e3, 3),
                'ms_per_step_max': round(max(round_s) / args.steps * 1e3, 3),
                'ms_per_step_each': [round(r / args.steps * 1e3, 2) for r in round_s],
                'note': 'each round = exactly --steps steps between barrier + '
                        'synchronize brackets, max over ranks; value = median round',
            },
            'ranks': {
                'decode_ms_per_step': [round(r[0], 3) for r in rank_diag],
                'decode_ms_per_step_spread': round(max(r[0] for r in rank_diag)
                                                   - min(r[0] for r in rank_diag), 3),
                'drain_wait_ms_per_round': [round(r[1], 3) for r in rank_diag],
                'gather_ms_median': [round(r[2], 3) for r in rank_diag],
                'gather_ms_max': [round(r[3], 3) for r in rank_diag],
                'host_cpus_pinned': [{'n': int(r[4]), 'first': int(r[5]), 'last': int(r[6])}
                                     for r in rank_diag],
                'launcher': ('bench.py (wenet_amd.dist.launch_local_ranks)'
                             if os.environ.get('WN_SELF_LAUNCHED') == '1' else
                             'torch.distributed.run' if world > 1 or dist_on else None),
                'note': 'per rank, median timed round: time until the rank\'s own last result '
                        'was on the host / --steps; wait for outstanding result gathers at '
                        'the round\'s end; one result gather (pack + all_gather + unpack on '
                        'the worker thread, incl. waiting for the slowest peer) over all '
                        'gathers of the run.  ms_per_step is the max over ranks of the whole '
                        'round, these are its parts',
            },
            'roofline': {
                'bound': 'mfma',
                'kernel': (f'{prof_name}, M={enc_rows} F={ffn} D={d_model}, hidden split '
                           f'{ffn_split}: {4 * enc_rows * ffn * d_model / 1e9:.2f} algorithmic '
                           f'GFLOP per launch (both contractions), x 6 executed as '
                           f'v_mfma_f32_32x32x16_bf16; peak = dense bf16 peak 2500 / 6'
                           if 'ffn_x6f' in prof_name else
                           f'{prof_name}, M={enc_rows} N={ffn} K={d_model}: '
                           f'{2 * enc_rows * ffn * d_model / 1e9:.2f} algorithmic GFLOP per '
                           f'launch, x 6 executed as v_mfma_f32_32x32x16_bf16; peak = dense '
                           f'bf16 peak 2500 / 6' if x6 else
                           f'{prof_name}, M={enc_rows} F={ffn} D={d_model}: '
                           f'{4 * enc_rows * ffn * d_model / 1e9:.2f} GFLOP per launch'
                           if 'fused' in prof_name else
                           f'{prof_name} (FFN w_1, M={enc_rows} N={ffn} K={d_model})'),
                'regime': 'achieved / frac / avg_launch_us: ONE decode in flight (a pass of plain '
                          'decode() steps right after the timed rounds -- the kernel alone on the '
                          'chip, what rocprofv3 --streams 1 reproduces); the same event pairs '
                          'inside the headline\'s rounds (--streams decodes in flight) are under '
                          'timed_rounds',
                'achieved': round(achieved, 2),
                'peak': peak,
                'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4),
                'launches': tot_launch,
                'avg_launch_us': round(tot_ms * 1e3 / max(tot_launch, 1), 2),
                'timing': 'HIP events on the launch stream around every 6th launch of the '
                          f'kernel, {roof_steps} plain decode() steps with ONE decode in '
                          'flight right after the timed rounds (reproduce: rocprofv3 '
                          '--kernel-trace --stats -- python bench.py --streams 1)',
                'timed_rounds': {
                    'launches': timed_launch,
                    'avg_launch_us': round(timed_ms * 1e3 / max(timed_launch, 1), 2),
                    'achieved': round((timed_flops / (timed_ms * 1e-3)) / 1e12
                                      if timed_ms > 0 else 0.0, 2),
                    'frac': round(((timed_flops / (timed_ms * 1e-3)) / 1e12
                                   if timed_ms > 0 else 0.0) / peak, 4),
                    'note': 'the same event pairs inside the timed rounds: with --streams > 1 '
                            'they also time what the other decode runs beside the kernel',
                },
                'traffic': None,
                'whole_decode_tflops': round(whole_flops / world / (ms_per_step * 1e-3)
                                             / 1e12, 2),
                'whole_decode_frac': round(whole_flops / world / (ms_per_step * 1e-3)
                                           / 1e12 / whole_peak, 4),
                'whole_decode_peak': whole_peak,
                'whole_decode_note': 'encoder + CTC-head contraction FLOPs of one GPU\'s '
                                     'batch (SURVEY.md 8d formula) / ms_per_step / '
                                     'whole_decode_peak (fp32 mode: the six-product ceiling, '
                                     'dense bf16 peak / 6)',
            },
        }
        # HBM traffic of that kernel from the PMC passes (tools/gpu_pmc.sh; cannot
        # be collected inside a timed run): bytes per launch, committed summary
        # one record per workload and dtype: profiles/pmc_roofline_kernels.json (written by
        # tools/pmc_table.py --key from the end-of-round PMC passes)
        pmc = os.path.join(ROOT, 'profiles', 'pmc_roofline_kernels.json')
        rec = None
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f).get(f'{args.workload}:{args.dtype}')
            kind = lambda n: ('ffn_x6f' if 'ffn_x6f' in n else 'ffn_fused' if 'ffn_fused' in n
                              else 'x6' if 'x6' in n else 'lp' if 'gemm_lp' in n else 'gemm')
            # (the bf16 / fp8 modes name their FFN w_1 launch "gemm (FFN w_1)": at these shapes
            # it IS the pipelined low-precision kernel the record was taken from)
            same = (kind(rec.get('kernel', '')) == kind(prof_name) or
                    (args.dtype != 'fp32' and kind(prof_name) == 'gemm' and
                     kind(rec.get('kernel', '')) == 'lp')) if rec is not None else False
            if rec is not None and not same:
                rec = None     # the committed counters describe another kernel
        if rec is not None:
            line['roofline']['traffic'] = rec['hbm_bytes_per_launch']
            line['roofline']['traffic_unit'] = 'bytes/launch (HBM read + write, PMC)'
            line['roofline']['traffic_source'] = ('profiles/pmc_roofline_kernels.json, '
                                                  'visit ' + str(rec.get('visit', '?')) +
                                                  ' (' + str(rec.get('table', '')) + ')')
            if 'ffn_x6f' in prof_name:
                # the ALGORITHM's bytes: X planes in once (6 B / element: LN(x) arrives as
                # its plane image), W_1 + W_2 planes once, the result Y out once (fp32); the
                # hidden tensor never leaves the registers.  What this launch geometry adds
                # -- Y leaves as `ffn_split` hidden-slice partials instead of once -- is NOT
                # algorithmic: it is listed beside it as launch_geometry_bytes
                line['roofline']['algorithmic_bytes'] = int(
                    6 * enc_rows * d_model + 4 * enc_rows * d_model
                    + 6 * 2 * ffn * d_model)
                line['roofline']['launch_geometry_bytes'] = int(
                    6 * enc_rows * d_model + 4 * enc_rows * d_model * ffn_split
                    + 6 * 2 * ffn * d_model)
            elif x6:
                # A planes in (6 B / element), W planes (L2-resident across row tiles),
                # hidden-tensor planes out
                line['roofline']['algorithmic_bytes'] = int(
                    6 * (enc_rows * d_model + ffn * d_model + enc_rows * ffn))
            elif 'gemm_lp' in rec.get('kernel', ''):
                # operands and result in their storage types: bf16 A / W / C (2 B), or MXFP8
                # (1 B + one scale byte per 32 elements)
                esz = 2.0 if args.dtype == 'bf16' else 1.0 + 1.0 / 32
                line['roofline']['algorithmic_bytes'] = int(
                    esz * (enc_rows * d_model + ffn * d_model + enc_rows * ffn))
            elif 'ffn_fused' in rec.get('kernel', ''):
                # fused v_mfma_f32 FFN: X in, S hidden-slice partials out, W_1 + W_2 once
                line['roofline']['algorithmic_bytes'] = int(
                    4 * (enc_rows * d_model * (1 + ffn_split) + 2 * ffn * d_model))
            else:
                line['roofline']['algorithmic_bytes'] = int(
                    4 * (enc_rows * d_model + ffn * d_model + enc_rows * ffn))
        if line['roofline'].get('traffic') and line['roofline'].get('algorithmic_bytes'):
            line['roofline']['traffic_over_algorithmic'] = round(
                line['roofline']['traffic'] / line['roofline']['algorithmic_bytes'], 2)
        if clock_ghz is not None:
            line['roofline']['shader_clock_ghz_sampled'] = round(clock_ghz, 3)
            line['roofline']['frac_at_sampled_clock'] = round(
                achieved / (peak * clock_ghz / NOMINAL_GHZ), 4)
            line['roofline']['clock_note'] = (
                f'peak is priced at the {NOMINAL_GHZ} GHz peak engine clock; the kernel\'s own '
                'cycle / real-time stamps (one block, last launch of two extra untimed steps) give '
                'the clock the power management held under this load -- frac_at_sampled_clock '
                'prices the same achieved rate against the peak at that clock')
        if x6:
            line['roofline']['executed_mfma_tflops'] = round(6 * achieved, 1)
            line['roofline']['fp32_mfma_peak'] = PEAK_TFLOPS['fp32']
            line['roofline']['whole_decode_frac_of_fp32_mfma_peak'] = round(
                line['roofline']['whole_decode_tflops'] / PEAK_TFLOPS['fp32'], 4)
        if f32_only is not None:
            line['f32_mfma_only'] = {
                'value': round(total_audio * args.steps / f32_only, 1),
                'ms_per_step': round(f32_only / args.steps * 1e3, 3),
                'note': 'same decode, every GEMM on v_mfma_f32_32x32x2_f32 (gemm_x6 = 0), '
                        'one round of --steps steps',
            }
        if plain is not None:
            line['plain_decode'] = {
                'value': round(total_audio * args.steps / plain, 1),
                'ms_per_step': round(plain / args.steps * 1e3, 3),
                'note': 'same batch through back-to-back ASRModel.decode() calls (one decode '
                        'in flight, as wenet/bin/recognize.py:289 drives the reference), one '
                        'round of --steps steps; the headline keeps --streams decodes in '
                        'flight (wenet_amd.pipeline.DecodePipeline)',
            }
        if two_chained is not None:
            line['two_in_flight_chained'] = {
                'value': round(total_audio * args.steps / two_chained, 1),
                'ms_per_step': round(two_chained / args.steps * 1e3, 3),
                'note': 'the headline form of rounds 2-5: two decodes in flight, encoder i + 1 '
                        'starts when encoder i has finished (DecodePipeline(n_streams=2)); one '
                        'round of --steps steps.  The headline keeps --streams decodes in flight '
                        'with free-running encoders (wenet_amd/pipeline.py)',
            }
        if nbest_leg is not None:
            line['nbest_materialised'] = {
                'value': round(total_audio * args.steps / nbest_leg, 1),
                'ms_per_step': round(nbest_leg / args.steps * 1e3, 3),
                'note': 'the headline pipeline with .nbest / .nbest_scores / .nbest_times of '
                        'every DecodeResult read inside the timed round (3 x B x beam Python '
                        'lists per step, built on first access: wenet_amd/search.py '
                        '_NBestBatch), one round of --steps steps',
            }
        # what the timed steps produced vs the real reference's answer
        ver = verify.verify_bench_output(args.workload, world, out, method)
        line['verified'] = ver.pop('verified')
        line['verify'] = ver
        line['verify']['scope'] = ('1-best tokens of every utterance of the LAST timed step '
                                   'against the real reference\'s (tests/golden/bench_*.npz); '
                                   'scores, n-best lists and per-frame rules are the -m gpu '
                                   'tests\' job (tests/test_gpu_bench_parity.py)')
        if reduced and line['verified'] is False:
            # reduced-precision modes are not bit-comparable with the fp32 reference
            line['verify']['note'] = ('reduced-precision run: token identity with the fp32 '
                                      'reference is reported, not required; the pass criterion of '
                                      'this mode is per frame against the oracle under the same '
                                      'operand rounding at this shape (tests/golden/'
                                      'bench_config5_{bf16,fp8}.npz, tests/test_gpu_bench_parity.py)')
        if e2e is not None:
            line['end_to_end'] = e2e
        if not args.no_cpu_baseline and world == 1 and not whisper:
            line['cpu_baseline'] = cpu_baseline(configs, sd, feats, lens, method,
                                                decode_kw, beam)
        print(json.dumps(line), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
