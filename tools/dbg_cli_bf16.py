"""debug: bf16 CLI crash reproduction (temporary)"""
import os, sys, json, wave, faulthandler
faulthandler.enable()
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wenet_amd import synthetic as S
from wenet_amd.bin import recognize as R
tmp = sys.argv[1]
streams = sys.argv[2]
dtype = sys.argv[3]
os.makedirs(tmp, exist_ok=True)
configs = S.make_configs('tiny_causal'); sd = S.make_state_dict(configs, 0)
V = configs['output_dim']
syms = ['<blank>', '<unk>'] + [f't{i}' for i in range(2, V - 1)] + ['<sos/eos>']
open(f'{tmp}/units.txt', 'w').write(''.join(f'{s} {i}\n' for i, s in enumerate(syms)))
cfg = dict(configs); cfg['tokenizer'] = 'char'
cfg['tokenizer_conf'] = dict(symbol_table_path=f'{tmp}/units.txt', non_lang_syms_path=None, connect_symbol=' ')
cfg['dataset_conf'] = dict(fbank_conf=dict(num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0))
open(f'{tmp}/train.yaml', 'w').write(yaml.safe_dump(cfg))
torch.save(sd, f'{tmp}/final.pt')
rng = np.random.RandomState(5)
lines = []
for i in range(5):
    n = int(rng.randint(16000, 36000)); t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * (180 + 70 * i) * t) + 0.05 * rng.randn(n)
    with wave.open(f'{tmp}/u{i}.wav', 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.clip(x * 32768, -32768, 32767).astype(np.int16).tobytes())
    lines.append(json.dumps(dict(key=f'utt{i}', wav=f'{tmp}/u{i}.wav', txt='')))
open(f'{tmp}/raw.list', 'w').write('\n'.join(lines) + '\n')
modes = sys.argv[4].split(',')
for rep, dt in enumerate(dtype.split('+')):
    rc = R.main(['--config', f'{tmp}/train.yaml', '--checkpoint', f'{tmp}/final.pt', '--test_data', f'{tmp}/raw.list',
                 '--result_dir', f'{tmp}/out', '--batch_size', '2', '--beam_size', '3', '--ctc_weight', '0.5',
                 '--streams', streams, '--dtype', dt, '--modes'] + modes)
    print('run', rep, dt, 'rc', rc, flush=True)
