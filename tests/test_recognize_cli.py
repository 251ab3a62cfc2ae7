"""Host logic of wenet_amd/bin/recognize.py and wenet_amd/tokenizer.py (no GPU):
argument handling, list reading, batching / in-batch order, the rank-part merge,
detokenisation against the reference's tokenizers."""
import json
import os

import pytest

from conftest import needs_reference
from wenet_amd.bin import recognize as R
from wenet_amd.tokenizer import Tokenizer, get_blank_id, init_tokenizer


def test_args_reject_modes_outside_the_path():
    base = ['--config', 'c', '--test_data', 'd', '--checkpoint', 'k', '--result_dir', 'o']
    a = R.get_args(base + ['--modes', 'ctc_greedy_search', 'attention_rescoring'])
    assert a.modes == ['ctc_greedy_search', 'attention_rescoring'] and a.batch_size == 16
    with pytest.raises(SystemExit):
        R.get_args(base + ['--modes', 'rnnt_greedy_search'])
    with pytest.raises(SystemExit):
        R.get_args(base + ['--modes', 'attention', '--data_type', 'shard'])


def test_data_list_batches_and_order(tmp_path):
    p = tmp_path / 'data.list'
    p.write_text('\n'.join(json.dumps(dict(key=f'u{i}', wav=f'/w/{i}.wav', txt='x'))
                           for i in range(7)) + '\n\n')
    entries = R.read_data_list(str(p))
    assert entries[3] == ('u3', '/w/3.wav') and len(entries) == 7
    b = R.static_batches(entries, 3)
    assert [len(x) for x in b] == [3, 3, 1]
    # processor.padding: longest first, ties keep list order
    assert R.padding_order([50, 80, 50, 90, 80]) == [3, 1, 4, 0, 2]
    p.write_text('{"wav": "x"}\n')
    with pytest.raises(ValueError):
        R.read_data_list(str(p))


def test_merge_parts_restores_batch_order(tmp_path):
    d = tmp_path / 'm'
    d.mkdir()
    (d / 'text.part0').write_text('0\tu0 a\n0\tu1 b\n2\tu4 e\n')
    (d / 'text.part1').write_text('1\tu2 c\n1\tu3 d\n')
    R.merge_parts(str(tmp_path), ['m'], 2, 3)
    assert (d / 'text').read_text() == 'u0 a\nu1 b\nu2 c\nu3 d\nu4 e\n'
    assert sorted(os.listdir(d)) == ['text']


def test_override_config_and_feature_checks():
    cfg = {'input_dim': 80, 'dataset_conf': {'fbank_conf': {'num_mel_bins': 80}},
           'encoder_conf': {'num_blocks': 12}}
    R.override_config(cfg, ['encoder_conf.num_blocks 6'])
    assert cfg['encoder_conf']['num_blocks'] == 6
    with pytest.raises(KeyError):
        R.override_config(cfg, ['encoder_conf.nope 1'])
    R.check_feature_conf(cfg)
    cfg['dataset_conf']['fbank_conf']['frame_shift'] = 20
    with pytest.raises(NotImplementedError):
        R.check_feature_conf(cfg)


def test_tokenizer_char_and_bpe_detokenize():
    table = {'<blank>': 0, '<unk>': 1, '▁he': 2, 'llo': 3, '▁wor': 4, 'ld': 5, '你': 6}
    bpe = Tokenizer(table, 'bpe')
    assert bpe.detokenize([2, 3, 4, 5]) == ('he llo wor ld'.replace('he llo', 'hello')
                                            .replace('wor ld', 'world'),
                                            ['▁he', 'llo', '▁wor', 'ld'])
    ch = Tokenizer(table, 'char')
    assert ch.detokenize([6, 6])[0] == '你你'
    assert ch.detokenize([2, 3])[0] == '▁hello'  # char units keep the mark
    assert Tokenizer(table, 'char', connect_symbol=' ').detokenize([6, 6])[0] == '你 你'
    with pytest.raises(KeyError):
        ch.detokenize([99])
    cfg = {}
    assert get_blank_id(cfg, table) == 0 and cfg['ctc_conf']['ctc_blank_id'] == 0
    with pytest.raises(AssertionError):
        get_blank_id({}, {'a': 1})


@needs_reference
def test_tokenizers_match_the_reference(tmp_path):
    from oracle import _ref_harness
    _ref_harness.install()
    from wenet.utils.init_tokenizer import init_tokenizer as ref_init
    units = tmp_path / 'units.txt'
    syms = ['<blank>', '<unk>', '▁he', 'llo', '▁wor', 'ld', '你', '好', '<sos/eos>']
    units.write_text(''.join(f'{s} {i}\n' for i, s in enumerate(syms)))
    for kind, extra in (('char', {}), ('char', {'connect_symbol': ' '}),
                        ('char', {'split_with_space': True})):
        cfg = {'tokenizer': kind,
               'tokenizer_conf': dict(symbol_table_path=str(units), non_lang_syms_path=None,
                                      **extra)}
        ref, got = ref_init(dict(cfg)), init_tokenizer(cfg)
        for ids in ([6, 7], [2, 3, 4, 5], []):
            assert ref.detokenize(ids) == got.detokenize(ids)
        assert ref.symbol_table == got.symbol_table
    spm = pytest.importorskip('sentencepiece')
    corpus = tmp_path / 'c.txt'
    corpus.write_text('\n'.join(['HELLO WORLD', 'THE CAT SAT'] * 40))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / 'b'),
                                   vocab_size=30, model_type='bpe')
    sp = spm.SentencePieceProcessor()
    sp.load(str(tmp_path / 'b.model'))
    units.write_text(''.join(f'{sp.id_to_piece(i)} {i}\n'
                             for i in range(sp.get_piece_size())))
    cfg = {'tokenizer': 'bpe',
           'tokenizer_conf': dict(symbol_table_path=str(units), non_lang_syms_path=None,
                                  bpe_path=str(tmp_path / 'b.model'))}
    ref, got = ref_init(dict(cfg)), init_tokenizer(cfg)
    ids = sp.encode('HELLO THE CAT')
    assert ref.detokenize(ids) == got.detokenize(ids)


def test_transcribe_cli_args():
    from wenet_amd.bin import transcribe as T
    a = T.get_args(['a.wav', '-m', '/models/x'])
    assert a.audio_file == 'a.wav' and a.model == '/models/x' and a.beam is None
    assert a.context_score == 6.0 and a.device == 'cuda'
    with pytest.raises(SystemExit):
        T.get_args(['a.wav'])                      # no model download: -m is required
    with pytest.raises(SystemExit):
        T.get_args(['a.wav', '-m', 'x', '--device', 'cpu'])
