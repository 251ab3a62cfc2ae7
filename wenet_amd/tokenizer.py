"""Token id <-> text for the result writer (the decode path only needs ids).

Mirrors what `init_tokenizer(configs)` (wenet/utils/init_tokenizer.py:26-60)
hands to `recognize.py` / the CLI for the two tokenizers the Conformer recipes
use:
  * `char`: CharTokenizer.detokenize = connect_symbol.join(symbols)
    (wenet/text/char_tokenizer.py:56-57, base_tokenizer.py:14-17);
  * `bpe`:  BpeTokenizer.detokenize = the same join with the sentencepiece
    word-boundary mark turned into spaces and stripped (bpe_tokenizer.py:48-51).
Text -> ids (training side) is out of scope, except for the biasing phrase list
(wenet_amd/context_graph.py).
"""
import os
from typing import Dict, List, Tuple


def read_symbol_table(path: str) -> Dict[str, int]:
    """wenet/utils/file_utils.py:61-68: `symbol id` per line."""
    table = {}
    with open(path, 'r', encoding='utf8') as f:
        for line in f:
            arr = line.strip().split()
            assert len(arr) == 2, f'bad symbol table line: {line!r}'
            table[arr[0]] = int(arr[1])
    return table


class Tokenizer:

    def __init__(self, symbol_table, kind: str = 'char', connect_symbol: str = '',
                 bpe_path: str = None):
        if not isinstance(symbol_table, dict):
            symbol_table = read_symbol_table(symbol_table)
        assert kind in ('char', 'bpe')
        self.kind = kind
        self.connect_symbol = connect_symbol if kind == 'char' else ''
        self.bpe_path = bpe_path
        self._symbol_table = symbol_table
        self.char_dict = {v: k for k, v in symbol_table.items()}

    @property
    def symbol_table(self) -> Dict[str, int]:
        return self._symbol_table

    def vocab_size(self) -> int:
        return len(self.char_dict)

    def ids2tokens(self, ids: List[int]) -> List[str]:
        return [self.char_dict[int(w)] for w in ids]  # KeyError like the reference

    def tokens2text(self, tokens: List[str]) -> str:
        text = self.connect_symbol.join(tokens)
        if self.kind == 'bpe':
            text = text.replace('▁', ' ').strip()
        return text

    def detokenize(self, ids: List[int]) -> Tuple[str, List[str]]:
        tokens = self.ids2tokens(ids)
        return self.tokens2text(tokens), tokens


def init_tokenizer(configs: dict, model_dir: str = None) -> Tokenizer:
    """init_tokenizer for `tokenizer: char | bpe`; with `model_dir`, paths of
    tokenizer_conf are first looked up next to the model like
    wenet/cli/model.py:41-45."""
    kind = configs.get('tokenizer', 'char')
    if kind not in ('char', 'bpe'):
        raise NotImplementedError(f'tokenizer {kind!r} (char and bpe are supported)')
    conf = dict(configs['tokenizer_conf'])
    if model_dir is not None:
        for key, value in conf.items():
            if isinstance(value, str):
                local = os.path.join(model_dir, os.path.basename(value))
                if os.path.exists(local):
                    conf[key] = local
    return Tokenizer(conf['symbol_table_path'], kind,
                     connect_symbol=conf.get('connect_symbol', ''),
                     bpe_path=conf.get('bpe_path'))


def get_blank_id(configs: dict, symbol_table: Dict[str, int]) -> int:
    """wenet/utils/ctc_utils.py:122-136."""
    ctc_conf = configs.setdefault('ctc_conf', {})
    if '<blank>' in symbol_table:
        if 'ctc_blank_id' in ctc_conf:
            assert ctc_conf['ctc_blank_id'] == symbol_table['<blank>']
        else:
            ctc_conf['ctc_blank_id'] = symbol_table['<blank>']
    else:
        assert 'ctc_blank_id' in ctc_conf, 'PLZ set ctc_blank_id in yaml'
    return ctc_conf['ctc_blank_id']
