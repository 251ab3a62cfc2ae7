"""Host logic of context biasing (wenet_amd/context_graph.py): the product's
own ContextGraph against the oracle's restatement (itself pinned against the
reference in tests/test_oracle.py), the flat tables the C ABI takes, and the
flattening of a reference-style graph object."""
import numpy as np
import pytest

from conftest import needs_reference
from oracle import wenet_oracle as O
from wenet_amd.context_graph import ContextGraph, FlatGraph, flatten, tokenize


def _random_phrases(rng, vocab, n):
    return [[int(t) for t in rng.randint(1, vocab, rng.randint(1, 6))]
            for _ in range(n)]


def _walk_flat(f: FlatGraph, state: int, token: int):
    """The device algorithm (ctc.hip ctx_step) over the flat tables."""
    edges = {(int(a), int(t)): int(b)
             for a, t, b in zip(f.edge_from, f.edge_token, f.edge_to)}
    n = edges.get((state, token))
    if n is not None:
        score = f.token_score[n]
    else:
        n = int(f.fail[state])
        c = edges.get((n, token))
        while c is None:
            n = int(f.fail[n])
            c = edges.get((n, token))
            if n == 0:
                break
        if c is not None:
            n = c
        score = f.node_score[n] - f.node_score[state]
    return score + f.output_score[n], n


def test_matches_oracle_graph_on_random_phrase_lists():
    rng = np.random.RandomState(1)
    for _ in range(40):
        vocab = int(rng.randint(3, 10))
        phrases = _random_phrases(rng, vocab, int(rng.randint(1, 15)))
        score = float(rng.choice([0.5, 2.0, 6.0]))
        g = ContextGraph(context_list=phrases, context_score=score)
        o = O.ContextGraph(phrases, score)
        f = g.flat()
        assert g.num_nodes == o.num_nodes and f.n_nodes == o.num_nodes + 1
        assert f.fail.tolist() == o.fail
        assert f.node_score.tolist() == o.node_score
        assert f.output_score.tolist() == o.output_score
        assert f.n_edges == o.num_nodes  # a trie: one arc into every non-root node
        for _ in range(10):
            s = so = sf = 0
            for tok in rng.randint(0, vocab + 1, 30):
                a, s = g.forward_one_step(s, int(tok))
                b, so = o.forward_one_step(so, int(tok))
                c, sf = _walk_flat(f, sf, int(tok))
                assert a == b == c and s == so == sf
                assert g.finalize(s) == o.finalize(so)


def test_tokenize_char_units(tmp_path):
    table = {'<blank>': 0, '<unk>': 1, 'a': 2, 'b': 3, '▁': 4, '你': 5}
    lines = ['ab a\n', ' b你z \n', '\n', 'zzz\n']
    path = tmp_path / 'ctx.txt'
    path.write_text(''.join(lines), encoding='utf8')
    assert tokenize(str(path), table) == O.tokenize_context(lines, table)
    assert tokenize(str(path), table) == [[2, 3, 4, 2], [3, 5, 1], [], [1, 1, 1]]
    table.pop('<unk>')
    assert tokenize(str(path), table) == [[2, 3, 4, 2], [3, 5], [], []]
    g = ContextGraph(str(path), table, None, 3.0)
    assert g.num_nodes == 6 and g.context_list[0] == [2, 3, 4, 2]


def test_constructor_needs_a_phrase_source():
    with pytest.raises(ValueError):
        ContextGraph()


def test_empty_graph_is_only_the_root():
    f = ContextGraph(context_list=[[]]).flat()
    assert f.n_nodes == 1 and f.n_edges == 0 and f.fail.tolist() == [0]


@needs_reference
def test_flatten_accepts_the_reference_graph_object():
    from oracle import gen_golden_context
    rng = np.random.RandomState(5)
    for _ in range(10):
        vocab = int(rng.randint(3, 9))
        phrases = _random_phrases(rng, vocab, int(rng.randint(1, 12)))
        ref = gen_golden_context.reference_graph(phrases, 2.0)
        f = flatten(ref)
        assert flatten(ref) is f  # cached on the object
        assert f.n_nodes == ref.num_nodes + 1
        for _ in range(10):
            rs, s = ref.root, 0
            for tok in rng.randint(0, vocab + 1, 30):
                a, rs = ref.forward_one_step(rs, int(tok))
                b, s = _walk_flat(f, s, int(tok))
                assert a == b
                assert -f.node_score[s] == ref.finalize(rs)[0]


@needs_reference
def test_tokenize_bpe_matches_reference(tmp_path):
    spm = pytest.importorskip('sentencepiece')
    from oracle import _ref_harness
    _ref_harness.install()
    from wenet.utils.context_graph import tokenize as ref_tokenize
    corpus = tmp_path / 'corpus.txt'
    corpus.write_text('\n'.join(['HELLO WORLD', 'THE CAT SAT ON THE MAT',
                                 'SPEECH RECOGNITION', 'A QUICK BROWN FOX'] * 30))
    prefix = str(tmp_path / 'bpe')
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix,
                                   vocab_size=40, model_type='bpe',
                                   character_coverage=1.0)
    sp = spm.SentencePieceProcessor()
    sp.load(prefix + '.model')
    table = {sp.id_to_piece(i): i for i in range(sp.get_piece_size())}
    table['你'] = len(table)
    ctx = tmp_path / 'ctx.txt'
    ctx.write_text('the cat\nhello 你 world\nzebra\n', encoding='utf8')
    assert tokenize(str(ctx), table, prefix + '.model') == \
        ref_tokenize(str(ctx), table, prefix + '.model')
