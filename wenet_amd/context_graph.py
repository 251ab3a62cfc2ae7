"""Context biasing graph for the GPU CTC prefix beam search.

Host-side mirror of the reference's ``ContextGraph``
(wenet/utils/context_graph.py:101-265): same constructor
(``context_list_path, symbol_table, bpe_model, context_score``), same automaton
(a trie of the biasing phrases with Aho-Corasick fail / output arcs, including
the reference's rule that a node is an end node only if it was CREATED as the
last token of a phrase).  The graph is kept as flat arrays, which is what
``wn_set_context_graph`` (include/wenet_amd.h) uploads; states are node ids
(0 = root).

``flatten(graph)`` also accepts the reference's own ``ContextGraph`` object (a
``root`` ContextState with ``next`` / ``fail`` / ``node_score`` / ...), so a
graph built by reference code can be handed to ``ASRModel.decode`` unchanged.
"""
import ctypes
import re
from collections import deque
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_CJK = re.compile(r'([一-鿿])')


def tokenize(context_list_path: str, symbol_table: Dict[str, int],
             bpe_model: Optional[str] = None) -> List[List[int]]:
    """context_graph.py:24-58: one phrase per line -> token ids.  Char units
    map ' ' to U+2581; with a BPE model the text is upper-cased, CJK characters
    stay single tokens and the rest goes through sentencepiece
    (text/tokenize_utils.py:18-19,28-64).  Symbols missing from the table become
    <unk> when the table has one and are dropped otherwise."""
    sp = None
    if bpe_model is not None:
        import sentencepiece as spm
        sp = spm.SentencePieceProcessor()
        sp.load(bpe_model)
    with open(context_list_path, 'r', encoding='utf8') as f:
        lines = f.readlines()
    unk = symbol_table.get('<unk>')
    phrases = []
    for line in lines:
        text = line.strip()
        if sp is None:
            units = ['▁' if ch == ' ' else ch for ch in text]
        else:
            units = []
            for piece in _CJK.split(text.upper()):
                if not piece.strip():
                    continue
                if _CJK.fullmatch(piece):
                    units.append(piece)
                else:
                    units.extend(sp.encode_as_pieces(piece))
        ids = []
        for u in units:
            if u in symbol_table:
                ids.append(symbol_table[u])
            elif unk is not None:
                ids.append(unk)
        phrases.append(ids)
    return phrases


class FlatGraph:
    """The arrays wn_set_context_graph takes (node 0 = root)."""

    def __init__(self, fail, node_score, output_score, token_score, edge_from,
                 edge_token, edge_to):
        self.fail = np.ascontiguousarray(fail, dtype=np.int32)
        self.node_score = np.ascontiguousarray(node_score, dtype=np.float64)
        self.output_score = np.ascontiguousarray(output_score, dtype=np.float64)
        self.token_score = np.ascontiguousarray(token_score, dtype=np.float64)
        self.edge_from = np.ascontiguousarray(edge_from, dtype=np.int32)
        self.edge_token = np.ascontiguousarray(edge_token, dtype=np.int32)
        self.edge_to = np.ascontiguousarray(edge_to, dtype=np.int32)

    @property
    def n_nodes(self) -> int:
        return int(self.fail.shape[0])

    @property
    def n_edges(self) -> int:
        return int(self.edge_from.shape[0])


class ContextGraph:
    """Biasing phrases as an Aho-Corasick automaton over token ids."""

    def __init__(self, context_list_path: Optional[str] = None,
                 symbol_table: Optional[Dict[str, int]] = None,
                 bpe_model: Optional[str] = None, context_score: float = 6.0,
                 context_list: Optional[Sequence[Sequence[int]]] = None):
        """Either ``context_list_path`` + ``symbol_table`` (the reference's
        signature) or ``context_list`` (token-id lists) directly."""
        self.context_score = context_score
        if context_list is None:
            if context_list_path is None or symbol_table is None:
                raise ValueError('ContextGraph: give context_list_path + '
                                 'symbol_table, or context_list')
            context_list = tokenize(context_list_path, symbol_table, bpe_model)
        self.context_list = [list(map(int, p)) for p in context_list]
        self._children: List[Dict[int, int]] = [{}]
        self._token = [-1]
        self._depth_bonus = [0.0]     # node_score
        self._match_bonus = [0.0]     # output_score
        self._arc_bonus = [0.0]       # token_score
        self._ends_phrase = [False]
        self._fail = [0]
        self._insert_phrases()
        self._link_suffixes()
        self._flat: Optional[FlatGraph] = None

    # -- construction ---------------------------------------------------------
    def _insert_phrases(self):
        # context_graph.py:157-172.  A node's end flag is fixed at creation.
        for phrase in self.context_list:
            at = 0
            for pos, tok in enumerate(phrase):
                nxt = self._children[at].get(tok)
                if nxt is None:
                    nxt = len(self._children)
                    closing = pos + 1 == len(phrase)
                    bonus = self._depth_bonus[at] + self.context_score
                    self._children[at][tok] = nxt
                    self._children.append({})
                    self._token.append(tok)
                    self._depth_bonus.append(bonus)
                    self._match_bonus.append(bonus if closing else 0)
                    self._arc_bonus.append(self.context_score)
                    self._ends_phrase.append(closing)
                    self._fail.append(0)
                at = nxt

    def _suffix_target(self, start: int, tok: int) -> int:
        """Follow fail arcs from `start` until `tok` can be consumed or the root
        is reached (the reference's loop shape, context_graph.py:190-203 and
        :237-243: the root ends the walk even if it was reached by a fail arc)."""
        at = start
        while tok not in self._children[at]:
            at = self._fail[at]
            if at == 0:
                break
        return self._children[at].get(tok, at)

    def _link_suffixes(self):
        # breadth first, context_graph.py:175-214
        todo = deque(self._children[0].values())
        while todo:
            parent = todo.popleft()
            for tok, node in self._children[parent].items():
                pf = self._fail[parent]
                if tok in self._children[pf]:
                    target = self._children[pf][tok]
                else:
                    target = self._suffix_target(self._fail[pf], tok)
                self._fail[node] = target
                hit = target
                while not self._ends_phrase[hit]:
                    hit = self._fail[hit]
                    if hit == 0:
                        hit = None
                        break
                if hit is not None:
                    self._match_bonus[node] += self._match_bonus[hit]
                todo.append(node)

    # -- the reference's query API (states are node ids) ----------------------
    @property
    def num_nodes(self) -> int:
        return len(self._children) - 1

    @property
    def root(self) -> int:
        return 0

    def forward_one_step(self, state: int, token: int) -> Tuple[float, int]:
        """context_graph.py:216-248."""
        nxt = self._children[state].get(token)
        if nxt is not None:
            bonus = self._arc_bonus[nxt]
        else:
            nxt = self._suffix_target(self._fail[state], token)
            bonus = self._depth_bonus[nxt] - self._depth_bonus[state]
        return bonus + self._match_bonus[nxt], nxt

    def finalize(self, state: int) -> Tuple[float, int]:
        """context_graph.py:250-265."""
        return -self._depth_bonus[state], 0

    def flat(self) -> FlatGraph:
        if self._flat is None:
            ef, et, eo = [], [], []
            for n, ch in enumerate(self._children):
                for tok, c in ch.items():
                    ef.append(n)
                    et.append(tok)
                    eo.append(c)
            self._flat = FlatGraph(self._fail, self._depth_bonus, self._match_bonus,
                                   self._arc_bonus, ef, et, eo)
        return self._flat


def flatten(graph) -> FlatGraph:
    """FlatGraph of a wenet_amd ContextGraph, a FlatGraph, or a reference-style
    graph object (``graph.root`` with ``next`` / ``fail`` / ``node_score`` /
    ``output_score`` / ``token_score`` per ContextState)."""
    if isinstance(graph, FlatGraph):
        return graph
    if isinstance(graph, ContextGraph):
        return graph.flat()
    cached = getattr(graph, '_wenet_amd_flat', None)
    if cached is not None:
        return cached
    root = graph.root
    order = [root]
    index = {id(root): 0}
    todo = deque([root])
    while todo:
        st = todo.popleft()
        for child in st.next.values():
            if id(child) not in index:
                index[id(child)] = len(order)
                order.append(child)
                todo.append(child)
    ef, et, eo = [], [], []
    for st in order:
        for tok, child in st.next.items():
            ef.append(index[id(st)])
            et.append(int(tok))
            eo.append(index[id(child)])
    flat = FlatGraph([index[id(st.fail)] for st in order],
                     [float(st.node_score) for st in order],
                     [float(st.output_score) for st in order],
                     [float(st.token_score) for st in order], ef, et, eo)
    try:
        graph._wenet_amd_flat = flat
    except AttributeError:
        pass
    return flat


def install(lib, handle, graph, stream_ptr) -> None:
    """wn_set_context_graph(handle, graph) -- `graph` None clears."""
    from wenet_amd import _lib
    if graph is None:
        _lib.check(lib.wn_set_context_graph(handle, 0, None, None, None, None, 0,
                                            None, None, None, stream_ptr),
                   'wn_set_context_graph')
        return
    g = flatten(graph)
    f64p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))  # noqa: E731
    _lib.check(
        lib.wn_set_context_graph(handle, g.n_nodes, _lib.i32p(g.fail),
                                 f64p(g.node_score), f64p(g.output_score),
                                 f64p(g.token_score), g.n_edges,
                                 _lib.i32p(g.edge_from), _lib.i32p(g.edge_token),
                                 _lib.i32p(g.edge_to), stream_ptr),
        'wn_set_context_graph')
