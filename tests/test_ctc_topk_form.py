"""The selection order of the CTC top-k kernels (csrc/ctc.hip), restated on the CPU.

`ctc_row_wave_kernel` keeps a row in registers (lane l owns elements l + 64 e) and runs k
rounds of: per-lane scan with strict `>` over e ascending, wave arg-max (larger value, lower
index on ties), retire the winner as NaN.  `ctc_row_wave2_kernel` (ctc_wave = 2) keeps the
per-lane maxima of groups of 8 elements and rescans only the winner's group.  Both must pick
what torch.topk-with-stable-ties picks: value descending, index ascending among equals -- the
order the reference's `ctc_probs.topk(beam_size)` feeds the prefix beam search
(wenet/models/transformer/search.py:152-153) for distinct values, and a deterministic one
for ties.  This test replays both procedures lane by lane (no GPU)."""
import numpy as np
import pytest

G = 8


def _lanes(row, epl):
    v = np.full((64, epl), np.nan, dtype=np.float32)
    for i, x in enumerate(row):
        v[i % 64, i // 64] = x
    return v


def _wave_best(cands):
    """vi_wave: larger value first, lower index on ties; (-inf, 0x7fffffff) when empty."""
    best = (-np.inf, 0x7fffffff)
    for val, idx in cands:
        if val > best[0] or (val == best[0] and idx < best[1]):
            best = (val, idx)
    return best


def flat_topk(row, k, epl):
    v = _lanes(row, epl)
    out = []
    for _ in range(k):
        cands = []
        for lane in range(64):
            bv, bi = -np.inf, 0x7fffffff
            for e in range(epl):
                if v[lane, e] > bv:            # NaN compares false
                    bv, bi = v[lane, e], e * 64 + lane
            cands.append((bv, bi))
        val, idx = _wave_best(cands)
        out.append(idx)
        if idx != 0x7fffffff:
            v[idx & 63, idx >> 6] = np.nan
    return out


def two_level_topk(row, k, epl):
    v = _lanes(row, epl)
    ng = epl // G

    def scan(lane, g):
        bv, bi = -np.inf, 0x7fffffff
        for j in range(G):
            e = g * G + j
            if v[lane, e] > bv:
                bv, bi = v[lane, e], e * 64 + lane
        return bv, bi

    gv = [[scan(lane, g) for g in range(ng)] for lane in range(64)]
    out = []
    for _ in range(k):
        cands = []
        for lane in range(64):
            bv, bi = -np.inf, 0x7fffffff
            for g in range(ng):
                if gv[lane][g][0] > bv:
                    bv, bi = gv[lane][g]
            cands.append((bv, bi))
        val, idx = _wave_best(cands)
        out.append(idx)
        te = idx >> 6
        tg = te >> 3
        if tg < ng:                              # (0x7fffffff: no group, nothing to retire)
            v[idx & 63, te] = np.nan
            for lane in range(64):               # every lane rescans that group
                gv[lane][tg] = scan(lane, tg)
    return out


def expected(row, k):
    order = sorted(range(len(row)), key=lambda i: (-row[i], i))
    order = [i for i in order if row[i] > -np.inf]
    return (order + [0x7fffffff] * k)[:k]


@pytest.mark.parametrize('V,epl,k,kind', [
    (4233, 72, 10, 'random'), (4233, 72, 10, 'ties'), (500, 8, 10, 'ties'),
    (5002, 96, 16, 'random'), (70, 8, 10, 'mostly_neg_inf'), (4608, 72, 10, 'ties'),
    (130, 8, 16, 'all_equal'),
])
def test_two_level_selection_is_the_flat_scan_and_the_stable_topk(V, epl, k, kind):
    rng = np.random.default_rng(V + k)
    row = rng.standard_normal(V).astype(np.float32)
    if kind == 'ties':
        row = np.round(row * 2.0) / 2.0          # many equal values, also across lanes / groups
    elif kind == 'mostly_neg_inf':
        row[:] = -np.inf
        row[[3, 67, 5]] = [1.0, 1.0, 2.0]        # fewer finite elements than k
    elif kind == 'all_equal':
        row[:] = 0.25
    want = expected(row, k)
    assert flat_topk(row, k, epl) == want
    assert two_level_topk(row, k, epl) == want
