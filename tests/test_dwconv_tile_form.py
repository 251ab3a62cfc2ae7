"""Index rule of the row-tiled depthwise convolution (csrc/encoder_kernels.hip,
dwconv_tiled_kernel) against the row-per-wave kernel's, restated on the CPU.

For output row `row` of the packed batch (utterance u, frame t = row - off[u]) and tap k the
row-per-wave kernel uses  x[off + tt] (tt = t + k - lpad inside the utterance), the pad frame
(tt < 0 and causal, or len <= tt < t_max) or nothing -- ConvolutionModule.forward's padding
(wenet/models/transformer/convolution.py:119-146).  The tiled kernel takes window row
i = r + j of the tile's R + 7 rows starting at packed row row0 + k0 - lpad, address clamped
into [0, M), and decides use / pad / skip per OUTPUT row.  Both must name the same source for
every (row, k), also where a tile straddles two utterances and at the ends of the tensor."""
import numpy as np
import pytest

R, TG = 4, 8


def per_row(M, row_utt, off, length, K, causal, t_max):
    lpad = K - 1 if causal else (K - 1) // 2
    out = {}
    for row in range(M):
        u = row_utt[row]
        if u < 0:
            continue
        t = row - off[u]
        if t >= length[u]:
            continue
        for k in range(K):
            tt = t + k - lpad
            if 0 <= tt < length[u]:
                out[(row, k)] = ('x', off[u] + tt)
            elif (tt < 0 and causal) or (length[u] <= tt < t_max):
                out[(row, k)] = ('pad', -1)
    return out


def tiled(M, row_utt, off, length, K, causal, t_max):
    lpad = K - 1 if causal else (K - 1) // 2
    out = {}
    for row0 in range(0, M, R):
        meta = []
        for r in range(R):
            row = row0 + r
            u = row_utt[row] if row < M else -1
            if u >= 0:
                meta.append((True and (row - off[u] < length[u]), row - off[u], length[u]))
            else:
                meta.append((False, 0, 0))
        for k0 in range(0, K, TG):
            window = [min(max(row0 + k0 - lpad + i, 0), M - 1) for i in range(R + TG - 1)]
            for r in range(R):
                on, t, ln = meta[r]
                for j in range(TG):
                    k = k0 + j
                    tt = t + k - lpad
                    if on and k < K:
                        if 0 <= tt < ln:
                            out[(row0 + r, k)] = ('x', window[r + j])
                        elif (tt < 0 and causal) or (ln <= tt < t_max):
                            out[(row0 + r, k)] = ('pad', -1)
    return out


@pytest.mark.parametrize('K,causal', [(8, True), (15, False), (31, False), (8, False), (33, True)])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_tile_rule_names_the_same_source_rows(K, causal, seed):
    rng = np.random.default_rng(seed * 100 + K)
    lens = list(rng.integers(1, 40, size=6)) + [1, 2, 3]
    rng.shuffle(lens)
    # packed rows: utterances back to back, a few skipped rows (row_utt = -1) in between
    row_utt, off, length = [], [], []
    for u, n in enumerate(lens):
        off.append(len(row_utt))
        # rows up to a padded count belong to the utterance, only the first n are frames
        pad = int(rng.integers(0, 3))
        length.append(int(n))
        row_utt += [u] * (int(n) + pad)
        if rng.random() < 0.3:
            row_utt += [-1] * int(rng.integers(1, 4))
    M = len(row_utt)
    t_max = max(lens) + 2
    a = per_row(M, row_utt, off, length, K, causal, t_max)
    b = tiled(M, row_utt, off, length, K, causal, t_max)
    assert a == b
