"""CPU check of the decomposition the GPU prefix-beam kernel is built on."""
import pytest
import torch

from oracle import wenet_oracle as O
from prefix_beam_formulation import emul


@pytest.mark.parametrize('seed0', [0, 100, 200])
def test_entry_formulation_equals_reference_loop(seed0):
    for trial in range(seed0, seed0 + 100):
        g = torch.Generator().manual_seed(trial)
        V = int(torch.randint(3, 8, (1, ), generator=g))
        T = int(torch.randint(5, 60, (1, ), generator=g))
        beam = int(torch.randint(1, min(V, 6) + 1, (1, ), generator=g))
        logits = torch.randn(T, V, generator=g) * 2
        logits[:, 0] += float(torch.rand(1, generator=g)) * 3
        for t in range(1, T, 2):
            logits[t] = logits[t - 1] + 0.1 * torch.randn(V, generator=g)
        logp = logits.log_softmax(-1)
        ref = O.ctc_prefix_beam_search(logp.unsqueeze(0), torch.tensor([T]),
                                       beam)[0]
        got = emul(logp, T, beam, canonical=True)
        assert [list(k) for k, _, _ in got] == [list(x) for x in ref.nbest]
        assert [t for _, _, t in got] == ref.nbest_times
        for (_, s, _), r in zip(got, ref.nbest_scores):
            assert s == r or abs(s - r) < 1e-9


@pytest.mark.parametrize('seed0', [0, 150])
def test_entry_formulation_with_context_graph(seed0):
    """Context biasing in the per-entry formulation: the entry takes its
    (state, bonus) from the first of its <= 3 contributions in the reference's
    loop order; rank on score + bonus; finalize at the end."""
    import numpy as np
    for trial in range(seed0, seed0 + 150):
        g = torch.Generator().manual_seed(trial)
        rng = np.random.RandomState(trial)
        V = int(torch.randint(3, 8, (1, ), generator=g))
        T = int(torch.randint(5, 50, (1, ), generator=g))
        beam = int(torch.randint(1, min(V, 6) + 1, (1, ), generator=g))
        logits = torch.randn(T, V, generator=g) * 2
        logits[:, 0] += float(torch.rand(1, generator=g)) * 3
        for t in range(1, T, 2):
            logits[t] = logits[t - 1] + 0.1 * torch.randn(V, generator=g)
        logp = logits.log_softmax(-1)
        phrases = [[int(t) for t in rng.randint(1, V, rng.randint(1, 5))]
                   for _ in range(rng.randint(1, 10))]
        graph = O.ContextGraph(phrases, float(rng.choice([0.5, 1.5, 4.0])))
        ref = O.ctc_prefix_beam_search(logp.unsqueeze(0), torch.tensor([T]), beam,
                                       0, graph)[0]
        got = emul(logp, T, beam, canonical=True, graph=graph)
        assert [list(k) for k, _, _ in got] == [list(x) for x in ref.nbest], trial
        assert [t for _, _, t in got] == ref.nbest_times
        for (_, s, _), r in zip(got, ref.nbest_scores):
            assert s == r or abs(s - r) < 1e-9
