import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
from wenet_amd import synthetic as S
from wenet_amd.model import ASRModel
configs = S.make_configs('aishell_u2pp'); sd = S.make_state_dict(configs, 0)
model = ASRModel(configs, sd, device='cuda:0')
feats, lens = S.make_bench_batch('config2', 1)
fd = feats.cuda()
for _ in range(5): model.decode(['ctc_prefix_beam_search'], fd, lens, beam_size=10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): model.decode(['ctc_prefix_beam_search'], fd, lens, beam_size=10)
torch.cuda.synchronize()
print('ms per decode', (time.perf_counter() - t0) / 30 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(30): model.decode(['ctc_prefix_beam_search'], fd, lens, beam_size=10)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
