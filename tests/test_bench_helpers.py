"""CPU tests of the measurement helpers bench.py relies on: the output check against the
committed real-reference goldens (bench_verify.py) and the algorithmic FLOP count
(bench.contraction_flops)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from wenet_amd import synthetic as S  # noqa: E402
import bench_verify as verify  # noqa: E402


def _meta(name):
    z = np.load(os.path.join(verify.GOLDEN_DIR, name + '.npz'))
    return json.loads(bytes(z['meta']).decode('utf8'))


@pytest.mark.parametrize('workload', ['config2', 'config3', 'config4'])
def test_verify_accepts_the_reference_answer_and_rejects_anything_else(workload):
    meta = _meta(f'bench_{workload}')
    method = S.BENCH_WORKLOADS[workload]['method']
    exp = verify._expected(meta, method, 1)
    res = [(i, list(e[0]), 0.0) for i, e in enumerate(exp)]
    v = verify.verify_bench_output(workload, 1, res, method)
    assert v['verified'] is True and v['identical'] == len(exp) and v['mismatched'] == []
    # one token changed in one utterance: not verified, and the utterance is named
    bad = [list(r) for r in res]
    k = next(i for i, r in enumerate(bad) if len(r[1]) > 0)
    bad[k][1] = list(bad[k][1])
    bad[k][1][0] = int(bad[k][1][0]) + 1
    v = verify.verify_bench_output(workload, 1, [tuple(r) for r in bad], method)
    assert v['verified'] is False and k in [m if isinstance(m, int) else m[0]
                                            for m in v['mismatched']]
    # a missing utterance is a failure, not a pass over fewer items
    v = verify.verify_bench_output(workload, 1, res[:-1], method)
    assert v['verified'] is False


def test_verify_runner_up_only_inside_the_tie_window():
    meta = _meta('bench_config2')
    exp = verify._expected(meta, 'ctc_prefix_beam_search', 1)
    res = [(i, list(e[0]), 0.0) for i, e in enumerate(exp)]
    # the runner-up of an utterance whose top-2 gap is far outside the window is a mismatch
    k = max(range(len(exp)), key=lambda i: exp[i][2] if exp[i][1] is not None else -1)
    assert exp[k][1] is not None and exp[k][2] > verify.NBEST_TIE
    res[k] = (k, list(exp[k][1]), 0.0)
    v = verify.verify_bench_output('config2', 1, res, 'ctc_prefix_beam_search')
    assert v['verified'] is False


def test_world_8_golden_covers_eight_groups_and_unknown_workloads_report_none():
    # configs[4] has a golden of its configured shape since round 3 (greedy tokens of the real
    # reference's 32-block encoder): the right answer verifies, a short list does not
    meta5 = _meta('bench_config5')
    good = [(i, list(t), 0.0) for i, t in enumerate(meta5['greedy'])]
    assert verify.verify_bench_output('config5', 1, good, 'ctc_greedy_search')['verified'] is True
    assert verify.verify_bench_output('config5', 1, [], 'ctc_greedy_search')['verified'] is False
    assert verify.verify_bench_output('config9', 1, [], 'ctc_greedy_search')['verified'] is None
    meta = _meta('bench_config2_w8')
    assert len(meta['groups']) == 8
    exp = verify._expected(meta, 'ctc_prefix_beam_search', 8)
    assert len(exp) == 8 * S.BENCH_WORKLOADS['config2']['batch']


def test_contraction_flops_matches_the_survey_figure():
    """11.776 GMAC for one 998-frame AIShell utterance (T' = 248)."""
    import bench
    configs = S.make_configs('aishell_u2pp')
    gmac = bench.contraction_flops(configs, [998]) / 2 / 1e9
    assert abs(gmac - 11.776) < 0.01
    # additive over utterances, quadratic only in the attention term
    a = bench.contraction_flops(configs, [998, 998])
    assert abs(a - 2 * bench.contraction_flops(configs, [998])) < 1.0


def test_freeze_host_heap_keeps_the_collector_enabled():
    import gc
    from wenet_amd.pipeline import freeze_host_heap
    try:
        n = freeze_host_heap()
        assert n > 0 and gc.isenabled() and gc.get_freeze_count() == n
    finally:
        gc.unfreeze()
    assert gc.get_freeze_count() == 0


def test_pmc_roofline_records_are_keyed_by_workload_and_dtype(tmp_path, monkeypatch):
    """profiles/pmc_roofline_kernels.json (bench.py's `roofline.traffic`): one record per
    '<workload>:<dtype>', filed by `tools/pmc_table.py --merge`; every committed record carries
    the fields bench.py reads and names the kernel family bench.py reports for that workload."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = json.load(open(os.path.join(root, 'profiles', 'pmc_roofline_kernels.json')))
    want = {'config2:fp32': 'ffn_x6f_kernel', 'config3:fp32': 'gemm_x6_kernel',
            'config4:fp32': 'gemm_x6_kernel', 'config5:bf16': 'gemm_lp_kernel<0',
            'config5:fp8': 'gemm_lp_kernel<1'}
    for key, fam in want.items():
        rec = table[key]
        assert fam in rec['kernel'], (key, rec['kernel'])
        # (the three fields are truncations of float averages: the sum may be off by one)
        assert abs(rec['hbm_bytes_per_launch'] - rec['hbm_read_bytes_per_launch'] -
                   rec['hbm_write_bytes_per_launch']) <= 1
        assert 0.0 < rec['mfma_busy'] < 1.0 and rec['avg_us'] > 0 and rec['visit']
    # --merge files a record under its key and leaves the others alone
    spec = importlib.util.spec_from_file_location('pmc_table', os.path.join(root, 'tools',
                                                                           'pmc_table.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    prof = tmp_path / 'profiles'
    prof.mkdir()
    (tmp_path / 'tools').mkdir()
    (prof / 'pmc_roofline_kernels.json').write_text(json.dumps({'config2:fp32': {'kernel': 'a'}}))
    rec = tmp_path / 'rec.json'
    rec.write_text(json.dumps({'key': 'config9:bf16', 'kernel': 'b', 'avg_us': 1.0}))
    monkeypatch.setattr(mod, '__file__', str(tmp_path / 'tools' / 'pmc_table.py'))
    mod.merge(str(rec))
    merged = json.loads((prof / 'pmc_roofline_kernels.json').read_text())
    assert merged == {'config2:fp32': {'kernel': 'a'}, 'config9:bf16': {'kernel': 'b',
                                                                        'avg_us': 1.0}}
