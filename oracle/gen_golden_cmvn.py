"""Golden for the CMVN loader: the reference's own test
resource test/resources/global_cmvn parsed by the reference's own loader
(wenet/utils/cmvn.py:21-93 load_cmvn).  Stores the raw statistics (the input
fixture: mean_stat, var_stat, frame_num) and the loader's output (means, istd), so
tests/ can run wenet_amd.model.load_cmvn on the same statistics without the
reference tree.  Test infrastructure; run in the build container:

    python oracle/gen_golden_cmvn.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_harness  # noqa: E402


def main():
    _ref_harness.install()
    from wenet.utils.cmvn import load_cmvn
    src = os.path.join(_ref_harness.REFERENCE_ROOT, 'test', 'resources', 'global_cmvn')
    with open(src) as f:
        st = json.load(f)
    out = load_cmvn(src, True)  # (means, istd)
    np.savez_compressed(
        os.path.join(ROOT, 'tests', 'golden', 'cmvn_reference_resource.npz'),
        mean_stat=np.asarray(st['mean_stat'], np.float64),
        var_stat=np.asarray(st['var_stat'], np.float64),
        frame_num=np.asarray(st['frame_num'], np.float64),
        means=np.asarray(out[0], np.float64), istd=np.asarray(out[1], np.float64))
    print('wrote cmvn_reference_resource.npz', len(out[0]))


if __name__ == '__main__':
    main()
