"""The GELU of the pipelined bf16 / MXFP8 GEMM (csrc/gemm_bf16p.hip, gelu_e5):
x Phi(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2) with erfc(z) = 2^(-z G(z)), G a degree-5
polynomial.  Restated in fp32 numpy with the constants READ FROM THE SOURCE, against the exact
erf form the reference computes (torch.nn.functional.gelu, activation_type 'gelu'): the absolute
error bound and the relative accuracy in the negative tail the kernel's comment promises."""
import os
import re

import numpy as np
import torch
from scipy.special import erfc

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'wenet_amd', 'csrc',
                   'gemm_bf16p.hip')


def _constants():
    txt = open(SRC).read()
    body = txt[txt.index('float gelu_e5(float x)'):]
    body = body[:body.index('return')]
    lead = re.search(r'fmaf\((-?[0-9.e-]+)f, z, (-?[0-9.e-]+)f\)', body)
    rest = re.findall(r'g = fmaf\(g, z, (-?[0-9.e-]+)f\)', body)
    clamp = re.search(r'fminf\(ax \* ([0-9.]+)f, ([0-9.]+)f\)', body)
    coef = [float(lead.group(1)), float(lead.group(2))] + [float(c) for c in rest]
    assert len(coef) == 6, coef
    return coef, float(clamp.group(1)), float(clamp.group(2))


def gelu_e5(x, coef, rs2, zmax):
    f = np.float32
    ax = np.abs(x).astype(f)
    z = np.minimum(ax * f(rs2), f(zmax))
    g = np.full_like(z, f(coef[0]))
    for c in coef[1:]:
        g = g * z + f(c)
    e = np.exp2(-z * g).astype(f)
    return (f(-0.5) * ax) * e + np.maximum(x, f(0))


def test_gelu_e5_against_the_exact_erf_form():
    coef, rs2, zmax = _constants()
    assert abs(rs2 - 2 ** -0.5) < 1e-7 and zmax == 6.5
    x = np.linspace(-12, 12, 1200001, dtype=np.float32)
    xd = x.astype(np.float64)
    ref = 0.5 * xd * erfc(-xd / np.sqrt(2.0))          # = x Phi(x), accurate in the tail
    assert np.abs(ref - torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()).max() < 1e-14
    got = gelu_e5(x, coef, rs2, zmax).astype(np.float64)
    err = np.abs(got - ref)
    assert err.max() < 5e-6, err.max()
    # relative accuracy where the value is small because Phi(x) is (the negative tail down to
    # the clamp): the erf form of Abramowitz-Stegun 7.1.26 loses it to cancellation
    tail = (x < -1.0) & (x > -8.5)
    rel = err[tail] / np.abs(ref[tail])
    assert rel.max() < 3.5e-3, rel.max()
    near = (x < -1.0) & (x > -5.6)
    assert (err[near] / np.abs(ref[near])).max() < 2e-3
    # beyond the clamp both are below 1e-17 in magnitude
    far = x < -9.5
    assert np.abs(got[far]).max() < 1e-17 and np.abs(ref[far]).max() < 1e-17
    # bf16-rounded outputs: at least as often the exactly rounded value as the A&S form
    def bf16(v):
        return torch.from_numpy(v.astype(np.float32)).to(torch.bfloat16).float().numpy()
    xs = np.linspace(-9, 12, 400001, dtype=np.float32)
    refb = bf16(torch.nn.functional.gelu(torch.from_numpy(xs).double()).numpy())
    gotb = bf16(gelu_e5(xs, coef, rs2, zmax))
    assert (gotb != refb).mean() < 0.15
