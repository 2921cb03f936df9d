"""Unit bounds of the fast-math device helpers, each against its exact counterpart (float64 torch), through
psl_selftest_math (include/pointslam_hip.h).  The parity tests bound these helpers end to end; the bounds here are what
lets somebody change one of them and know at once whether it still is what the kernels assume.

Reference counterparts: torch.sin / torch.cos in GaussianFourierFeatureTransform (src/conv_onet/models/decoder.py:33-36),
nn.Softplus(beta=100) (decoder.py:124,231,335; threshold 20), torch.optim.Adam's step with a zero gradient
(Mapper.py:394-402,556).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SINCOS, SOFTPLUS, SOFTPLUS_NB, SOFTPLUS_GRAD, ADAM_REPLAY = 0, 1, 2, 3, 4


def _run(kind, x, out_cols=1):
    from point_slam_amd import _lib
    dev = torch.device("cuda:0")
    xin = x.to(dev).contiguous()
    n = xin.shape[0]
    out = torch.empty(n * out_cols, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().psl_selftest_math(kind, _lib.ptr(xin), _lib.ptr(out), n, _lib.stream_ptr()), "psl_selftest_math")
    torch.cuda.synchronize()
    return out.cpu().reshape(n, out_cols) if out_cols > 1 else out.cpu()


def _ulp_err(got, exact64):
    """|got - exact| in units of the fp32 spacing at the exact value (floor: the spacing at 2^-126 for exact zeros)."""
    e32 = exact64.to(torch.float32)
    spacing = torch.from_numpy(np.spacing(np.abs(e32.numpy()).astype(np.float32))).double()
    return (got.double() - exact64).abs() / spacing


def test_fast_sincosf_against_libm_over_the_phase_range():
    """|x| <= 2e4 rad covers the phases the decoders produce: 2 pi p . B with B ~ N(0, 25^2 .. 32^2) and room-scale p
    ('up to ~1e4 rad').  Absolute error <= 1.2e-7 (one fp32 ulp of 1.0) everywhere; in ulps of the exact
    value <= 2 wherever |value| >= 1e-3 (near a zero of sin/cos a fixed ABSOLUTE error is what a reduced argument gives --
    libm's sinf has the same property for large arguments)."""
    g = torch.Generator().manual_seed(5)
    x = torch.cat([
        (torch.rand(2_000_000, generator=g) * 2 - 1) * 2.0e4,          # the whole range, uniformly
        (torch.rand(500_000, generator=g) * 2 - 1) * 8.0,              # small arguments: no reduction error to hide behind
        torch.arange(-4000, 4001, dtype=torch.float32) * (np.pi / 4),  # the polynomial interval boundaries
        torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e-8, 0.5, 1.0, 2.0e4, -2.0e4]),
    ]).float()
    got = _run(SINCOS, x, 2)
    xd = x.double()
    for col, exact in ((0, torch.sin(xd)), (1, torch.cos(xd))):
        err = (got[:, col].double() - exact).abs()
        assert float(err.max()) <= 1.2e-7, (col, float(err.max()))
        big = exact.abs() >= 1e-3
        ulps = _ulp_err(got[:, col][big], exact[big])
        assert float(ulps.max()) <= 2.0, (col, float(ulps.max()))
    # sin^2 + cos^2 = 1 to rounding: catches a quadrant mix-up that per-value errors near +-1 would hide
    assert float((got[:, 0].double() ** 2 + got[:, 1].double() ** 2 - 1).abs().max()) < 4e-7


def test_fast_sincosf_huge_and_non_finite_arguments():
    """|x| >= 1e5: one double-precision reduction by 2 pi first (selected, not branched to).  The result is the sine of the
    fp32 ARGUMENT to <= 4e-7 up to 1e8 (n = x / 2 pi <= 1.6e7 is exact in double, n x 2 pi is off by n x 2.4e-16 <= 4e-9, the
    reduced argument is rounded to fp32 once: <= 1.2e-7); inf / NaN give NaN."""
    g = torch.Generator().manual_seed(6)
    x = torch.cat([10 ** (5 + 3 * torch.rand(200_000, generator=g)), -(10 ** (5 + 3 * torch.rand(200_000, generator=g)))]).float()
    got = _run(SINCOS, x, 2)
    xd = x.double()
    assert float((got[:, 0].double() - torch.sin(xd)).abs().max()) < 4e-7
    assert float((got[:, 1].double() - torch.cos(xd)).abs().max()) < 4e-7
    bad = _run(SINCOS, torch.tensor([float("inf"), float("-inf"), float("nan")]), 2)
    assert bool(torch.isnan(bad).all())


def test_softplus100_against_torch():
    """nn.Softplus(beta=100, threshold=20) over the range the hidden activations live in ([-1, 1]) and beyond:
    |error| <= 4e-9 + 3e-7 |exact| -- an absolute floor for the tiny activations (log1p through v_log_f32 right above the
    t = 2^-10 switch: 6e-10) and ~2 ulp for the ones that matter (the exponent 100 x log2(e) is rounded twice before
    v_exp_f32: 1.3e-6 relative in t at 100 x = 20).  The branch-free twin used inside the register-chained kernels is
    bit-identical to the branchy one."""
    g = torch.Generator().manual_seed(7)
    x = torch.cat([(torch.rand(2_000_000, generator=g) * 2 - 1), (torch.rand(500_000, generator=g) * 2 - 1) * 0.25,
                   torch.linspace(-3, 3, 60001), torch.tensor([0.0, 0.2, 0.2000001, 0.1999999, -0.5, 50.0, -50.0])]).float()
    a, b = _run(SOFTPLUS, x), _run(SOFTPLUS_NB, x)
    assert torch.equal(a, b)
    exact = torch.nn.functional.softplus(x.double(), beta=100, threshold=20)
    err = (a.double() - exact).abs()
    excess = err - (4e-9 + 3e-7 * exact.abs())
    assert float(excess.max()) <= 0, (float(err.max()), float(x[excess.argmax()]))
    assert bool((a >= 0).all())


def test_softplus100_grad_from_output_against_autograd():
    """The backward kernels rebuild sigmoid(100 x) from the SAVED OUTPUT y = softplus(x): 1 - exp(-100 y).  Against autograd
    of torch's softplus in float64 at the same x: absolute error <= 5e-7 over [-1, 1] (y itself carries ~2 ulp, and d/dy of
    the expression is 100 (1 - sigma) <= 100; a numpy emulation with every hardware op 1 ulp off gives 2e-7)."""
    g = torch.Generator().manual_seed(8)
    x = torch.cat([(torch.rand(1_000_000, generator=g) * 2 - 1), (torch.rand(500_000, generator=g) * 2 - 1) * 0.1]).float()
    y = _run(SOFTPLUS_NB, x)
    got = _run(SOFTPLUS_GRAD, y)
    exact = torch.sigmoid(100.0 * x.double())
    exact = torch.where(100.0 * x.double() > 20.0, torch.ones_like(exact), exact)      # torch's threshold branch: slope 1
    err = (got.double() - exact).abs()
    assert float(err.max()) <= 5e-7, float(err.max())
    assert bool(((got >= 0) & (got <= 1)).all())


def test_lazy_adam_replay_against_the_ieee_step():
    """A row that receives no gradient for k iterations is stepped k times at once when it is next needed, with v_rcp / v_sqrt
    instead of the IEEE quotient and root (psl_adam.h: adam_replay).  m and v must be BIT-identical to the dense kernel's
    sequence.  The parameter differs by (a) the approximation of each increment (rcp, sqrt: 1 ulp each; the fma saves one
    rounding) and (b) one rounding of p per step in either sequence: |sum of increments, replay - IEEE| <=
    1e-6 x |sum of increments| + steps x ulp(p)."""
    g = torch.Generator().manual_seed(9)
    n = 200_000
    p = torch.randn(n, generator=g) * 0.3
    scale = 10 ** (-6 + 5 * torch.rand(n, generator=g))                # gradient scale of the row
    m = torch.randn(n, generator=g) * scale
    v = (0.1 + 3.9 * torch.rand(n, generator=g)) * scale ** 2           # |m| / sqrt(v) = O(1), as Adam keeps it
    lr = 10 ** (-4 + 2 * torch.rand(n, generator=g))
    t = torch.randint(1, 400, (n,), generator=g).float()
    lr_bc1 = lr / (1 - 0.9 ** t)
    sqrt_bc2 = torch.sqrt(1 - 0.999 ** t)
    steps = torch.randint(1, 65, (n,), generator=g).float()            # a prefetch block is 64 iterations
    x = torch.stack([p, m, v, lr_bc1, sqrt_bc2, steps], 1).float().contiguous()
    out = _run(ADAM_REPLAY, x, 6)
    assert torch.equal(out[:, 1], out[:, 4]) and torch.equal(out[:, 2], out[:, 5])      # m, v bit-identical
    inc_replay = out[:, 0].double() - x[:, 0].double()
    inc_ieee = out[:, 3].double() - x[:, 0].double()
    ulp_p = torch.from_numpy(np.spacing(np.maximum(np.abs(out[:, 3].numpy()), np.abs(x[:, 0].numpy())))).double()
    excess = (inc_replay - inc_ieee).abs() - (1e-6 * inc_ieee.abs() + x[:, 5].double() * ulp_p)
    assert float(excess.max()) <= 0, float(excess.max())
    assert float(inc_ieee.abs().max()) > 1e-3                             # the rows really moved
