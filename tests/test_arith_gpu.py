"""The scalar functions and lane collectives whose bits decide where an LD_MMA solve stops (csrc/mmm_arith.h, dev_math.h),
evaluated on the device through mmm_debug_math and compared BIT FOR BIT with the host: the same header compiled by gcc
(oracle/mmm_twin.c), IEEE division / square root of numpy, and the summation trees written out in Python."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(mmm, ctx, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    bp = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    mmm._lib.check(mmm.lib().mmm_debug_math(ctx.h, op, a.size, a, None if bp is None else bp.ctypes.data, out), ctx.h, "debug_math")
    return out


def _bits_equal(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    bad = x.view(np.int64) != y.view(np.int64)
    bad &= ~(np.isnan(x) & np.isnan(y))
    assert not bad.any(), "%d of %d values differ; first: %r vs %r" % (bad.sum(), x.size, x[bad][:3], y[bad][:3])


@pytest.fixture(scope="module")
def ctx(mmm):
    return mmm.Context(0)


def test_exp_log_digamma_bits(mmm, oracle, ctx):
    rng = np.random.default_rng(11)
    L = oracle.lib()
    xs = np.concatenate([rng.uniform(-40, 40, 200000), rng.uniform(-745, 709, 20000), rng.normal(0, 1e-4, 2000), [0.0, -0.0, 710.0, -746.0, 1e-320]])
    ref = np.empty_like(xs); L.orc_ar_exp_vec(xs.size, xs, ref)
    _bits_equal(_dev(mmm, ctx, 0, xs), ref)
    np.testing.assert_allclose(ref[:200000], np.exp(xs[:200000]), rtol=3e-16)
    xs = np.concatenate([rng.uniform(1e-7, 30, 200000), 10.0 ** rng.uniform(-300, 300, 20000), 1.0 + rng.normal(0, 1e-6, 2000)])
    ref = np.empty_like(xs); L.orc_ar_log_vec(xs.size, xs, ref)
    _bits_equal(_dev(mmm, ctx, 1, xs), ref)
    np.testing.assert_allclose(ref, np.log(xs), rtol=4e-16, atol=1e-18)
    xs = np.concatenate([rng.uniform(1e-3, 200, 100000), 10.0 ** rng.uniform(-7, 8, 20000)])
    ref = np.empty_like(xs); L.orc_ar_digamma_vec(xs.size, xs, ref)
    _bits_equal(_dev(mmm, ctx, 2, xs), ref)


def test_table_exp_log_bits_and_accuracy(mmm, oracle, ctx):
    """ar_exp_tab / ar_log_tab (the exp / log of the LD_MMA objectives: 128-entry tables, no division): the device evaluates them from LDS,
    the CPU restatement from a static array of the same generated numbers -- bit for bit, including the ends of the range (v_ldexp_f64 against
    the host's ldexp on subnormal results) -- and they are as accurate as their header claims (exp < 1 ulp; log: absolute 2.5e-15 up to x = 30)."""
    import mpmath as mp
    rng = np.random.default_rng(21)
    L = oracle.lib()
    xs = np.concatenate([rng.uniform(-40, 40, 200000), rng.uniform(-760, 712, 40000), rng.normal(0, 1e-4, 2000), rng.uniform(-745.2, -707, 20000),
                         [0.0, -0.0, 709.7, 709.78, 709.79, 710.0, 1e5, -1e5, 1e300, -1e300, np.inf, -np.inf, np.nan, -745.0, -745.13, -745.14, -746.0, -750.0, 1e-320]])
    ref = np.empty_like(xs); L.orc_ar_exptab_vec(xs.size, xs, ref)
    _bits_equal(_dev(mmm, ctx, 9, xs), ref)
    with np.errstate(over="ignore"):
        np.testing.assert_allclose(ref[:200000], np.exp(xs[:200000]), rtol=2.3e-16)
    assert ref[-19 + 2] > 1e308 and np.isinf(ref[-19 + 4]) and np.isinf(ref[-19 + 8]) and ref[-19 + 9] == 0.0 and np.isnan(ref[-19 + 12]) and ref[-19 + 11] == 0.0
    mp.mp.dps = 40
    worst = 0.0
    for x, y in zip(xs[:3000], ref[:3000]):                 # against 40-digit values: < 1 ulp
        t = mp.exp(mp.mpf(float(x)))
        worst = max(worst, float(abs(mp.mpf(float(y)) - t) / t) / 2.0 ** -52)
    assert worst < 1.0, worst
    xs = np.concatenate([rng.uniform(1e-7, 30, 200000), 10.0 ** rng.uniform(-300, 300, 20000), 1.0 + rng.normal(0, 1e-6, 2000), [1.0, 2.0, 0.5, 1e-7]])
    ref = np.empty_like(xs); L.orc_ar_logtab_vec(xs.size, xs, ref)
    _bits_equal(_dev(mmm, ctx, 10, xs), ref)
    assert np.max(np.abs(ref[:200000] - np.log(xs[:200000]))) < 2.5e-15
    worst = max(float(abs(mp.mpf(float(y)) - mp.log(mp.mpf(float(x))))) for x, y in zip(xs[:3000], ref[:3000]))
    assert worst < 2.5e-15, worst


def test_division_and_sqrt_are_ieee(mmm, ctx):
    """dev_div / dev_sqrt (the compiler's sequences without range handling) are correctly rounded in the range the MMA step
    algebra works in -- the host side of the parity tests uses plain `/` and sqrt()."""
    rng = np.random.default_rng(12)
    a = rng.normal(0, 1, 300000) * 10.0 ** rng.uniform(-12, 12, 300000)
    b = rng.normal(0, 1, 300000) * 10.0 ** rng.uniform(-12, 12, 300000)
    b[b == 0] = 1.0
    _bits_equal(_dev(mmm, ctx, 3, a, b), a / b)
    x = np.abs(a)
    _bits_equal(_dev(mmm, ctx, 4, x), np.sqrt(x))
    _bits_equal(_dev(mmm, ctx, 4, np.zeros(64)), np.zeros(64))


def _tree(v):
    v = list(v)
    while len(v) > 1:
        v = [v[2 * i] + v[2 * i + 1] for i in range(len(v) // 2)]
    return v[0]


def test_lane_sums_are_balanced_trees(mmm, ctx):
    """group_sum<L> = balanced tree over adjacent pairs (what oracle/mmm_twin.c writes); wave_sum = xor-butterfly 32,16,...,1."""
    rng = np.random.default_rng(13)
    x = rng.normal(0, 1, 64 * 50) * 10.0 ** rng.uniform(-8, 8, 64 * 50)
    for op, L in ((5, 16), (6, 32), (7, 64)):
        ref = np.repeat([_tree(x[i:i + L]) for i in range(0, x.size, L)], L)
        _bits_equal(_dev(mmm, ctx, op, x), ref)
    ref = []
    for i in range(0, x.size, 64):
        t = x[i:i + 64].copy()
        off = 32
        while off:
            t = t + t[np.arange(64) ^ off]
            off >>= 1
        ref.append(t)
    _bits_equal(_dev(mmm, ctx, 8, x), np.concatenate(ref))
