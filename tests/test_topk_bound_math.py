"""The error bound behind the threshold + collect top-K form (id-grec_amd/csrc/idg_score_bf16.inc), checked
numerically on the CPU: with u^ = bf16(u), v^ = bf16(v) (round to nearest even),

    | sum_k u^_k v^_k  -  sum_k u_k v_k |  <=  2^-7 (1 + 2^-9) sum_k |u_k v_k|  <=  2^-7 (1 + 2^-9) ||u|| ||v||

(bf16 keeps 8 significant bits: either operand moves by at most 2^-8 relative) and the constant the kernels use, bound_c(d)
= 0.00786 + 2e-7 d, leaves room above it for the fp32 accumulation on either
side (d 2^-24 for the fmaf chain, (d + 16) 2^-23 for the matrix cores, whatever their order) and for the rounding of the
bound product.  Also: rounding a positive float UP to bf16 the way bf16_round_up does never lands below it."""
import numpy as np
import pytest


def bf16_rne(x):
    """float32 -> bf16 (round to nearest even) -> float32"""
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def bf16_up(x):
    """the smallest bf16 >= x for x >= 0 (bits + 0xFFFF, low half cleared: idg_score_bf16.inc bf16_round_up)"""
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0xFFFF) >> 16) << 16).astype(np.uint32).view(np.float32)


def bound_c(d):
    return np.float32(0.00786) + np.float32(2.0e-7) * np.float32(d)


@pytest.mark.parametrize("d", [64, 128, 256])
def test_bf16_product_error_is_under_the_bound(d):
    rng = np.random.default_rng(d)
    worst = 0.0
    for trial in range(400):
        scale_u = 10.0 ** rng.uniform(-6, 4)
        scale_v = 10.0 ** rng.uniform(-6, 4)
        kind = trial % 4
        if kind == 0:  # plain Gaussian rows
            u, v = rng.standard_normal(d) * scale_u, rng.standard_normal(d) * scale_v
        elif kind == 1:  # every element in the worst spot between two bf16 values (just under the midpoint above 1), signs aligned
            w = (1.0 + 2.0 ** -8 * (1 - 2.0 ** -12)) * 2.0 ** rng.integers(-8, 8, d)  # (u parallel to v: sum |u v| = ||u|| ||v||)
            u, v = w * 2.0 ** rng.integers(-20, 20), w * 2.0 ** rng.integers(-20, 20)  # (powers of two keep the mantissas)
        elif kind == 2:  # a few huge features among tiny ones
            u, v = rng.standard_normal(d) * scale_u, rng.standard_normal(d) * scale_v
            u[rng.integers(0, d, 3)] *= 1e4
            v[rng.integers(0, d, 3)] *= 1e4
        else:  # cancelling products: the exact score is small, the error is not
            u = np.abs(rng.standard_normal(d)) * scale_u
            v = rng.standard_normal(d) * scale_v
            v[1::2] = -v[0::2] * u[0::2] / u[1::2]
        u, v = u.astype(np.float32), v.astype(np.float32)
        X = np.dot(u.astype(np.float64), v.astype(np.float64))
        S = np.dot(bf16_rne(u).astype(np.float64), bf16_rne(v).astype(np.float64))
        abs_sum = np.dot(np.abs(u).astype(np.float64), np.abs(v).astype(np.float64))
        norms = np.linalg.norm(u.astype(np.float64)) * np.linalg.norm(v.astype(np.float64))
        assert abs(S - X) <= 2.0 ** -7 * (1 + 2.0 ** -9) * abs_sum * (1 + 1e-12)
        assert abs_sum <= norms * (1 + 1e-12)
        worst = max(worst, abs(S - X) / norms)
    # what is left of bound_c(d) after the accumulation terms still covers the operand rounding
    spare = float(bound_c(d)) - (d * 2.0 ** -24 * 1.01 + (d + 16) * 2.0 ** -23 * 1.01)
    assert worst <= 2.0 ** -7 * (1 + 2.0 ** -9) < spare
    assert worst > 0.0077  # (the worst-spot rows do come close: the constant is not slack by a factor)


def test_round_up_to_bf16_never_rounds_down():
    rng = np.random.default_rng(0)
    x = np.concatenate([np.abs(rng.standard_normal(100000)).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 100000).astype(np.float32),
                        np.array([0.0, 1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7, 3.3895314e38], dtype=np.float32)])
    up = bf16_up(x)
    assert (up >= x).all()
    assert (up.view(np.uint32) & 0xFFFF == 0).all()  # a bf16
    assert (up <= x * np.float32(1 + 2.0 ** -7)).all() or np.isinf(up).any()  # at most one bf16 step above


def row_norm_up(v, lanes=8):
    """idg_score_bf16.inc row_norm_up restated in float32 numpy: the squares of the row scaled by a power of two that
    brings its largest element into [0.5, 1), eight chained fmaf per lane, a tree over the lanes, the root scaled back and
    padded by 1.000004, floored at 2^-40.  Returns (bound, regular)."""
    v = np.asarray(v, dtype=np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        mx = np.float32(np.max(np.abs(np.where(np.isnan(v), np.float32(0), v)))) if v.size else np.float32(0)  # (fmaxf drops NaNs)
        ex = 0 if not np.isfinite(mx) or mx == 0 else int(np.frexp(mx)[1])
        sv = np.ldexp(v, -ex).astype(np.float32)
        parts = []
        for piece in sv.reshape(-1, 8):  # one lane = 8 features, chained (fmaf: the square is exact in float64)
            acc = np.float32(0)
            for x in piece:
                acc = np.float32(np.float64(x) * np.float64(x) + np.float64(acc))
            parts.append(acc)
        parts = np.array(parts, dtype=np.float32)
        while len(parts) > 1:  # butterfly over the lanes
            parts = (parts[0::2] + parts[1::2]).astype(np.float32)
        nr = np.ldexp(np.float32(np.sqrt(parts[0], dtype=np.float32) * np.float32(1.000004)), ex).astype(np.float32)
    regular = bool(nr <= np.float32(2.0 ** 60))
    return max(np.float32(nr), np.float32(2.0 ** -40)) if not np.isnan(nr) else np.float32(2.0 ** -40), regular


@pytest.mark.parametrize("d", [64, 256])
def test_row_norm_bound_neither_underflows_nor_overflows(d):
    """sqrt(sum v^2) of a row of elements below ~1e-23 underflowed to 0 and the bound collapsed to 1e-30
    while bf16's operand error stayed relative.  The max-scaled form is an upper bound of the norm at every scale, within
    5e-6 of it (above the 2^-40 floor), and reports rows the bound arithmetic cannot take."""
    rng = np.random.default_rng(d)
    for trial in range(300):
        scale = 10.0 ** rng.uniform(-37, 16)
        v = (rng.standard_normal(d) * scale).astype(np.float32)
        if trial % 3 == 1:
            v[rng.integers(0, d, 4)] *= np.float32(1e-12)  # elements far below the largest: lost to the scaling, harmlessly
        if trial % 3 == 2:
            v[:] = np.float32(scale)  # a constant row
        true = float(np.linalg.norm(v.astype(np.float64)))
        got, regular = row_norm_up(v)
        assert regular
        assert float(got) >= true, (scale, float(got), true)
        if true > 2.0 ** -39:
            assert float(got) <= true * (1 + 5e-6), (scale, float(got), true)
    # what the old form did on such a row
    tiny = np.full(d, 1e-25, dtype=np.float32)
    assert np.float32(np.sqrt(np.sum(tiny * tiny, dtype=np.float32))) == 0 and float(row_norm_up(tiny)[0]) >= np.sqrt(d) * 1e-25
    for bad in (np.nan, np.inf, -np.inf, 3e38):
        v = rng.standard_normal(d).astype(np.float32)
        v[7] = bad
        assert not row_norm_up(v)[1], bad
    v = np.full(d, 2.0 ** 58, dtype=np.float32)  # norm 2^61 (d = 64) / 2^62: finite, but products of two such rows overflow
    assert not row_norm_up(v)[1]
    assert row_norm_up(np.zeros(d, dtype=np.float32)) == (np.float32(2.0 ** -40), True)


def test_survivor_cut_lies_below_tau_minus_two_eps():
    """the finish keeps a candidate when UB >= cut, cut = fl(tau - fl(fl(2.000002 cu) vmax)) taken one float
    DOWN.  It must never exceed the real number tau - 2 cu vmax — also when |tau| >> eps, where the subtraction's rounding
    (half an ulp of tau) is larger than the 2e-6 relative pad on eps; without the step down it does exceed it."""
    rng = np.random.default_rng(5)
    n = 200000
    tau = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 4, n)).astype(np.float32)
    cu = (np.abs(rng.standard_normal(n)) * 10.0 ** rng.uniform(-9, 0, n)).astype(np.float32)
    vmax = (np.abs(rng.standard_normal(n)) * 10.0 ** rng.uniform(-3, 2, n)).astype(np.float32)
    two_eps = ((np.float32(2.000002) * cu).astype(np.float32) * vmax).astype(np.float32)
    plain = (tau - two_eps).astype(np.float32)
    cut = np.nextafter(plain, np.float32(-np.inf))
    exact = tau.astype(np.float64) - 2.0 * cu.astype(np.float64) * vmax.astype(np.float64)
    assert (two_eps.astype(np.float64) >= 2.0 * cu.astype(np.float64) * vmax.astype(np.float64)).all()
    assert (cut.astype(np.float64) <= exact).all()
    assert (plain.astype(np.float64) > exact).any()  # (the case the step down is there for does occur)
    # ... and the step costs one ulp: nothing that could admit a crowd of extra survivors
    assert (plain.astype(np.float64) - cut.astype(np.float64) <= np.abs(plain.astype(np.float64)) * 2.0 ** -22 + 1e-44).all()
