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
