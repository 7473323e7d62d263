"""Known-answer tests of the DEVICE field and group primitives against the oracle, word for word:
the inline-assembly Montgomery product (mont_mul_gfx950.inc), the portable CIOS product, add, sub,
neg, to/from Montgomery for Fq and Fr; G1 mixed add / full add / double including the P + P,
P + (-P) and infinity branches."""
import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import load_srs, rand_fr_wire

pytestmark = pytest.mark.gpu


def _edge_values(mod):
    vals = [0, 1, 2, mod - 1, mod - 2, (1 << 256) % mod, (1 << 255) % mod, (mod - 1) // 2, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF,
            (1 << 253) - 1, mod - (1 << 32), 0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF % mod]
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = opy.int_to_limbs(v)
    return out


@pytest.mark.parametrize("field,mod", [("fq", opy.P), ("fr", opy.R)])
def test_field_ops_match_oracle(gpu, field, mod):
    n = 1 << 16
    a = rand_fr_wire(n, 1)
    b = rand_fr_wire(n, 2)
    e = _edge_values(mod)
    ea = np.repeat(e, len(e), axis=0)
    eb = np.tile(e, (len(e), 1))
    a = np.concatenate([ea, a])
    b = np.concatenate([eb, b])
    mul_asm = gpu.field_op(field, 0, a, b)
    mul_c = gpu.field_op(field, 3, a, b)
    assert np.array_equal(mul_asm, mul_c)
    assert np.array_equal(gpu.field_op(field, 1, a, b), gpu.field_op(field, 8, a, b))   # asm add vs portable
    assert np.array_equal(gpu.field_op(field, 2, a, b), gpu.field_op(field, 9, a, b))   # asm sub vs portable
    add = gpu.field_op(field, 1, a, b)
    sub = gpu.field_op(field, 2, a, b)
    sqr = gpu.field_op(field, 4, a, b)
    neg = gpu.field_op(field, 5, a, b)
    frm = gpu.field_op(field, 6, a, b)
    tom = gpu.field_op(field, 7, a, b)
    to_int = lambda row: opy.limbs_to_int(row)
    idx = list(range(len(ea))) + list(range(len(ea), len(a), 97))
    for i in idx:
        x, y = to_int(a[i]), to_int(b[i])
        assert to_int(mul_asm[i]) == opy.mont_mul(x, y, mod), (field, i)
        assert to_int(add[i]) == (x + y) % mod
        assert to_int(sub[i]) == (x - y) % mod
        assert to_int(sqr[i]) == opy.mont_mul(x, x, mod)
        assert to_int(neg[i]) == (-x) % mod
        assert to_int(frm[i]) == opy.from_mont(x, mod)
        assert to_int(tom[i]) == opy.to_mont(x, mod)
    # the whole random block against the C oracle product
    fn = oc.lib.oracle_fq_mul if field == "fq" else oc.lib.oracle_fr_mul
    want = np.zeros(4, dtype=np.uint64)
    for i in range(len(ea), len(a), 7):
        fn(oc._p(np.ascontiguousarray(a[i])), oc._p(np.ascontiguousarray(b[i])), oc._p(want))
        assert np.array_equal(mul_asm[i], want)


def test_asm_product_equals_portable_on_a_million(gpu):
    n = 1 << 20
    a = rand_fr_wire(n, 11)
    b = rand_fr_wire(n, 12)
    for field in ("fq", "fr"):
        assert np.array_equal(gpu.field_op(field, 0, a, b), gpu.field_op(field, 3, a, b))


def test_group_ops_match_oracle(gpu):
    wire, pts = load_srs("lagrange-srs-4096.bin")
    n = 512
    a, b = wire[:n].copy(), wire[n:2 * n].copy()
    pa, pb = list(pts[:n]), list(pts[n:2 * n])
    # forced edge cases: P + P, P + (-P), inf + Q, P + inf, inf + inf
    b[0] = a[0]; pb[0] = pa[0]
    b[1] = oc.points_from_affine([opy.g1_neg(pa[1])])[0]; pb[1] = opy.g1_neg(pa[1])
    a[2] = 0; pa[2] = None
    b[3] = 0; pb[3] = None
    a[4] = 0; b[4] = 0; pa[4] = pb[4] = None
    aff = lambda j: oc.jac_to_affine_ints(j)
    madd = gpu.g1_op(0, a, b)
    full = gpu.g1_op(1, a, b)
    dbl = gpu.g1_op(2, a, b)
    msub = gpu.g1_op(3, a, b)
    twice = gpu.g1_op(4, a, b)
    qadd, qtwice, qmix = gpu.g1_op(5, a, b), gpu.g1_op(6, a, b), gpu.g1_op(7, a, b)     # four-lane addition (ecquad.hpp)
    q29 = [gpu.g1_op(op, a, b) for op in (8, 9, 10, 11, 12, 13)]                                # ... on 29-bit limbs (ecquad29.hpp)
    l29 = [gpu.g1_op(op, a, b) for op in range(14, 22)]        # the re-limbed operands and typed bounds of ec29l.hpp: one lane (14..17), quads (18..21)
    for i in range(n):
        s = opy.g1_add(pa[i], pb[i])
        for base in (0, 4):
            assert aff(l29[base][i]) == s, (base, i)
            assert aff(l29[base + 1][i]) == opy.g1_add(s, s), (base, i)
            assert aff(l29[base + 2][i]) == opy.g1_add(s, opy.g1_add(pa[i], opy.g1_neg(pb[i]))), (base, i)
            assert aff(l29[base + 3][i]) == s, (base, i)                # ((a + b) - (a + b)) + a + (b + infinity)
        assert aff(qadd[i]) == s, i
        assert aff(qtwice[i]) == opy.g1_add(s, s), i
        assert aff(qmix[i]) == opy.g1_add(s, opy.g1_add(pa[i], opy.g1_neg(pb[i]))), i
        assert aff(q29[0][i]) == s, i
        assert aff(q29[1][i]) == opy.g1_add(s, s), i
        assert aff(q29[2][i]) == opy.g1_add(s, opy.g1_add(pa[i], opy.g1_neg(pb[i]))), i
        s2 = opy.g1_add(s, s)
        assert aff(q29[3][i]) == opy.g1_add(s2, s2), i                # two doublings of an addition's (lazy) result
        assert aff(q29[4][i]) == s2, i
        a2 = opy.g1_add(pa[i], pa[i])
        assert aff(q29[5][i]) == opy.g1_add(a2, a2), i                # a doubling of a doubling's result
        assert aff(madd[i]) == s, i
        assert aff(full[i]) == s, i
        assert aff(dbl[i]) == opy.g1_add(pa[i], pa[i]), i
        assert aff(msub[i]) == opy.g1_add(pa[i], opy.g1_neg(pb[i])), i
        assert aff(twice[i]) == opy.g1_add(s, s), i


# ---- the 9 x 29-bit-limb representation used inside the hot loops (fp29.hpp) ---------------------
@pytest.mark.parametrize("field,mod", [("fq", opy.P), ("fr", opy.R)])
def test_l29_representation_matches_canonical_arithmetic(gpu, field, mod):
    """Products, lazy additions/subtractions, carry normalisation, canonicalisation and the
    2^256 <-> 2^261 Montgomery-form maps of the 29-bit-limb code agree word for word with the
    canonical 32-bit-limb arithmetic (itself checked against the oracle above) -- on edge values
    (0, 1, M-1, all-ones limbs, ...) in every combination and on 2^16 random pairs."""
    n = 1 << 16
    a = rand_fr_wire(n, 31)
    b = rand_fr_wire(n, 32)
    e = _edge_values(mod)
    a = np.concatenate([np.repeat(e, len(e), axis=0), a])
    b = np.concatenate([np.tile(e, (len(e), 1)), b])
    assert np.array_equal(gpu.field_op(field, 10, a, b), gpu.field_op(field, 3, a, b))     # mul
    assert np.array_equal(gpu.field_op(field, 11, a, b), gpu.field_op(field, 8, a, b))     # add
    assert np.array_equal(gpu.field_op(field, 12, a, b), gpu.field_op(field, 9, a, b))     # sub, 4M offset
    assert np.array_equal(gpu.field_op(field, 13, a, b), gpu.field_op(field, 9, a, b))     # sub, 12M offset
    assert np.array_equal(gpu.field_op(field, 15, a, b), a if False else gpu.field_op(field, 1, a, np.zeros_like(a)))   # form round trip == a mod M
    assert np.array_equal(gpu.field_op(field, 16, a, b), gpu.field_op(field, 17, a, b))    # sqr == mul(t, t), lazy t
    three_b = gpu.field_op(field, 8, b, gpu.field_op(field, 8, b, b))
    assert np.array_equal(gpu.field_op(field, 18, a, b), gpu.field_op(field, 9, a, three_b))   # x - 3y, offset 4M/T3
    assert np.array_equal(gpu.field_op(field, 19, a, b), gpu.field_op(field, 9, a, b))     # x - y, offset 2M/T1
    dual = gpu.field_op(field, 8, gpu.field_op(field, 3, a, b), gpu.field_op(field, 3, gpu.field_op(field, 8, a, b), a))
    assert np.array_equal(gpu.field_op(field, 20, a, b), dual)                             # x*y + (x+y)*x, one reduction
    for asm_op, cpp_op in ((10, 21), (16, 22), (20, 23)):                                   # assembly chains == C++ forms
        assert np.array_equal(gpu.field_op(field, asm_op, a, b), gpu.field_op(field, cpp_op, a, b))
    # lazy chain (op 14): t = 2*mul261(x - y, x + y) - mul261(x - y, y * 2^5), result t * 2^-5, where
    # mul261(u, v) = u*v*2^-261 -- un-normalised operands at the documented limb bounds
    got = gpu.field_op(field, 14, a, b)
    to_int = lambda row: opy.limbs_to_int(row)
    i261, i5 = pow(1 << 261, -1, mod), pow(32, -1, mod)
    for i in list(range(len(e) * len(e))) + list(range(len(e) * len(e), len(a), 499)):
        x, y = to_int(a[i]) % mod, to_int(b[i]) % mod
        d = (x - y) % mod
        want = (2 * d * (x + y) * i261 - d * (y * 32) * i261) * i5 % mod
        assert to_int(got[i]) == want, (field, i)


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_constant_operand_product(gpu, field):
    """fp29.hpp mulc (Shoup's form on 29-bit limbs, the NTT's twiddle product): x * w mod M as plain integers for canonical w and its
    companion floor(w 2^261 / M); the first operand anywhere below 2^256 (op 24) or lazy, 2 (x + 4M) with limbs near 2^31.7 (op 25)."""
    import random
    mod = opy.R if field == "fr" else opy.P
    rng = random.Random(5)
    xs = [rng.randrange(1 << 256) for _ in range(300)] + [0, 1, mod - 1, mod, mod + 1, (1 << 256) - 1, 2 * mod, 5 * mod]
    ws = [rng.randrange(mod) for _ in range(300)] + [mod - 1, 0, 1, 2, mod - 2, (1 << 253), mod // 2, 3]
    to_wire = lambda vals: np.array([[(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)] for v in vals], dtype=np.uint64)
    from_wire = lambda a: [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]
    a, b = to_wire(xs), to_wire(ws)
    assert from_wire(gpu.field_op(field, 24, a, b)) == [x * w % mod for x, w in zip(xs, ws)]
    assert from_wire(gpu.field_op(field, 25, a, b)) == [2 * (x + 4 * mod) * w % mod for x, w in zip(xs, ws)]
    assert from_wire(gpu.field_op(field, 26, a, b)) == [((w << 261) // mod) % (1 << 256) for w in ws]
    # the passes' lazy reduction (op 27: reduce3 of x + w + 4M, raw): same residue, below 3M, for any 256-bit operands
    ys = [rng.randrange(1 << 256) for _ in range(300)] + [(1 << 256) - 1, 0, 5 * mod, mod, mod - 1, (1 << 256) - 1, 1, 2 * mod + 1]
    got = from_wire(gpu.field_op(field, 27, a, to_wire(ys)))
    assert all(g < 3 * mod and (g - x - y) % mod == 0 for g, x, y in zip(got, xs, ys))


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_the_product_entry_point_runs_the_same_primitives(gpu, field):
    """uzk_field_op_device (include/uzkge_gpu.h: the host mirrors' seven plain operations) against the test hook of the same opcode."""
    a, b = rand_fr_wire(64, 5), rand_fr_wire(64, 6)          # < 2^252: valid words of either field
    for op in (0, 1, 2, 4, 5, 6, 7):
        assert np.array_equal(gpu.field_elementwise(field, op, a, b), gpu.field_op(field, op, a, b)), op


@pytest.mark.parametrize("field,mod", [("fq", opy.P), ("fr", opy.R)])
def test_typed_lazy_arithmetic_and_its_free_conversions(gpu, field, mod):
    """lz29.hpp: a canonical wire word re-limbed at bit offset -5 is the 2^261-form; values return by exact division by 2^5 / 2^10.
    Ops 28..33 against the 8 x 32-bit primitives (themselves held to the oracle above) on random and edge operands (0, 1, M - 1,
    (M - 1) / 2, limb-boundary values, all pairs)."""
    n = 1 << 14
    e = _edge_values(mod)
    a = np.concatenate([rand_fr_wire(n, 11), np.repeat(e, len(e), axis=0)])
    b = np.concatenate([rand_fr_wire(n, 12), np.tile(e, (len(e), 1))])
    mul, sub, sqr_b = gpu.field_op(field, 0, a, b), gpu.field_op(field, 2, a, b), gpu.field_op(field, 4, b, b)
    assert np.array_equal(gpu.field_op(field, 28, a, b), a)
    assert np.array_equal(gpu.field_op(field, 29, a, b), mul)
    assert np.array_equal(gpu.field_op(field, 30, a, b), sub)
    assert np.array_equal(gpu.field_op(field, 31, a, b), gpu.field_op(field, 1, mul, sqr_b))
    assert np.array_equal(gpu.field_op(field, 32, a, b), a)
    two_a = gpu.field_op(field, 1, a, a)
    assert np.array_equal(gpu.field_op(field, 33, a, b), gpu.field_op(field, 2, gpu.field_op(field, 1, two_a, two_a), b))
