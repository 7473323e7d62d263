"""TEST INFRASTRUCTURE ONLY -- the optimal ate pairing on BN254, restated in plain Python integers.

Why it is here: the reference holds no fixture for the polynomial-opening half of the hot path
(`batch_prove`, uzkge/src/poly_commit/pcs.rs:107-168: evaluations, linear combination, division by
X - z, fold, FFT, Lagrange commit, blind factors).  But the reference's parameter file
`parameters/srs-padding.bin` carries, behind its G1 powers, the two G2 elements H and [tau]H
(`public_parameter_group_2`, kzg_poly_commitment.rs:176,195-198), and the reference's own verifier is
one pairing equation over them (kzg_poly_commitment.rs:344-371):

        e(C - [v]G, H)  ==  e(pi, [tau]H - [z]H)

A KZG opening proof is unique given (C, z, v): pi = [q(tau)]G with q = (f - v)/(X - z).  So an opening
produced by the HIP path that satisfies this equation over the reference's OWN G2 data is pinned by
reference-held data exactly as the commitments are pinned by the Lagrange identities -- without running
arkworks.  Only tests import this file (tests/test_oracle_pairing.py, tests/test_gpu_kzg_pairing.py).

The arithmetic lives in a third-party dependency that is absent from /root/reference: `ark-ec-zypher` /
`ark-bn254-zypher` 0.4 (`Bn254::pairing`, `Bn254::multi_pairing`; Cargo.toml:28-33).  Restated here from
the published algorithm (Vercauteren, "Optimal pairings", 2010; Beuchat et al., "High-speed software
implementation of the optimal ate pairing over Barreto-Naehrig curves", 2010):
   Fq2 = Fq[u]/(u^2 + 1),  Fq12 = Fq2[w]/(w^6 - xi), xi = 9 + u,
   twist E'/Fq2: y^2 = x^3 + 3/xi (D-type), untwist (x', y') -> (x' w^2, y' w^3),
   Miller loop over 6x + 2 (x = 4965661367192848881), two Frobenius line steps, final exponentiation
   (p^12 - 1)/r by plain square-and-multiply.
Anchored on reference data by tests/test_oracle_pairing.py: the file's first G2 element is the standard
generator, both lie on the twist, e(srs[i+1], H) == e(srs[i], [tau]H) (the reference's own parameter test,
kzg_poly_commitment.rs:440-470, run on its own file), and bilinearity on small multiples.
"""
from typing import List, Optional, Sequence, Tuple

from bn254_py import P, R, g1_neg  # same directory; test infra only

Fq2 = Tuple[int, int]                 # c0 + c1 u
G2Affine = Optional[Tuple[Fq2, Fq2]]  # None = infinity
Fq12 = List[Fq2]                      # sum_k c_k w^k, k < 6, w^6 = xi

BN_X = 4965661367192848881
ATE_LOOP = 6 * BN_X + 2
XI: Fq2 = (9, 1)

G2_GEN: G2Affine = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


# ---- Fq2 ----------------------------------------------------------------------------------------
def f2_add(a: Fq2, b: Fq2) -> Fq2: return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a: Fq2, b: Fq2) -> Fq2: return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a: Fq2) -> Fq2: return ((-a[0]) % P, (-a[1]) % P)
def f2_conj(a: Fq2) -> Fq2: return (a[0], (-a[1]) % P)
def f2_scale(a: Fq2, k: int) -> Fq2: return (a[0] * k % P, a[1] * k % P)


def f2_mul(a: Fq2, b: Fq2) -> Fq2:
    t0, t1 = a[0] * b[0], a[1] * b[1]
    return ((t0 - t1) % P, ((a[0] + a[1]) * (b[0] + b[1]) - t0 - t1) % P)


def f2_sqr(a: Fq2) -> Fq2:
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2_inv(a: Fq2) -> Fq2:
    d = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a: Fq2, e: int) -> Fq2:
    r: Fq2 = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


TWIST_B: Fq2 = f2_mul((3, 0), f2_inv(XI))      # 3 / xi


# ---- G2 (affine on the twist) -------------------------------------------------------------------
def g2_is_on_curve(q: G2Affine) -> bool:
    if q is None:
        return True
    x, y = q
    return f2_sqr(y) == f2_add(f2_mul(f2_sqr(x), x), TWIST_B)


def g2_neg(q: G2Affine) -> G2Affine:
    return None if q is None else (q[0], f2_neg(q[1]))


def g2_add(a: G2Affine, b: G2Affine) -> G2Affine:
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if a[1] != b[1] or a[1] == (0, 0):
            return None
        lam = f2_mul(f2_scale(f2_sqr(a[0]), 3), f2_inv(f2_scale(a[1], 2)))
    else:
        lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))


def g2_mul(q: G2Affine, k: int) -> G2Affine:
    k %= R
    acc: G2Affine = None
    while k:
        if k & 1:
            acc = g2_add(acc, q)
        q = g2_add(q, q)
        k >>= 1
    return acc


def parse_srs_g2(data: bytes) -> List[G2Affine]:
    """The G2 tail of KZGCommitmentScheme::to_unchecked_bytes (kzg_poly_commitment.rs:205-225): after
    `u32 len_g1 | u32 len_g2 | len_g1 x 64 B`, len_g2 uncompressed ark-serialize G2 points of 128 B --
    x.c0 | x.c1 | y.c0 | y.c1, 32 B little-endian each, flags in the top two bits of the last byte
    (bit 7: y is the lexicographically larger root, bit 6: infinity)."""
    n1 = int.from_bytes(data[0:4], "little")
    n2 = int.from_bytes(data[4:8], "little")
    off = 8 + 64 * n1
    assert len(data) == off + 128 * n2, "unexpected parameter file size"
    out: List[G2Affine] = []
    for i in range(n2):
        b = bytearray(data[off + 128 * i: off + 128 * (i + 1)])
        flags = b[127] & 0xC0
        b[127] &= 0x3F
        if flags & 0x40:
            out.append(None)
            continue
        c = [int.from_bytes(b[32 * k: 32 * k + 32], "little") for k in range(4)]
        assert all(v < P for v in c)
        out.append(((c[0], c[1]), (c[2], c[3])))
    return out


# ---- Fq12 = Fq2[w] / (w^6 - xi) -----------------------------------------------------------------
F12_ONE: Fq12 = [(1, 0)] + [(0, 0)] * 5


def f12_mul(a: Fq12, b: Fq12) -> Fq12:
    t = [[0, 0] for _ in range(11)]
    for i in range(6):
        ai = a[i]
        if ai == (0, 0):
            continue
        for j in range(6):
            bj = b[j]
            if bj == (0, 0):
                continue
            m = f2_mul(ai, bj)
            t[i + j][0] += m[0]
            t[i + j][1] += m[1]
    out = []
    for k in range(6):
        lo = (t[k][0] % P, t[k][1] % P)
        if k < 5:
            hi = f2_mul((t[k + 6][0] % P, t[k + 6][1] % P), XI)
            lo = f2_add(lo, hi)
        out.append(lo)
    return out


def f12_pow(a: Fq12, e: int) -> Fq12:
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


FINAL_EXP = (P ** 12 - 1) // R
assert (P ** 12 - 1) % R == 0

# Frobenius on the twist: pi(x', y') = (conj(x') g12, conj(y') g13), pi^2(x', y') = (x' g22, y' g23)
G12 = f2_pow(XI, (P - 1) // 3)
G13 = f2_pow(XI, (P - 1) // 2)
G22 = f2_pow(XI, (P * P - 1) // 3)
G23 = f2_pow(XI, (P * P - 1) // 2)


def _line(t: Tuple[Fq2, Fq2], q: Tuple[Fq2, Fq2], p: Tuple[int, int]):
    """Line through the twist points t, q (tangent when equal) evaluated at the untwisted G1 point p, and t + q.
    With slope lam' on the twist the untwisted line is  yP - lam' xP w + (lam' xT - yT) w^3.
    A vertical line (q = -t) returns (xP - xT w^2, infinity)."""
    xp, yp = p
    if t[0] == q[0] and t[1] != q[1]:
        return [(xp, 0), (0, 0), f2_neg(t[0]), (0, 0), (0, 0), (0, 0)], None
    if t == q:
        lam = f2_mul(f2_scale(f2_sqr(t[0]), 3), f2_inv(f2_scale(t[1], 2)))
    else:
        lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), t[0]), q[0])
    y3 = f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1])
    ell = [(yp, 0), f2_scale(f2_neg(lam), xp), (0, 0), f2_sub(f2_mul(lam, t[0]), t[1]), (0, 0), (0, 0)]
    return ell, (x3, y3)


def miller_loop(p, q: G2Affine) -> Fq12:
    """f_{6x+2,Q}(P) * l_{[6x+2]Q, pi(Q)}(P) * l_{[6x+2]Q + pi(Q), -pi^2(Q)}(P); 1 when either point is infinity."""
    if p is None or q is None:
        return F12_ONE
    f = F12_ONE
    t = q
    for bit in bin(ATE_LOOP)[3:]:
        ell, t = _line(t, t, p)
        f = f12_mul(f12_mul(f, f), ell)
        if bit == "1":
            ell, t = _line(t, q, p)
            f = f12_mul(f, ell)
    q1 = (f2_mul(f2_conj(q[0]), G12), f2_mul(f2_conj(q[1]), G13))
    q2 = (f2_mul(q[0], G22), f2_neg(f2_mul(q[1], G23)))      # -pi^2(Q)
    ell, t = _line(t, q1, p)
    f = f12_mul(f, ell)
    ell, _ = _line(t, q2, p)
    return f12_mul(f, ell)


def final_exponentiation(f: Fq12) -> Fq12:
    return f12_pow(f, FINAL_EXP)


def pairing(p, q: G2Affine) -> Fq12:
    """e(P, Q), P an affine G1 point (x, y) or None, Q an affine twist point or None."""
    return final_exponentiation(miller_loop(p, q))


def pairing_product_is_one(pairs: Sequence[Tuple[object, G2Affine]]) -> bool:
    """prod e(P_i, Q_i) == 1 with one final exponentiation (Bn254::multi_pairing, kzg_poly_commitment.rs:412-418)."""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return final_exponentiation(f) == F12_ONE


def kzg_verify(g1_0, g2_0: G2Affine, g2_1: G2Affine, cm, point: int, ev: int, proof, g1_mul, g1_add) -> bool:
    """KZGCommitmentScheme::verify (kzg_poly_commitment.rs:344-371):
    e(cm - [eval] g1_0, g2_0) == e(proof, g2_1 - [point] g2_0), as one product  e(lhs, g2_0) e(-proof, rhs) == 1."""
    lhs = g1_add(cm, g1_neg(g1_mul(g1_0, ev % R))) if ev % R else cm
    rhs = g2_add(g2_1, g2_neg(g2_mul(g2_0, point)))
    return pairing_product_is_one([(lhs, g2_0), (g1_neg(proof), rhs)])
