"""Pure-Python big-integer ORACLE for the uzkge PlonK hot path (BN254 G1 MSM + Fr NTT).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`uzkge_amd/`, `include/`) may import
this file; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg do,
and only as the checker.

What it restates (reference = /root/reference, zypher-game/uzkge @ 2025-02-12):
  * `G1Projective::msm(&points_raw, &coefs)`      uzkge/src/poly_commit/kzg_poly_commitment.rs:287-290
  * `domain.fft(&coefs)` / `domain.ifft(&values)` uzkge/src/poly_commit/field_polynomial.rs:583-597
  * coset variants (`mul_var` then fft)           uzkge/src/poly_commit/field_polynomial.rs:470-477,589-607
  * domain choice 2^k / 3*2^k                     uzkge/src/poly_commit/field_polynomial.rs:554-567
  * SRS binary format                             uzkge/src/poly_commit/kzg_poly_commitment.rs:206-264

The arithmetic itself lives in un-vendored crates (ark-ec-zypher / ark-poly-zypher /
ark-ff-zypher / ark-bn254-zypher, all `version = "0.4"`, Cargo.toml:28-38, no lockfile), so
this file restates the *published* algorithms (Montgomery form with R = 2^256, short
Weierstrass group law on y^2 = x^3 + 3, DFT over <omega_n> with omega_n = g^((r-1)/n), g = 5).
Both operations have a unique mathematical result, so any correct implementation is bit-exact
with arkworks once the representation and omega are fixed.

Pinning (tests/test_oracle_pinning.py): the oracle is checked against the reference's own
parameter files (tests/golden/lagrange-srs-*.bin, srs-padding.bin):
  sum_i L_i == G,  sum_i omega^i L_i == [tau]G  (pins omega and natural ordering),
  MSM(lagrange_srs, NTT(c)) == MSM(monomial_srs, c)  (pins NTT and MSM together).

Pure-Python loops: use for small cases only (n <= a few thousand).
"""
from __future__ import annotations

import struct
from typing import Iterable, List, Optional, Sequence, Tuple

# ---------------------------------------------------------------------------------------
# Constants (SURVEY.md section 8 constants block; re-derived below in _self_check()).
# ---------------------------------------------------------------------------------------
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq modulus
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr modulus
MONT_BITS = 256
MONT_R = 1 << MONT_BITS
FQ_R = MONT_R % P
FR_R = MONT_R % R
FQ_RINV = pow(MONT_R, -1, P)
FR_RINV = pow(MONT_R, -1, R)
FQ_INV64 = (-pow(P, -1, 1 << 64)) % (1 << 64)  # 0x87d20782e4866389
FR_INV64 = (-pow(R, -1, 1 << 64)) % (1 << 64)  # 0xc2e1f593efffffff
FQ_INV32 = FQ_INV64 & 0xFFFFFFFF
FR_INV32 = FR_INV64 & 0xFFFFFFFF
CURVE_B = 3
G1_GEN = (1, 2)
FR_GENERATOR = 5  # multiplicative generator used by ark-bn254 FrConfig
FR_TWO_ADICITY = 28
FR_TWO_ADIC_ROOT = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)

Affine = Optional[Tuple[int, int]]  # None == point at infinity


# ---------------------------------------------------------------------------------------
# Field helpers: canonical ints <-> Montgomery 4x64 LE limbs (the C-ABI wire format).
# ---------------------------------------------------------------------------------------
def to_mont(x: int, mod: int) -> int:
    return (x << MONT_BITS) % mod


def from_mont(x: int, mod: int) -> int:
    return (x * (FQ_RINV if mod == P else FR_RINV)) % mod


def int_to_limbs(x: int) -> Tuple[int, int, int, int]:
    return tuple((x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4))


def limbs_to_int(l: Sequence[int]) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def mont_mul(a: int, b: int, mod: int) -> int:
    """Montgomery product a*b*R^-1 mod `mod` (what ark-ff's MontBackend::mul_assign returns)."""
    return (a * b * (FQ_RINV if mod == P else FR_RINV)) % mod


def mont_mul_cios32(a: int, b: int, mod: int) -> int:
    """Word-level CIOS (32-bit words) restatement -- mirrors the device kernel step by step;
    used to KAT the big-int shortcut above against the limb algorithm."""
    inv = FQ_INV32 if mod == P else FR_INV32
    al = [(a >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    bl = [(b >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    pl = [(mod >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    t = [0] * 10
    for i in range(8):
        c = 0
        for j in range(8):
            s = t[j] + al[j] * bl[i] + c
            t[j] = s & 0xFFFFFFFF
            c = s >> 32
        s = t[8] + c
        t[8] = s & 0xFFFFFFFF
        t[9] = s >> 32
        m = (t[0] * inv) & 0xFFFFFFFF
        s = t[0] + m * pl[0]
        c = s >> 32
        for j in range(1, 8):
            s = t[j] + m * pl[j] + c
            t[j - 1] = s & 0xFFFFFFFF
            c = s >> 32
        s = t[8] + c
        t[7] = s & 0xFFFFFFFF
        t[8] = t[9] + (s >> 32)
    res = sum(t[i] << (32 * i) for i in range(9))
    if res >= mod:
        res -= mod
    return res


# ---------------------------------------------------------------------------------------
# G1 (y^2 = x^3 + 3 over Fq), canonical (non-Montgomery) affine coordinates.
# ---------------------------------------------------------------------------------------
def g1_is_on_curve(pt: Affine) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % P == 0


def g1_neg(pt: Affine) -> Affine:
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_add(a: Affine, b: Affine) -> Affine:
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


# Jacobian for speed in scalar-mul / MSM (no inversion per add).
def _jac_double(p):
    X, Y, Z = p
    if Z == 0:
        return p
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return _jac_double(p)
        return (1, 1, 0)
    H = (U2 - U1) % P
    I = (2 * H) * (2 * H) % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


def _to_jac(a: Affine):
    return (1, 1, 0) if a is None else (a[0], a[1], 1)


def jac_to_affine(p) -> Affine:
    X, Y, Z = p
    if Z % P == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_mul(pt: Affine, k: int) -> Affine:
    k %= R
    acc = (1, 1, 0)
    base = _to_jac(pt)
    while k:
        if k & 1:
            acc = _jac_add(acc, base)
        base = _jac_double(base)
        k >>= 1
    return jac_to_affine(acc)


def msm_naive(points: Sequence[Affine], scalars: Sequence[int]) -> Affine:
    """sum_i scalars[i] * points[i] by one double-and-add per term -- the definition the
    reference's own `test_commit` uses (kzg_poly_commitment.rs:526-548)."""
    acc = (1, 1, 0)
    for pt, k in zip(points, scalars):
        if pt is None or k % R == 0:
            continue
        acc = _jac_add(acc, _to_jac(g1_mul(pt, k)))
    return jac_to_affine(acc)


def msm_pippenger(points: Sequence[Affine], scalars: Sequence[int], c: int = 8) -> Affine:
    """Bucket method; independent second implementation used to cross-check msm_naive."""
    nwin = (254 + c - 1) // c
    total = (1, 1, 0)
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = _jac_double(total)
        buckets = [(1, 1, 0)] * (1 << c)
        for pt, k in zip(points, scalars):
            if pt is None:
                continue
            d = ((k % R) >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = _jac_add(buckets[d], _to_jac(pt))
        run = (1, 1, 0)
        acc = (1, 1, 0)
        for d in range((1 << c) - 1, 0, -1):
            run = _jac_add(run, buckets[d])
            acc = _jac_add(acc, run)
        total = _jac_add(total, acc)
    return jac_to_affine(total)


# ---------------------------------------------------------------------------------------
# Evaluation domains and NTT (natural order in and out).
# ---------------------------------------------------------------------------------------
def domain_supported(n: int) -> bool:
    """Radix2 (n = 2^k) or MixedRadix (n = 3 * 2^k) -- field_polynomial.rs:554-567.
    Fr's multiplicative group has 2-adicity 28 and 3-adicity 2; uzkge only uses 3^0 / 3^1."""
    if n <= 0:
        return False
    m = n
    if m % 3 == 0:
        m //= 3
    return m & (m - 1) == 0 and m <= (1 << FR_TWO_ADICITY)


def root_of_unity(n: int) -> int:
    """group_gen of the size-n domain: g^((r-1)/n) with g = 5.  For n = 2^k this equals
    TWO_ADIC_ROOT_OF_UNITY^(2^(28-k)) (ark-ff `get_root_of_unity`); for 3*2^k it is the
    large-subgroup root raised the same way.  The 2^k case is pinned by the reference's
    Lagrange SRS files (see module docstring)."""
    assert domain_supported(n), n
    return pow(FR_GENERATOR, (R - 1) // n, R)


def dft_naive(vals: Sequence[int], n: int, inverse: bool = False) -> List[int]:
    """O(n^2) definition: out[i] = sum_j vals[j] * w^(i*j); input zero-padded to n."""
    w = root_of_unity(n)
    if inverse:
        w = pow(w, -1, R)
    v = list(vals) + [0] * (n - len(vals))
    out = []
    for i in range(n):
        wi = pow(w, i, R)
        acc = 0
        x = 1
        for j in range(n):
            acc = (acc + v[j] * x) % R
            x = x * wi % R
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, R)
        out = [o * ninv % R for o in out]
    return out


def _ntt_rec(v: List[int], w: int) -> List[int]:
    n = len(v)
    if n == 1:
        return v
    if n % 2 == 0:
        even = _ntt_rec(v[0::2], w * w % R)
        odd = _ntt_rec(v[1::2], w * w % R)
        out = [0] * n
        x = 1
        h = n // 2
        for i in range(h):
            t = x * odd[i] % R
            out[i] = (even[i] + t) % R
            out[i + h] = (even[i] - t) % R
            x = x * w % R
        return out
    assert n == 3
    w2 = w * w % R
    a, b, c = v
    return [(a + b + c) % R, (a + b * w + c * w2) % R, (a + b * w2 + c * w2 * w2) % R]


def ntt(vals: Sequence[int], n: int, inverse: bool = False) -> List[int]:
    """`EvaluationDomain::fft` / `ifft` on a size-n domain: zero-pad, transform, natural order
    (field_polynomial.rs:583-597).  Canonical ints in and out."""
    assert len(vals) <= n and domain_supported(n)
    w = root_of_unity(n)
    if inverse:
        w = pow(w, -1, R)
    v = [x % R for x in vals] + [0] * (n - len(vals))
    if n % 3 == 0 and n > 3:
        # decimate by 3 once, then radix-2 below
        m = n // 3
        w3 = pow(w, 3, R)
        subs = [_ntt_rec(v[k::3], w3) for k in range(3)]
        out = [0] * n
        for i in range(n):
            wi = pow(w, i, R)
            out[i] = (subs[0][i % m] + wi * subs[1][i % m] + wi * wi % R * subs[2][i % m]) % R
    else:
        out = _ntt_rec(v, w)
    if inverse:
        ninv = pow(n, -1, R)
        out = [o * ninv % R for o in out]
    return out


def coset_ntt(vals: Sequence[int], n: int, k: int) -> List[int]:
    """coset_fft_with_domain: evaluate p(k*X) on the domain (field_polynomial.rs:589-591)."""
    x = 1
    scaled = []
    for c in vals:
        scaled.append(c * x % R)
        x = x * k % R
    return ntt(scaled, n)


def coset_intt(vals: Sequence[int], n: int, k_inv: int) -> List[int]:
    """coset_ifft_with_domain: ifft then mul_var(k_inv) (field_polynomial.rs:601-607)."""
    c = ntt(vals, n, inverse=True)
    x = 1
    out = []
    for ci in c:
        out.append(ci * x % R)
        x = x * k_inv % R
    return out


def poly_eval(coefs: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(coefs):
        acc = (acc * x + c) % R
    return acc


# ---------------------------------------------------------------------------------------
# Reference SRS file format (kzg_poly_commitment.rs:206-264; ark-serialize uncompressed G1:
# x LE32 || y LE32, flags in the top two bits of the last byte: bit7 = y-sign, bit6 = infinity).
# ---------------------------------------------------------------------------------------
def parse_srs_g1(data: bytes) -> List[Affine]:
    len1, _len2 = struct.unpack_from("<II", data, 0)
    pts: List[Affine] = []
    off = 8
    for _ in range(len1):
        xb = data[off : off + 32]
        yb = bytearray(data[off + 32 : off + 64])
        flags = yb[31] & 0xC0
        yb[31] &= 0x3F
        off += 64
        if flags & 0x40:
            pts.append(None)
        else:
            pts.append((int.from_bytes(xb, "little"), int.from_bytes(bytes(yb), "little")))
    return pts


def affine_to_wire(pt: Affine) -> bytes:
    """C-ABI `uzk_g1_affine`: x,y Montgomery 4x64 LE; infinity = all-zero."""
    if pt is None:
        return b"\0" * 64
    return to_mont(pt[0], P).to_bytes(32, "little") + to_mont(pt[1], P).to_bytes(32, "little")


def wire_to_affine(b: bytes) -> Affine:
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:64], "little")
    if x == 0 and y == 0:
        return None
    return (from_mont(x, P), from_mont(y, P))


def jac_wire_to_affine(b: bytes) -> Affine:
    """C-ABI `uzk_g1_jac` (x,y,z Montgomery) -> canonical affine."""
    X, Y, Z = (from_mont(int.from_bytes(b[32 * i : 32 * i + 32], "little"), P) for i in range(3))
    return jac_to_affine((X, Y, Z))


def fr_to_wire(xs: Iterable[int]) -> bytes:
    return b"".join(to_mont(x % R, R).to_bytes(32, "little") for x in xs)


def wire_to_fr(b: bytes) -> List[int]:
    return [from_mont(int.from_bytes(b[i : i + 32], "little"), R) for i in range(0, len(b), 32)]


def _self_check() -> None:
    assert FQ_INV64 == 0x87D20782E4866389 and FR_INV64 == 0xC2E1F593EFFFFFFF
    assert FQ_R == 0x0E0A77C19A07DF2F666EA36F7879462C0A78EB28F5C70B3DD35D438DC58F0D9D
    assert FR_R == 0x0E0A77C19A07DF2F666EA36F7879462E36FC76959F60CD29AC96341C4FFFFFFB
    assert FR_TWO_ADIC_ROOT == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert root_of_unity(1 << 14) == 20619701001583904760601357484951574588621083236087856586626117568842480512645
    assert root_of_unity(98304) == 12335946018549440216557516484429761123556309106189083444736381984701768732337
    assert g1_is_on_curve(G1_GEN)


_self_check()
