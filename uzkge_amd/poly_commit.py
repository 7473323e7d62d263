"""Host-side mirror of the reference's polynomial layer for the hot path, over the C ABI.

  FpPolynomial                 uzkge/src/poly_commit/field_polynomial.rs:13-17, 86-90, 154-159,
                               198-209 (eval), 470-477 (mul_var), 554-607 (domains, fft wrappers)
  KZGCommitmentSchemeBN254     uzkge/src/poly_commit/kzg_poly_commitment.rs:170-313
                               (from_unchecked_bytes :228-264, commit :278-293,
                                apply_blind_factors :299-313)

Same names, argument meaning and error behaviour as the Rust (errors are `UzkgeError` with the
reference's variant names).  All heavy work goes to the GPU through `backend`; what stays here is
what stays in Rust in the real integration: trimming, zero-padding, domain choice, length checks.
Coefficients are numpy uint64 arrays [n, 4] in the wire format (Montgomery, 4 LE limbs).
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from . import backend as B
from .errors import UzkgeError

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_MASK64 = (1 << 64) - 1


def _to_wire(x: int, mod: int) -> np.ndarray:
    v = (x << 256) % mod
    return np.array([(v >> (64 * i)) & _MASK64 for i in range(4)], dtype=np.uint64)


def _wire_int(row: np.ndarray) -> int:
    return sum(int(row[i]) << (64 * i) for i in range(4))


def fr_from_int(x: int) -> np.ndarray:
    """canonical integer -> Montgomery wire limbs."""
    return _to_wire(x % FR_MODULUS, FR_MODULUS)


def fr_to_int(row: np.ndarray) -> int:
    return _wire_int(row) * pow(1 << 256, -1, FR_MODULUS) % FR_MODULUS


def fr_neg(rows: np.ndarray) -> np.ndarray:
    """-x on wire-format rows (Montgomery form is linear: r - x)."""
    out = np.zeros_like(rows)
    for i, row in enumerate(rows.reshape(-1, 4)):
        v = _wire_int(row)
        v = (FR_MODULUS - v) % FR_MODULUS
        out.reshape(-1, 4)[i] = [(v >> (64 * k)) & _MASK64 for k in range(4)]
    return out


class FpPolynomial:
    """Dense polynomial over BN254 Fr, coefficients low to high."""

    def __init__(self, coefs: np.ndarray):
        self.coefs = np.ascontiguousarray(coefs, dtype=np.uint64).reshape(-1, 4)

    # -- construction -------------------------------------------------------------------------
    @classmethod
    def from_coefs(cls, coefs: np.ndarray) -> "FpPolynomial":
        """Trims trailing zeros like the reference (field_polynomial.rs:86-90,154-159); the zero
        polynomial keeps one zero coefficient."""
        c = np.ascontiguousarray(coefs, dtype=np.uint64).reshape(-1, 4)
        nz = np.flatnonzero(c.any(axis=1))
        keep = int(nz[-1]) + 1 if nz.size else 1
        if c.shape[0] == 0:
            c = np.zeros((1, 4), dtype=np.uint64)
        return cls(c[:keep].copy())

    @classmethod
    def from_ints(cls, ints: Sequence[int]) -> "FpPolynomial":
        return cls.from_coefs(np.stack([fr_from_int(x) for x in ints]) if len(ints) else np.zeros((0, 4), np.uint64))

    def get_coefs_ref(self) -> np.ndarray:
        return self.coefs

    def degree(self) -> int:
        nz = np.flatnonzero(self.coefs.any(axis=1))
        return int(nz[-1]) if nz.size else 0

    # -- small host-side helpers ---------------------------------------------------------------
    def eval(self, point: int) -> int:
        """Horner (field_polynomial.rs:198-209), canonical ints."""
        acc = 0
        for row in self.coefs[::-1]:
            acc = (acc * point + fr_to_int(row)) % FR_MODULUS
        return acc

    # -- domains (field_polynomial.rs:554-567) ------------------------------------------------
    @staticmethod
    def evaluation_domain(num_coeffs: int) -> Optional[int]:
        assert num_coeffs > 0 and num_coeffs & (num_coeffs - 1) == 0
        return num_coeffs if B.domain_supported(num_coeffs) else None

    @staticmethod
    def quotient_evaluation_domain(num_coeffs: int) -> Optional[int]:
        m = num_coeffs // 3 if num_coeffs % 3 == 0 else num_coeffs
        assert num_coeffs > 0 and m & (m - 1) == 0
        return num_coeffs if B.domain_supported(num_coeffs) else None

    # -- transforms ---------------------------------------------------------------------------
    def _padded(self, n: int) -> np.ndarray:
        out = np.zeros((n, 4), dtype=np.uint64)
        out[: self.coefs.shape[0]] = self.coefs
        return out

    def fft(self, num_coeffs: int) -> Optional[np.ndarray]:
        """field_polynomial.rs:570-580."""
        assert num_coeffs > self.degree()
        dom = (self.evaluation_domain(num_coeffs) if num_coeffs & (num_coeffs - 1) == 0
               else self.quotient_evaluation_domain(num_coeffs))
        if dom is None:
            return None
        return self.fft_with_domain(dom)

    def fft_with_domain(self, domain: int) -> np.ndarray:
        """`domain.fft(&self.coefs)` (field_polynomial.rs:583-586): zero-pad, natural order."""
        assert domain > self.degree()
        return B.ntt(self._padded(domain)[:domain] if self.coefs.shape[0] <= domain else self.coefs[:domain])

    def coset_fft_with_domain(self, domain: int, k: np.ndarray) -> np.ndarray:
        """fft of p(kX) (field_polynomial.rs:589-591); the k^j scaling is fused on the device."""
        assert domain > self.degree()
        return B.ntt(self._padded(domain), coset_shift=k)

    @classmethod
    def ifft_with_domain(cls, domain: int, values: np.ndarray) -> "FpPolynomial":
        """`domain.ifft(values)` then from_coefs (field_polynomial.rs:594-597)."""
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        buf = np.zeros((domain, 4), dtype=np.uint64)
        buf[: v.shape[0]] = v
        return cls.from_coefs(B.ntt(buf, inverse=True))

    @classmethod
    def coset_ifft_with_domain(cls, domain: int, values: np.ndarray, k_inv: np.ndarray) -> "FpPolynomial":
        """ifft then mul_var(k_inv) (field_polynomial.rs:601-607)."""
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        buf = np.zeros((domain, 4), dtype=np.uint64)
        buf[: v.shape[0]] = v
        return cls.from_coefs(B.ntt(buf, inverse=True, coset_shift=k_inv))

    def __eq__(self, other) -> bool:
        return isinstance(other, FpPolynomial) and np.array_equal(
            FpPolynomial.from_coefs(self.coefs).coefs, FpPolynomial.from_coefs(other.coefs).coefs)


def parse_srs_g1_wire(data: bytes) -> np.ndarray:
    """The G1 section of a reference SRS blob as wire-format affine points [len, 8] (host only): u32 len_g1 | u32 len_g2 |
    len_g1 x (x LE32 || y LE32, flags in the top two bits of the last byte) | G2 points (kzg_poly_commitment.rs:228-264)."""
    if len(data) < 8:
        raise UzkgeError(N.UZK_ERR_PARAMETER, "DeserializationError: short SRS blob")
    len1, _len2 = struct.unpack_from("<II", data, 0)
    if len(data) < 8 + 64 * len1:
        raise UzkgeError(N.UZK_ERR_PARAMETER, "DeserializationError: truncated G1 section")
    out = np.zeros((len1, 8), dtype=np.uint64)
    for i in range(len1):
        off = 8 + 64 * i
        x = int.from_bytes(data[off:off + 32], "little")
        yb = bytearray(data[off + 32:off + 64])
        flags = yb[31] & 0xC0
        yb[31] &= 0x3F
        if flags & 0x40:
            continue                       # infinity -> (0, 0)
        y = int.from_bytes(bytes(yb), "little")
        out[i, :4] = _to_wire(x, FQ_MODULUS)
        out[i, 4:] = _to_wire(y, FQ_MODULUS)
    return out


def srs_params_wire(srs_blob: bytes, size: int) -> np.ndarray:
    """load_srs_params (uzkge/src/gen_params/mod.rs:151-183) as a wire array (host only): powers 0..2050, the identity
    up to `size`, then the three padding powers size, size + 1, size + 2."""
    if size > 16384:
        raise UzkgeError(N.UZK_ERR_PARAMETER, "ParameterError: size exceeds the embedded SRS")
    g1 = parse_srs_g1_wire(srs_blob)
    out = np.zeros((max(size + 3, 2051), 8), dtype=np.uint64)
    out[:2051] = g1[:2051]
    pad = {4096: 2051, 8192: 2054, 16384: 2057}.get(size)
    if pad is not None:
        out[size:size + 3] = g1[pad:pad + 3]
    return out


class KZGCommitmentSchemeBN254:
    """KZG over BN254 with the G1 powers resident on the GPU."""

    def __init__(self, g1_wire: np.ndarray):
        self.public_parameter_group_1 = np.ascontiguousarray(g1_wire, dtype=np.uint64).reshape(-1, 8)
        self._srs = B.Srs.from_host(self.public_parameter_group_1)

    @classmethod
    def from_unchecked_bytes(cls, data: bytes) -> "KZGCommitmentSchemeBN254":
        """Reference SRS blob: u32 len_g1 | u32 len_g2 | len_g1 x (x LE32 || y LE32, flags in the
        top two bits of the last byte) | G2 points (kzg_poly_commitment.rs:228-264).  The G2 part is
        only used by the verifier and is ignored here."""
        return cls(parse_srs_g1_wire(data))

    def max_degree(self) -> int:
        return self.public_parameter_group_1.shape[0] - 1

    def commit(self, polynomial: FpPolynomial) -> np.ndarray:
        """C = sum_i coef_i * SRS_i (kzg_poly_commitment.rs:278-293).  DegreeError when
        degree + 1 > SRS length; an all-zero polynomial commits to infinity."""
        degree = polynomial.degree()
        if degree + 1 > self.public_parameter_group_1.shape[0]:
            raise UzkgeError(N.UZK_ERR_DEGREE, "polynomial degree exceeds the SRS")
        return B.msm(self._srs, polynomial.get_coefs_ref()[: degree + 1])

    def apply_blind_factors(self, commitment: np.ndarray, blinds: np.ndarray, zeroing_degree: int) -> np.ndarray:
        """C += sum_i b_i * (SRS[i] - SRS[zeroing_degree + i]) (kzg_poly_commitment.rs:299-313)."""
        b = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(-1, 4)
        if b.shape[0] == 0:
            return commitment
        plus = B.msm(self._srs, b, offset=0)
        minus = B.msm(self._srs, fr_neg(b), offset=zeroing_degree)
        return B.g1_fold(np.stack([np.asarray(commitment, dtype=np.uint64).reshape(12), plus, minus]))

    def release(self) -> None:
        self._srs.release()


# ---------------------------------------------------------------------------------------------------
# The callers of the two primitives inside the prover: the Lagrange-basis commit path.
#   prover commit closure        uzkge/src/plonk/prover.rs:125-149
#   PolyComScheme::batch_prove   uzkge/src/poly_commit/pcs.rs:107-168
#   split_t_and_commit           uzkge/src/plonk/helpers.rs:1323-1408
# Fiat-Shamir (Keccak transcript) and the prover's RNG are out of scope: the
# challenge `alpha` and the random blinds are arguments here, where the Rust draws them.
# ---------------------------------------------------------------------------------------------------
def fr_add_rows(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a + b on wire-format rows (Montgomery form is linear)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(a)
    for i in range(a.shape[0]):
        v = (_wire_int(a[i]) + _wire_int(b[i])) % FR_MODULUS
        out[i] = [(v >> (64 * k)) & _MASK64 for k in range(4)]
    return out


def max_power_of_2(degree: int) -> int:
    """The reference's loop `for i in (0..=degree).rev() { if (i & (i - 1)) == 0 {..break} }`
    (pcs.rs:139-145, helpers.rs:1367-1373): the largest power of two <= degree.  degree = 0 (a constant quotient) leaves
    the loop with 0 in a release build (0 & usize::MAX == 0): the fold then has no domain and the caller's `fft(.., 0)`
    fails -- PCSProveEvalError / FFTError in the reference, UzkgeError(FFT) here -- instead of an assertion."""
    if degree <= 0:
        return 0
    return 1 << (degree.bit_length() - 1)


def commit_folded_lagrange(pcs: KZGCommitmentSchemeBN254, lagrange_pcs: KZGCommitmentSchemeBN254, coefs: np.ndarray,
                           degree: int) -> np.ndarray:
    """The tail shared by batch_prove (pcs.rs:137-166, `degree` = q.degree()) and split_t_and_commit
    (helpers.rs:1366-1394, `degree` = coefs.len()): fold the coefficients from max_power_of_2 on back onto
    the low ones, fft(N), commit the evaluations over the Lagrange SRS, undo the fold with blind factors."""
    c = np.ascontiguousarray(coefs, dtype=np.uint64).reshape(-1, 4)
    npow = max_power_of_2(degree)
    if npow == 0:
        raise UzkgeError(N.UZK_ERR_FFT, "no evaluation domain for a degree-0 polynomial (max_power_of_2 = 0)")
    blinds = fr_neg(c[npow:])
    new_coefs = c[:npow].copy()
    if blinds.shape[0]:
        new_coefs[: blinds.shape[0]] = fr_add_rows(new_coefs[: blinds.shape[0]], c[npow:])    # coefs[i] - blinds[i]
    sub_q = FpPolynomial.from_coefs(new_coefs)
    q_eval = sub_q.fft(npow)
    if q_eval is None:
        raise UzkgeError(N.UZK_ERR_FFT, "no evaluation domain for the folded polynomial")
    cm = lagrange_pcs.commit(FpPolynomial.from_coefs(q_eval))
    return pcs.apply_blind_factors(cm, blinds, npow)


def batch_prove(pcs: KZGCommitmentSchemeBN254, lagrange_pcs: Optional[KZGCommitmentSchemeBN254],
                polys: Sequence[FpPolynomial], point: np.ndarray, alpha: np.ndarray):
    """PolyComScheme::batch_prove's default body (pcs.rs:107-168) with `alpha` given instead of drawn from the
    transcript.  Returns (commitment of q, evaluations p_k(point) [len(polys), 4]).  The evaluations, the linear
    combination and the division by X - point run on the device (uzk_open_quotient)."""
    assert len(polys) > 0
    n = max(p.coefs.shape[0] for p in polys)
    stack = np.zeros((len(polys), n, 4), dtype=np.uint64)
    for k, p in enumerate(polys):
        stack[k, : p.coefs.shape[0]] = p.coefs
    q_rows, evals = B.open_quotient(stack, point, alpha)
    q = FpPolynomial.from_coefs(q_rows)
    if lagrange_pcs is not None:
        return commit_folded_lagrange(pcs, lagrange_pcs, q.coefs, q.degree()), evals
    return pcs.commit(q), evals


def split_t_and_commit(pcs: KZGCommitmentSchemeBN254, lagrange_pcs: Optional[KZGCommitmentSchemeBN254], t: FpPolynomial,
                       n_wires_per_gate: int, n: int, rands: np.ndarray):
    """helpers.rs:1323-1408 with the prover's random blinds `rands` [n_wires_per_gate, 4] given: chunk i of t gets
    + rand_i * X^n and - rand_(i-1); every chunk is committed (Lagrange path: fold, fft, commit, blinds).
    Returns (commitments [n_wires_per_gate, 12], chunk polynomials)."""
    coefs_all = t.get_coefs_ref()
    coefs_len = coefs_all.shape[0]
    rands = np.ascontiguousarray(rands, dtype=np.uint64).reshape(n_wires_per_gate, 4)
    prev = np.zeros((1, 4), dtype=np.uint64)
    cms, polys = [], []
    for i in range(n_wires_per_gate):
        start = i * n
        end = coefs_len if i == n_wires_per_gate - 1 else (i + 1) * n
        coefs = coefs_all[start:min(coefs_len, end)].copy() if start < coefs_len else np.zeros((0, 4), np.uint64)
        if i != n_wires_per_gate - 1:
            padded = np.zeros((n + 1, 4), dtype=np.uint64)
            padded[: coefs.shape[0]] = coefs
            coefs = padded
            coefs[n] = fr_add_rows(coefs[n:n + 1], rands[i:i + 1])[0]
            coefs[0] = fr_add_rows(coefs[0:1], fr_neg(prev))[0]
        elif coefs.shape[0] == 0:
            coefs = fr_neg(prev)
        else:
            coefs[0] = fr_add_rows(coefs[0:1], fr_neg(prev))[0]
        prev = rands[i:i + 1]
        if lagrange_pcs is not None:
            cm = commit_folded_lagrange(pcs, lagrange_pcs, coefs, coefs.shape[0])
        else:
            cm = pcs.commit(FpPolynomial.from_coefs(coefs))
        cms.append(cm)
        polys.append(FpPolynomial.from_coefs(coefs))
    return np.stack(cms), polys


class ProverCommit:
    """The commit closure of prover_with_lagrange (prover.rs:125-149; twin in indexer.rs:284-299): the Lagrange
    SRS is used iff it has exactly n_constraints bases; then C = MSM(lagrange, evals) + blind factors, else the
    monomial commit of the coefficient polynomial."""

    def __init__(self, pcs: KZGCommitmentSchemeBN254, lagrange_pcs: Optional[KZGCommitmentSchemeBN254], n_constraints: int):
        self.pcs = pcs
        self.n_constraints = n_constraints
        self.lagrange_pcs = lagrange_pcs if (lagrange_pcs is not None and lagrange_pcs.max_degree() + 1 == n_constraints) else None

    def __call__(self, evals: np.ndarray, coef_polynomial: FpPolynomial, blinds: np.ndarray) -> np.ndarray:
        if self.lagrange_pcs is not None:
            cm = self.lagrange_pcs.commit(FpPolynomial.from_coefs(evals))
            return self.pcs.apply_blind_factors(cm, blinds, self.n_constraints)
        return self.pcs.commit(coef_polynomial)


def preprocess_tables(commit: ProverCommit, evals: np.ndarray, m: int, k1: np.ndarray, keep_on_device: bool = False):
    """The per-table loop of the preprocessing -- indexer_with_lagrange (uzkge/src/plonk/indexer.rs:316-470: the
    permutation, selector, boolean, Anemoi, ecc and shuffle tables) and refresh_prover_params_public_key
    (shuffle/src/gen_params/params.rs:88-121) -- for a stack `evals` [k, n, 4] of evaluation tables:
        coefs_i      = FpPolynomial::ifft_with_domain(domain_n, evals_i)
        coset_evals_i = coefs_i.coset_fft_with_domain(domain_m, k[1])
        cm_i         = commit(evals_i, coefs_i)        (the closure without blinds, indexer.rs:284-299)
    All k tables go through ONE batched inverse transform, ONE batched coset transform on the m-domain and ONE batched
    MSM; the tables cross PCIe once.  Returns (coefs [k, n, 4], coset_evals [k, m, 4], cms [k, 12]); with
    `keep_on_device` the first two are int64 CUDA tensors (the quotient kernel's resident inputs, uzk_t_quotient_device)
    instead of host arrays."""
    import torch
    ev_h = np.ascontiguousarray(evals, dtype=np.uint64)
    assert ev_h.ndim == 3 and ev_h.shape[2] == 4
    k, n = ev_h.shape[0], ev_h.shape[1]
    if n != commit.n_constraints or m % n != 0:
        raise UzkgeError(N.UZK_ERR_PARAMETER, "tables must have n_constraints rows and the quotient domain must be a multiple of n")
    if not (B.domain_supported(n) and B.domain_supported(m)):
        raise UzkgeError(N.UZK_ERR_FFT, "no evaluation domain of that size")
    ev = torch.from_numpy(ev_h.view(np.int64)).cuda()
    coefs = torch.empty_like(ev)
    coset = torch.zeros((k, m, 4), dtype=torch.int64, device=ev.device)
    torch.cuda.synchronize()                       # the library runs on its own stream
    B.ntt_batch_device(ev.data_ptr(), coefs.data_ptr(), n, k, inverse=True, sync=True)
    coset[:, :n] = coefs
    torch.cuda.synchronize()
    B.ntt_batch_device(coset.data_ptr(), coset.data_ptr(), m, k, coset_shift=k1)
    if commit.lagrange_pcs is not None:
        cms = B.msm_batch_device(commit.lagrange_pcs._srs, ev.data_ptr(), n, k)
    else:
        if n > commit.pcs.public_parameter_group_1.shape[0]:
            raise UzkgeError(N.UZK_ERR_DEGREE, "polynomial degree exceeds the SRS")
        cms = B.msm_batch_device(commit.pcs._srs, coefs.data_ptr(), n, k)
    B.sync()
    if keep_on_device:
        return coefs, coset, cms
    return (coefs.cpu().numpy().view(np.uint64), coset.cpu().numpy().view(np.uint64), cms)


def hide_polynomial(polynomial: FpPolynomial, blinds: np.ndarray, zeroing_degree: int) -> FpPolynomial:
    """helpers.rs:139-158 with the random blinds given: adds (b_0 + b_1 X + ...) * (X^zeroing_degree - 1)."""
    b = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(-1, 4)
    size = max(polynomial.coefs.shape[0], zeroing_degree + b.shape[0])
    c = np.zeros((size, 4), dtype=np.uint64)
    c[: polynomial.coefs.shape[0]] = polynomial.coefs
    for i in range(b.shape[0]):
        c[i] = fr_add_rows(c[i:i + 1], b[i:i + 1])[0]
        c[zeroing_degree + i] = fr_add_rows(c[zeroing_degree + i:zeroing_degree + i + 1], fr_neg(b[i:i + 1]))[0]
    return FpPolynomial.from_coefs(c)


def load_srs_params(srs_blob: bytes, size: int) -> KZGCommitmentSchemeBN254:
    """uzkge/src/gen_params/mod.rs:151-183: the monomial SRS of a size-`size` circuit from the embedded blob --
    powers 0..2050, the identity up to `size`, then the three padding powers size, size+1, size+2."""
    return KZGCommitmentSchemeBN254(srs_params_wire(srs_blob, size))
