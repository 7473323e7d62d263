#!/usr/bin/env python3
"""Generates tests/golden/vectors_v1.npz from the oracle (oracle/bn254_py.py big-int definitions and
oracle/bn254_oracle.c), the set SURVEY.md 8(c) lists: field KATs, G1 additions incl. edge cases, MSM at
n in {1, 2, 33, 1024, 4096} over the first points of the reference's lagrange-srs-4096.bin with
uniform / zero / one / r-1 / boolean-heavy scalars, NTT / inverse / coset at n in {1, 2, 4, 16, 48,
1024, 2^14}.  Inputs and expected outputs, wire format (Montgomery, 4 x u64 LE); every expected value
is produced by the big-int code where that is fast enough and cross-checked against the C oracle.
Run from the repo root:  python tests/golden/make_vectors.py"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bn254_py as opy
import oracle_c as oc
from util import rand_fr_wire, load_srs


def aff_wire(pts):
    return oc.points_from_affine(pts)


def main():
    out = {}
    # ---- field KATs (Fr and Fq): a*b, a+b, a-b on edge values and random ones, big-int definitions
    for name, mod in (("fr", opy.R), ("fq", opy.P)):
        rng = np.random.default_rng(11 if name == "fr" else 12)
        edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << 253) % mod, (1 << 128) - 1]
        a = edge * len(edge) + [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(200)]
        b = [y for y in edge for _ in edge] + [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(200)]
        out[f"{name}_a"] = oc.fr_from_ints(a, mod); out[f"{name}_b"] = oc.fr_from_ints(b, mod)
        out[f"{name}_mul"] = oc.fr_from_ints([x * y % mod for x, y in zip(a, b)], mod)
        out[f"{name}_add"] = oc.fr_from_ints([(x + y) % mod for x, y in zip(a, b)], mod)
        out[f"{name}_sub"] = oc.fr_from_ints([(x - y) % mod for x, y in zip(a, b)], mod)
    for i in range(out["fr_a"].shape[0]):      # the C oracle agrees with the big-int definitions
        assert np.array_equal(oc.fr_mul(out["fr_a"][i], out["fr_b"][i]), out["fr_mul"][i])
        assert np.array_equal(oc.fq_mul(out["fq_a"][i], out["fq_b"][i]), out["fq_mul"][i])
    # ---- G1: a + b for generic pairs, P + P, P + (-P), P + inf, inf + P, inf + inf (affine in, affine out)
    G = opy.G1_GEN
    ks = [1, 2, 3, 5, 7, 11, 0xDEADBEEF, opy.R - 1, opy.R - 2, 12345678901234567890]
    P = [opy.g1_mul(G, k) for k in ks]
    A = P + [P[3], P[4], P[5], None, None] + [P[0]]
    B = P[1:] + P[:1] + [P[3], opy.g1_neg(P[4]), None, P[6], None] + [opy.g1_mul(G, opy.R - 1)]
    out["g1_a"] = aff_wire(A); out["g1_b"] = aff_wire(B)
    out["g1_sum"] = aff_wire([opy.g1_add(x, y) for x, y in zip(A, B)])
    # ---- MSM over the reference's Lagrange SRS points
    srs_wire, srs_pts = load_srs("lagrange-srs-4096.bin")
    one = oc.fr_from_ints([1])[0]; zero = np.zeros(4, dtype=np.uint64); rm1 = oc.fr_from_ints([opy.R - 1])[0]
    rng = np.random.default_rng(7)
    for n in (1, 2, 33, 1024, 4096):
        sets = {"uniform": rand_fr_wire(n, 100 + n), "zero": np.tile(zero, (n, 1)), "one": np.tile(one, (n, 1)),
                "rminus1": np.tile(rm1, (n, 1))}
        mix = rand_fr_wire(n, 200 + n); sel = rng.integers(0, 10, n)
        mix[sel < 5] = zero; mix[(sel >= 5) & (sel < 8)] = one
        sets["boolean_heavy"] = mix
        for kind, sc in sets.items():
            ref = oc.msm_pippenger(srs_wire[:n], sc, 0, 8)
            if n <= 33:     # big-int definition for the small ones
                want = opy.msm_naive(srs_pts[:n], oc.fr_to_ints(sc))
                assert oc.jac_to_affine_ints(ref) == want
            out[f"msm_{n}_{kind}_scalars"] = sc
            out[f"msm_{n}_{kind}_affine"] = aff_wire([oc.jac_to_affine_ints(ref)])
    # ---- NTT: forward, inverse, coset forward (shift 7) -- natural order, omega = 5^((r-1)/n)
    for n in (1, 2, 4, 16, 48, 1024, 1 << 14):
        x = rand_fr_wire(n, 300 + n)
        fwd = oc.ntt(x)
        if n <= 48:
            assert oc.fr_to_ints(fwd) == opy.dft_naive(oc.fr_to_ints(x), n)
        out[f"ntt_{n}_in"] = x; out[f"ntt_{n}_fwd"] = fwd; out[f"ntt_{n}_inv"] = oc.ntt(x, inverse=True)
        k = oc.fr_from_ints([7])[0]
        out[f"ntt_{n}_coset7"] = oc.ntt(oc.mul_var(x, k))
    np.savez_compressed(os.path.join(HERE, "vectors_v1.npz"), **out)
    print(f"wrote {len(out)} arrays")


if __name__ == "__main__":
    main()
