"""uzkge_amd/csrc/host_ec64.hpp -- the 4 x 64-bit host arithmetic of the MSM's final Horner step and of uzk_g1_fold --
against the portable host code of fp256.hpp / ec.hpp (tests/cpp/test_host_ec64.cpp, plain g++, no GPU), and uzk_g1_fold
itself against the oracle."""
import os
import subprocess

import numpy as np

import bn254_py as opy
import oracle_c as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_ec64_against_portable_host_code():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_ec64")
    src = os.path.join(ROOT, "tests", "cpp", "test_host_ec64.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "uzkge_amd", "csrc"), src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:]


def test_g1_fold_matches_oracle_sum():
    """uzk_g1_fold is host-only (no GPU needed): sum of Jacobian partials incl. infinity, a repeated and an opposite point."""
    from uzkge_amd import backend as b
    ks = [5, 7, 7, 11, 1 << 40]
    pts = [opy.g1_mul(opy.G1_GEN, k) for k in ks]
    neg = (pts[3][0], (opy.P - pts[3][1]) % opy.P)
    aff = oc.points_from_affine(pts + [neg])
    jac = np.zeros((len(aff) + 1, 12), dtype=np.uint64)
    one = oc.points_from_affine([opy.G1_GEN])[0][:4]           # Montgomery 1 = x(G1)
    for i, a in enumerate(aff):
        jac[i, :8] = a
        jac[i, 8:] = one
    jac[-1, :4] = one; jac[-1, 4:8] = one                   # infinity: z = 0
    got = oc.jac_to_affine_ints(b.g1_fold(jac))
    want = opy.g1_mul(opy.G1_GEN, 5 + 7 + 7 + (1 << 40))        # 11 G cancels against -11 G
    assert got == want
    assert oc.jac_to_affine_ints(b.g1_fold(jac[-1:])) is None
