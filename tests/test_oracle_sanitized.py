"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile `asan`): the checker the GPU parity tests
lean on is itself run through a memory-error detector -- field arithmetic, the Pippenger MSM with its thread pool against the
naive sum, NTT round trips at power-of-two and 3 * 2^k sizes with and without threads, Horner, z_poly and the opening quotient,
plus a Lagrange identity on one of the reference's parameter files.  GPU sanitizers are not available on this pool; this is the
CPU build's turn.  Runs in a child process (the sanitizer runtime has to be loaded before the interpreter's first allocation)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, os
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import bn254_py as opy
import oracle_c as oc
from util import load_srs, rand_fr_wire
assert "asan" in oc._LIB
wire, pts = load_srs("lagrange-srs-4096.bin")
# field and group arithmetic against the pure-Python oracle
a, b = rand_fr_wire(2, 5)
ai, bi = (int.from_bytes(x.tobytes(), "little") * pow(1 << 256, -1, opy.R) % opy.R for x in (a, b))
assert [int.from_bytes(oc.fr_mul(a, b).tobytes(), "little") * pow(1 << 256, -1, opy.R) % opy.R] == [ai * bi % opy.R]
assert np.array_equal(oc.fr_mul(oc.fr_inv(a), a).reshape(4), oc.fr_from_ints([1])[0])
# MSM: Pippenger (1 and 4 threads, several window widths) against the naive sum; zero scalars; the Lagrange identity sum L_i = G
for n, c in ((1, 0), (33, 4), (257, 7), (1000, 0)):
    s = rand_fr_wire(n, 7 + n); s[::5] = 0
    want = oc.jac_to_affine_ints(oc.msm_naive(wire[:n], s))
    for th in (1, 4):
        assert oc.jac_to_affine_ints(oc.msm_pippenger(wire[:n], s, c, th)) == want, (n, c, th)
assert oc.jac_to_affine_ints(oc.msm_pippenger(wire, oc.fr_from_ints([1] * 4096), 0, 4)) == opy.G1_GEN
# NTT: round trips, threads, a 3 * 2^k size, the coset form
for n in (1, 2, 16, 1024, 3 * 256, 4096):
    x = rand_fr_wire(n, 100 + n)
    for th in (1, 4):
        f = oc.ntt(x, threads=th)
        assert np.array_equal(oc.ntt(f, inverse=True, threads=th), x), (n, th)
    assert np.array_equal(oc.ntt(x, threads=1), oc.ntt(x, threads=4))
x = rand_fr_wire(64, 3); k = rand_fr_wire(1, 4)[0]
assert np.array_equal(oc.poly_eval(x, k).reshape(4), oc.poly_eval(np.ascontiguousarray(x), k).reshape(4))
q, ev, ok = oc.open_quotient(rand_fr_wire(3 * 50, 9).reshape(3, 50, 4), a, b)
assert ok and q.shape[0] >= 49
print("sanitized oracle ok")
"""


def test_oracle_checks_pass_under_address_and_ub_sanitizers(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    lib = os.path.join(ROOT, "oracle", "liboracle_bn254_asan.so")
    assert os.path.exists(lib)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan next to gcc")
    env = dict(os.environ, UZK_ORACLE_LIB=lib, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
