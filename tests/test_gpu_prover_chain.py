"""A checked prover-round chain: every device-resident step one 52-card proof issues (tools/prover_chain.py, the
stand-in for BASELINE config #4 -- uzkge/src/plonk/prover.rs:88-394 cannot run here: no Rust toolchain), with each
commitment, evaluation vector and intermediate polynomial compared against the same chain on the CPU oracle.
The call mix is the reference's (which polynomials are transformed / committed / evaluated / combined / opened, in which
batches, at which lengths; split_t with chunk n + 2); circuit polynomials and witness are synthetic (random elements of the
real shapes: n = 2^14, 6n = 98304); the SRS files are the reference's (`lagrange-srs-{8192,16384}.bin`, `srs-padding.bin`),
so every commitment is an MSM over reference bases."""
import os
import sys

import numpy as np
import pytest

import bn254_py as opy
import oracle_c as oc
from util import affine_of

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _add_blinds(coefs, blinds, n):
    from chain_oracle import add_blinds
    return add_blinds(coefs, blinds, n)


def _check_chain_against(c, o, want):
    """c: ProverChain(keep_blinds=True) after run(), o: its outputs, want: chain_oracle.oracle_chain(inputs)."""
    snap = c.snapshot()
    assert np.array_equal(snap["tables"], want["tables"][: snap["tables"].shape[0]])     # setup: the circuit's coset tables (46, or 21 without the shuffle feature)
    assert np.array_equal(snap["coefs"], want["coefs"]) and not snap["coefs_beyond"].any()       # hidden coefficient polynomials
    for key in ("cm_w_wsel", "cm_z", "cm_t", "cm_q"):
        got = oc.points_from_affine([affine_of(j) for j in o[key]])
        assert np.array_equal(got, want[key]), key
    for key in ("z_evals", "coset_evals", "t_quotient", "t", "chunks", "r", "quotients"):
        assert np.array_equal(snap[key], want[key]), key
    for key in ("t_blinds", "q_blinds", "evals"):
        assert np.array_equal(o[key], want[key]), key


def test_prover_round_chain_matches_oracle_chain(gpu):
    from chain_oracle import oracle_chain
    from prover_chain import ChainInputs, ProverChain
    inp = ChainInputs(1 << 14, 11)
    c = ProverChain(inputs=inp, keep_blinds=True)
    try:
        _check_chain_against(c, c.run(), oracle_chain(inp))
        # the general pipeline (no window table) commits to the same points
        c2 = ProverChain(inputs=inp, precompute=False)
        try:
            o2 = c2.run()
            for key in ("cm_w_wsel", "cm_z", "cm_t", "cm_q"):
                assert [affine_of(j) for j in o2[key]] == [affine_of(j) for j in c.out[key]], key
        finally:
            c2.release()
    finally:
        c.release()


@pytest.mark.parametrize("n", [4096, 8192])
def test_quotient_without_shuffle_vectors(gpu, n):
    """A circuit without the "shuffle" feature (zmatchmaking, helpers.rs:437 #[cfg(feature = "shuffle")]; its circuit has
    n = 8192 constraints, matchmaking/src/build_cs.rs:68-99): the 28 vectors of terms 12..18 are passed as NULL; the result
    equals the full formula with those vectors zero."""
    from prover_chain import ChainInputs, ProverChain
    inp = ChainInputs(n, 5)
    c = ProverChain(inputs=inp, shuffle=False, precompute=False)
    try:
        c.run()
        snap = c.snapshot()
        assert sum(1 for p in c.tq_ptrs if not p) == 28
        assert snap["tables"].shape[0] == 21                                   # a circuit without the feature has no shuffle / ECC tables
        vecs = np.concatenate([snap["coset_evals"], snap["tables"], np.zeros((25,) + snap["tables"].shape[1:], dtype=np.uint64)])
        for slot in list(range(5, 8)) + list(range(31, 56)):
            vecs[slot] = 0
        want = oc.t_quotient(n, 6, vecs, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv)
        assert np.array_equal(snap["t_quotient"], want)
        # a partial set of shuffle vectors is an argument error, not a silent zero
        from uzkge_amd import UzkgeError
        bad = list(c.tq_ptrs)
        bad[5] = c.d_coset.ptr
        with pytest.raises(UzkgeError):
            gpu.t_quotient_device(n, 6, bad, c.alpha, c.beta, c.gamma, c.k, c.anemoi_g, c.anemoi_g_inv, c.edwards_a, c.z_h_inv,
                                  c.d_tq.ptr)
    finally:
        c.release()


def test_zmatchmaking_sized_chain_without_shuffle_terms(gpu):
    """The whole checked chain at zmatchmaking's size (n = 8192, commits over the reference's lagrange-srs-8192.bin and the
    padding powers at index 2054 of srs-padding.bin) with the shuffle-feature terms absent."""
    from chain_oracle import oracle_chain
    from prover_chain import ChainInputs, ProverChain
    inp = ChainInputs(8192, 23)
    c = ProverChain(inputs=inp, shuffle=False, keep_blinds=True)
    try:
        _check_chain_against(c, c.run(), oracle_chain(inp, shuffle=False))
    finally:
        c.release()


def test_lincomb_and_hide_primitives(gpu):
    import torch
    _host = lambda t: t.cpu().numpy().view(np.uint64)
    n = 1000
    from util import rand_fr_wire
    polys = [rand_fr_wire(ln, 40 + i) for i, ln in enumerate((n, n - 7, 3, n + 5))]
    scal = rand_fr_wire(4, 50)
    d = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
    out = torch.empty((n + 5, 4), dtype=torch.int64, device="cuda")
    gpu.poly_lincomb_device([t.data_ptr() for t in d], [p.shape[0] for p in polys], scal, out.data_ptr(), n + 5)
    gpu.sync()                                            # asynchronous on the library stream
    want = [0] * (n + 5)
    si = oc.fr_to_ints(scal)
    for s_k, p in zip(si, polys):
        for j, v in enumerate(oc.fr_to_ints(p)):
            want[j] = (want[j] + s_k * v) % opy.R
    assert oc.fr_to_ints(_host(out)) == want
    co = rand_fr_wire(n + 3, 60); co[n:] = 0
    bl = rand_fr_wire(3, 61)
    dco = torch.from_numpy(co.view(np.int64)).cuda()
    gpu.hide_polynomial_device(dco.data_ptr(), n + 3, bl, n)
    gpu.sync()
    assert oc.fr_to_ints(_host(dco)) == _add_blinds(oc.fr_to_ints(co[:n]), oc.fr_to_ints(bl), n)
