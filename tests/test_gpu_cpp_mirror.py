"""Builds and runs the C++ restatement of the reference's boundary tests (tests/cpp) against the
header-only host mirror include/uzkge_poly_commit.hpp.  The binary links the product library and,
for the naive side of the checks, the CPU oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    out = os.path.join(ROOT, "tests", "cpp", "test_poly_commit")
    src = os.path.join(ROOT, "tests", "cpp", "test_poly_commit.cpp")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call([
        "g++", "-O2", "-std=c++17", "-o", out, src,
        "-L" + os.path.join(ROOT, "uzkge_amd"), "-luzkge_gpu",
        "-L" + os.path.join(ROOT, "oracle"), "-loracle_bn254",
        "-Wl,-rpath," + os.path.join(ROOT, "uzkge_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
        "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
    ])
    return out


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_mirror_reference_tests(gpu):
    exe = _build()
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "srs-padding.bin"),
                        os.path.join(ROOT, "tests", "golden", "lagrange-srs-4096.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def _build_rounds():
    """tests/cpp/prover_rounds.cpp: one proof's hot-path sequence issued from C++ through the C ABI ALONE -- compiled by plain
    g++ against include/uzkge_gpu.h, linked against libuzkge_gpu.so and nothing else (the command lives in __graft_entry__)."""
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import rounds_build_command
    cmd = rounds_build_command()
    subprocess.check_call(cmd)
    return cmd[cmd.index("-o") + 1]


def test_cpp_prover_rounds_compiles_and_links():
    assert os.path.exists(_build_rounds())


def test_cpp_prover_rounds_needs_only_the_c_abi():
    """The device-resident flow is reachable from a host language through include/uzkge_gpu.h alone: the driver's source
    names no HIP header or symbol, its compile line names no HIP compiler or library, and the binary's own NEEDED list holds
    libuzkge_gpu.so but not the HIP runtime (which only the library itself depends on)."""
    import re
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import rounds_build_command
    src = open(os.path.join(ROOT, "tests", "cpp", "prover_rounds.cpp")).read()
    code = re.sub(r"//.*", "", src)
    assert not re.search(r"\bhip[A-Z_]\w*|hip_runtime|<hip/", code), "prover_rounds.cpp still uses the HIP runtime"
    cmd = rounds_build_command()
    assert cmd[0] == "g++" and not any("hip" in a.lower() or "rocm" in a.lower() for a in cmd[1:] if not a.startswith(ROOT) and not a.startswith("-L" + ROOT) and not a.startswith("-Wl,-rpath," + ROOT))
    exe = _build_rounds()
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    libs = re.findall(r"\(NEEDED\).*\[(.*?)\]", needed)
    assert any("libuzkge_gpu" in l for l in libs) and not any("amdhip" in l or "hsa" in l for l in libs), libs


def _write_inputs(inp, d, shuffle=True, precompute=True):
    import numpy as np
    def put(name, a, dtype=np.uint64):
        np.ascontiguousarray(a, dtype=dtype).tofile(os.path.join(d, name + ".bin"))
    put("meta", [inp.n, int(shuffle), int(precompute)])
    put("bases", inp.bases)
    put("evals9", np.concatenate([inp.w_evals.reshape(-1, 4), inp.wsel_evals.reshape(-1, 4), inp.pi_evals]))
    put("perm", inp.perm, np.uint32)
    put("table_polys", inp.table_polys)
    put("k", inp.k)
    put("scalars", np.stack([inp.beta, inp.gamma, inp.alpha, inp.zeta, inp.alpha_open, inp.alpha_open2, inp.anemoi_g, inp.anemoi_g_inv,
                             inp.edwards_a, inp.k1_inv, inp.zeta_omega]))
    put("z_h_inv", inp.z_h_inv)
    put("blinds8", np.concatenate([inp.blinds_w, inp.blinds_wsel])); put("blinds_z", inp.blinds_z)
    put("t_rands", inp.t_rands); put("r_scalars", inp.r_scalars)


def _run_rounds_and_check(tmp_path, shuffle):
    import sys
    import numpy as np
    for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from prover_chain import N_TABLES, ChainInputs
    V3 = np.load(os.path.join(ROOT, "tests", "golden", "vectors_v3.npz"))
    n = int(V3["n"][0])
    inp = ChainInputs(n, int(V3["seed"][0]))
    _write_inputs(inp, str(tmp_path), shuffle=shuffle)
    exe = _build_rounds()
    # 3 timed chains, then 3 host threads (one context each, one shared SRS) running 3 chains each at the same time: every
    # thread must end with the single-threaded chain's commitments and evaluations
    r = subprocess.run([exe, str(tmp_path), "3", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout and '"threads_agree_with_single": true' in r.stdout, r.stdout + r.stderr
    rd = lambda name, shape: np.fromfile(os.path.join(str(tmp_path), "out_" + name + ".bin"), dtype=np.uint64).reshape(shape)
    m, cs = 6 * n, n + 8
    small = {key: rd(key, (cnt, 12)) for key, cnt in (("cm_w_wsel", 8), ("cm_z", 1), ("cm_t", 5), ("cm_q", 2))}
    small.update(evals=rd("evals", (-1, 4)), t_blinds=rd("t_blinds", (5, 3, 4)), q_blinds=rd("q_blinds", (2, 3, 4)))
    big = {"coefs": rd("coefs", (10, m, 4))[:, : n + 3], "coset_evals": rd("coset_evals", (10, m, 4)), "t_quotient": rd("t_quotient", (m, 4)),
           "t": rd("t", (m, 4)), "z_evals": rd("z_evals", (n, 4)), "r": rd("r", (n + 3, 4)), "chunks": rd("chunks", (5, cs, 4)),
           "quotients": rd("quotients", (2, cs, 4)), "tables": rd("tables", (-1, m, 4))}
    assert N_TABLES == 46
    return V3, small, big


@pytest.mark.gpu
def test_cpp_prover_rounds_match_frozen_outputs(gpu, tmp_path):
    """The C++ driver (plain g++, the C ABI only) runs ChainInputs(4096, 7) and must reproduce tests/golden/vectors_v3.npz
    (commitments over the reference's SRS files, evaluations, blinds, digests of the intermediates) -- the same fixture the
    Python chain is held to."""
    from test_gpu_golden import check_against_frozen
    V3, small, big = _run_rounds_and_check(tmp_path, shuffle=True)
    assert big["tables"].shape[0] == 46
    check_against_frozen(V3, small, big)


@pytest.mark.gpu
def test_cpp_prover_rounds_without_shuffle_terms(gpu, tmp_path):
    """The same driver for a circuit without the "shuffle" feature (zmatchmaking): a circuit of 21 slots (the 25 shuffle / ECC
    tables do not exist, the quotient kernel gets 28 NULL slots), 15 evaluations, 19 polynomials in r, 12 in the opening at
    zeta; rounds 1-2 are unchanged."""
    import numpy as np
    V3, small, big = _run_rounds_and_check(tmp_path, shuffle=False)
    assert big["tables"].shape[0] == 21 and small["evals"].shape[0] == 15
    assert np.array_equal(small["evals"][:15], V3["evals"][:15])             # the first 15 evaluations do not depend on the feature
    from test_gpu_golden import affine_of, oc
    for key in ("cm_w_wsel", "cm_z"):
        assert np.array_equal(oc.points_from_affine([affine_of(j) for j in small[key]]), V3[key]), key
