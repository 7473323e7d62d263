"""The Rust side of the drop-in exists as files (rust/): a -sys crate, the arkworks glue module and the call-site
patch.  No Rust toolchain here, so what can drift is checked textually: the generated bindings against the header,
the functions the glue calls against the bindings, and the patch against the reference's files when present."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust")


def _ffi():
    return open(os.path.join(RUST, "uzkge-gpu-sys", "src", "ffi.rs")).read()


def test_bindings_are_regenerated_from_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"])
    assert r.returncode == 0, "rust/uzkge-gpu-sys/src/ffi.rs is stale: run tools/gen_rust_bindings.py"


def test_every_header_symbol_is_bound_with_the_same_arity():
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "uzkge_gpu.h")).read(), flags=re.S)
    ffi = _ffi()
    decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(uzk_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr)}
    bound = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (uzk_[a-z0-9_]+)\((.*?)\)", ffi)}
    assert set(decl) == set(bound) and len(decl) >= 40
    for name, args in decl.items():
        n_c = 0 if args.strip() in ("", "void") else args.count(",") + 1
        n_r = 0 if not bound[name].strip() else bound[name].count(",") + 1
        assert n_c == n_r, name
    from uzkge_amd import _native as N
    assert set(bound) == set(N.PROTOTYPES) and not any(b.startswith("uzk_test_") for b in bound)     # test hooks are not bound


def test_struct_layouts_match_the_header():
    ffi = _ffi()
    assert "pub struct uzk_g1_affine {\n    pub x: [u64; 4],\n    pub y: [u64; 4],\n}" in ffi
    assert "pub struct uzk_g1_jac {\n    pub x: [u64; 4],\n    pub y: [u64; 4],\n    pub z: [u64; 4],\n}" in ffi
    assert "pub vec: [*const c_void; UZK_TQ_NVEC]" in ffi and "pub const UZK_TQ_NVEC: usize = 56;" in ffi
    import ctypes
    from uzkge_amd._native import QuotientArgs
    assert ctypes.sizeof(QuotientArgs) == 8 + 56 * 8 + (3 + 5 + 3 + 16) * 32       # the same block in the ctypes mirror


def _calls(src, name_re):
    """[(name, n_args)] for every call `name(...)` in src whose name matches name_re; arguments counted at nesting depth 0."""
    out = []
    for m in re.finditer(r"\b(" + name_re + r")\s*\(", src):
        i, depth, args, cur = m.end(), 1, 0, ""
        while i < len(src) and depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                args += 1 if cur.strip() else 0
                cur = ""
                i += 1
                continue
            if depth:
                cur += ch
            i += 1
        out.append((m.group(1), args + (1 if cur.strip() else 0)))
    return out


def _strip_rust_comments(src):
    return re.sub(r"//.*", "", src)


def test_glue_and_wrappers_only_call_bound_functions():
    ffi = _ffi()
    decl = {m.group(1): (0 if not m.group(2).strip() else m.group(2).count(",") + 1) for m in re.finditer(r"pub fn (uzk_[a-z0-9_]+)\((.*?)\)", ffi)}
    consts = set(re.findall(r"pub const (UZK_[A-Z0-9_]+)", ffi))
    lib = _strip_rust_comments(open(os.path.join(RUST, "uzkge-gpu-sys", "src", "lib.rs")).read())
    used = _calls(lib, r"uzk_[a-z0-9_]+")
    assert len(used) >= 20
    for name, n_args in used:                                   # every FFI call of the wrappers: bound, and with that arity
        assert name in decl, name
        assert decl[name] == n_args, (name, decl[name], n_args)
    for c in re.findall(r"\b(UZK_[A-Z0-9_]+)\b", lib):
        assert c in consts, c
    wrappers = {}
    for m in re.finditer(r"pub fn ([a-z_0-9]+)\(", lib):          # parameter lists hold tuple types: count at depth 0
        name, n_params = _calls(lib[m.start():], r"fn " + m.group(1))[0]
        has_self = re.match(r"\s*&?(mut )?self\b", lib[m.end():]) is not None
        wrappers[m.group(1)] = n_params - (1 if has_self else 0)
    types = set(re.findall(r"pub (?:struct|enum|type) ([A-Za-z]+)", lib)) | set(re.findall(r"pub struct ([a-z_0-9]+)", ffi))
    for fname in ("gpu.rs", "gpu_prover.rs"):
        glue = _strip_rust_comments(open(os.path.join(RUST, "uzkge-glue", fname)).read())
        for name, n_args in _calls(glue, r"sys::[a-z_0-9]+"):   # free functions of the -sys crate: exist, same arity
            short = name.split("::")[1]
            if short.startswith("uzk_"):
                assert decl.get(short) == n_args, (fname, name, n_args)
            else:
                assert wrappers.get(short) == n_args, (fname, name, n_args, wrappers.get(short))
        for t, meth in re.findall(r"sys::([A-Z][A-Za-z]+)::([a-z_0-9]+)", glue):
            assert t in types and (re.search(rf"pub fn {meth}\(", lib) or meth[0].isupper() or t == "Error"), (fname, t, meth)
        for c in re.findall(r"sys::(UZK_[A-Z0-9_]+)", glue):
            assert c in consts, (fname, c)
        for meth in re.findall(r"\b(?:srs|bases)\.([a-z_]+)\(", glue):
            assert meth == "clone" or re.search(rf"pub fn {meth}\(", lib), (fname, meth)
    # the generator check: group_gen is compared with the library's, once per domain size, and a mismatch falls back
    glue = open(os.path.join(RUST, "uzkge-glue", "gpu.rs")).read()
    assert "same_generator" in glue and "domain_group_gen" in glue
    # ADVICE r2 / VERDICT r2: no panic on a device error, no registry lock across a device call, eviction, size-aware table
    assert "panic!" not in _strip_rust_comments(glue) and "Arc<sys::Srs>" in glue and "SRS_CACHE_CAP" in glue and "PRECOMPUTE_MAX_LEN" in glue
    assert "release_srs" in glue


def _round_sequence(src, pattern):
    return [m.group(1) for m in re.finditer(pattern, src)]


def test_rust_prover_drives_the_library_rounds_like_the_cpp_driver():
    """One implementation of the five rounds -- the library's.  rust/uzkge-glue/gpu_prover.rs and tests/cpp/prover_rounds.cpp (the
    driver the GPU tests hold to frozen outputs and to the reference's verifier) both issue uzk_prove_round1..5, in order, once
    per proof, and nothing else between them that touches the device; the Rust wrappers they go through call exactly those
    entry points."""
    cpp = re.sub(r"//.*", "", open(os.path.join(ROOT, "tests", "cpp", "prover_rounds.cpp")).read())
    body = cpp[cpp.index("void chain(int source"):cpp.index("std::vector<uint64_t> digest()")]
    assert _round_sequence(body, r"\buzk_prove_round(\d)\(") == list("12345")
    lib = open(os.path.join(RUST, "uzkge-gpu-sys", "src", "lib.rs")).read()
    for k in "12345":                                           # wrapper roundK -> uzk_prove_roundK and nothing else
        m = re.search(rf"pub fn round{k}\(.*?\n    }}\n", lib, flags=re.S)
        assert m and re.findall(r"\b(uzk_[a-z0-9_]+)\(", m.group(0)) == [f"uzk_prove_round{k}"], k
    glue = _strip_rust_comments(open(os.path.join(RUST, "uzkge-glue", "gpu_prover.rs")).read())
    rounds = glue[glue.index("fn rounds<"):]
    assert _round_sequence(rounds, r"prover\.round(\d)\(") == list("12345")
    # no device call of the old call-by-call flow is left in the glue: the only sys:: items it names are the three handles' types,
    # their constructors and the slot constants
    assert set(re.findall(r"sys::([A-Za-z_0-9]+)", glue)) <= {"Circuit", "Prover", "Context", "Error", "uzk_g1_affine", "uzk_g1_jac", "uzk_circuit_desc",
                                                               "UZK_CIRCUIT_SLOTS", "UZK_CS_Q", "UZK_CS_S", "UZK_CS_L1", "UZK_CS_QB", "UZK_CS_QPRK",
                                                               "UZK_CS_QPK", "UZK_CS_QG", "UZK_CS_QECC", "domain_group_gen"}
    assert len(glue.splitlines()) < 520                        # marshalling for any number of lanes, not orchestration (was 786 lines of it)


def test_every_prover_thread_gets_a_context_and_a_shared_prover():
    """the reference proves one proof per call from application threads (prover.rs:88-100, shuffle/src/sdk.rs:196-214).
    The glue gives each such thread its own context (made current before anything else touches the device) and a prover of ONE
    proof -- the kind the library shares between threads that prove at the same time (uzk_coalesce_config) -- and a host that holds
    several witnesses can pass them together (prove_batch: one prng and one transcript per proof)."""
    glue = _strip_rust_comments(open(os.path.join(RUST, "uzkge-glue", "gpu_prover.rs")).read())
    assert "static CONTEXT: RefCell<Option<sys::Context>>" in glue
    ctx_fn = glue[glue.index("fn thread_context()"):glue.index("fn thread_prover(")]
    assert "sys::Context::new()" in ctx_fn and "make_current()" in ctx_fn
    lanes_fn = glue[glue.index("fn prove_lanes<"):glue.index("fn rounds<")]
    assert lanes_fn.index("thread_context()") < lanes_fn.index("resident(") < lanes_fn.index("thread_prover(n, lanes)")
    assert "sys::Prover::new(n as u32, batch as u32)" in glue                       # batch = 1: uzk_prover_create's shared kind
    prove = glue[glue.index("pub(super) fn prove<"):glue.index("pub fn prove_batch<")]
    assert "prove_lanes::<R, PCS, CS>(&mut prngs, &mut transcripts" in prove and "&[cs]" in prove
    batch = glue[glue.index("pub fn prove_batch<"):glue.index("fn prove_lanes<")]
    assert "prngs: &mut [R]" in batch and "transcripts: &mut [Transcript]" in batch
    lib = open(os.path.join(RUST, "uzkge-gpu-sys", "src", "lib.rs")).read()
    assert "pub fn coalesce_config(" in lib and "pub fn on_device(" in lib and "pub fn new_private(" in lib
    patch = open(os.path.join(RUST, "uzkge-gpu.patch")).read()
    assert "prove_batch as gpu_prove_batch" in patch


def test_circuit_identity_is_the_verifier_key_not_an_address():
    """VERDICT r3 / ADVICE r3: the resident circuit is found by its verifier-key commitments; the public-key commitments name the
    tables it currently holds, and a mismatch replaces the twelve tables (exclusive guard; round 1 re-checks under the shared one) before the
    proof starts -- `refresh_prover_params_public_key` swaps them in place once per game (shuffle/src/gen_params/params.rs:57-129)."""
    glue = _strip_rust_comments(open(os.path.join(RUST, "uzkge-glue", "gpu_prover.rs")).read())
    key_fn = glue[glue.index("fn circuit_key"):glue.index("fn public_key_of")]
    for field in ("cm_q_vec", "cm_s_vec", "cm_qb", "cm_prk_vec", "cm_q_ecc", "cm_shuffle_generator_vec", "cs_size"):
        assert field in key_fn, field
    assert "cm_shuffle_public_key_vec" not in key_fn and "cm_shuffle_public_key_vec" in glue[glue.index("fn public_key_of"):glue.index("fn coefs_of")]
    assert "as *const PlonkProverParams" not in glue and "as usize, n" not in glue          # no address in any key
    rounds = glue[glue.index("fn rounds<"):]
    # round 1 runs under the SHARED guard, after the check, in the same block (an exclusive lock across round 1 --
    # the one place provers of other threads can join a cohort -- made every cohort built through the glue one lane wide);
    # replacing the tables takes the exclusive guard, re-checks, and goes round the loop to the shared guard again
    rd, check_eq, r1, wr, check_ne, update = (rounds.index(t) for t in ("entry.read()", "r.public_key == public_key", "prover.round1(", "entry.write()",
                                                                       "r.public_key != public_key", "update_tables(sys::UZK_CS_QPK"))
    assert rd < check_eq < r1 < wr < check_ne < update and "r.public_key = public_key" in rounds[update:update + 400]
    assert "entry.lock()" not in glue and "let cms = loop {" in rounds
    assert rounds[rd:r1].count("{") - rounds[rd:r1].count("}") >= 1           # round 1 is inside the block that holds the read guard
    refresh = glue[glue.index("pub fn refresh_public_key"):glue.index("pub fn release_circuits")]
    assert "entry.write()" in refresh and "refresh_tables(sys::UZK_CS_QPK" in refresh and "r.public_key = bytes_of(&cms)" in refresh
    # no data-dependent panic where an error or the CPU path is available: no assert / expect / indexing-by-panic helpers
    assert not re.search(r"\bassert(_eq)?!|\.expect\(|panic!", glue)
    # the unwraps left are the reference's own "safe unwrap"s (prover.rs:197,213,243,301, helpers.rs:1045) and one Option known to be Some
    allowed = ("insert_beta_gamma(beta, gamma).unwrap()", "insert_alpha(alpha).unwrap()", "insert_zeta(zeta).unwrap()", "insert_u(u).unwrap()",
               "guard.as_ref().unwrap()", "eval_selector_multipliers(&w_refs).unwrap()")
    assert glue.count(".unwrap()") == sum(glue.count(a) for a in allowed) == len(allowed)


def test_patch_targets_the_cited_call_sites():
    patch = open(os.path.join(RUST, "uzkge-gpu.patch")).read()
    for f in PATCHED:
        assert f"+++ b/{f}" in patch, f
    assert patch.count("crate::gpu::fft(") == 4 and "crate::gpu::commit(" in patch and 'gpu = ["uzkge-gpu-sys"]' in patch
    assert "super::gpu_prover::prove(" in patch and "fn as_kzg_bn254" in patch and "fn commitment_from_g1" in patch
    assert "pub(super) fn r_poly_or_comm" in patch and "mod gpu_prover;" in patch
    assert "uzkge::plonk::gpu_refresh_public_key(" in patch and 'gpu = ["uzkge/gpu"]' in patch          # the refresh loop's device hook
    # the indexer's six per-table steps each ask the device once (and fall through to their CPU code on None)
    assert patch.count("crate::gpu::preprocess_tables(") == 1 and patch.count("gpu_tables(") == 6 and patch.count("let gpu_tables = ") == 2
    added = "\n".join(l for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    assert "panic!" not in added and ".unwrap()" not in added


PATCHED = ("Cargo.toml", "shuffle/Cargo.toml", "shuffle/src/gen_params/params.rs", "uzkge/Cargo.toml", "uzkge/src/lib.rs", "uzkge/src/plonk/mod.rs", "uzkge/src/plonk/helpers.rs", "uzkge/src/plonk/prover.rs", "uzkge/src/plonk/indexer.rs",
           "uzkge/src/poly_commit/pcs.rs", "uzkge/src/poly_commit/kzg_poly_commitment.rs", "uzkge/src/poly_commit/field_polynomial.rs")


@pytest.mark.skipif(not os.path.isdir("/root/reference/uzkge"), reason="reference tree not present (GPU box)")
def test_patch_applies_to_the_reference(tmp_path):
    if not shutil.which("patch"):
        pytest.skip("no patch(1)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_rust_patch.py"), "--check"])
    assert r.returncode == 0, "rust/uzkge-gpu.patch is stale: run tools/make_rust_patch.py"
    for f in PATCHED:
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join("/root/reference", f), dst)
    r = subprocess.run(["patch", "-p1", "-i", os.path.join(RUST, "uzkge-gpu.patch")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # the glue names items of the reference: they exist where it looks for them, with the visibility it needs
    helpers = open(tmp_path / "uzkge/src/plonk/helpers.rs").read()
    for item in ("pub(super) fn r_poly_or_comm", "pub(super) fn first_lagrange_poly", "pub(super) struct PlonkChallenges"):
        assert item in helpers, item
    prover = open(tmp_path / "uzkge/src/plonk/prover.rs").read()
    hook = prover.index("super::gpu_prover::prove(")
    assert prover.index("transcript_init_plonk(") < hook < prover.index("// 1. Build the PI polynomial")
    glue = open(os.path.join(RUST, "uzkge-glue", "gpu_prover.rs")).read()
    indexer = open("/root/reference/uzkge/src/plonk/indexer.rs").read()
    for field in set(re.findall(r"\bp\.([a-z_][a-z_0-9]*)", glue)) | set(re.findall(r"prover_params\.([a-z_][a-z_0-9]*)", glue)):
        assert re.search(rf"pub {field}:", indexer), f"PlonkProverParams has no field {field}"
    for field in set(re.findall(r"\bvp\.([a-z_][a-z_0-9]*)", glue)):
        assert re.search(rf"pub {field}:", indexer), f"PlonkVerifierParams has no field {field}"
    params_rs = open(tmp_path / "shuffle/src/gen_params/params.rs").read()
    hook = params_rs.index("uzkge::plonk::gpu_refresh_public_key(")
    assert params_rs.index("compute_shuffle_public_key_selectors()") < hook < params_rs.index("let q_shuffle_public_key_polys: Vec<FpPolynomial<Fr>>")
    proof_fields = re.findall(r"^            ([a-z_0-9]+)[,:]", glue[glue.index("proofs.push(PlonkProof {"):glue.index("Ok(proofs)")], flags=re.M)
    for field in proof_fields:
        assert re.search(rf"pub {field}:", indexer), f"PlonkProof has no field {field}"
    assert len(proof_fields) == 14
