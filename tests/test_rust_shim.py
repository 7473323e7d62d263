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
    assert set(bound) == set(N.PROTOTYPES)


def test_struct_layouts_match_the_header():
    ffi = _ffi()
    assert "pub struct uzk_g1_affine {\n    pub x: [u64; 4],\n    pub y: [u64; 4],\n}" in ffi
    assert "pub struct uzk_g1_jac {\n    pub x: [u64; 4],\n    pub y: [u64; 4],\n    pub z: [u64; 4],\n}" in ffi
    assert "pub vec: [*const c_void; UZK_TQ_NVEC]" in ffi and "pub const UZK_TQ_NVEC: usize = 56;" in ffi
    import ctypes
    from uzkge_amd._native import QuotientArgs
    assert ctypes.sizeof(QuotientArgs) == 8 + 56 * 8 + (3 + 5 + 3 + 16) * 32       # the same block in the ctypes mirror


def test_glue_and_wrappers_only_call_bound_functions():
    ffi_fns = set(re.findall(r"pub fn (uzk_[a-z0-9_]+)\(", _ffi()))
    lib = open(os.path.join(RUST, "uzkge-gpu-sys", "src", "lib.rs")).read()
    used = set(re.findall(r"\b(uzk_[a-z0-9_]+)\(", lib))
    assert used and used <= ffi_fns, used - ffi_fns
    glue = open(os.path.join(RUST, "uzkge-glue", "gpu.rs")).read()
    wrappers = set(re.findall(r"pub fn ([a-z_]+)", lib))
    for call in re.findall(r"\bsys::([a-z_]+)\(", glue):
        assert call in wrappers, call
    for meth in re.findall(r"\bsrs\.([a-z_]+)\(", glue):
        assert re.search(rf"pub fn {meth}\(", lib), meth
    # the generator check: group_gen is compared with the library's, once per domain size
    assert "assert_same_generator" in glue and "domain_group_gen" in glue


def test_patch_targets_the_cited_call_sites():
    patch = open(os.path.join(RUST, "uzkge-gpu.patch")).read()
    for f in ("uzkge/Cargo.toml", "uzkge/src/lib.rs", "uzkge/src/poly_commit/kzg_poly_commitment.rs",
              "uzkge/src/poly_commit/field_polynomial.rs"):
        assert f"+++ b/{f}" in patch
    assert patch.count("crate::gpu::fft(") == 4 and "crate::gpu::commit(" in patch and 'gpu = ["uzkge-gpu-sys"]' in patch


@pytest.mark.skipif(not os.path.isdir("/root/reference/uzkge"), reason="reference tree not present (GPU box)")
def test_patch_applies_to_the_reference(tmp_path):
    if not shutil.which("patch"):
        pytest.skip("no patch(1)")
    for f in ("Cargo.toml", "uzkge/Cargo.toml", "uzkge/src/lib.rs", "uzkge/src/poly_commit/kzg_poly_commitment.rs",
              "uzkge/src/poly_commit/field_polynomial.rs"):
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join("/root/reference", f), dst)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(RUST, "uzkge-gpu.patch")], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
