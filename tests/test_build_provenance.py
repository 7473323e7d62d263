"""The GPU box receives prebuilt, git-ignored binaries.  Two checks tie them to the tree:
  * the library carries a hash of every source it is built from (uzk_version(), stamped by the Makefile); it is recomputed
    here from the files -- runs on the CPU box and on the GPU box alike, seconds;
  * a from-scratch build of a COPY of the sources (`make clean && make`) reproduces the in-tree library byte for byte
    (hipcc output is deterministic) -- CPU box only, about two minutes on 8 cores; UZK_SKIP_CLEAN_BUILD=1 skips it."""
import glob
import hashlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "uzkge_amd", "csrc")


def _source_hash() -> str:
    files = []
    for pat in ("*.hip", "*.cpp", "*.hpp", "*.inc"):
        files += [os.path.basename(f) for f in glob.glob(os.path.join(CSRC, pat))]
    files += ["Makefile", "../../include/uzkge_gpu.h", "../../include/uzkge_gpu_test.h"]
    h = hashlib.sha256()
    for f in sorted(set(files)):                 # GNU make's $(sort ...): plain byte order, duplicates removed
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def test_library_is_stamped_with_the_hash_of_its_sources():
    from uzkge_amd import _native as N
    version = N.lib.uzk_version().decode()
    assert version.endswith("src:" + _source_hash()), (
        f"{version}: libuzkge_gpu.so was built from other sources than the tree holds (hash now {_source_hash()}); "
        "run `make -C uzkge_amd/csrc`")


@pytest.mark.gpu
def test_library_on_the_gpu_box_is_stamped_with_the_hash_of_its_sources(gpu):
    test_library_is_stamped_with_the_hash_of_its_sources()


@pytest.mark.skipif(os.environ.get("UZK_SKIP_CLEAN_BUILD") == "1", reason="UZK_SKIP_CLEAN_BUILD=1")
@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_clean_build_reproduces_the_shipped_library(tmp_path):
    from uzkge_amd import backend
    if backend.device_count() > 0:
        pytest.skip("CPU-box check (the GPU box runs the stamp test)")
    shutil.copytree(CSRC, tmp_path / "uzkge_amd" / "csrc", ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    build = str(tmp_path / "uzkge_amd" / "csrc")
    subprocess.check_call(["make", "-s", "-C", build, "clean"])
    jobs = str(max(2, min(8, os.cpu_count() or 2)))
    subprocess.check_call(["make", "-s", "-j", jobs, "-C", build], stderr=subprocess.DEVNULL)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    fresh, shipped = str(tmp_path / "uzkge_amd" / "libuzkge_gpu.so"), os.path.join(ROOT, "uzkge_amd", "libuzkge_gpu.so")
    assert sha(fresh) == sha(shipped), "the in-tree libuzkge_gpu.so is not what a clean build of the tree produces"
