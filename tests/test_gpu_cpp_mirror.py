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
