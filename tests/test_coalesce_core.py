"""The synchronisation core of the shared provers (uzkge_amd/csrc/coalesce_core.hpp) on the CPU: built with plain g++ -- it has
no HIP in it -- and driven by a fake backend from many threads with stragglers, abandoned proofs, lanes that fail alone and
provers that come and go (tests/cpp/coalesce_core_test.cpp).  Once plain, once under ThreadSanitizer: the host-side concurrency
of the library's busiest lock is checked by a race detector, not only by GPU stress runs."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "coalesce_core_test.cpp")


def _build(tmp_path, flags):
    exe = os.path.join(str(tmp_path), "coalesce_core_test")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", *flags, "-o", exe, SRC], check=True)
    return exe


def _run(exe, args, timeout):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (lines[0] if lines else {})


@pytest.mark.parametrize("threads,proofs,lanes", [(12, 300, 4), (5, 400, 8), (16, 150, 2)])
def test_every_caller_gets_its_own_proof(tmp_path, threads, proofs, lanes):
    r, res = _run(_build(tmp_path, []), (threads, proofs, lanes), 300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert res["bad"] == 0 and res["proofs"] > 0 and res["cohorts_opened"] == res["cohorts_closed"]
    assert 1 < res["widest_round"] <= lanes                      # callers did share rounds, never more than the configured lanes
    assert res["lanes_run"] > res["rounds_run"]
    assert res["moved_out"] > 0 and res["failed_on_purpose"] > 0 and res["abandoned"] > 0      # every path was taken


def test_scheduling_policy_with_a_stub_backend(tmp_path):
    """What the GPU suite used to assert with wall-clock bounds on a foreign host: a lone prover never waits, a full cohort leaves
    at once, a caller that stays away is moved out after straggler_wait and the one on time finishes its round before the dawdler
    is back (order of events), two-wide rounds, exactly one lane moved out, both proofs right."""
    r = subprocess.run([_build(tmp_path, []), "policy"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_no_data_race_under_thread_sanitizer(tmp_path):
    r, res = _run(_build(tmp_path, ["-fsanitize=thread"]), (8, 120, 4), 600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:4000]
    assert res["bad"] == 0 and res["widest_round"] > 1


@pytest.mark.parametrize("threads", [8, 16, 32])
def test_callers_with_unrelated_phases_converge_on_full_teams(tmp_path, threads):
    """tests/cpp/coalesce_core_sim.cpp: a backend that only takes time the way the GPU does (a round costs base + per-lane time,
    four at once), threads that start out of phase and prove back to back.  After a warm-up every round must serve (nearly) a
    whole team -- threads / 4 callers -- and gathering must cost next to nothing: the one-proof API then runs as few, full
    lockstep sequences."""
    exe = os.path.join(str(tmp_path), "coalesce_core_sim")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "coalesce_core_sim.cpp")], check=True)
    r = subprocess.run([exe, str(threads), "8", "4", "2000", "1.0"], capture_output=True, text=True, timeout=120)
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ideal_lanes_per_round"] == threads / 4
    assert res["lanes_per_round"] >= 0.9 * res["ideal_lanes_per_round"], res
    assert res["gather_us_per_cohort"] < 1000 and res["moved_out"] == 0, res


@pytest.mark.parametrize("threads,want", [(5, 2.5), (6, 2.0), (7, 7 / 3)])
def test_no_team_of_one_once_provers_outnumber_teams(tmp_path, threads, want):
    """Five to seven provers over four teams: a lone prover beside pairs made the whole slower than four threads on the GPU
    (profiles/r05_shared_odd_thread_counts.txt), so the teams become min(groups, provers / 2) -- 5 -> (3, 2), 6 -> (2, 2, 2),
    7 -> (3, 2, 2) -- and the rounds serve provers / teams callers on average."""
    exe = os.path.join(str(tmp_path), "coalesce_core_sim")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "coalesce_core_sim.cpp")], check=True)
    r = subprocess.run([exe, str(threads), "8", "4", "2000", "1.0"], capture_output=True, text=True, timeout=120)
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["lanes_per_round"] >= 0.9 * want, res
    assert res["moved_out"] == 0, res
