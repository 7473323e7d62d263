#!/usr/bin/env python3
"""bench.py -- headline measurement for the MI355X uzkge backend.

Metric (BASELINE.json): BN254 G1 MSM points/s at 2^24 points per GPU (synthetic random points
and uniform scalars generated on device, resident in HBM before the timed region), plus the
Fr NTT at 2^22 reported under "extra".  One step = one MSM over this rank's 2^24 points; with
N > 1 ranks (one process per GPU, torch.distributed / RCCL) every rank owns its own point chunk
(weak scaling, a 2^24*N-point MSM in total), the 96-byte partial sums are all-gathered and folded
on every rank -- EC addition is not an RCCL reduce op.

`python bench.py --gpus N` starts the N ranks ITSELF (N child processes, one per GPU, launched before this
process touches HIP or torch; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment); under an
external launcher (torchrun sets RANK) the process is a rank straight away.  `--total-log-n K` is
the strong-scaling mode of BASELINE config #5: 2^K points in total, 2^K / N per rank.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (msm_accumulate):
achieved = 96 B/point (32 B scalar + 64 B affine base) x points per launch / the
kernel's mean duration, measured live with HIP events on the library stream during the timed
steps.  `cpu_baseline` times the CPU oracle (own C restatement, NOT arkworks) on a bounded sample
of the same workload on this node's host cores.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MADS_PER_MIXED_ADD = 1467   # v_mad_u64_u32 per mixed addition (ec29.hpp: 6 products, 2 squarings, 1 dual product)
NUM_SIMDS = 1024
LANES = 64
BOOST_SCLK_MHZ = 2390
# The roofline that binds both hot loops is vector-instruction ISSUE: a SIMD issues at most one vector instruction per
# quad-cycle (4 cycles), whatever the instruction (the SQ counters show SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU for these
# kernels, v_mad_u64_u32 included: profiles/sq_counters.json).  Peak = 1024 SIMDs x 64 lanes x sclk / 4 lane-instructions/s.


def _issue_peak_t(sclk_mhz: float) -> float:
    return NUM_SIMDS * LANES * sclk_mhz * 1e6 / 4.0 / 1e12


def _sample_clocks(work, sync, seconds: float = 2.0):
    """Median sclk (MHz) and socket power (W) from rocm-smi while `work` repeats; None if unavailable."""
    import re
    import shutil
    import subprocess
    import threading
    if not shutil.which("rocm-smi"):
        return None
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            if m:
                samples.append((int(m.group(1)), float(pw.group(1)) if pw else None))

    th = threading.Thread(target=sampler)
    t0 = time.perf_counter()
    work(); sync()
    th.start()
    while time.perf_counter() - t0 < seconds:
        work()
    sync()
    stop[0] = True
    th.join()
    samples = samples[1:] if len(samples) > 2 else samples      # the first one may predate the ramp
    if not samples:
        return None
    sclk = sorted(x[0] for x in samples)[len(samples) // 2]
    pws = sorted(x[1] for x in samples if x[1] is not None)
    return {"sclk_mhz": sclk, "power_w": pws[len(pws) // 2] if pws else None, "samples": len(samples)}


class _ClockSampler:
    """Shader clock (MHz) and socket power (W) sampled in the background WHILE a region runs (the timed steps): a driver line then
    carries the clock and power of its own box during its own measurement -- the pool's boxes differ by several per cent under
    the power cap.  Reads through librocm_smi64 in-process (rsmi_dev_gpu_clk_freq_get / rsmi_dev_current_socket_power_get: what
    `rocm-smi --showclocks --showpower` prints, without a process per sample); the rocm-smi tool itself is the fallback.  (The
    amdgpu hwmon files were tried first and read 102 MHz under full load on this pool: not the shader clock.)"""

    def __init__(self, device_index: int = 0, period_s: float = 0.01):
        self.samples, self._stop, self._th, self.period = [], False, None, period_s
        self.source, self._lib, self._dev = None, None, device_index
        try:
            import ctypes

            class Freq(ctypes.Structure):
                _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                            ("frequency", ctypes.c_uint64 * 33)]
            lib = ctypes.CDLL("librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                self._lib, self._Freq, self._ct = lib, Freq, ctypes
                if self._read() is not None:
                    self.source = "librocm_smi64"
        except Exception:
            self._lib = None
        if self.source is None:
            import shutil
            self._lib = None
            if shutil.which("rocm-smi"):
                self.source = "rocm-smi"

    def _read(self):
        if self._lib is not None:
            ct = self._ct
            f = self._Freq()
            if self._lib.rsmi_dev_gpu_clk_freq_get(ct.c_uint32(self._dev), ct.c_int(0), ct.byref(f)) != 0 or f.current >= 33:      # RSMI_CLK_TYPE_SYS
                return None
            pw = ct.c_uint64(0)
            watts = None
            if self._lib.rsmi_dev_current_socket_power_get(ct.c_uint32(self._dev), ct.byref(pw)) == 0:
                watts = pw.value / 1e6
            return f.frequency[f.current] / 1e6, watts
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return None
        m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        return (float(m.group(1)), float(pw.group(1)) if pw else None) if m else None

    def _loop(self):
        while not self._stop:
            t = time.perf_counter()
            r = self._read()
            if r:
                self.samples.append((t, r[0], r[1]))
            if self._lib is not None:
                time.sleep(self.period)

    def start(self):
        import threading
        if self.source:
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def stop(self, t0: float, t1: float):
        """Median over the samples taken inside [t0, t1] (perf_counter values); None when there were none."""
        self._stop = True
        if self._th:
            self._th.join(timeout=15)
        inside = [x for x in self.samples if t0 <= x[0] <= t1]
        if not inside:
            return None
        sclk = sorted(x[1] for x in inside)[len(inside) // 2]
        pws = sorted(x[2] for x in inside if x[2] is not None)
        return {"sclk_mhz_timed": round(sclk), "power_w_timed": round(pws[len(pws) // 2], 1) if pws else None, "samples": len(inside),
                "source": self.source, "what": "median shader clock and socket power sampled in the background during the timed steps"}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_ranks: int, argv) -> int:
    """Start `n_ranks` copies of this script as rank processes and wait for them.  Runs in a parent
    that has not imported torch or the backend (no HIP call), so nothing GPU-initialised is ever
    forked or exec'd.  Returns the first non-zero exit code (the other ranks are then terminated)."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "UZK_BENCH_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = set(range(n_ranks))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--total-log-n", type=int, default=0,
                    help="strong scaling: log2 of the TOTAL point count, split evenly over the ranks (config #5: 26)")
    ap.add_argument("--scalars", choices=["uniform", "mix"], default="uniform",
                    help="scalar set of the headline loop: (A) uniform or (B) the prover-like mix of BASELINE.md section 4")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank contributes (rank + 1) * G; exercises the launcher, the collective and the fold")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=24, help="log2 MSM points per GPU")
    ap.add_argument("--ntt-log-n", type=int, default=22)
    ap.add_argument("--points", choices=["random", "arith"], default="random")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--no-extras", action="store_true", help="skip the window-table and prover-shape extras")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # parent: no torch, no HIP -- only children touch the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


def _host_cores() -> int:
    """Host threads this process can really run at once: the affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants 16 CPUs' worth of time -- more threads than that only contend)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def _kernel_sources_sha() -> str:
    """sha256 over the sources of the two hot kernels: ties committed counter files to the code they measured."""
    import hashlib
    h = hashlib.sha256()
    for f in ("msm.hip", "ec29.hpp", "fp29.hpp", "mul29_gfx950.inc", "ntt.hip"):
        with open(os.path.join(ROOT, "uzkge_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _load_counters(name: str):
    """profiles/<name> if it was measured on the kernel sources of this tree, else (None, reason)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "no counter file"
    if d.get("kernel_sources_sha") != _kernel_sources_sha():
        return None, f"stale: {name} was measured on other kernel sources ({d.get('kernel_sources_sha')})"
    return d, None


def _gather_proofs(dist, world, mine):
    """One proof per device is the realistic multi-GPU mode at the prover's size: every rank proves on its own GPU,
    nothing is exchanged, the job's rate is the sum.  `mine`: this rank's {"proofs_per_s": ..} (or {"error": ..})."""
    per = [mine]
    if world > 1:
        objs = [None] * world
        dist.all_gather_object(objs, mine)
        per = objs
    rates = [p.get("proofs_per_s") for p in per]
    return {"what": "every rank runs tests/cpp/prover_rounds on its own GPU in the reference's call pattern -- host_threads_per_rank host threads (min(32, 2 x host_cores / ranks)), each calling "
                    "uzk_prove_round1..5 on its own prover of ONE proof from the default context, its own witness uploaded from pinned memory "
                    "every proof; the library shares the rounds (uzk_coalesce_config defaults) -- one circuit resident per process, no collective",
            "per_rank": per, "proofs_per_s_total": (sum(rates) if all(r is not None for r in rates) else None)}


def _run_prover_rounds(extra_args, visible_device=None, n_log=14, timeout=300):
    """tests/cpp/prover_rounds (built by __graft_entry__.build(): plain g++, the C ABI only) on ChainInputs(2^n_log, 11); returns its
    JSON lines.  visible_device: the child sees only that GPU (an entry of the parent's own visible list)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tests", "cpp", "prover_rounds")
    if not os.path.exists(exe):
        raise RuntimeError("tests/cpp/prover_rounds is not built (python -c 'import __graft_entry__ as g; g.build()')")
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import prover_chain
    from test_gpu_cpp_mirror import _write_inputs
    env = dict(os.environ)
    if visible_device is not None:
        seen = [v for v in env.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
        env["HIP_VISIBLE_DEVICES"] = seen[visible_device] if visible_device < len(seen) else str(visible_device)
    with tempfile.TemporaryDirectory() as td:
        _write_inputs(prover_chain.ChainInputs(1 << n_log, 11), td, precompute=True)
        r = subprocess.run([exe, td] + [str(a) for a in extra_args], capture_output=True, text=True, timeout=timeout, env=env)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError((r.stdout + r.stderr)[-400:])
    return lines


SQ_MEASURED_LOG_N, SQ_MEASURED_WINDOWS = 24, 15     # the launch the SQ counter pass of tools/profile_round.sh measures (bench.py defaults)


def _alu_roofline(n: int, cbits: int, nwin: int, acc_avg_ms: float, plain_workload: bool, clocks_timed=None) -> dict:
    """The binding roofline of msm_accumulate: vector-instruction issue (see _issue_peak_t).  Vector instructions per mixed addition come
    from the stamped SQ counter pass (profiles/sq_counters.json: measured on ONE launch of 2^24 points x 15 windows); they are a property
    of the addition, not of the launch, so they price any launch of the same kernel on uniform scalars with the automatic window plan --
    `valu_insts_measured_at` says where they were counted.  The achieved rate uses THIS run's mean kernel duration."""
    adds = float(n) * nwin
    sq, why = _load_counters("sq_counters.json")
    alu = {"unit": "T lane vector-instructions/s (issue: one per SIMD per 4 cycles)", "mixed_adds_per_launch": int(adds),
           "window_bits": cbits, "windows": nwin, "mads_per_mixed_add": MADS_PER_MIXED_ADD,
           "peak_at_boost_clock": round(_issue_peak_t(BOOST_SCLK_MHZ), 2), "boost_sclk_mhz": BOOST_SCLK_MHZ}
    insts_per_add = None
    if sq is not None and "msm_accumulate" in sq and plain_workload and nwin > 0:
        ent = sq["msm_accumulate"]
        at_adds = float(ent.get("measured_mixed_adds", (1 << SQ_MEASURED_LOG_N) * SQ_MEASURED_WINDOWS))
        insts_per_add = ent["SQ_INSTS_VALU"] * LANES / at_adds        # wave instructions x 64 lanes / lane-additions
        alu["valu_insts_measured_at"] = (f"2^{ent.get('measured_log_n', SQ_MEASURED_LOG_N)} points x {ent.get('measured_windows', SQ_MEASURED_WINDOWS)} windows "
                                         f"({sq.get('profile_tag')}); this launch: {n} points x {nwin} windows of {cbits} bits")
    alu_achieved = 0.0
    if insts_per_add:
        alu_achieved = adds * insts_per_add / (acc_avg_ms * 1e-3) / 1e12 if acc_avg_ms > 0 else 0.0
        alu.update({"valu_insts_per_mixed_add": round(insts_per_add, 1), "achieved": round(alu_achieved, 3),
                    "frac_at_boost_clock": round(alu_achieved / _issue_peak_t(BOOST_SCLK_MHZ), 4),
                    "mad_share_of_instructions": round(MADS_PER_MIXED_ADD / insts_per_add, 4),
                    "sq_counters": sq["msm_accumulate"]})
    else:
        alu["sq_counters_note"] = why or "no SQ counter pass for this configuration"
    if clocks_timed:
        alu.update(clocks_timed)
        alu["peak_at_timed_clock"] = round(_issue_peak_t(clocks_timed["sclk_mhz_timed"]), 2)
        if alu_achieved:
            alu["frac_at_timed_clock"] = round(alu_achieved / _issue_peak_t(clocks_timed["sclk_mhz_timed"]), 4)
    alu["_achieved"] = alu_achieved
    return alu


def _cpu_threads(world: int) -> int:
    """Host threads of the CPU baseline leg: every CPU this process may use at N = 1; under N > 1 ranks the other ranks wait at the
    closing barrier while rank 0 times the port -- one CPU is left to each of them (a waiting rank may spin)."""
    return max(1, _host_cores() - (world - 1))


def _cpu_baseline_msm(oc, hp, hs, threads: int, world: int):
    """The C port (oracle/bn254_oracle.c: Pippenger, own restatement, NOT arkworks) timed on a bounded sample of the workload."""
    m = hp.shape[0]
    tc = time.perf_counter()
    ref = oc.msm_pippenger(hp, hs, 0, threads)
    cpu_s = time.perf_counter() - tc
    lg = m.bit_length() - 1
    return ref, {"value": m / cpu_s, "unit": "points/s", "cores": threads, "kind": "port",
                 "sample": f"first 2^{lg} points/scalars of rank 0's workload, Pippenger in oracle/bn254_oracle.c (own CPU restatement, not arkworks), "
                           + ("one thread per host CPU this process may use (affinity mask capped by the cgroup quota)" if world == 1 else
                              f"timed on rank 0 while the other {world - 1} ranks wait at the closing barrier; threads = host CPUs - {world - 1}"),
                 "host_cores": _host_cores(), "seconds": round(cpu_s, 3)}


def _cpu_baseline_proof(oc, np, threads: int, reps: int = 2) -> dict:
    """One proof's MSMs and transforms (16 MSM of n = 2^14 points over the reference's Lagrange SRS, 10 iFFT(n) +
    7 FFT(n), 10 FFT(6n) + 1 iFFT(6n)) through the C port on the host cores -- the part of `prover_with_lagrange` this backend replaces
    call for call (about 78 % of the reference's modular multiplications).  A LOWER bound of a CPU prover's time: the
    quotient evaluation, z_poly, the openings' divisions and the transcript are not in it.  Not arkworks."""
    import bn254_py as opy
    with open(os.path.join(ROOT, "tests", "golden", "lagrange-srs-16384.bin"), "rb") as f:
        hp = oc.points_from_affine(opy.parse_srs_g1(f.read()))
    n = hp.shape[0]
    rng = np.random.default_rng(14)

    def rand(m):
        a = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        return a
    sc16 = [rand(n) for _ in range(16)]
    vn = [rand(n) for _ in range(17)]
    v6 = [rand(6 * n) for _ in range(11)]
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        for v in sc16:
            oc.msm_pippenger(hp, v, 0, threads)
        t_msm = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i, v in enumerate(vn):
            oc.ntt(v, inverse=i < 10, threads=threads)
        for i, v in enumerate(v6):
            oc.ntt(v, inverse=i == 10, threads=threads)
        t_ntt = time.perf_counter() - t0
        if best is None or t_msm + t_ntt < best[0] + best[1]:
            best = (t_msm, t_ntt)
    total = best[0] + best[1]
    return {"value": round(total * 1e3, 2), "unit": "ms per proof (MSM + NTT part only)", "cores": threads, "kind": "port",
            "covers": "MSM + NTT only (about 78 % of the reference's modular multiplications, SURVEY.md 3.5): a lower bound of a CPU proof",
            "msm_ms": round(best[0] * 1e3, 2), "ntt_ms": round(best[1] * 1e3, 2), "proofs_per_s_upper_bound": round(1.0 / total, 2),
            "sample": f"one proof's 16 MSM(2^14, reference Lagrange SRS, uniform scalars) + 17 NTT(2^14) + 11 NTT(6 x 2^14) through oracle/bn254_oracle.c, "
                      f"best of {reps} passes; uniform scalars are the MSM's worst case (a real witness is mostly 0 / 1)"}


def dry_run_rank(args, rank, world, dist, np) -> None:
    """--dry-run: the launcher, the rendezvous, the all-gather of 96-byte partials and the fold, with no GPU:
    rank r contributes (r + 1) * G (built with the host-side fold of the C ABI)."""
    import torch
    from uzkge_amd import backend as b
    one_q = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)
    two_q = np.array([0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)
    g = np.concatenate([one_q, two_q, one_q])                      # G = (1, 2, 1) in Montgomery form
    from uzkge_amd.sharded import ShardedCommitter
    part = b.g1_fold(np.stack([g] * (rank + 1)))
    total = ShardedCommitter(None).exchange(part)                  # the product's exchange: all-gather + fold on every rank
    aff = b.g1_to_affine(total)
    # the proofs-per-device extra of a real N > 1 run: every rank reports its own rate, rank 0 prints the sum (here: 100 (r + 1))
    proofs = _gather_proofs(dist, world, {"rank": rank, "proofs_per_s": 100.0 * (rank + 1)})
    if rank == 0:
        # the two objects an N > 1 line must carry, built by the SAME code as a real run while the other ranks wait at the barrier:
        # cpu_baseline (the C port on the reference's 2^14-point Lagrange SRS here -- no device points exist) and roofline.alu
        # (strong-scaling shape of config #5 on 8 GPUs: 2^23 points per GPU; nominal duration and clock, nothing was timed)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import bn254_py as opy
        import oracle_c as oc   # checker / reported baseline only
        with open(os.path.join(ROOT, "tests", "golden", "lagrange-srs-16384.bin"), "rb") as f:
            hp = oc.points_from_affine(opy.parse_srs_g1(f.read()))
        rng = np.random.default_rng(5)
        hs = rng.integers(0, 1 << 62, size=(hp.shape[0], 4), dtype=np.uint64)
        _, cpu_baseline = _cpu_baseline_msm(oc, hp, hs, _cpu_threads(world), world)
        alu = _alu_roofline(1 << 23, 17, 15, 8.5, True, {"sclk_mhz_timed": 2050, "source": "dry run: nominal duration (8.5 ms) and clock, nothing was timed"})
        alu.pop("_achieved")
        print(json.dumps({"metric": "bn254_g1_msm_points_per_sec", "value": 0.0, "unit": "points/s", "n_gpus": world,
                          "dry_run": True, "dist_world_size": dist.get_world_size() if world > 1 else 1,
                          "fold_of_rank_partials_affine": [int(x) for x in aff],
                          "expect": f"{world * (world + 1) // 2} * G", "host_cores": _host_cores(),
                          "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "alu": alu}, "cpu_baseline": cpu_baseline,
                          "extra": {"proofs_per_device": proofs}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args) -> None:
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # UZK_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks then share a device and exchange their partials through host memory); the driver's runs
    # use the default: one rank per GPU over RCCL.
    backend = os.environ.get("UZK_BENCH_BACKEND", "gloo" if args.dry_run else "nccl")
    if args.dry_run:
        if world > 1:
            import torch.distributed as dist  # type: ignore
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        dry_run_rank(args, rank, world, dist, np)
        return
    # UZK_BENCH_FORCE_DIST=1: a single rank still joins a process group and runs the collective (RCCL world of one):
    # the exchange path can be exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("UZK_BENCH_FORCE_DIST") == "1"
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    if use_dist:
        import torch.distributed as dist  # type: ignore
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from uzkge_amd import backend as b

    b.init(dev_index)
    if args.window_bits:
        b.set_msm_window_bits(args.window_bits)

    strong = args.total_log_n > 0
    if strong:
        total = 1 << args.total_log_n
        if total % world:
            raise SystemExit(f"--total-log-n {args.total_log_n}: 2^{args.total_log_n} points do not split over {world} ranks")
        n = total // world
    else:
        n = 1 << args.log_n
    seed = 0x755A6B67655F6D73 + rank   # documented SplitMix64 seed, per-rank chunk
    pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
    sc = torch.empty((n, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if args.points == "random":
        b.synth_points_random(pts.data_ptr(), n, seed)
    else:
        k = np.array([seed & 0xFFFFFFFFFFFFFFFF, 1, 0, 0], dtype=np.uint64)
        b.synth_points_arith(pts.data_ptr(), n, k)
    if args.scalars == "mix":
        b.synth_scalars_mix(sc.data_ptr(), n, seed ^ 0x5CA1AB1E)
    else:
        b.synth_scalars(sc.data_ptr(), n, seed ^ 0x5CA1AB1E)
    srs = b.Srs.from_device(pts.data_ptr(), n)
    # the product's sharded commit (uzkge_amd/sharded.py): this rank's MSM, the 96-byte all-gather over RCCL / xGMI, the fold
    # of the N partial sums on every rank; it keeps the two wall clocks that say where a step's time goes on THIS rank
    from uzkge_amd.sharded import ShardedCommitter
    committer = ShardedCommitter(srs, device=dev, force_collective=use_dist)

    def step():
        return committer.commit_device(sc.data_ptr(), n)

    def fence():
        b.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        b.sync()
        torch.cuda.synchronize()

    def timed_steps(fn, steps):
        fence()
        t0 = time.perf_counter()
        res = None
        for _ in range(steps):
            res = fn()
        fence()
        mine = el = time.perf_counter() - t0           # this rank's clock around the steps and their fences -- nothing else
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, mine, res

    for _ in range(args.warmup):
        step()
    b.profile_reset()
    b.profile_enable(True)
    committer.reset_clocks()
    sampler = _ClockSampler(local_rank).start() if rank == 0 else None
    t_local = time.perf_counter()
    elapsed, local_elapsed, result = timed_steps(step, args.steps)     # local_elapsed: this rank's own clock, taken BEFORE the MAX all-reduce
    clocks_timed = sampler.stop(t_local, t_local + local_elapsed) if sampler else None
    b.profile_enable(False)
    prof = b.profile_table()
    # per-rank view of the timed region, so that a multi-GPU run explains itself: every rank's ms per step, its MSM share and
    # its exchange share (all-gather of the 96-byte partials + fold; expected << 0.1 ms, i.e. flat weak scaling)
    mine = {"rank": rank, "ms_per_step": round(local_elapsed / args.steps * 1e3, 4), "msm_ms_per_step": round(committer.msm_s / args.steps * 1e3, 4),
            "exchange_ms_per_step": round(committer.exchange_s / args.steps * 1e3, 4)}
    per_rank = [mine]
    if use_dist:
        objs = [None] * world
        dist.all_gather_object(objs, mine)
        per_rank = sorted(objs, key=lambda o: o["rank"])

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed

    # which devices the ranks really ran on (so the record shows N distinct GPUs behind the collective)
    dev_ids = [dev_index]
    dev_uuids = [str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", ""))]
    if use_dist:
        objs = [None] * world
        dist.all_gather_object(objs, (dev_index, dev_uuids[0]))
        dev_ids = [o[0] for o in objs]
        dev_uuids = [o[1] for o in objs]

    acc_cnt, acc_ms = prof.get("msm_accumulate", (0, 0.0))
    acc_avg_ms = acc_ms / max(acc_cnt, 1)
    achieved = (96.0 * n) / (acc_avg_ms * 1e-3) / 1e9 if acc_cnt else 0.0
    # HBM bytes per launch: rocprofv3 PMC passes summarised in profiles/pmc_traffic.json, used only when that file
    # was measured on the kernel sources of this tree (it records their hash) and on this configuration
    traffic, traffic_note = None, None
    pmc, why = _load_counters("pmc_traffic.json")
    if pmc is None:
        traffic_note = why
    else:
        ent = pmc.get("msm_accumulate", {}).get(str(n.bit_length() - 1))
        if ent and args.points == "random" and not args.window_bits and args.scalars == "uniform" and n & (n - 1) == 0:
            traffic = ent["traffic_bytes"]
        else:
            traffic_note = "no PMC pass for this configuration"
    # The kernel's real bound is vector-instruction issue (_alu_roofline): ~2090 vector instructions per mixed addition, 1467 of
    # them v_mad_u64_u32 (ec29.hpp), counted by the SQ pass of the same kernel; the achieved rate uses THIS run's mean duration.
    cbits, nwin = (args.window_bits, 0)
    try:
        cbits, nwin = b.msm_plan_info(n)
    except Exception:
        pass
    plain = args.scalars == "uniform" and not args.window_bits and n & (n - 1) == 0 and n >= (1 << 21)     # the c = 17 plan of the counted launch
    alu = _alu_roofline(n, cbits, nwin, acc_avg_ms if acc_cnt else 0.0, plain, clocks_timed)
    alu_achieved = alu.pop("_achieved")
    roofline = {
        "bound": "hbm", "kernel": "msm_accumulate", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
        "avg_launch_ms": round(acc_avg_ms, 4), "algorithmic_bytes_per_launch": 96 * n,
        "alu": alu,
        "note": "vector-issue bound (254-bit modular arithmetic, ~2090 instructions per mixed addition), not HBM bound: `alu` is the "
                "binding roofline, `frac` the HBM one the contract asks for; see DESIGN.md 3.1",
    }
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    # device kernels (HIP events) and, prefixed host_, the host-side sections of the call (wall clock)
    kernels = {k: {"launches": v[0], "avg_ms": round(v[1] / max(v[0], 1), 4)} for k, v in sorted(prof.items())}

    extra = {"msm_kernels": kernels}
    solo = rank == 0 and world == 1 and not args.no_extras

    # ---- BASELINE config #5 next to the weak-scaling headline: 2^26 points in total, split over the ranks ------
    if world > 1 and not strong and not args.no_extras and  (1 << 26) // world <= n:
        m5 = (1 << 26) // world

        def step5():
            return committer.commit_device(sc.data_ptr(), m5)
        step5()
        el5, _ = timed_steps(step5, max(3, args.steps // 2))
        extra["strong_2p26"] = {"what": "BASELINE config #5: 2^26 points in total, point-chunk sharded over the ranks, "
                                        "RCCL all-gather of the partial sums, fold on every rank",
                                "points_per_gpu": m5, "total_points": m5 * world, "steps": max(3, args.steps // 2),
                                "ms_per_step": round(el5 / max(3, args.steps // 2) * 1e3, 4),
                                "value": m5 * world * max(3, args.steps // 2) / el5, "unit": "points/s", "scaling": "strong"}

    # ---- proofs per second, one prover process per device (the realistic scaling mode at n = 2^14) --------
    if world > 1 and not args.no_extras:
        # host threads per rank: the reference's call pattern needs a caller thread per proof in flight; 32 per device when the host
        # has them, never more than twice this rank's share of the CPUs the job may use (8 ranks x 32 threads on a 16-CPU grant would
        # measure the host, not the GPUs) -- the line prints the host budget beside the rate
        p_threads = max(2, min(32, 2 * _host_cores() // world))
        try:
            lines = _run_prover_rounds([10, p_threads, 1, "shared", 0], visible_device=dev_index)
            mine_p = {"rank": rank, "device": dev_index, "host_threads": p_threads, "proofs_per_s": lines[-1]["proofs_per_s"], "ms_per_proof_single": lines[0]["ms_per_chain"],
                      "proofs_per_shared_round": lines[-1].get("proofs_per_shared_round"), "threads_agree_with_single": lines[-1]["threads_agree_with_single"]}
        except Exception as e:
            mine_p = {"rank": rank, "device": dev_index, "error": str(e)[-300:]}
        proofs = _gather_proofs(dist, world, mine_p)
        if rank == 0:
            proofs["host_cores"] = _host_cores()
            proofs["host_threads_per_rank"] = p_threads
            extra["proofs_per_device"] = proofs
            extra["proofs_per_s_total"] = proofs["proofs_per_s_total"]

    # ---- NTT 2^22 (single GPU path; replicas only under N > 1) --------------------------------
    if not args.no_ntt and rank == 0:
        nn = 1 << args.ntt_log_n
        x = torch.empty((nn, 4), dtype=torch.int64, device=dev)
        y = torch.empty((nn, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        b.synth_scalars(x.data_ptr(), nn, 0x4E5454)
        b.ntt_device(x.data_ptr(), y.data_ptr(), nn, sync=True)          # warm-up (builds the plan)
        b.ntt_device(y.data_ptr(), y.data_ptr(), nn, inverse=True, sync=True)
        roundtrip_ok = bool(torch.equal(x, y))
        # throughput: back-to-back transforms, no per-kernel events (they add bubbles between the three
        # short passes); then a short profiled loop for the per-kernel durations
        reps_t = 200
        b.sync()
        t1 = time.perf_counter()
        for _ in range(reps_t):
            b.ntt_device(x.data_ptr(), y.data_ptr(), nn)
        b.sync()
        ntt_s = (time.perf_counter() - t1) / reps_t
        b.sync()
        t1 = time.perf_counter()
        for _ in range(reps_t // 2):
            b.ntt_device(x.data_ptr(), y.data_ptr(), nn)
            b.ntt_device(y.data_ptr(), y.data_ptr(), nn, inverse=True)
        b.sync()
        rt_s = (time.perf_counter() - t1) / (reps_t // 2)
        b.profile_reset()
        b.profile_enable(True)
        reps = 20
        for _ in range(reps):
            b.ntt_device(x.data_ptr(), y.data_ptr(), nn)
        b.sync()
        b.profile_enable(False)
        p2 = b.profile_table()
        kern_ms = sum(v[1] for k, v in p2.items() if k.startswith("ntt_pass")) / reps
        ntt_traffic = None
        if pmc is not None:
            ntt_traffic = pmc.get("ntt_pass", {}).get(str(args.ntt_log_n), {}).get("traffic_bytes_per_transform")
        extra["ntt"] = {
            "metric": "bn254_fr_ntt_elements_per_sec", "log_n": args.ntt_log_n,
            "value": nn / ntt_s, "ms_per_transform": round(ntt_s * 1e3, 4),
            "forward_plus_inverse_ms": round(rt_s * 1e3, 4),
            "kernel_ms_per_transform": round(kern_ms, 4), "roundtrip_bit_exact": roundtrip_ok,
            # per transform = back-to-back pass kernels: priced on the un-instrumented wall time per
            # transform (the event-bracketed kernel durations below do not overlap their ramps and sum to more)
            "roofline": {"bound": "hbm", "achieved": round(64.0 * nn / ntt_s / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(64.0 * nn / ntt_s / 1e9 / HBM_PEAK_GBS, 5),
                         "hbm_read_frac": round((32.0 * nn / (HBM_PEAK_GBS * 1e9)) / ntt_s, 5),
                         "traffic": ntt_traffic},
            "kernels": {k: {"launches": v[0], "avg_ms": round(v[1] / max(v[0], 1), 4)} for k, v in sorted(p2.items())},
        }
        sq_ntt, _ = _load_counters("sq_counters.json")
        if sq_ntt is not None and "ntt_pass" in sq_ntt:
            extra["ntt"]["sq_counters"] = sq_ntt["ntt_pass"]
        if not args.no_extras:
            # the sizes below the headline: 2^17 .. 2^21 run in two passes of 9 .. 11 bits, 2^22 keeps three
            sizes = {}
            for lg in (18, 20, 21):
                m = 1 << lg
                b.ntt_device(x.data_ptr(), y.data_ptr(), m, sync=True)
                t1 = time.perf_counter()
                for _ in range(100):
                    b.ntt_device(x.data_ptr(), y.data_ptr(), m)
                b.sync()
                sizes[f"2^{lg}"] = round((time.perf_counter() - t1) / 100 * 1e3, 4)
            extra["ntt"]["ms_per_transform_other_sizes"] = sizes

    # ---- sustained shader clock under this workload (rank 0, N = 1 only) --------------------------
    # The MAD peak above was measured with sub-millisecond kernels at the boost clock; the 18 ms
    # accumulation kernel runs the package into its power limit and the clock drops.  Sample
    # rocm-smi while the same MSM repeats for ~2 s and restate the ALU fraction at that clock.
    if solo:
        try:
            clk = _sample_clocks(lambda: b.msm_device(srs, sc.data_ptr(), n), b.sync)
            if clk:
                roofline["alu"]["sustained_sclk_mhz"] = clk["sclk_mhz"]
                roofline["alu"]["sustained_power_w"] = clk["power_w"]
                peak_s = _issue_peak_t(clk["sclk_mhz"])
                roofline["alu"]["peak_at_sustained_clock"] = round(peak_s, 2)
                if alu_achieved:
                    roofline["alu"]["frac_at_sustained_clock"] = round(alu_achieved / peak_s, 4)
        except Exception as e:
            roofline["alu"]["clock_probe_error"] = str(e)

    # ---- the other rows of BASELINE.md section 5 and section 4's set (B), end-to-end timings (rank 0, N = 1 only) ----
    if solo:
        def time_msm(handle, d_sc, m, reps=5):
            b.msm_device(handle, d_sc, m)
            b.sync()
            t = time.perf_counter()
            for _ in range(reps):
                r = b.msm_device(handle, d_sc, m)
            b.sync()
            return (time.perf_counter() - t) / reps, r
        try:     # scalar set (B): the prover-like mix, same points, same size
            sc_mix = torch.empty((n, 4), dtype=torch.int64, device=dev)
            b.synth_scalars_mix(sc_mix.data_ptr(), n, seed ^ 0x5CA1AB1E)
            s_mix, _ = time_msm(srs, sc_mix.data_ptr(), n)
            extra["msm_prover_mix"] = {"what": "scalar set (B): 50 % zero, 20 % one, 10 % r-1, 10 % < 2^16, 10 % uniform (placeholder proportions)",
                                       "log_n": n.bit_length() - 1, "ms_per_msm": round(s_mix * 1e3, 4), "points_per_sec": n / s_mix}
            del sc_mix
        except Exception as e:
            extra["msm_prover_mix"] = {"error": str(e)}
        try:     # 2^20 (config #2's size): a prefix of the same workload
            if n >= (1 << 20):
                s20, _ = time_msm(srs, sc.data_ptr(), 1 << 20, reps=10)
                extra["msm_2p20"] = {"ms_per_msm": round(s20 * 1e3, 4), "points_per_sec": (1 << 20) / s20}
                # the window table at this size (uzk_srs_precompute): where does it stop paying?  (gpu.rs registers it up to 2^15 bases)
                srs20 = b.Srs.from_device(pts.data_ptr(), 1 << 20)
                try:
                    for cbits in (20, 22):
                        srs20.precompute(cbits)
                        s20t, r20t = time_msm(srs20, sc.data_ptr(), 1 << 20, reps=10)
                        extra["msm_2p20"][f"window_table_c{cbits}_ms"] = round(s20t * 1e3, 4)
                finally:
                    srs20.release()
        except Exception as e:
            extra["msm_2p20"] = {"error": str(e)}
        try:     # 2^26 points on this one GPU (BASELINE config #5's total size; the N > 1 runs shard it): chunks of 2^24 into one bucket set
            if n == (1 << 24) and world == 1:
                n26 = 1 << 26
                pts26 = torch.empty((n26, 8), dtype=torch.int64, device=dev)
                sc26 = torch.empty((n26, 4), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                b.synth_points_random(pts26.data_ptr(), n26, seed ^ 0x26)
                b.synth_scalars(sc26.data_ptr(), n26, seed ^ 0x2626)
                srs26 = b.Srs.from_device(pts26.data_ptr(), n26)
                try:
                    s26, _ = time_msm(srs26, sc26.data_ptr(), n26, reps=2)
                    extra["msm_2p26"] = {"ms_per_msm": round(s26 * 1e3, 3), "points_per_sec": n26 / s26,
                                         "what": "2^26 random points + uniform scalars resident on ONE GPU: four chunks of 2^24 into one bucket set (msm_run_chunked)"}
                finally:
                    srs26.release()
                    del pts26, sc26
                    torch.cuda.empty_cache()
        except Exception as e:
            extra["msm_2p26"] = {"error": str(e)}
        try:     # PCIe-inclusive: the host-pointer entry points (scalars / vector cross PCIe inside the call)
            hs = sc.cpu().numpy().view(np.uint64).reshape(-1, 4)
            def host_msm_ms(reps=3):
                b.msm(srs, hs)                       # workspaces of this path
                t = time.perf_counter()
                for _ in range(reps):
                    b.msm(srs, hs)
                return (time.perf_counter() - t) / reps
            pc_s = host_msm_ms()
            b.tune("msm_stream_log", -1)
            plain_s = host_msm_ms(2)
            b.tune("msm_stream_log", 0)
            ent = {"msm_host_scalars_ms": round(pc_s * 1e3, 3), "msm_points_per_sec": n / pc_s,
                   "msm_host_scalars_upload_then_compute_ms": round(plain_s * 1e3, 3),
                   "what": "uzk_msm_g1 / uzk_ntt_fr with pageable host buffers: upload + compute + download inside the call; the MSM "
                           "streams its scalars in point chunks under the previous chunk's accumulation (one bucket set, one reduction), "
                           "upload_then_compute = the same with streaming switched off; one 2^22 transform moves 128 MiB each way, which "
                           "cannot overlap for a single vector: its floor is two PCIe transfers"}
            if "ntt" in extra:
                hx = np.ascontiguousarray(x.cpu().numpy().view(np.uint64).reshape(-1, 4))
                b.ntt_inplace(hx)                    # uzk_ntt_fr transforms the caller's vector in place
                t = time.perf_counter()
                b.ntt_inplace(hx, inverse=True)
                b.ntt_inplace(hx)
                ent["ntt_host_vector_ms"] = round((time.perf_counter() - t) * 1e3 / 2, 3)
                del hx
            extra["pcie_inclusive"] = ent
            del hs
        except Exception as e:
            extra["pcie_inclusive"] = {"error": str(e)}

    # ---- the single-process sharded MSM of the C ABI (uzk_msm_g1_sharded: what ONE Rust process with several GPUs links) ----------
    if solo:
        try:
            m22 = min(n, 1 << 22)
            hp22 = pts[:m22].cpu().numpy().view(np.uint64).reshape(-1, 8)
            hs22 = sc[:m22].cpu().numpy().view(np.uint64).reshape(-1, 4)
            one22 = b.Srs.from_host(hp22)

            def wall(fn, reps=5):
                fn(); fn()
                t = time.perf_counter()
                for _ in range(reps):
                    r = fn()
                return (time.perf_counter() - t) / reps, r
            # the same plan on both sides: a 2^22-point vector is streamed in point chunks by uzk_msm_g1, its 2^21-point halves are
            # not (msm_stream_min_log) -- with streaming off everywhere the comparison is the sharding alone
            b.tune("msm_stream_log", -1)
            single_s, single_r = wall(lambda: b.msm(one22, hs22))
            b.tune("msm_stream_log", 0)
            single_streamed_s, _ = wall(lambda: b.msm(one22, hs22))
            ent = {"what": "uzk_msm_g1_sharded at 2^22 points, host scalars (pageable), VIRTUAL shards -- every chunk on device 0, a context, a stream and a "
                           "persistent host thread each, host fold of the 96-byte partials -- against uzk_msm_g1 over the same vector on one "
                           "context (single_streamed_ms: the call as shipped, scalars streamed under the accumulation; single_ms: streaming off).  "
                           "Every chunk streams its own scalars.  shards_N_overhead = shards_N_ms / single_streamed_ms - 1: on ONE GPU the chunks "
                           "share one PCIe link and one chip, so every chunk's first sub-chunk upload is exposed and the per-chunk fixed work (bucket "
                           "reduction, sort set-up) is serialised -- an upper bound of what N devices with a link each would see; the sharded "
                           "form's own cost (hand-over to persistent threads, host fold of 96-byte partials) is microseconds",
                   "log_n": m22.bit_length() - 1, "single_ms": round(single_s * 1e3, 3), "single_streamed_ms": round(single_streamed_s * 1e3, 3)}
            for chunks in (2, 8):
                sh = b.ShardedSrs(hp22, [0] * chunks, -1)
                try:
                    sh_s, sh_r = wall(lambda: sh.msm(hs22))
                    ent[f"shards_{chunks}_ms"] = round(sh_s * 1e3, 3)
                    ent[f"shards_{chunks}_overhead"] = round(sh_s / single_streamed_s - 1.0, 4)
                    ent[f"shards_{chunks}_vs_upload_then_compute"] = round(sh_s / single_s - 1.0, 4)
                    ent[f"shards_{chunks}_same_commitment"] = bool(np.array_equal(b.g1_to_affine(sh_r), b.g1_to_affine(single_r)))
                finally:
                    sh.release()
            one22.release()
            extra["msm_sharded_c_abi"] = ent
            del hp22, hs22
        except Exception as e:
            extra["msm_sharded_c_abi"] = {"error": str(e)[-300:]}
        finally:
            b.tune("msm_stream_log", 0)

    # ---- opt-in window-table mode and the real prover's call mix (rank 0, N = 1 only) ----------
    if solo:
        try:
            srs.precompute(20 if n >= (1 << 22) else 0)
            b.msm_device(srs, sc.data_ptr(), n)
            b.sync()
            t2 = time.perf_counter()
            for _ in range(3):
                pre_res = b.msm_device(srs, sc.data_ptr(), n)
            b.sync()
            pre_s = (time.perf_counter() - t2) / 3
            same = bool(np.array_equal(b.g1_to_affine(pre_res), b.g1_to_affine(result)))
            extra["msm_precomputed"] = {"ms_per_msm": round(pre_s * 1e3, 4), "points_per_sec": n / pre_s,
                                        "window_bits": 20 if n >= (1 << 22) else "auto",
                                        "same_commitment_as_general_mode": same,
                                        "note": "uzk_srs_precompute: window table resident in HBM (opt-in; not the headline)"}
        except Exception as e:   # e.g. not enough HBM for the table at a larger --log-n
            extra["msm_precomputed"] = {"error": str(e)}
        finally:
            b.tune("msm_no_precompute", 1)      # the parity sample below runs the general mode again
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import prover_shape
            extra["prover_shape"] = prover_shape.run(reps=3)
            extra["prover_shape"]["with_window_table"] = prover_shape.run(reps=3, precompute=True)
            extra["prover_shape"]["what"] = ("one 52-card proof's hot-path calls (n = 2^14: 16 MSM, 17 NTT(n), 11 NTT(6n)), "
                                             "device-resident data: call by call vs batched entry points, ms; with_window_table = "
                                             "the same after uzk_srs_precompute (static SRS)")
        except Exception as e:
            extra["prover_shape"] = {"error": str(e)}
        try:     # the checked chain of tests/test_gpu_prover_chain.py, timed (stand-in for config #4's prove time)
            import prover_chain
            ch = prover_chain.ProverChain()
            for _ in range(3):
                ch.run()
            b.sync()
            t3 = time.perf_counter()
            for _ in range(5):
                ch.run()
            b.sync()
            chain_ms = (time.perf_counter() - t3) / 5 * 1e3
            b.profile_reset(); b.profile_enable(True); ch.run(); b.sync(); b.profile_enable(False)
            ktab = b.profile_table()
            ch.release()
            extra["prover_chain"] = {"what": "every device-resident step of one 52-card proof chained with the reference's call mix "
                                             "(tools/prover_chain.py: 9+1 iFFT, 8+1+5+2 commits with blinds, z_poly, 10 coset FFT(6n), quotient "
                                             "kernel, coset iFFT, split_t with chunk n+2, 19 evaluations, r_poly over 43 polynomials, openings "
                                             "of 16 and 4 polynomials), synthetic circuit, Python (ctypes) glue included",
                                     "ms_per_chain": round(chain_ms, 3),
                                     "device_kernel_ms": round(sum(ms for k, (cnt, ms) in ktab.items() if not k.startswith("host_")), 3)}
        except Exception as e:
            extra["prover_chain"] = {"error": str(e)}
        try:     # the same chain issued from compiled host code (tests/cpp/prover_rounds.cpp, built by __graft_entry__.build())
            lines = _run_prover_rounds([20, 4, 1, "private", 0])
            extra["prover_rounds_cpp"] = dict(lines[0], what="proofs through uzk_circuit_create / uzk_prove_round1..5 from C++ (plain g++, the C ABI only).  First "
                                              "the latency of ONE proof on a prover that owns its lane: median of five timed blocks after 0.5 s of warm-up; "
                                              "ms_per_chain = witness resident, ms_per_chain_with_witness_upload = the 8n witness elements uploaded from pinned "
                                              "memory every proof.  Then throughput, where EVERY proof has its own witness, public inputs, blinds, challenges and "
                                              "r_poly scalars, its witness is uploaded from pinned memory at the start of every proof, and every thread's proofs are "
                                              "compared with the single-threaded proof of the same inputs: shared_N = N host threads, each calling round 1..5 on its own "
                                              "prover of one proof from the default context -- the reference's call pattern (prover.rs:88-100, sdk.rs:196-214), the "
                                              "library shares the rounds (uzk_coalesce_config defaults; proofs_per_shared_round says how well); shared_32_skewed = the "
                                              "same on the witness classes of SURVEY F7 (50 % zero, 20 % one, 10 % minus one, 10 % < 2^16, 10 % uniform); lockstep8 = the "
                                              "explicit API, four threads x eight witnesses per uzk_prove_round call (uzk_prover_create(n, 8)), and lockstep8_skewed; "
                                              "private_4 = four threads, each a prover that owns its lane on its own context.  proofs_per_s_total = shared_32: what a "
                                              "caller of prover_with_lagrange gets.")
            if len(lines) > 1:
                extra["prover_rounds_cpp"]["private_4"] = lines[1]
            for key, a in (("shared_8", [10, 8, 1, "shared", 0]), ("shared_16", [10, 16, 1, "shared", 0]), ("shared_32", [10, 32, 1, "shared", 0]),
                           ("shared_32_skewed", [10, 32, 1, "shared", 1]), ("lockstep8", [10, 4, 8, "lockstep", 0]), ("lockstep8_skewed", [10, 4, 8, "lockstep", 1])):
                # clock and socket power while the throughput section runs (the last 40 % of the driver's run: after its single-proof
                # part and the threads' warm-up): joules per proof beside proofs per second
                smp = _ClockSampler(dev_index).start() if key in ("shared_32", "lockstep8") else None
                t_a = time.perf_counter()
                res_k = _run_prover_rounds(a)[-1]
                t_b = time.perf_counter()
                if smp is not None:
                    st_k = smp.stop(t_a + 0.6 * (t_b - t_a), t_b)
                    if st_k and st_k.get("power_w_timed") and res_k.get("proofs_per_s"):
                        res_k["sclk_mhz"] = st_k["sclk_mhz_timed"]
                        res_k["socket_power_w"] = st_k["power_w_timed"]
                        res_k["joules_per_proof_socket"] = round(st_k["power_w_timed"] / res_k["proofs_per_s"], 4)
                extra["prover_rounds_cpp"][key] = res_k
            extra["proofs_per_s_total"] = extra["prover_rounds_cpp"]["shared_32"]["proofs_per_s"]
        except Exception as e:
            extra["prover_rounds_cpp"] = {"error": str(e)}
        try:     # the per-game refresh of the twelve public-key tables (params.rs:88-121) as one device call
            import prover_chain
            inp = prover_chain.ChainInputs(1 << 14, 11)
            cir = b.Circuit(inp.n, inp.lagrange_wire, inp.bases[inp.n:], inp.perm, inp.k, inp.anemoi_g, inp.anemoi_g_inv, inp.edwards_a,
                            [inp.table_polys[i] for i in range(prover_chain.N_TABLES)], precompute=True)
            ev = np.ascontiguousarray(inp.table_polys[prover_chain.T_QPK:prover_chain.T_QPK + 12])
            res = {}
            for name, kw in (("device_only", dict(want_polys=False, want_coset=False)), ("with_polys_download", dict(want_polys=True, want_coset=False)),
                             ("with_polys_and_coset_download", dict(want_polys=True, want_coset=True))):
                cir.refresh_tables(b.CS_QPK, ev, **kw)
                t4 = time.perf_counter()
                for _ in range(5):
                    cir.refresh_tables(b.CS_QPK, ev, **kw)
                res[name + "_ms"] = round((time.perf_counter() - t4) / 5 * 1e3, 3)
            b.profile_reset(); b.profile_enable(True); cir.refresh_tables(b.CS_QPK, ev, want_polys=False); b.sync(); b.profile_enable(False)
            res["device_kernel_ms"] = round(sum(ms for k, (cnt, ms) in b.profile_table().items() if not k.startswith("host_")), 3)
            cir.release()
            extra["public_key_refresh"] = dict(res, what="uzk_circuit_refresh_tables at n = 2^14: 12 evaluation vectors uploaded (6 MiB) -> batched iFFT(n) -> "
                                               "batched coset FFT(6n) -> batched Lagrange commit, the circuit's tables replaced (copy on write); wall ms per call "
                                               "incl. the upload; refresh_prover_params_public_key runs this loop call by call on the CPU once per game")
        except Exception as e:
            extra["public_key_refresh"] = {"error": str(e)}

    # ---- CPU baseline + parity on a bounded sample (rank 0; under N > 1 the other ranks wait at the closing barrier) ----------
    cpu_baseline = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c as oc   # checker / reported baseline only

        # N = 1: the whole default workload (a few seconds on the node's host cores); N > 1: the first 2^22 points (about a second)
        m = min(n, 1 << 24) if world == 1 else min(n, 1 << 22)
        hp = pts[:m].cpu().numpy().view(np.uint64).reshape(-1, 8)
        hs = sc[:m].cpu().numpy().view(np.uint64).reshape(-1, 4)
        cores = _cpu_threads(world)
        ref, cpu_baseline = _cpu_baseline_msm(oc, hp, hs, cores, world)
        got = b.msm_device(srs, sc.data_ptr(), m)
        cpu_baseline["gpu_matches_cpu_on_sample"] = oc.jac_to_affine_ints(ref) == oc.jac_to_affine_ints(got)
        if "ntt" in extra:
            nn = 1 << args.ntt_log_n
            hx = x.cpu().numpy().view(np.uint64).reshape(-1, 4)
            tc = time.perf_counter()
            ref_ntt = oc.ntt(hx, threads=cores)
            ntt_cpu_s = time.perf_counter() - tc
            b.ntt_device(x.data_ptr(), y.data_ptr(), nn, sync=True)
            extra["ntt"]["cpu_baseline"] = {
                "value": nn / ntt_cpu_s, "unit": "elements/s", "cores": cores, "kind": "port",
                "sample": f"one full 2^{args.ntt_log_n} forward transform", "seconds": round(ntt_cpu_s, 3),
                "gpu_matches_cpu": bool(np.array_equal(y.cpu().numpy().view(np.uint64).reshape(-1, 4), ref_ntt)),
            }

        # ---- CPU baseline of the third metric (prove ms at n = 2^14): the MSM + NTT part of ONE proof on the C port -------------------
        if isinstance(extra.get("prover_rounds_cpp"), dict) and "error" not in extra["prover_rounds_cpp"]:
            try:
                extra["prover_rounds_cpp"]["cpu_baseline"] = _cpu_baseline_proof(oc, np, cores)
            except Exception as e:
                extra["prover_rounds_cpp"]["cpu_baseline"] = {"error": str(e)[-300:]}

    if rank == 0:
        lg = n.bit_length() - 1 if n & (n - 1) == 0 else None
        per = f"2^{lg}" if lg is not None else str(n)
        line = {
            "metric": "bn254_g1_msm_points_per_sec", "value": value, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u256 (254-bit modular integers: 8 x u32 Montgomery limbs in HBM, 9 x 29-bit limbs in the two hot loops)",
            "data": "synthetic",
            "config": {"workload": f"BN254 G1 MSM, {per} {args.points} points + {args.scalars} scalars per GPU, "
                                   f"bases resident in HBM" + (f" (2^{args.total_log_n} points in total)" if strong else ""),
                       "points_per_gpu": n, "total_points": n * world,
                       "sharding": "point-chunk per rank, all-gather of 96-byte partial sums, host fold" if world > 1 else "single GPU"},
            "dist_world_size": dist.get_world_size() if use_dist else 1,
            "collective_backend": (dist.get_backend() if use_dist else None), "host_cores": _host_cores(),
            "device_ids": dev_ids, "device_uuids": dev_uuids,
            # every rank's own view of the timed region: ms per step, of which its MSM and of which the exchange (the 96-byte
            # all-gather + the fold, including the wait for the slowest rank).  `ms_per_step` above is the maximum over ranks.
            "per_rank": per_rank,
            "exchange_ms_per_step_max": max(r["exchange_ms_per_step"] for r in per_rank),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extra": extra,
            "result_is_infinity": bool(not result[8:12].any()),
            "result_affine_sha256": hashlib.sha256(np.ascontiguousarray(b.g1_to_affine(result)).tobytes()).hexdigest()[:16],
        }
        print(json.dumps(line))
    srs.release()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
