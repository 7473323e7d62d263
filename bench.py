#!/usr/bin/env python3
"""bench.py -- headline measurement for the MI355X uzkge backend.

Metric (BASELINE.json): BN254 G1 MSM points/s at 2^24 points per GPU (synthetic random points
and uniform scalars generated on device, resident in HBM before the timed region), plus the
Fr NTT at 2^22 reported under "extra".  One step = one MSM over this rank's 2^24 points; with
N > 1 ranks (one process per GPU, torch.distributed / RCCL) every rank owns its own point chunk
(weak scaling, a 2^24*N-point MSM in total), the 96-byte partial sums are all-gathered and folded
on every rank -- EC addition is not an RCCL reduce op.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (msm_accumulate):
achieved = 96 B/point (32 B scalar + 64 B affine base) x points per launch / the
kernel's mean duration, measured live with HIP events on the library stream during the timed
steps.  `cpu_baseline` times the CPU oracle (own C restatement, NOT arkworks) on a bounded sample
of the same workload on this node's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MAD_PEAK_T = 30.4        # T lane-MAD/s, measured (tools/microbench/int_rates.hip, profiles/r01i_microbench.txt)
MADS_PER_MIXED_ADD = 1467
CYC_PER_MAD = 5.17        # issue cycles per wave instruction at 4 waves/SIMD (int_rates.hip)
CYC_PER_OTHER_VALU = 3.4   # mean over the loop's other VALU instructions (v_and 2.7, v_lshrrev_b64 4.25, v_mul_lo 4.46, ...)
OTHER_VALU_PER_MIXED_ADD = 586   # tools/isa_hist.py on the accumulation loop; plus 176 s_nop after asm statements
NUM_SIMDS = 1024
MAD_PEAK_SCLK_MHZ = 2390  # shader clock during that sub-millisecond microbenchmark (rocm-smi: 2388-2393 MHz)


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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=24, help="log2 MSM points per GPU")
    ap.add_argument("--ntt-log-n", type=int, default=22)
    ap.add_argument("--points", choices=["random", "arith"], default="random")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--no-extras", action="store_true", help="skip the window-table and prover-shape extras")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # UZK_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    # (ranks then share a device and exchange their partials through host memory); the driver's runs
    # use the default: one rank per GPU over RCCL.
    backend = os.environ.get("UZK_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world > 1:
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

    n = 1 << args.log_n
    seed = 0x755A6B67655F6D73 + rank   # documented SplitMix64 seed, per-rank chunk
    pts = torch.empty((n, 8), dtype=torch.int64, device=dev)
    sc = torch.empty((n, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if args.points == "random":
        b.synth_points_random(pts.data_ptr(), n, seed)
    else:
        import ctypes  # noqa: F401
        k = np.array([seed & 0xFFFFFFFFFFFFFFFF, 1, 0, 0], dtype=np.uint64)
        b.synth_points_arith(pts.data_ptr(), n, k)
    b.synth_scalars(sc.data_ptr(), n, seed ^ 0x5CA1AB1E)
    srs = b.Srs.from_device(pts.data_ptr(), n)
    gather_in = torch.zeros(96, dtype=torch.uint8, device=coll_dev)
    gather_out = torch.zeros(96 * world, dtype=torch.uint8, device=coll_dev)

    def step():
        part = b.msm_device(srs, sc.data_ptr(), n)
        if world == 1:
            return part
        gather_in.copy_(torch.from_numpy(part.view(np.uint8)))
        dist.all_gather_into_tensor(gather_out, gather_in)
        allp = gather_out.cpu().numpy().view(np.uint64).reshape(world, 12)
        return b.g1_fold(allp)

    def fence():
        b.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        b.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    b.profile_reset()
    b.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    result = None
    for _ in range(args.steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    b.profile_enable(False)
    prof = b.profile_table()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed

    acc_cnt, acc_ms = prof.get("msm_accumulate", (0, 0.0))
    acc_avg_ms = acc_ms / max(acc_cnt, 1)
    achieved = (96.0 * n) / (acc_avg_ms * 1e-3) / 1e9 if acc_cnt else 0.0
    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), valid for
    # the configuration they were collected on (same kernel, same log_n, default window bits)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        ent = pmc.get("msm_accumulate", {}).get(str(args.log_n))
        if ent and args.points == "random" and not args.window_bits:
            traffic = ent["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        traffic = None
    # The kernel's real bound is the 64-bit multiply-add pipe: every mixed addition is 1467
    # v_mad_u64_u32 per lane (ec29.hpp: 6 products, 2 squarings, 1 dual product on 9 x 29-bit limbs)
    # against the measured issue peak of that instruction (tools/microbench/int_rates.hip:
    # 5.17 cycles per wave instruction per SIMD at 4 waves/SIMD = 30.4 T lane-MAD/s chip-wide).
    cbits, nwin = (args.window_bits, 0)
    try:
        cbits, nwin = b.msm_plan_info(n)
    except Exception:
        pass
    mads = float(MADS_PER_MIXED_ADD) * n * nwin
    alu_achieved = mads / (acc_avg_ms * 1e-3) / 1e12 if acc_cnt and nwin else 0.0
    roofline = {
        "bound": "hbm", "kernel": "msm_accumulate", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
        "avg_launch_ms": round(acc_avg_ms, 4), "algorithmic_bytes_per_launch": 96 * n,
        "alu": {"unit": "T lane-MAD/s (v_mad_u64_u32)", "achieved": round(alu_achieved, 3), "peak": MAD_PEAK_T,
                "frac": round(alu_achieved / MAD_PEAK_T, 4), "mads_per_mixed_add": MADS_PER_MIXED_ADD,
                "mixed_adds_per_launch": n * nwin, "window_bits": cbits, "windows": nwin},
        "note": "integer-ALU bound (254-bit modular arithmetic on v_mad_u64_u32), not HBM bound: `alu` is the "
                "binding roofline, `frac` the HBM one the contract asks for; see DESIGN.md 3.1",
    }
    # device kernels (HIP events) and, prefixed host_, the host-side sections of the call (wall clock)
    kernels = {k: {"launches": v[0], "avg_ms": round(v[1] / max(v[0], 1), 4)} for k, v in sorted(prof.items())}

    extra = {"msm_kernels": kernels}

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
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                ntt_traffic = json.load(f).get("ntt_pass", {}).get(str(args.ntt_log_n), {}).get("traffic_bytes_per_transform")
        except (OSError, ValueError):
            ntt_traffic = None
        extra["ntt"] = {
            "metric": "bn254_fr_ntt_elements_per_sec", "log_n": args.ntt_log_n,
            "value": nn / ntt_s, "ms_per_transform": round(ntt_s * 1e3, 4),
            "kernel_ms_per_transform": round(kern_ms, 4), "roundtrip_bit_exact": roundtrip_ok,
            # per transform = three back-to-back pass kernels: priced on the un-instrumented wall time per
            # transform (the event-bracketed kernel durations below do not overlap their ramps and sum to more)
            "roofline": {"bound": "hbm", "achieved": round(64.0 * nn / ntt_s / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(64.0 * nn / ntt_s / 1e9 / HBM_PEAK_GBS, 5),
                         "hbm_read_frac": round((32.0 * nn / (HBM_PEAK_GBS * 1e9)) / ntt_s, 5),
                         "traffic": ntt_traffic},
            "kernels": {k: {"launches": v[0], "avg_ms": round(v[1] / max(v[0], 1), 4)} for k, v in sorted(p2.items())},
        }

    # ---- sustained shader clock under this workload (rank 0, N = 1 only) --------------------------
    # The MAD peak above was measured with sub-millisecond kernels at the boost clock; the 18 ms
    # accumulation kernel runs the package into its power limit and the clock drops.  Sample
    # rocm-smi while the same MSM repeats for ~2 s and restate the ALU fraction at that clock.
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            clk = _sample_clocks(lambda: b.msm_device(srs, sc.data_ptr(), n), b.sync)
            if clk:
                roofline["alu"]["sustained_sclk_mhz"] = clk["sclk_mhz"]
                roofline["alu"]["sustained_power_w"] = clk["power_w"]
                roofline["alu"]["boost_sclk_mhz"] = MAD_PEAK_SCLK_MHZ
                peak_s = MAD_PEAK_T * clk["sclk_mhz"] / MAD_PEAK_SCLK_MHZ
                roofline["alu"]["peak_at_sustained_clock"] = round(peak_s, 2)
                roofline["alu"]["frac_at_sustained_clock"] = round(alu_achieved / peak_s, 4) if peak_s else None
                # whole instruction stream of the loop (ISA count: 1467 MADs + 586 other VALU instructions + 176
                # s_nop per addition) priced at the measured issue costs, against the SIMD cycles the launch had
                need = (MADS_PER_MIXED_ADD * CYC_PER_MAD + OTHER_VALU_PER_MIXED_ADD * CYC_PER_OTHER_VALU + 176) * (n * nwin / 64.0)
                have = acc_avg_ms * 1e-3 * NUM_SIMDS * clk["sclk_mhz"] * 1e6
                roofline["alu"]["valu_issue_frac_at_sustained_clock"] = round(need / have, 4) if have else None
        except Exception as e:
            roofline["alu"]["clock_probe_error"] = str(e)

    # ---- opt-in window-table mode and the real prover's call mix (rank 0, N = 1 only) ----------
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            srs.precompute(20 if args.log_n >= 22 else 0)
            b.msm_device(srs, sc.data_ptr(), n)
            b.sync()
            t2 = time.perf_counter()
            for _ in range(3):
                pre_res = b.msm_device(srs, sc.data_ptr(), n)
            b.sync()
            pre_s = (time.perf_counter() - t2) / 3
            same = bool(np.array_equal(b.g1_to_affine(pre_res), b.g1_to_affine(result)))
            extra["msm_precomputed"] = {"ms_per_msm": round(pre_s * 1e3, 4), "points_per_sec": n / pre_s,
                                        "window_bits": 20 if args.log_n >= 22 else "auto",
                                        "same_commitment_as_general_mode": same,
                                        "note": "uzk_srs_precompute: window table resident in HBM (opt-in; not the headline)"}
        except Exception as e:   # e.g. not enough HBM for the table at a larger --log-n
            extra["msm_precomputed"] = {"error": str(e)}
        finally:
            b.tune("msm_no_precompute", 1)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import prover_shape
            extra["prover_shape"] = prover_shape.run(reps=3)
            extra["prover_shape"]["what"] = ("one 52-card proof's hot-path calls (n = 2^14: 16 MSM, 17 NTT(n), 11 NTT(6n)), "
                                             "device-resident data: call by call vs batched entry points, ms")
        except Exception as e:
            extra["prover_shape"] = {"error": str(e)}

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only) -----------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c as oc   # checker / reported baseline only

        m = min(n, 1 << 24)   # the whole default workload: about 6 s on 16 host threads
        hp = pts[:m].cpu().numpy().view(np.uint64).reshape(-1, 8)
        hs = sc[:m].cpu().numpy().view(np.uint64).reshape(-1, 4)
        cores = min(16, os.cpu_count() or 1)
        tc = time.perf_counter()
        ref = oc.msm_pippenger(hp, hs, 0, cores)
        cpu_s = time.perf_counter() - tc
        got = b.msm_device(srs, sc.data_ptr(), m)
        parity = oc.jac_to_affine_ints(ref) == oc.jac_to_affine_ints(got)
        cpu_baseline = {
            "value": m / cpu_s, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"first 2^{m.bit_length() - 1} points/scalars of the same workload, Pippenger in oracle/bn254_oracle.c "
                      f"(own CPU restatement, not arkworks)", "seconds": round(cpu_s, 3),
            "gpu_matches_cpu_on_sample": parity,
        }
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

    if rank == 0:
        line = {
            "metric": "bn254_g1_msm_points_per_sec", "value": value, "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (8 x u32 Montgomery limbs)", "data": "synthetic",
            "config": {"workload": f"BN254 G1 MSM, 2^{args.log_n} {args.points} points + uniform scalars per GPU, "
                                   f"bases resident in HBM", "points_per_gpu": n, "total_points": n * world,
                       "sharding": "point-chunk per rank, all-gather of 96-byte partial sums, host fold" if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extra": extra,
            "result_is_infinity": bool(not result[8:12].any()),
        }
        print(json.dumps(line))
    srs.release()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
