#!/usr/bin/env python3
"""Builds gpurun_out/<tag>_pmc_traffic.json and <tag>_sq_counters.json from the per-kernel PMC summaries of
tools/profile_round.sh, stamped with the hash of the kernel sources of THIS tree (bench.py ignores counter files whose
stamp differs from the sources it runs on).  usage: python tools/make_counter_files.py <tag>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out")


def table(name):
    rows = {}
    p = os.path.join(out, f"{tag}_pmc_{name}.csv")
    if not os.path.exists(p):
        return rows
    for r in csv.DictReader(open(p)):
        rows.setdefault(r["Kernel"].replace("uzk::", ""), {})[r["Counter"]] = (int(r["Launches"]), float(r["Mean_Value_Per_Launch"]))
    return rows


sha = bench._kernel_sources_sha()
fetch, write, sq, grbm = table("fetch_size"), table("write_size"), table("sq"), table("grbm")
KIB = 1024.0
traffic = {"_doc": "HBM traffic per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of `python3 bench.py "
                   "--steps 1 --warmup 0 --no-cpu-baseline --no-extras`; tools/profile_round.sh). FETCH_SIZE / WRITE_SIZE are KiB. gfx950 correction "
                   "(MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of a coalesced 16 B/lane stream -> x2 for the NTT passes; the bucket accumulation is "
                   "dominated by 64-byte gathers, for which the calibration of round 1 (tools/microbench/gather_calib.hip, "
                   "profiles/r01e_pmc_fetch_size_calibration.csv) measured a factor of 1.003 -> x1. WRITE_SIZE is exact.",
           "kernel_sources_sha": sha, "profile_tag": tag}
acc = "msm_accumulate29_kernel"
if acc in fetch and acc in write:
    f, w = fetch[acc]["FETCH_SIZE"][1], write[acc]["WRITE_SIZE"][1]
    traffic["msm_accumulate"] = {"24": {"fetch_size_kib": f, "write_size_kib": w, "fetch_factor": 1.0, "traffic_bytes": int((f + w) * KIB),
                                        "pippenger_gather_bytes": (1 << 24) * 15 * 68, "profile": f"profiles/{tag}_pmc_fetch_size.csv, profiles/{tag}_pmc_write_size.csv"}}
per = {}
for k in fetch:
    if k.startswith("ntt_pass29_kernel") and k in write:
        per[k] = (2.0 * fetch[k]["FETCH_SIZE"][1] + write[k]["WRITE_SIZE"][1]) * KIB
if per:
    # one 2^22 transform = first pass + two later passes (8 + 8 + 6): the <8,false> and <6,false> kernels run once each
    total = sum(per.values())
    traffic["ntt_pass"] = {"22": {"per_kernel_traffic_bytes": per, "fetch_factor": 2.0, "traffic_bytes_per_transform": int(total),
                                  "note": "coalesced 16 B/lane streams: FETCH_SIZE x2; one transform = the three pass kernels once each; algorithmic bytes 64 B/element = 268 MB"}}
json.dump(traffic, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)

sqj = {"_doc": "SQ / GRBM counters per launch (rocprofv3 --pmc, one pass per block; tools/profile_round.sh): means over the launches of the pass. "
               "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves (MI355X_MICROARCH.md); derived: "
               "valu_active_frac_of_wave_time = ACTIVE_INST_VALU / WAVE_CYCLES (share of a resident wave's time in which it is issuing vector "
               "instructions; with W waves per SIMD an issue-bound SIMD shows about 1 / W).  On this chip SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU for these "
               "kernels: every vector instruction -- v_mad_u64_u32 included -- occupies its SIMD's issue port for one quad-cycle (4 cycles), and a SIMD issues "
               "at most one per quad-cycle.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Hence valu_issue_util = 4 * ACTIVE_INST_VALU / 1024 SIMDs / "
               "(GRBM_GUI_ACTIVE / 8): the fraction of all vector issue slots of the launch that carried an instruction -- the binding roofline of both hot loops.",
       "kernel_sources_sha": sha, "profile_tag": tag}
for label, kern in (("msm_accumulate", "msm_accumulate29_kernel"), ("ntt_pass", None)):
    kerns = [kern] if kern else [k for k in sq if k.startswith("ntt_pass29_kernel")]
    for k in kerns:
        if k not in sq:
            continue
        c = {n: v[1] for n, v in sq[k].items()}
        g = {n: v[1] for n, v in grbm.get(k, {}).items()}
        ent = {"kernel": k, "launches": next(iter(sq[k].values()))[0], **{n: round(v, 1) for n, v in c.items()}, **{n: round(v, 1) for n, v in g.items()}}
        if c.get("SQ_WAVE_CYCLES"):
            ent["valu_active_frac_of_wave_time"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"], 4)
            ent["wait_inst_frac_of_wave_time"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
            ent["wait_any_frac_of_wave_time"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 4)
        if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
            ent["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        if g.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
            ent["valu_issue_util"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (g["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if label == "msm_accumulate":
            sqj[label] = ent
        else:
            sqj.setdefault(label, {})[k] = ent
json.dump(sqj, open(os.path.join(out, f"{tag}_sq_counters.json"), "w"), indent=1)
print(json.dumps({"traffic": {k: v for k, v in traffic.items() if not k.startswith("_")}, "sq": {k: v for k, v in sqj.items() if not k.startswith("_")}})[:3000])
