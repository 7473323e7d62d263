#!/usr/bin/env python3
"""Where the streams of a prover_rounds throughput run stand idle.

usage: rounds_stream_gaps.py <kernel_trace.csv> [memory_copy_trace.csv]

Reads rocprofv3's per-dispatch kernel trace (and, optionally, its memory-copy trace), keeps the middle half of the run (steady
state), and prints
  * per queue: the share of the window in which one of its kernels (or copies) was running,
  * the share of the window in which k = 0, 1, 2, ... kernels ran at once on the device,
  * the idle gaps between two consecutive operations of a queue, grouped by (operation before -> operation after): how many per
    1000 kernels, the mean gap and the share of that queue-time they make up -- a gap is host time (launch latency, a wait for a
    result, the host's part of a round) or the wait of the caller for its team.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").split("::")[-1][:34]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    print("columns:", ", ".join(rows[0].keys()))
    ops = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows]
    copies = []
    if len(sys.argv) > 2:
        for r in csv.DictReader(open(sys.argv[2])):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "copy"))))
    # prover_rounds first proves alone on the default context (the queue of the very first kernel); the throughput section runs on
    # the other queues (the threads' contexts, or the library's own for shared provers): keep that section only
    first_q = min(ops)[2]
    section = [o for o in ops if o[2] != first_q]
    if section:
        t0, t1 = min(o[0] for o in section), max(o[1] for o in section)
        ops = [o for o in ops if o[0] >= t0 and o[1] <= t1]
    t0 = min(o[0] for o in ops)
    t1 = max(o[1] for o in ops)
    lo, hi = t0 + (t1 - t0) // 4, t1 - (t1 - t0) // 4
    win = hi - lo
    ops = sorted(o for o in ops if o[0] >= lo and o[1] <= hi)
    print(f"window {win / 1e6:.1f} ms of {(t1 - t0) / 1e6:.1f} ms, {len(ops)} kernels in it")

    by_q = defaultdict(list)
    for o in ops:
        by_q[o[2]].append(o)
    print("\nqueue            kernels   busy share   mean kernel us")
    gaps = defaultdict(lambda: [0, 0])
    q_time = 0
    for q, lst in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in lst)
        span = lst[-1][1] - lst[0][0]
        q_time += span
        print(f"{q:>14s} {len(lst):9d}   {busy / span:9.3f}   {busy / len(lst) / 1e3:10.1f}")
        names = defaultdict(lambda: [0, 0])
        for s_, e_, _, nm in lst:
            names[nm][0] += 1
            names[nm][1] += e_ - s_
        print("                 " + "; ".join(f"{nm} x{c} {t / 1e6:.1f} ms" for nm, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:9]))
        for a, b in zip(lst, lst[1:]):
            g = b[0] - a[1]
            if g > 0:
                gaps[(a[3], b[3])][0] += 1
                gaps[(a[3], b[3])][1] += g

    # how many kernels at once
    ev = []
    for s, e, _, _ in ops:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, at = 0, lo, defaultdict(int)
    for t, d in ev:
        at[depth] += t - last
        last = t
        depth += d
    at[depth] += hi - last
    print("\nkernels running at once: share of the window")
    mean = 0.0
    for k in sorted(at):
        print(f"  {k}: {at[k] / win:.3f}")
        mean += k * at[k] / win
    print(f"  mean {mean:.2f}")

    total_gap = sum(v[1] for v in gaps.values())
    print(f"\nidle between consecutive kernels of a queue: {total_gap / q_time:.3f} of the queues' time; the largest contributors")
    print(f"{'kernel before':36s} {'kernel after':36s} {'per 1000 k':>10s} {'mean us':>9s} {'share':>7s}")
    for (a, b), (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{a:36s} {b:36s} {1000.0 * cnt / len(ops):10.2f} {tot / cnt / 1e3:9.1f} {tot / q_time:7.3f}")

    batches = sum(1 for o in ops if o[3].startswith("t_quotient")) or 1
    names = defaultdict(lambda: [0, 0])
    for s_, e_, _, nm in ops:
        names[nm][0] += 1
        names[nm][1] += e_ - s_
    total = sum(v[1] for v in names.values())
    print(f"\nper round-3 launch (one group of proofs; {batches} in the window): {len(ops) / batches:.1f} kernels, {total / batches / 1e3:.1f} us of kernel time")
    for nm, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {c / batches:6.2f} x {t / c / 1e3:8.1f} us = {t / batches / 1e3:8.1f} us  {t / total:6.3f}  {nm}")

    if copies:
        inwin = [c for c in copies if c[0] >= lo and c[1] <= hi]
        by_dir = defaultdict(lambda: [0, 0])
        for s, e, d in inwin:
            by_dir[d][0] += 1
            by_dir[d][1] += e - s
        print("\nmemory copies in the window (count, total ms, share of the window)")
        for d, (cnt, tot) in sorted(by_dir.items(), key=lambda kv: -kv[1][1]):
            print(f"  {d:28s} {cnt:7d} {tot / 1e6:9.2f} {tot / win:7.3f}")


if __name__ == "__main__":
    main()
