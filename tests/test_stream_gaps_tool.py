"""tools/rounds_stream_gaps.py on a hand-made rocprofv3 kernel trace: the numbers DESIGN.md 3.5 quotes come out of this script, so its
arithmetic (the throughput section's window, busy shares, kernels at once, gaps by kernel pair) is held to a case worked by hand."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_busy_shares_concurrency_and_gaps(tmp_path):
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp"]
    # queue 1: the single-proof section before anything else (must be left out of the window)
    rows.append("KERNEL_DISPATCH,0,1,uzk::warm_kernel(int),0,1000000")
    # queues 2 and 3: 40 periods of 100 us; queue 2 runs a(0..60) then idles 40, queue 3 runs b(50..100): overlap 10 per period
    for k in range(40):
        t = 10_000_000 + 100_000 * k
        rows.append(f"KERNEL_DISPATCH,0,2,void uzk::a_kernel<4>(int),{t},{t + 60_000}")
        rows.append(f"KERNEL_DISPATCH,0,3,uzk::t_quotient_split_kernel(int),{t + 50_000},{t + 100_000}")
    trace = tmp_path / "kernel_trace.csv"
    trace.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rounds_stream_gaps.py"), str(trace)], capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    assert not any("warm_kernel" in ln for ln in lines), out
    q2 = next(ln for ln in lines if ln.split()[:1] == ["2"])
    q3 = next(ln for ln in lines if ln.split()[:1] == ["3"])
    assert abs(float(q2.split()[2]) - 0.6) < 0.02 and abs(float(q3.split()[2]) - 0.5) < 0.02, out
    at = {int(ln.split(":")[0]): float(ln.split(":")[1]) for ln in lines if ln.strip()[:2] in ("0:", "1:", "2:")}
    assert abs(at[2] - 0.10) < 0.02 and abs(at[1] - 0.90) < 0.03 and at.get(0, 0.0) < 0.02, out
    gap = next(ln for ln in lines if ln.startswith("a_kernel<4>") and "a_kernel<4>" in ln[30:])
    assert abs(float(gap.split()[-2]) - 40.0) < 0.5, out            # 40 us between two a_kernels
    assert any("per round-3 launch" in ln for ln in lines)
