#!/usr/bin/env python3
"""Register / LDS / scratch usage of every gfx950 kernel in libuzkge_gpu.so (reads the embedded code objects' notes).
usage: python tools/kernel_meta.py [filter]"""
import os, struct, subprocess, sys, tempfile, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = open(os.path.join(ROOT, "uzkge_amd", "libuzkge_gpu.so"), "rb").read()
pos, flt = 0, (sys.argv[1] if len(sys.argv) > 1 else "")
rows = []
while True:
    i = d.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
    if i < 0: break
    pos = i + 24
    n = struct.unpack_from("<Q", d, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", d, off); off += 24
        t = d[off:off + tl].decode(); off += tl
        if "gfx950" not in t or s == 0: continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(d[i + o:i + o + s]); f.flush()
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for ln in out.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
            if not m: continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and cur.get("name"): rows.append(cur); cur = {}
            if k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "agpr_count", "max_flat_workgroup_size"): cur[k] = v
        if cur.get("name"): rows.append(cur)
seen = set()
for r in rows:
    nm = r.get("name", "")
    if flt not in nm or nm in seen: continue
    seen.add(nm)
    short = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().split("(")[0][-60:]
    print(f"{short:60s} vgpr {r.get('vgpr_count','?'):>4} agpr {r.get('agpr_count','?'):>3} sgpr {r.get('sgpr_count','?'):>4} scratch {r.get('private_segment_fixed_size','?'):>5} lds {r.get('group_segment_fixed_size','?'):>6} spill {r.get('vgpr_spill_count','?')}")
