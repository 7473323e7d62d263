import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from uzkge_amd import backend as b
b.init(0)
n, m = 1 << 14, 98304
B = 40
buf = torch.empty((B * m, 4), dtype=torch.int64, device="cuda"); out = torch.empty((B * m, 4), dtype=torch.int64, device="cuda"); ref = torch.empty((B * m, 4), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
b.synth_scalars(buf.data_ptr(), B * m, 5)
kk = torch.empty((1, 4), dtype=torch.int64, device="cuda"); b.synth_scalars(kk.data_ptr(), 1, 77); k = kk.cpu().numpy().view(np.uint64).reshape(4)
cases = [("coset fft 98304 x10", lambda o: b.ntt_batch_device(buf.data_ptr(), o, m, 10, coset_shift=k), 10 * m),
         ("coset fft 98304 x12", lambda o: b.ntt_batch_device(buf.data_ptr(), o, m, 12, coset_shift=k), 12 * m),
         ("coset fft 98304 x40", lambda o: b.ntt_batch_device(buf.data_ptr(), o, m, 40, coset_shift=k), 40 * m),
         ("coset ifft 98304 x1", lambda o: b.ntt_device(buf.data_ptr(), o, m, inverse=True, coset_shift=k), m),
         ("ifft 2^14 x10", lambda o: b.ntt_batch_device(buf.data_ptr(), o, n, 10, inverse=True), 10 * n),
         ("ifft 2^14 x40", lambda o: b.ntt_batch_device(buf.data_ptr(), o, n, 40, inverse=True), 40 * n),
         ("fft 2^14 x64", lambda o: b.ntt_batch_device(buf.data_ptr(), o, n, 64), 64 * n),
         ("fft 2^18", lambda o: b.ntt_device(buf.data_ptr(), o, 1 << 18), 1 << 18),
         ("fft 2^19", lambda o: b.ntt_device(buf.data_ptr(), o, 1 << 19), 1 << 19),
         ("fft 2^20", lambda o: b.ntt_device(buf.data_ptr(), o, 1 << 20), 1 << 20),
         ("fft 2^21", lambda o: b.ntt_device(buf.data_ptr(), o, 1 << 21), 1 << 21),
         ("fft 2^22", lambda o: b.ntt_device(buf.data_ptr(), o, 1 << 21), 1 << 21)]
vals = [0, 2, 4, 6]
for name, fn, cnt in cases:
    b.tune("ntt_stagger", 0); fn(ref.data_ptr()); b.sync()
    t = {v: 1e9 for v in vals}
    for v in vals[1:]:
        b.tune("ntt_stagger", v); fn(out.data_ptr()); b.sync()
        assert bool(torch.equal(out[:cnt], ref[:cnt])), (name, v)
    for rnd in range(5):
        for v in vals:
            b.tune("ntt_stagger", v); fn(out.data_ptr()); b.sync()
            t0 = time.perf_counter()
            for _ in range(40): fn(out.data_ptr())
            b.sync(); t[v] = min(t[v], (time.perf_counter() - t0) / 40)
    print(f"{name:22s} " + "  ".join(f"stagger={v}: {t[v]*1e6:7.1f} us" for v in vals), flush=True)
b.tune("ntt_stagger", 4)
