import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (R, os.path.join(R, "tests"), os.path.join(R, "oracle"), os.path.join(R, "tools")):
    sys.path.insert(0, p)
from uzkge_amd import backend as b
b.init(0)
import test_gpu_circuit_rounds as T
import test_gpu_contexts as C
t0 = time.time(); n = 0
for i in range(int(os.environ.get("ITERS", "12"))):
    T.test_provers_on_several_contexts_while_the_tables_are_being_swapped(b); n += 1
    T.test_tables_are_copy_on_write_for_a_proof_in_flight(b); n += 1
    print("iteration", i, "ok", round(time.time() - t0, 1), "s", flush=True)
print(n, "runs, no failure")
