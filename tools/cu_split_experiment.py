"""EXPERIMENT: the near field on k compute units of every 32 and the far field on the others, side by side (FMMBEM_CU_SPLIT=k,
hipExtStreamCreateWithCUMask).  usage: python tools/cu_split_experiment.py [p] [k ...]   -- one child process per setting."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
import fmm_bem_relaxed_amd as fb
p = int(sys.argv[1])
v = np.concatenate([fb.unit_sphere(9), fb.unit_sphere(9, center=(2.5, 0, 0))])
plan = fb.FMM_plan(fb.LaplaceSphericalBEM(p, 3), v, fb.FMMOptions())
x = torch.from_numpy(np.random.default_rng(1).standard_normal(len(v))).cuda()
y = torch.empty_like(x)
for _ in range(5): plan.execute_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream, p=p)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): plan.execute_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream, p=p)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 40 * 1e3
plan.set_timing(1)
for _ in range(10): plan.execute_device(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream, p=p)
torch.cuda.synchronize()
st = plan.stats()
np.save(sys.argv[2], y.cpu().numpy())
print("%%.3f ms per matvec; stage ms: near %%.3f p2m %%.3f m2m %%.3f m2l %%.3f l2l %%.3f l2p %%.3f" %% (ms, st["ms_near"], st["ms_p2m"], st["ms_m2m"], st["ms_m2l"], st["ms_l2l"], st["ms_l2p"]))
""" % ROOT


def run(p, k, out):
    env = dict(os.environ)
    env.pop("FMMBEM_CU_SPLIT", None)
    env.pop("FMMBEM_NEAR_CUS", None)
    if k:
        env["FMMBEM_CU_SPLIT"] = str(k)
        env["FMMBEM_NEAR_CUS"] = str(8 * k)
    r = subprocess.run([sys.executable, "-c", CHILD, str(p), out], env=env, capture_output=True, text=True)
    return (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]


if __name__ == "__main__":
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ks = [int(a) for a in sys.argv[2:]] or [8, 12, 16, 20, 24]
    print("p = %d" % p, flush=True)
    print("  serial (the product)            ", run(p, 0, "/tmp/y_base.npy"), flush=True)
    base = np.load("/tmp/y_base.npy")
    for k in ks:
        line = run(p, k, "/tmp/y_k.npy")
        same = os.path.exists("/tmp/y_k.npy") and np.array_equal(np.load("/tmp/y_k.npy"), base)
        print("  near on %2d of 32 CUs, far on %2d: " % (k, 32 - k), line, "| bitwise equal" if same else "| DIFFERS", flush=True)
