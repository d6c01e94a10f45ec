import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import fmm_bem_relaxed_amd as fb
v = np.concatenate([fb.unit_sphere(9, center=(3.0 * i, 0.0, 0.0)) for i in range(2)])
x = torch.rand(len(v), dtype=torch.float64).cuda(); y = torch.empty_like(x)
for ov in ("0", "1", "2"):
    os.environ["FMMBEM_OVERLAP_NEAR"] = ov
    K = fb.LaplaceSphericalBEM(12, 3); plan = fb.FMM_plan(K, v, p_max=12)
    for p in (1, 2, 3, 4, 6, 10):
        for _ in range(3): plan.execute_torch(x, out=y, p=p)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.execute_torch(x, out=y, p=p)
        e1.record(); torch.cuda.synchronize()
        print("overlap", ov, "p", p, "matvec ms %.3f" % (e0.elapsed_time(e1) / 20), flush=True)
    plan.close()
