import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import fmm_bem_relaxed_amd as fb
from oracle import oracle as O
from conftest import drand48, rel_l2
def two():
    v = np.concatenate([fb.unit_sphere(5), fb.unit_sphere(5, center=(3.0, 0.0, 0.0))])
    K = fb.LaplaceSphericalBEM(10, 3)
    pl = fb.FMM_plan(K, v); o = O.Oracle(v)
    x = drand48(len(v), seed=11)
    y = pl.execute(x); print('two spheres', rel_l2(y,o.matvec(x,10)))
def shards():
    v = fb.unit_sphere(6)
    K = fb.LaplaceSphericalBEM(10, 3)
    x = drand48(len(v))
    fullp = fb.FMM_plan(K, v)
    full = fullp.execute(x)
    o=O.Oracle(v); print('full vs oracle', rel_l2(full,o.matvec(x,10)))
    for world in (2, 4):
        acc = np.zeros_like(full)
        for rank in range(world):
            pp = fb.FMM_plan(K, v, shard=(rank, world))
            part = pp.execute(x)
            part2 = pp.execute(x)
            print(world, rank, 'repeat equal', np.array_equal(part,part2))
            acc += part
        d=np.abs(acc-full); print(world,'equal',np.array_equal(acc,full),'max diff',d.max(),'n diff',(d>0).sum(), 'rel', rel_l2(acc,full))
two(); shards(); two()
