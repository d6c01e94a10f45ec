import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import fmm_bem_relaxed_amd as fb
from oracle import oracle as O
from conftest import drand48, rel_l2
v = np.concatenate([fb.unit_sphere(5), fb.unit_sphere(5, center=(3.0, 0.0, 0.0))])
K = fb.LaplaceSphericalBEM(10, 3)
pl = fb.FMM_plan(K, v); o = O.Oracle(v)
x = drand48(len(v), seed=11)
y = pl.execute(x); yo=o.matvec(x,10)
print('matvec', rel_l2(y,yo))
M,L=pl.expansions('M',10),pl.expansions('L',10); Mo,Lo=o.expansions(10,'M'),o.expansions(10,'L')
b=pl.boxes()
eM=np.abs(M[:,0]-Mo[:,0]).max(axis=1)/(np.abs(Mo[:,0]).max(axis=1)+1e-300)
eL=np.abs(L[:,0]-Lo[:,0]).max(axis=1)/(np.abs(Lo[:,0]).max(axis=1)+1e-300)
for lev in range(b['level'].max()+1):
    m=b['level']==lev
    print(lev, m.sum(), 'leafs',b['leaf'][m].sum(),'errM',eM[m].max(),'errL',eL[m].max())
bad=np.where(eM>1e-10)[0]; print('bad M boxes',bad[:10], b['leaf'][bad[:10]], b['level'][bad[:10]])
