"""The training loop of highdim_pde/lambaem.jl on the CPU restatement (oracle): u0(x0) over the ADAM iterations.
TEST INFRASTRUCTURE (imports the oracle); reference run: u0 = 1.99 after 500 iterations, 3.85 after 1000, 4.59 after 1300."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import _sde_oracle as S
rng=np.random.default_rng(0)
th=S.glorot_params(100,110,rng)
D=S.desc(abstol=0.1,reltol=0.1,seed=0)
m=np.zeros_like(th); v=np.zeros_like(th); b1,b2=0.9,0.999; b1p,b2p=b1,b2
t0=time.time()
for it in range(4000):
    r=S.loss_grad(D,100,np.zeros(100),th,it=it,nthreads=8)
    g=r['grad']
    if it%25==0: print(it, "loss %.4f u0 %.4f steps %.0f t=%.0f"%(r['loss'], r['u0'], r['stats'][:,1].mean(), time.time()-t0), flush=True)
    m=b1*m+(1-b1)*g; v=b2*v+(1-b2)*g*g
    th=(th-(m/(1-b1p))/(np.sqrt(v/(1-b2p))+1e-8)*0.03).astype(np.float32)
    b1p*=b1; b2p*=b2
