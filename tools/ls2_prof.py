"""cycle profile of one trip of seirls2::seir_ls2_adj_kernel (a library built with -DUDE_LS2_CLOCKS: UDE_LIB_VARIANT=ls2clk)"""
import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
import universal_differential_equations_amd as U
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
w = bench.synth_inputs_other('seir', N, 0, dev)
ens = U.DeviceEnsemble(w['f'], w['alg'], w['tspan'], w['t'], w['u0'], data=w['data'], row_mask=w['mask'], **w['tol'])
th = torch.tensor(w['theta'], dtype=torch.float64, device=dev)
ens.loss_grad(th); torch.cuda.synchronize()
ens.eng.set_trace(0, 64)
ens.loss_grad(th); torch.cuda.synchronize()
tr = ens.eng.get_trace().ravel()
names = ['A / B / C0 row phases', 'barrier 1', 'matrix phase (3 barriers inside) + barrier', 'factor copy + D row phase', 'barrier in front of E', 'E step-end passes', 'barrier behind E', 'F state machine']
tot = tr[:8].sum()
print('N', N, 'block 0 wave 0: trips', int(tr[8]), 'cycles/trip', tot / max(tr[8], 1))
for n_, v in zip(names, tr[:8]):
    print('  %-46s %10.0f cycles/trip  %5.1f %%' % (n_, v / max(tr[8], 1), 100 * v / tot))
print('kernel ms', ens.kernel_ms())
