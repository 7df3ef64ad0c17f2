"""cycle profile of one trip of seirlf::seir_lsf_adj_kernel (a library built with -DUDE_LSF_CLOCKS: UDE_LIB_VARIANT=lsfclk)"""
import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
import universal_differential_equations_amd as U
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
w = bench.synth_inputs_other('seir', N, 0, dev)
ens = U.DeviceEnsemble(w['f'], w['alg'], w['tspan'], w['t'], w['u0'], data=w['data'], row_mask=w['mask'], sensealg=U.FastInterpolatingAdjoint(), **w['tol'])
th = torch.tensor(w['theta'], dtype=torch.float64, device=dev)
ens.loss_grad(th); torch.cuda.synchronize()
ens.eng.set_trace(0, 64)
ens.loss_grad(th); torch.cuda.synchronize()
tr = ens.eng.get_trace().ravel()
names = ['A take + B stage state + bcast', 'C0 locate / interpolate / inputs', 'barrier 1 (pre-matrix)', 'L1 mfma + tanh + barrier 2', 'L2 mfma + tanh (+W2T fetch) + barrier 3',
         'L2T mfma + d1 + gx tree + accumulate GEMMs', 'barrier 4', 'D row phase + state machine']
tot = tr[:8].sum()
print('N', N, 'block 0 wave 0: trips', int(tr[8]), 'cycles/trip', tot / max(tr[8], 1))
for n_, v in zip(names, tr[:8]):
    print('  %-46s %10.0f cycles/trip  %5.1f %%' % (n_, v / max(tr[8], 1), 100 * v / tot))
print('kernel ms', ens.kernel_ms())
