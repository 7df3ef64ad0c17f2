"""debug: Float32 device vs oracle, first point of divergence"""
import json, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
f32 = np.float32
g = json.load(open("tests/golden/Scenario_3_recovery_0.005.json"))
s = g["solution"]
prob = U.ODEProblem(models.rc_ode(26, 0.01, 1.0, 0.04, dtype="float32"), np.array(s["u0"], dtype=f32), tuple(s["tspan"]), [])
for ts in ([0.0, 0.5, 1.0], s["t"]):
    sol = U.solve(prob, U.Tsit5(), saveat=np.array(ts, dtype=f32))
    out, st, rc = O.solve_ensemble(O.kpp_true(26, 0.01, 1.0, 0.04, dtype=1), O.opts(O.TSIT5), s["u0"], s["tspan"], [], ts, dtype=f32)
    a = np.asarray(sol).T; b = out[0]
    d = sol.destats
    print("dev stats", d.nf, d.naccept, d.nreject, "oracle", st[0][:4])
    bad = [i for i in range(len(ts)) if not np.array_equal(a[i], b[i])]
    print("first differing save idx", bad[:3], "of", len(ts))
    if bad:
        i = bad[0]; print(np.abs(a[i]-b[i]).max(), np.nonzero(a[i]!=b[i])[0])
# rhs
X = np.array(s["u"], dtype=f32)[:4]
f = models.rc_ode(26, 0.01, 1.0, 0.04, dtype="float32")
du = U.rhs(f, X, [])
ref = np.array([O.rhs(O.kpp_true(26, 0.01, 1.0, 0.04, dtype=1), [], X[i], dtype=f32) for i in range(4)])
print("rhs equal", np.array_equal(du, ref), np.abs(du-ref).max())
# short span
for tf in (0.01, 0.0625, 0.25):
    p2 = U.ODEProblem(f, np.array(s["u0"], dtype=f32), (0.0, tf), [])
    sol = U.solve(p2, U.Tsit5(), saveat=np.array([0.0, tf], dtype=f32))
    out, st, rc = O.solve_ensemble(O.kpp_true(26, 0.01, 1.0, 0.04, dtype=1), O.opts(O.TSIT5), s["u0"], [0.0, tf], [], [0.0, tf], dtype=f32)
    d = sol.destats
    print(tf, "dev", d.nf, d.naccept, d.nreject, "oracle", st[0][:3], "equal", np.array_equal(np.asarray(sol).T, out[0]), np.abs(np.asarray(sol).T-out[0]).max())
# hudson: forward u, then adjoint pieces
HBg = json.load(open("tests/golden/Hudson_Bay_recovery.json"))
X = np.array(HBg["X"]["data_colmajor"], dtype=f32).reshape(21, 2)
t = np.array(HBg["t"], dtype=f32)
th = np.array(HBg["trained_parameters"], dtype=f32)
fh = models.ude_dynamics(models.hudson_chain(), trainable="both", dtype="float32")
prob = U.ODEProblem(fh, X[0], (float(t[0]), float(t[-1])), th)
for sense, osense in ((U.ForwardDiffSensitivity(), 1), (None, 0)):
    r = U.loss_and_gradient(prob, U.Vern7(), X[None], saveat=t, abstol=1e-6, reltol=1e-6, sensealg=sense)
    ref = O.loss_grad_ensemble(O.lv_ude_hudson(1), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=osense), X[0], [t[0], t[-1]], th, t, X[None], dtype=f32)
    print("hudson sense", osense, "stats", r.stats, ref["stats"], "u eq", np.array_equal(r.u, ref["u"]), "lpt", r.loss_per_traj, ref["loss_per_traj"],
          "gu0", r.grad_u0, ref["grad_u0"], "gth rel", np.linalg.norm(r.grad_theta-ref["grad_theta"])/np.linalg.norm(ref["grad_theta"]))
