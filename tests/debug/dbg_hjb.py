"""debug aid: compare the HJB device path with the oracle per trajectory"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _sde_oracle as S
from universal_differential_equations_amd import pde

def run(M, kw, tspan=(0.0, 1.0), seed=2, bias=0.02, it=3, x0s=0.0):
    rng = np.random.default_rng(seed)
    alg = pde.NNPDENS(100, 110)
    th = alg.init_params(rng)
    if bias:
        th = (th + bias * rng.standard_normal(th.size)).astype(np.float32)
    x0 = (x0s * rng.standard_normal(100)).astype(np.float32)
    prob = pde.TerminalPDEProblem(pde.hjb(1.0), x0, tspan)
    r = pde.loss_and_gradient(prob, alg, pde.LambaEM(), th, M, it=it, allow_failures=True, **kw)
    ref = S.loss_grad(S.desc(tspan=tspan, **kw), M, prob.x0, th, it=it, nthreads=8)
    print("u0", r.u0, ref["u0"], "loss", r.loss, ref["loss"])
    print("retcode", r.retcode.tolist(), ref["retcode"].tolist())
    bad = [j for j in range(M) if not np.array_equal(r.stats[j], ref["stats"][j])]
    print("stats mismatch traj", bad)
    for j in bad[:5]:
        print(j, r.stats[j], ref["stats"][j])
    badx = [j for j in range(M) if not np.array_equal(r.XT[j], ref["XT"][j])]
    badu = [j for j in range(M) if r.uT[j] != ref["uT"][j]]
    print("XT mismatch", badx, "uT mismatch", badu)
    for j in badu[:5]:
        print(j, r.uT[j], ref["uT"][j])
    badl = [j for j in range(M) if r.loss_traj[j] != ref["loss_traj"][j]]
    print("loss_traj mismatch", badl)
    g, gr = r.grad, ref["grad"]
    print("grad rel", np.linalg.norm(g - gr) / np.linalg.norm(gr), "max abs", np.abs(g - gr).max(), np.abs(gr).max())
    np0 = alg.num_params()[0]
    print("grad u0 part rel", np.linalg.norm(g[:np0] - gr[:np0]) / max(np.linalg.norm(gr[:np0]), 1e-30),
          "sg part rel", np.linalg.norm(g[np0:] - gr[np0:]) / np.linalg.norm(gr[np0:]))
    print("kernel ms", r.kernel_ms)
    # step-level comparison of trajectory jt
    import ctypes as C
    from universal_differential_equations_amd.sciml import Engine
    eng = Engine.get(0)
    prep = np.zeros(2, dtype=np.float32)
    eng.L.ude_hjb_debug_read(eng.h, 0, 0, 2, prep.ctypes.data)
    jt = 0
    pth = S.path(S.desc(tspan=tspan, **kw), prob.x0, th, it=it, traj=jt)
    print("dt_init dev %.9g oracle %.9g" % (prep[1], pth["dt"][0]))
    cap = 512
    n = pth["n"]
    rec = np.zeros((n, 104), dtype=np.float32)
    eng.L.ude_hjb_debug_read(eng.h, 1, jt * cap * 104, n * 104, rec.ctypes.data)
    tdev = rec[:, 100]
    first_t = next((i for i in range(n) if tdev[i] != pth["t"][i]), None)
    first_x = next((i for i in range(n) if not np.array_equal(rec[i, :100], pth["X"][i])), None)
    print("first t mismatch step", first_t, "first X mismatch step", first_x, "of", n)
    for i in ([first_t] if first_t is not None else []) + ([first_x] if first_x is not None else []):
        print(" step", i, "t", tdev[i], pth["t"][i], "dt_prev", pth["dt"][i - 1] if i else None, "X[:3]", rec[i, :3], pth["X"][i][:3])
    print("stats traj", jt, r.stats[jt])
    for ev in range(0, 12):
        out = np.zeros(100)
        eng.L.ude_hjb_normals(eng.h, C.c_uint64(kw.get("seed", 0)), it, jt, ev, 100, out.ctypes.data)
        o = S.normals(kw.get("seed", 0), it, jt, ev, 100)
        nd = int((out != o).sum())
        nf = int((out.astype(np.float32) != o.astype(np.float32)).sum())
        if nd:
            k = int(np.argmax(out != o))
            print("  normals event", ev, "double mismatches", nd, "float mismatches", nf, "first", k, repr(out[k]), repr(o[k]))
    if first_x is not None:
        i = first_x
        bad = np.nonzero(rec[i, :100] != pth["X"][i])[0]
        print("  X mismatch comps at step", i, bad[:10], "count", bad.size)
        print("  oracle dW_{i-1}", pth["dW"][i - 1][bad[:4]], "dt_{i-1}", pth["dt"][i - 1], "X_{i-1}", pth["X"][i - 1][bad[:4]], "dev X_{i-1}", rec[i - 1, bad[:4]])
        print("  dev X_i", rec[i, bad[:4]].tolist(), "oracle X_i", pth["X"][i][bad[:4]].tolist())
    e4 = np.zeros((n, 100), dtype=np.float32)
    eng.L.ude_hjb_debug_read(eng.h, 5, jt * cap * 100, n * 100, e4.ctypes.data)
    np0 = alg.num_params()[0]
    for i in range(min(n, 3)):
        z = S.net(100, 110, th[np0:], np.append(pth["X"][i], pth["t"][i]))
        e4o = np.float32(2.0) * np.float32(1.0) * pth["dt"][i] * z + pth["dW"][i]
        print(" step", i, "e4 max diff", np.abs(e4[i] - e4o).max(), "dW0 oracle", pth["dW"][i][:2])

if __name__ == "__main__":
    run(5, dict(abstol=0.1, reltol=0.1, seed=9))
    run(33, dict(abstol=0.05, reltol=0.05, seed=1, qmax=10.0), tspan=(0.0, 0.5), seed=3, bias=0.0, it=0, x0s=0.1)
