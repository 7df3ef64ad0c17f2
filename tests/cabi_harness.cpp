// cabi_harness.cpp -- drives the drop-in boundary (include/udecore.h) from plain C++, exactly as a Julia `ccall` would:
// host buffers, column-major arrays, no Python, no torch.  Built and run by tests/test_gpu_cabi_harness.py (-m gpu).
//   Array(solve(prob, Tsit5(); saveat))                               -> ude_solve_ensemble       (scenario_1.jl:40-41 shape)
//   loss(theta) and its adjoint gradient                              -> ude_loss_grad_ensemble   (seir_exposure.jl:144-147 shape)
//   a Float32 solve (scenario_3.jl:43-57)                             -> ude_solve_ensemble with dtype = 1
//   one deep-BSDE loss + gradient (highdim_pde/lambaem.jl:14-34)      -> ude_hjb_loss_grad
// Self-check: the adjoint gradient against central differences of the loss computed through ude_solve_ensemble.
// Prints one JSON line with the numbers the Python side compares with the oracle.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/udecore.h"

int main() {
    ude_ctx* ctx = nullptr;
    if (ude_create(0, &ctx) != UDE_OK) { fprintf(stderr, "ude_create failed\n"); return 2; }
    ude_model_desc m;
    memset(&m, 0, sizeof m);
    m.kind = UDE_KIND_LV_TRUE;       // lotka!, LotkaVolterra/scenario_1.jl:30-34
    m.n_state = 2;
    m.n_param = 4;
    m.lin_idx[0] = m.lin_idx[1] = -1;
    ude_solve_opts o;
    memset(&o, 0, sizeof o);
    o.alg = UDE_ALG_VERN7;
    o.abstol = 1e-10;
    o.reltol = 1e-10;
    const int N = 3, ns = 7, n = 2, np = 4;
    const double u0[N * n] = {0.44249296, 4.6280594, 0.5, 4.0, 0.4, 5.0};
    const double tspan[2] = {0.0, 3.0};
    double theta[np] = {1.3, 0.9, 0.8, 1.8};
    double saveat[ns];
    for (int i = 0; i < ns; ++i) saveat[i] = 0.5 * i;
    std::vector<double> truth(n * ns * N), pred(n * ns * N), gth(np), gu0(n * N), lpt(N);
    std::vector<int64_t> stats(UDE_NSTATS * N);
    std::vector<int32_t> rc(N);
    if (ude_solve_ensemble(ctx, &m, &o, N, u0, tspan, theta, saveat, ns, truth.data(), stats.data(), rc.data()) != UDE_OK) {
        fprintf(stderr, "solve: %s\n", ude_last_error(ctx));
        return 3;
    }
    // a perturbed parameter vector is "the model", the solution at p_ is "the data"
    double th2[np] = {1.2, 1.0, 0.7, 1.9};
    double loss = 0.0;
    if (ude_loss_grad_ensemble(ctx, &m, &o, N, u0, tspan, th2, saveat, ns, truth.data(), nullptr, &loss, lpt.data(), gth.data(),
                               gu0.data(), pred.data(), stats.data(), rc.data()) != UDE_OK) {
        fprintf(stderr, "loss_grad: %s\n", ude_last_error(ctx));
        return 4;
    }
    auto loss_of = [&](const double* th) -> double {
        std::vector<double> out(n * ns * N);
        std::vector<int64_t> st(UDE_NSTATS * N);
        std::vector<int32_t> r(N);
        if (ude_solve_ensemble(ctx, &m, &o, N, u0, tspan, th, saveat, ns, out.data(), st.data(), r.data()) != UDE_OK) return (double)NAN;
        double s = 0.0;
        for (size_t i = 0; i < out.size(); ++i) s += (out[i] - truth[i]) * (out[i] - truth[i]);
        return s;
    };
    double worst = 0.0;
    for (int i = 0; i < np; ++i) {
        double tp[np], tm[np];
        memcpy(tp, th2, sizeof tp);
        memcpy(tm, th2, sizeof tm);
        tp[i] += 1e-6;
        tm[i] -= 1e-6;
        const double fd = (loss_of(tp) - loss_of(tm)) / 2e-6;
        worst = fmax(worst, fabs(fd - gth[i]) / fmax(1.0, fabs(fd)));
    }
    // an unsupported descriptor must fail loudly through the same boundary
    ude_model_desc bad = m;
    bad.kind = UDE_KIND_LV_UDE;
    bad.n_layers = 2;
    bad.dims[0] = 2; bad.dims[1] = 7; bad.dims[2] = 2;
    const int rc_bad = ude_model_supported(ctx, &bad, &o, 1);
    // ---- a Float32 problem through the same entry point (ude_model_desc.dtype = 1: every real array is float) ----
    // rc_ode of LotkaVolterra/scenario_3.jl:43-57 on 26 points, Tsit5 at default tolerances
    ude_model_desc mf;
    memset(&mf, 0, sizeof mf);
    mf.kind = UDE_KIND_KPP_TRUE;
    mf.dtype = 1;
    mf.n_state = 26;
    mf.lin_idx[0] = mf.lin_idx[1] = -1;
    {
        const float dx = 0.04f, dx2 = dx * dx;
        const float off = (float)(1.0 / (double)dx2), dia = (float)(-2.0 / (double)dx2);
        mf.consts[0] = (double)(0.01f * off); mf.consts[1] = (double)(0.01f * dia); mf.consts[2] = 1.0;
    }
    ude_solve_opts of;
    memset(&of, 0, sizeof of);
    of.alg = UDE_ALG_TSIT5;
    float u0f[26], satf[3] = {0.0f, 2.5f, 5.0f}, outf[26 * 3], thf[1] = {0.0f};
    for (int i = 0; i < 26; ++i) {  // a bump from +, *, / only (no libm: the Python side must form the same bits)
        const float xx = (float)i / 25.0f, om = 1.0f - xx;
        u0f[i] = ((16.0f * xx) * xx) * (om * om);
    }
    const double tspanf[2] = {0.0, 5.0};
    int64_t stf[UDE_NSTATS];
    int32_t rcf = -1;
    // (the real-valued array parameters are ude_real* = void*: with dtype = 1 they carry float data)
    if (ude_solve_ensemble(ctx, &mf, &of, 1, u0f, tspanf, thf, satf, 3, outf, stf, &rcf) != UDE_OK) {
        fprintf(stderr, "f32 solve: %s\n", ude_last_error(ctx));
        return 5;
    }
    // ---- one deep-BSDE loss + gradient (highdim_pde/lambaem.jl:14-34) with a parameter vector from a small LCG ----
    ude_hjb_desc hd;
    memset(&hd, 0, sizeof hd);
    hd.d = 100; hd.hls = 110; hd.adaptive = 1; hd.seed = 77;
    hd.lambda = 1.0; hd.sigma = (double)sqrtf(2.0f); hd.t0 = 0.0; hd.t1 = 1.0; hd.abstol = 0.1; hd.reltol = 0.1;
    int32_t np0 = 0, np1 = 0;
    ude_hjb_num_params(hd.d, hd.hls, &np0, &np1);
    std::vector<float> hth((size_t)np0 + np1), hgrad((size_t)np0 + np1), hx0(100, 0.0f), huT(5), hXT(500);
    uint32_t lcg = 12345u;
    for (auto& v : hth) { lcg = lcg * 1664525u + 1013904223u; v = ((float)((lcg >> 8) & 0xFFFFu) / 65536.0f - 0.5f) * 0.2f; }
    double hloss = 0.0;
    float hu0 = 0.0f;
    std::vector<double> hlt(5);
    std::vector<int64_t> hst(4 * 5);
    std::vector<int32_t> hrc(5);
    if (ude_hjb_loss_grad(ctx, &hd, 5, hx0.data(), hth.data(), 3u, &hloss, hgrad.data(), &hu0, huT.data(), hXT.data(), hlt.data(), hst.data(),
                          hrc.data()) != UDE_OK) {
        fprintf(stderr, "hjb: %s\n", ude_last_error(ctx));
        return 6;
    }
    double hgn = 0.0;
    for (float v : hgrad) hgn += (double)v * (double)v;
    printf("{\"f32_nf\": %lld, \"f32_naccept\": %lld, \"f32_nreject\": %lld, \"f32_rc\": %d, \"f32_u_end_13\": %.9g, "
           "\"hjb_np\": %d, \"hjb_loss\": %.17g, \"hjb_u0\": %.9g, \"hjb_uT0\": %.9g, \"hjb_naccept0\": %lld, \"hjb_nreject0\": %lld, "
           "\"hjb_grad_norm\": %.9g, ",
           (long long)stf[0], (long long)stf[1], (long long)stf[2], (int)rcf, (double)outf[2 * 26 + 13], (int)(np0 + np1), hloss, (double)hu0,
           (double)huT[0], (long long)hst[1], (long long)hst[2], sqrt(hgn));
    printf("\"version\": %d, \"loss\": %.17g, \"loss_direct\": %.17g, \"grad\": [%.17g, %.17g, %.17g, %.17g], \"fd_worst\": %.3g, "
           "\"nf0\": %lld, \"naccept0\": %lld, \"rc_unsupported\": %d, \"pred00\": %.17g}\n",
           ude_version(), loss, loss_of(th2), gth[0], gth[1], gth[2], gth[3], worst, (long long)stats[0], (long long)stats[1], rc_bad,
           pred[0]);
    ude_destroy(ctx);
    return 0;
}
