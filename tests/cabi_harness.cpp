// cabi_harness.cpp -- drives the drop-in boundary (include/udecore.h) from plain C++, exactly as a Julia `ccall` would:
// host buffers, column-major arrays, no Python, no torch.  Built and run by tests/test_gpu_cabi_harness.py (-m gpu).
//   Array(solve(prob, Tsit5(); saveat))                               -> ude_solve_ensemble       (scenario_1.jl:40-41 shape)
//   loss(theta) and its adjoint gradient                              -> ude_loss_grad_ensemble   (seir_exposure.jl:144-147 shape)
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
    printf("{\"version\": %d, \"loss\": %.17g, \"loss_direct\": %.17g, \"grad\": [%.17g, %.17g, %.17g, %.17g], \"fd_worst\": %.3g, "
           "\"nf0\": %lld, \"naccept0\": %lld, \"rc_unsupported\": %d, \"pred00\": %.17g}\n",
           ude_version(), loss, loss_of(th2), gth[0], gth[1], gth[2], gth[3], worst, (long long)stats[0], (long long)stats[1], rc_bad,
           pred[0]);
    ude_destroy(ctx);
    return 0;
}
