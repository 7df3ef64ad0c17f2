/*
 * ude_oracle_adj.h -- ensemble wrappers around the adjoint, type-generic (REAL / FN; TEST INFRASTRUCTURE).
 * a10/a11: predict + loss + InterpolatingAdjoint gradient over an ensemble sharing theta
 * (seir_exposure.jl:137-147, Fisher-KPP-CNN.jl:134-143; ensemble shape per SURVEY.md 8(d)).
 */
static int FN(vjp_common)(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                           const REAL* u0, const REAL* tspan, const REAL* theta,
                           const REAL* saveat, int32_t ns, const REAL* cot, const REAL* data,
                           const uint8_t* mask, REAL* loss, REAL* loss_per_traj, REAL* u_out,
                           REAL* grad_theta, REAL* grad_u0, int64_t* stats, int32_t* retcode,
                           int32_t nthreads) {
    const int n = m->n_state, np = m->n_param;
    int nt = nthreads > 1 ? nthreads : 1;
#ifndef _OPENMP
    nt = 1;
#endif
    double* gacc = (double*)calloc((size_t)nt * np, sizeof(double)); /* ensemble sums in double for both types */
    double* lacc = (double*)calloc(nt, sizeof(double));
    REAL* gone = (REAL*)calloc((size_t)nt * np, sizeof(REAL));
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* g = gacc + (size_t)tid * np;
        REAL* g1 = gone + (size_t)tid * np;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t j = 0; j < N; ++j) {
            int64_t st[UDEO_NSTATS] = {0};
            REAL L = 0;
            for (int i = 0; i < np; ++i) g1[i] = 0;
            int rc = FN(vjp_one)(m, o, theta, u0 + (size_t)j * n, tspan[0], tspan[1], saveat, ns,
                                 cot ? cot + (size_t)j * n * ns : 0, data ? data + (size_t)j * n * ns : 0,
                                 mask, &L, u_out ? u_out + (size_t)j * n * ns : 0, g1,
                                 grad_u0 ? grad_u0 + (size_t)j * n : 0, st);
            for (int i = 0; i < np; ++i) g[i] += (double)g1[i];
            lacc[tid] += (double)L;
            if (loss_per_traj) loss_per_traj[j] = L;
            if (stats) memcpy(stats + (size_t)j * UDEO_NSTATS, st, sizeof(st));
            if (retcode) retcode[j] = rc;
            if (rc != UDEO_RET_SUCCESS) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                fail = 1;
            }
        }
    }
    double L = 0;
    for (int i = 0; i < np; ++i) {
        double a = 0;
        for (int t = 0; t < nt; ++t) a += gacc[(size_t)t * np + i]; /* fixed thread order: deterministic for a given nthreads */
        grad_theta[i] = (REAL)a;
    }
    for (int t = 0; t < nt; ++t) L += lacc[t];
    if (loss) *loss = (REAL)L;
    free(gacc);
    free(lacc);
    free(gone);
    return fail;
}

int FN(udeo_vjp_ensemble)(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                          const REAL* u0, const REAL* tspan, const REAL* theta,
                          const REAL* saveat, int32_t ns, const REAL* cotangent, REAL* u_out,
                          REAL* grad_theta, REAL* grad_u0, int64_t* stats, int32_t* retcode,
                          int32_t nthreads) {
    return FN(vjp_common)(m, o, N, u0, tspan, theta, saveat, ns, cotangent, 0, 0, 0, 0, u_out,
                           grad_theta, grad_u0, stats, retcode, nthreads);
}

int FN(udeo_loss_grad_ensemble)(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                                const REAL* u0, const REAL* tspan, const REAL* theta,
                                const REAL* saveat, int32_t ns, const REAL* data,
                                const uint8_t* row_mask, REAL* loss, REAL* loss_per_traj,
                                REAL* grad_theta, REAL* grad_u0, REAL* u_out, int64_t* stats,
                                int32_t* retcode, int32_t nthreads) {
    return FN(vjp_common)(m, o, N, u0, tspan, theta, saveat, ns, 0, data, row_mask, loss,
                           loss_per_traj, u_out, grad_theta, grad_u0, stats, retcode, nthreads);
}

#ifdef UDEO_ADJ_F64_ONLY
int udeo_solve_dense_f64(const udeo_model_desc* m, const udeo_solve_opts* o, const double* u0,
                         const double* tspan, const double* theta, int32_t cap, double* t_steps,
                         double* u_steps, double* k_steps, int64_t* stats) {
    dense_f64 d;
    d.cap = cap; d.n = m->n_state; d.nk = o->alg == UDEO_ALG_TSIT5 ? 7 : 16; d.nsteps = 0;
    d.t = t_steps; d.u = u_steps; d.k = k_steps; d.dt = 0;
    int64_t st[UDEO_NSTATS] = {0};
    int rc = solve_one_f64(m, o, theta, u0, tspan[0], tspan[1], 0, 0, 0, &d, st);
    if (stats) memcpy(stats, st, sizeof(st));
    return rc == UDEO_RET_SUCCESS ? d.nsteps : -rc;
}
#endif
