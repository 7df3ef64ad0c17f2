/*
 * sde_oracle_impl.h -- type-generic body of the N1 oracle (TEST INFRASTRUCTURE).  Included twice by sde_oracle.c with
 * REAL = float (the reference's Float32 problem, highdim_pde/lambaem.jl:9-10) and REAL = double (finite-difference
 * checks of the restatement itself).  See sde_oracle.h for what is restated and why parity is unpinned.
 */

/* sum over n <= 128 components: binary tree over adjacent index pairs of the zero-padded vector */
static REAL NAME(tsum)(const REAL* v, int n) {
    REAL b[128];
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = 0; i < P; ++i) b[i] = i < n ? v[i] : (REAL)0;
    for (int s = 1; s < P; s <<= 1)
        for (int i = 0; i < P; i += 2 * s) b[i] = b[i] + b[i + s];
    return b[0];
}

/* Dense layer, Flux layout W (out x in, column-major) then b: out_j = act(chain(b_j; W[j,k] a_k, k ascending)).
 * relu: NNlib's relu(x) = max(zero(x), x) (Flux 0.9, highdim_pde/Manifest.toml:264).  Its derivative under Tracker /
 * ForwardDiff is the partial of `max` with respect to its SECOND argument, `0 > x ? 0 : 1`: ONE at x == 0 [UP?].  That
 * convention is not a curiosity here: with x0 = 0 (lambaem.jl:9) and Flux's zero-initialised biases every pre-activation of
 * the u0 chain -- and of the first step's sigma^T grad u evaluation at (X, t) = (0, 0) -- is exactly 0 at the first
 * iteration; with relu'(0) = 0 only the output bias of u0 would ever receive a gradient there.
 * The output keeps the SIGN BIT of a non-positive pre-activation (-0.0 for a negative one, +0.0 for +0) (value 0 either way; a product with +-0 never changes a
 * sum that starts from a bias): RELU_ON(a) = "the pre-activation was >= 0" is what the reverse sweeps need. */
#define RELU_ON(a) (!signbit(a))
static void NAME(dense)(const REAL* W, const REAL* b, int in, int out, const REAL* a, REAL* o, int relu) {
    for (int j = 0; j < out; ++j) {
        REAL acc = b[j];
        for (int k = 0; k < in; ++k) acc = FMA(W[j + (size_t)k * out], a[k], acc);
        o[j] = relu ? (acc > (REAL)0 ? acc : (signbit(acc) ? (REAL)-0.0 : (REAL)0)) : acc;
    }
}

typedef struct {
    int d, H;
    const REAL *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4; /* sigma^T grad u chain: (d+1)->H->H->H->d (lambaem.jl:27-30) */
    const REAL *U1, *c1, *U2, *c2, *U3, *c3;            /* u0 chain: d->H->H->1 (lambaem.jl:23-25) */
} NAME(Nets);

static void NAME(nets_init)(NAME(Nets)* n, int d, int H, const REAL* th) {
    n->d = d; n->H = H;
    const REAL* p = th;
    n->U1 = p; p += (size_t)H * d; n->c1 = p; p += H;
    n->U2 = p; p += (size_t)H * H; n->c2 = p; p += H;
    n->U3 = p; p += H; n->c3 = p; p += 1;
    n->W1 = p; p += (size_t)H * (d + 1); n->b1 = p; p += H;
    n->W2 = p; p += (size_t)H * H; n->b2 = p; p += H;
    n->W3 = p; p += (size_t)H * H; n->b3 = p; p += H;
    n->W4 = p; p += (size_t)d * H; n->b4 = p; p += d;
}

/* z = sigma^T grad u([X; t]); activations returned for the reverse sweep */
static void NAME(sg_fwd)(const NAME(Nets)* n, const REAL* xin, REAL* a1, REAL* a2, REAL* a3, REAL* z) {
    NAME(dense)(n->W1, n->b1, n->d + 1, n->H, xin, a1, 1);
    NAME(dense)(n->W2, n->b2, n->H, n->H, a1, a2, 1);
    NAME(dense)(n->W3, n->b3, n->H, n->H, a2, a3, 1);
    NAME(dense)(n->W4, n->b4, n->H, n->d, a3, z, 0);
}

static REAL NAME(sumsq)(const REAL* v, int n) {
    REAL t[128];
    for (int i = 0; i < n; ++i) t[i] = v[i] * v[i];
    return NAME(tsum)(t, n);
}

typedef struct {
    REAL lam, sig, t0, t1, abstol, reltol, qmin, qmax, gamma, qoldinit, beta1, beta2, dtmax;
    int maxiters, cap;
} NAME(Par);

static void NAME(par_init)(NAME(Par)* p, const udeo_hjb_desc* D) {
    p->lam = (REAL)D->lambda; p->sig = (REAL)D->sigma; p->t0 = (REAL)D->t0; p->t1 = (REAL)D->t1;
    p->abstol = (REAL)D->abstol; p->reltol = (REAL)D->reltol;
    p->qmin = (REAL)(D->qmin > 0 ? D->qmin : 0.2);
    p->qmax = (REAL)(D->qmax > 0 ? D->qmax : 1.125);
    p->gamma = (REAL)(D->gamma > 0 ? D->gamma : 0.9);
    p->qoldinit = (REAL)(D->qoldinit > 0 ? D->qoldinit : 1e-4);
    p->beta1 = (REAL)(D->beta1 > 0 ? D->beta1 : 0.7);
    p->beta2 = (REAL)(D->beta2 > 0 ? D->beta2 : 0.4);
    p->dtmax = (REAL)(D->dtmax > 0 ? D->dtmax : D->t1 - D->t0);
    p->maxiters = D->maxiters > 0 ? D->maxiters : 1000000;
    p->cap = D->max_steps > 0 ? D->max_steps : 4096;
}

/* sde_determine_initdt restated for h0 = [x0; u0] (identical for every trajectory); 2 network evaluations */
static REAL NAME(initdt)(const NAME(Nets)* n, const NAME(Par)* p, const REAL* x0, REAL u0) {
    const int d = n->d;
    REAL xin[129], a1[128], a2[128], a3[128], z0[128], zB[128], tmp[128];
    const REAL sku = FMA(FABS(u0), p->reltol, p->abstol);
    for (int c = 0; c < d; ++c) {
        const REAL sk = FMA(FABS(x0[c]), p->reltol, p->abstol);
        const REAL q = x0[c] / sk;
        tmp[c] = q * q;
    }
    const REAL qu = u0 / sku;
    const REAL d0 = SQRT((NAME(tsum)(tmp, d) + qu * qu) / (REAL)(d + 1));
    for (int c = 0; c < d; ++c) xin[c] = x0[c];
    xin[d] = p->t0;
    NAME(sg_fwd)(n, xin, a1, a2, a3, z0);
    const REAL F0 = p->lam * NAME(sumsq)(z0, d);
    const REAL s3 = (REAL)3 * p->sig, s6 = (REAL)6 * p->sig;
    /* d1: (d+1) x d matrix max(|f0 + 3g0|, |f0 - 3g0|) ./ sk; rows c < d: only the diagonal 3 sigma; row u: |F0| + 3|z0_k| */
    for (int c = 0; c < d; ++c) {
        const REAL sk = FMA(FABS(x0[c]), p->reltol, p->abstol);
        const REAL q = s3 / sk;
        tmp[c] = q * q;
    }
    REAL sA = NAME(tsum)(tmp, d);
    for (int k = 0; k < d; ++k) {
        const REAL q = (FABS(F0) + (REAL)3 * FABS(z0[k])) / sku;
        tmp[k] = q * q;
    }
    const REAL d1 = SQRT((sA + NAME(tsum)(tmp, d)) / (REAL)((d + 1) * d));
    REAL dt0 = (d0 < (REAL)1e-5 || d1 < (REAL)1e-5) ? (REAL)1e-6 : (d0 / d1) / (REAL)100;
    if (dt0 > p->dtmax) dt0 = p->dtmax;
    /* u1 = h0 + dt0 f0: X unchanged; f1, g1 at t0 + dt0 */
    xin[d] = p->t0 + dt0;
    NAME(sg_fwd)(n, xin, a1, a2, a3, zB);
    const REAL F1 = p->lam * NAME(sumsq)(zB, d);
    for (int c = 0; c < d; ++c) {
        const REAL sk = FMA(FABS(x0[c]), p->reltol, p->abstol);
        const REAL q = s6 / sk; /* max(|3s - 3s|, |3s + 3s|) on the diagonal, f1 - f0 = 0 */
        tmp[c] = q * q;
    }
    sA = NAME(tsum)(tmp, d);
    const REAL dF = FABS(F1 - F0);
    for (int k = 0; k < d; ++k) {
        const REAL g0 = (REAL)3 * z0[k], g1 = (REAL)3 * zB[k];
        const REAL m1 = FABS(g0 - g1), m2 = FABS(g0 + g1);
        const REAL q = (dF + (m1 > m2 ? m1 : m2)) / sku;
        tmp[k] = q * q;
    }
    const REAL d2 = SQRT((sA + NAME(tsum)(tmp, d)) / (REAL)((d + 1) * d)) / dt0;
    const REAL mx = d1 > d2 ? d1 : d2;
    REAL dt1;
    if (mx <= (REAL)1e-15) {
        dt1 = dt0 * (REAL)1e-3;
        if (dt1 < (REAL)1e-6) dt1 = (REAL)1e-6;
    } else {
        dt1 = (REAL)udeo_pow10(-(2.0 + udeo_log10((double)mx)) / 1.0); /* order + 1/2 = 1 */
    }
    REAL r = (REAL)100 * dt0;
    if (dt1 < r) r = dt1;
    if (p->dtmax < r) r = p->dtmax;
    return r;
}

typedef struct {
    int nacc, nrej, ret;
    int64_t nf, ndraw;
} NAME(TrajStat);

static void NAME(draw)(uint64_t seed, uint32_t iter, uint32_t traj, uint32_t* ev, int d, REAL* out, NAME(TrajStat)* st) {
    double nb[128];
    udeo_hjb_normals(seed, iter, traj, *ev, d, nb);
    for (int c = 0; c < d; ++c) out[c] = (REAL)nb[c];
    *ev += 1;
    st->ndraw += 1;
}

/* one trajectory; records the accepted steps (t_n, dt_n, X_n, dW_n) when rec_* != NULL */
static void NAME(traj)(const udeo_hjb_desc* D, const NAME(Nets)* n, const NAME(Par)* p, const REAL* x0, REAL u0, REAL dt_init,
                       uint32_t iter, uint32_t traj, REAL* XT, REAL* uT, REAL* rec_t, REAL* rec_dt, REAL* rec_X, REAL* rec_dW,
                       REAL* rec_EE, NAME(TrajStat)* st) {
    const int d = n->d;
    REAL X[128], Xn[128], dW[128], nrm[128], xin[129], a1[128], a2[128], a3[128], z[128], z2[128], z3[128], tmp[128];
    REAL stackL[UDEO_HJB_STACK];
    REAL* stackW = (REAL*)malloc(sizeof(REAL) * UDEO_HJB_STACK * 128);
    int nstack = 0;
    uint32_t ev = 0;
    REAL t = p->t0, u = u0, dt = dt_init, qold = p->qoldinit, q11 = (REAL)1;
    int last = 0, iterc = 0;
    memset(st, 0, sizeof *st);
    for (int c = 0; c < d; ++c) X[c] = x0[c];
    {
        const REAL rem = p->t1 - t;
        if (dt >= rem) { dt = rem; last = 1; }
        NAME(draw)(D->seed, iter, traj, &ev, d, nrm, st);
        const REAL s = SQRT(dt);
        for (int c = 0; c < d; ++c) dW[c] = s * nrm[c];
    }
    for (;;) {
        iterc += 1;
        if (iterc > p->maxiters) { st->ret = UDEO_HJB_RET_MAXITERS; break; }
        const REAL sq = SQRT(dt);
        for (int c = 0; c < d; ++c) xin[c] = X[c];
        xin[d] = t;
        NAME(sg_fwd)(n, xin, a1, a2, a3, z);
        st->nf += 1;
        const REAL Sz = NAME(sumsq)(z, d);
        const REAL F = p->lam * Sz;
        for (int c = 0; c < d; ++c) tmp[c] = z[c] * dW[c];
        const REAL zdW = NAME(tsum)(tmp, d);
        for (int c = 0; c < d; ++c) Xn[c] = FMA(p->sig, dW[c], X[c]);
        const REAL un = FMA(dt, F, u) + zdW;
        REAL EE = (REAL)0, q = (REAL)1;
        int accept = 1;
        if (D->adaptive) {
            xin[d] = t + dt;
            NAME(sg_fwd)(n, xin, a1, a2, a3, z2);
            const REAL F2 = p->lam * NAME(sumsq)(z2, d);
            const REAL Ed = (dt * (F2 - F)) * (REAL)0.5;
            const REAL gs = SQRT(FMA((REAL)d * p->sig, p->sig, Sz)); /* ||G||_F */
            const REAL cc = gs * sq;
            for (int c = 0; c < d; ++c) xin[c] = X[c] + cc;
            xin[d] = t;
            NAME(sg_fwd)(n, xin, a1, a2, a3, z3);
            st->nf += 2;
            /* non-diagonal noise: the diffusion enters the estimate through SCALAR norms [UP?] --
             *   g_sized = ||G(h)||_F, ggprime = (||G(utilde)||_F - g_sized) / sqrt(dt), En = ggprime * internalnorm(dW.^2) / 2
             * (internalnorm = RMS), one number added to EVERY component's residual; the drift part Ed lives on the u row only */
            const REAL gs3 = SQRT(FMA((REAL)d * p->sig, p->sig, NAME(sumsq)(z3, d)));
            const REAL ggp = (gs3 - gs) / sq;
            for (int c = 0; c < d; ++c) { const REAL w2 = dW[c] * dW[c]; tmp[c] = w2 * w2; }
            const REAL nW2 = SQRT(NAME(tsum)(tmp, d) / (REAL)d);
            const REAL En = (ggp * nW2) * (REAL)0.5;
            for (int c = 0; c < d; ++c) {
                const REAL a0 = FABS(X[c]), a1x = FABS(Xn[c]);
                const REAL r = En / FMA((a0 > a1x ? a0 : a1x), p->reltol, p->abstol);
                tmp[c] = r * r;
            }
            const REAL au = FABS(u), aun = FABS(un);
            const REAL res = (Ed + En) / FMA((au > aun ? au : aun), p->reltol, p->abstol);
            EE = SQRT((NAME(tsum)(tmp, d) + res * res) / (REAL)(d + 1));
            if (EE == (REAL)0) {
                q = (REAL)1 / p->qmax;
                q11 = (REAL)1; /* (unused on this branch) */
            } else {
                q11 = (REAL)udeo_fastpow((double)EE, (double)p->beta1);
                q = q11 / (REAL)udeo_fastpow((double)qold, (double)p->beta2);
                q = q / p->gamma;
                const REAL lo = (REAL)1 / p->qmax, hi = (REAL)1 / p->qmin;
                if (q > hi) q = hi;
                if (q < lo) q = lo;
            }
            accept = EE <= (REAL)1;
            if (EE != EE) { st->ret = UDEO_HJB_RET_UNSTABLE; break; }
        }
        if (accept) {
            if (st->nacc >= p->cap) { st->ret = UDEO_HJB_RET_STORE_OVERFLOW; break; }
            if (rec_t) {
                rec_t[st->nacc] = t; rec_dt[st->nacc] = dt;
                if (rec_EE) rec_EE[st->nacc] = EE;
                for (int c = 0; c < d; ++c) { rec_X[(size_t)st->nacc * d + c] = X[c]; rec_dW[(size_t)st->nacc * d + c] = dW[c]; }
            }
            st->nacc += 1;
            t = last ? p->t1 : t + dt;
            u = un;
            int bad = un != un;
            for (int c = 0; c < d; ++c) { X[c] = Xn[c]; bad = bad || (Xn[c] != Xn[c]); }
            if (bad) { st->ret = UDEO_HJB_RET_UNSTABLE; break; }
            if (t >= p->t1) break;
            REAL dtn = dt;
            if (D->adaptive) {
                qold = EE > p->qoldinit ? EE : p->qoldinit;
                dtn = dt / q;
                if (dtn > p->dtmax) dtn = p->dtmax;
            }
            const REAL rem = p->t1 - t;
            last = 0;
            if (dtn >= rem) { dtn = rem; last = 1; }
            /* the increment over [t, t + dtn]: whole stack pieces, the last one bridged, the rest fresh */
            REAL acch = (REAL)0;
            for (int c = 0; c < d; ++c) dW[c] = (REAL)0;
            while (nstack > 0 && acch < dtn) {
                REAL* top = stackW + (size_t)(nstack - 1) * 128;
                const REAL L = stackL[nstack - 1];
                if (acch + L <= dtn) {
                    acch = acch + L;
                    for (int c = 0; c < d; ++c) dW[c] = dW[c] + top[c];
                    nstack -= 1;
                } else {
                    const REAL rl = dtn - acch, fr = rl / L;
                    NAME(draw)(D->seed, iter, traj, &ev, d, nrm, st);
                    const REAL sd = SQRT(((REAL)1 - fr) * rl);
                    for (int c = 0; c < d; ++c) {
                        const REAL w = FMA(fr, top[c], sd * nrm[c]);
                        top[c] = top[c] - w;
                        dW[c] = dW[c] + w;
                    }
                    stackL[nstack - 1] = L - rl;
                    acch = dtn;
                }
            }
            if (acch < dtn) {
                NAME(draw)(D->seed, iter, traj, &ev, d, nrm, st);
                const REAL sd = SQRT(dtn - acch);
                for (int c = 0; c < d; ++c) dW[c] = FMA(sd, nrm[c], dW[c]);
            }
            dt = dtn;
        } else {
            st->nrej += 1;
            REAL den = q11 / p->gamma;
            const REAL iq = (REAL)1 / p->qmin;
            if (iq < den) den = iq;
            const REAL dtn = dt / den, fr = dtn / dt;
            if (nstack >= UDEO_HJB_STACK) { st->ret = UDEO_HJB_RET_STACK_OVERFLOW; break; }
            NAME(draw)(D->seed, iter, traj, &ev, d, nrm, st);
            const REAL sd = SQRT(((REAL)1 - fr) * dtn);
            REAL* top = stackW + (size_t)nstack * 128;
            for (int c = 0; c < d; ++c) {
                const REAL w = FMA(fr, dW[c], sd * nrm[c]);
                top[c] = dW[c] - w;
                dW[c] = w;
            }
            stackL[nstack] = dt - dtn;
            nstack += 1;
            dt = dtn;
            last = 0;
        }
    }
    for (int c = 0; c < d; ++c) XT[c] = X[c];
    *uT = u;
    free(stackW);
}

/* reverse sweep of one accepted step: theta_sg gradient += ubar * d(u_{n+1})/d(theta); X never depends on theta
 * (mu = 0, sigma constant), so the columns (trajectory, step) are independent */
static void NAME(step_bwd)(const NAME(Nets)* n, const NAME(Par)* p, const REAL* X, REAL t, REAL dt, const REAL* dW, REAL ubar,
                           double* g /* theta_sg gradient, layout of theta_sg */) {
    const int d = n->d, H = n->H;
    REAL xin[129], a1[128], a2[128], a3[128], z[128], d4[128], d3[128], d2[128], d1[128];
    for (int c = 0; c < d; ++c) xin[c] = X[c];
    xin[d] = t;
    NAME(sg_fwd)(n, xin, a1, a2, a3, z);
    const REAL coef = ((REAL)2 * p->lam) * dt;
    for (int c = 0; c < d; ++c) d4[c] = ubar * FMA(coef, z[c], dW[c]);
    /* delta_l = (W_{l+1}^T delta_{l+1}) .* relu'(a_l): fma chains over the rows of W_{l+1} in ascending order from 0 */
    for (int i = 0; i < H; ++i) {
        REAL acc = (REAL)0;
        for (int c = 0; c < d; ++c) acc = FMA(n->W4[c + (size_t)i * d], d4[c], acc);
        d3[i] = RELU_ON(a3[i]) ? acc : (REAL)0;
    }
    for (int i = 0; i < H; ++i) {
        REAL acc = (REAL)0;
        for (int k = 0; k < H; ++k) acc = FMA(n->W3[k + (size_t)i * H], d3[k], acc);
        d2[i] = RELU_ON(a2[i]) ? acc : (REAL)0;
    }
    for (int i = 0; i < H; ++i) {
        REAL acc = (REAL)0;
        for (int k = 0; k < H; ++k) acc = FMA(n->W2[k + (size_t)i * H], d2[k], acc);
        d1[i] = RELU_ON(a1[i]) ? acc : (REAL)0;
    }
    double* gp = g;
    for (int k = 0; k <= d; ++k) for (int j = 0; j < H; ++j) gp[j + (size_t)k * H] += (double)d1[j] * (double)xin[k];
    gp += (size_t)H * (d + 1);
    for (int j = 0; j < H; ++j) gp[j] += (double)d1[j];
    gp += H;
    for (int k = 0; k < H; ++k) for (int j = 0; j < H; ++j) gp[j + (size_t)k * H] += (double)d2[j] * (double)a1[k];
    gp += (size_t)H * H;
    for (int j = 0; j < H; ++j) gp[j] += (double)d2[j];
    gp += H;
    for (int k = 0; k < H; ++k) for (int j = 0; j < H; ++j) gp[j + (size_t)k * H] += (double)d3[j] * (double)a2[k];
    gp += (size_t)H * H;
    for (int j = 0; j < H; ++j) gp[j] += (double)d3[j];
    gp += H;
    for (int k = 0; k < H; ++k) for (int c = 0; c < d; ++c) gp[c + (size_t)k * d] += (double)d4[c] * (double)a3[k];
    gp += (size_t)d * H;
    for (int c = 0; c < d; ++c) gp[c] += (double)d4[c];
}

/* u0 = u0 net(x0) and (g != NULL) g += U * d(u0)/d(theta_u0) */
static REAL NAME(u0_net)(const NAME(Nets)* n, const REAL* x0, REAL U, double* g) {
    const int d = n->d, H = n->H;
    REAL a1[128], a2[128], o[1], d2[128], d1[128];
    NAME(dense)(n->U1, n->c1, d, H, x0, a1, 1);
    NAME(dense)(n->U2, n->c2, H, H, a1, a2, 1);
    NAME(dense)(n->U3, n->c3, H, 1, a2, o, 0);
    if (g) {
        for (int i = 0; i < H; ++i) d2[i] = RELU_ON(a2[i]) ? n->U3[i] * U : (REAL)0;
        for (int i = 0; i < H; ++i) {
            REAL acc = (REAL)0;
            for (int k = 0; k < H; ++k) acc = FMA(n->U2[k + (size_t)i * H], d2[k], acc);
            d1[i] = RELU_ON(a1[i]) ? acc : (REAL)0;
        }
        double* gp = g;
        for (int k = 0; k < d; ++k) for (int j = 0; j < H; ++j) gp[j + (size_t)k * H] += (double)d1[j] * (double)x0[k];
        gp += (size_t)H * d;
        for (int j = 0; j < H; ++j) gp[j] += (double)d1[j];
        gp += H;
        for (int k = 0; k < H; ++k) for (int j = 0; j < H; ++j) gp[j + (size_t)k * H] += (double)d2[j] * (double)a1[k];
        gp += (size_t)H * H;
        for (int j = 0; j < H; ++j) gp[j] += (double)d2[j];
        gp += H;
        for (int k = 0; k < H; ++k) gp[k] += (double)U * (double)a2[k];
        gp += H;
        gp[0] += (double)U;
    }
    return o[0];
}

/* g(X) = log(0.5 + 0.5 |X|^2) (lambaem.jl:14), the log through the ARITH-SPEC double kernel */
static REAL NAME(gfun)(const REAL* X, int d) {
    const REAL S = NAME(sumsq)(X, d);
    return (REAL)udeo_log((double)FMA((REAL)0.5, S, (REAL)0.5));
}

static int NAME(loss_grad)(const udeo_hjb_desc* D, int64_t M, const REAL* x0, const REAL* theta, uint32_t iter, double* loss,
                           REAL* grad, REAL* u0_out, REAL* uT, REAL* XT, double* loss_traj, int64_t* stats, int32_t* retcode,
                           int32_t nthreads) {
    if (!D || !x0 || !theta || M <= 0 || D->d < 1 || D->d > 127 || D->hls < 1 || D->hls > 128) return -1;
    const int d = D->d, H = D->hls;
    int32_t np0, np1;
    udeo_hjb_num_params(d, H, &np0, &np1);
    NAME(Nets) n;
    NAME(nets_init)(&n, d, H, theta);
    NAME(Par) p;
    NAME(par_init)(&p, D);
    const REAL u0 = NAME(u0_net)(&n, x0, (REAL)0, NULL);
    if (u0_out) *u0_out = u0;
    REAL dt_init = (REAL)D->dt;
    if (D->adaptive && !(D->dt > 0)) dt_init = NAME(initdt)(&n, &p, x0, u0);
    if (!D->adaptive && !(D->dt > 0)) return -1;
    double* gsum = grad ? (double*)calloc((size_t)np0 + np1, sizeof(double)) : NULL;
    double total = 0.0, Usum = 0.0;
    int nfail = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        double* gl = grad ? (double*)calloc((size_t)np1, sizeof(double)) : NULL;
        const int cap = p.cap;
        REAL* rt = (REAL*)malloc(sizeof(REAL) * cap);
        REAL* rdt = (REAL*)malloc(sizeof(REAL) * cap);
        REAL* rX = (REAL*)malloc(sizeof(REAL) * (size_t)cap * d);
        REAL* rW = (REAL*)malloc(sizeof(REAL) * (size_t)cap * d);
        double ltot = 0.0, lU = 0.0;
        int lfail = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t j = 0; j < M; ++j) {
            NAME(TrajStat) st;
            REAL Xf[128], uf;
            NAME(traj)(D, &n, &p, x0, u0, dt_init, iter, (uint32_t)j, Xf, &uf, rt, rdt, rX, rW, NULL, &st);
            if (stats) {
                stats[j * UDEO_HJB_NSTATS + 0] = st.nf;
                stats[j * UDEO_HJB_NSTATS + 1] = st.nacc;
                stats[j * UDEO_HJB_NSTATS + 2] = st.nrej;
                stats[j * UDEO_HJB_NSTATS + 3] = st.ndraw;
            }
            if (retcode) retcode[j] = st.ret;
            if (uT) uT[j] = uf;
            if (XT) for (int c = 0; c < d; ++c) XT[(size_t)j * d + c] = Xf[c];
            if (st.ret != UDEO_HJB_RET_SUCCESS) {
                lfail += 1;
                if (loss_traj) loss_traj[j] = 0.0;
                continue;
            }
            const REAL e = NAME(gfun)(Xf, d) - uf;
            const REAL lj = e * e;
            if (loss_traj) loss_traj[j] = (double)lj;
            ltot += (double)lj;
            if (grad) {
                const REAL ubar = ((REAL)-2 * e) / (REAL)M; /* d(mean_j e_j^2)/d(u_T) */
                lU += (double)ubar;
                for (int s = 0; s < st.nacc; ++s) NAME(step_bwd)(&n, &p, rX + (size_t)s * d, rt[s], rdt[s], rW + (size_t)s * d, ubar, gl);
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total += ltot;
            Usum += lU;
            nfail += lfail;
            if (grad) for (int i = 0; i < np1; ++i) gsum[np0 + i] += gl[i];
        }
        free(gl); free(rt); free(rdt); free(rX); free(rW);
    }
    if (loss) *loss = nfail ? INFINITY : total / (double)M;
    if (grad) {
        NAME(u0_net)(&n, x0, (REAL)Usum, gsum);
        for (int i = 0; i < np0 + np1; ++i) grad[i] = (REAL)gsum[i];
        free(gsum);
    }
    return nfail ? -5 : 0;
}
