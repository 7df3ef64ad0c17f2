// ude_model_node.h -- the pure neural ODE of the SEIR script (included by ude_models.h, Float64 translation units only).
//
// dudt_node(u,p,t)  SEIR_exposure/seir_exposure.jl:53-66:
//   ann_node = FastChain(FastDense(7,64,tanh), FastDense(64,64,tanh), FastDense(64,64,tanh), FastDense(64,7))
//   dS,dE,dI,dR,dD = ann_node([S/N,E,I,R,N,D/N,C], p)   (the FIRST FIVE of the seven outputs)
//   dN = -mu*N ; dC = sigma*E
// theta = initial_params layout [W1 (64x7, column-major); b1; W2 (64x64); b2; W3 (64x64); b3; W4 (7x64); b4] = 9287.
// trained through concrete_solve(prob_node, Vern7(), ...; sensealg = InterpolatingAdjoint(...)) (seir_exposure.jl:69-73).
//
// One wavefront per trajectory, lane j = hidden neuron j of all three hidden layers (the SeirUde<64> scheme with two
// 64x64 layers): W2 and W3 sit in padded LDS copies shared by the block's trajectories (row AND column reads
// conflict-free), activations cross lanes through a wave-private LDS broadcast row, 64-term hidden dots follow the
// ARITH-SPEC wide-dot rule (4 blocks of 16, block sums left to right), reductions to replicated scalars (the 7-row
// output layer, the input cotangent) are wavefront tree sums.  The parameter cotangent is DEFERRED: every adjoint stage
// stores its factors (a1 a2 delta2 delta3 per lane and x, delta4 once per wave in LDS; a3 and delta1 per lane in HBM) and the RK-weighted
// sums of the 146 slots per lane are formed at the end of the step, fused with the error norm and the candidate mu
// (mu itself in HBM, two columns that swap on acceptance).
#pragma once
#ifndef NODE_BARRIER
#define NODE_BARRIER asm volatile("" ::: "memory")
#endif

namespace ude {

template <int G>
struct SeirNode {
    static_assert(G == 64, "neural-ODE kernel: one wavefront per trajectory");
    static constexpr int H = 64, NIN = 7, NOUT = 7, NBLK = 4;
    static constexpr int NS = 7;
    static constexpr int NEXTRA = NIN + 3 + NOUT + 1;  // W1[j,:], b1, b2, b3, W4[:,j], b4[j] (lanes 0..6)
    static constexpr int NSL = 2 * H + NEXTRA;          // 146 slots per lane
    static constexpr bool STATE_DISTRIBUTED = false;
    static constexpr bool THETA_GLOBAL = false, FUSED_ACC = true, SLOTS_GLOBAL = true, CPL = true, DEFERRED = true;
    static constexpr bool DADJ_K_FROM_DENSE = false;
    // Stage factors: only the stages whose B or BT weight is nonzero are kept (Tsit5: all 7; Vern7: 8 of 10), in COMPACTED slots
    // (AdjSys passes popcount(mask below s); a stage nobody reads lands in the slot of the next kept stage and is overwritten by
    // it).  Per stage FOUR full rows stay in LDS -- a1 and a2 (read by every lane: the broadcast factors of the W2 / W3 blocks),
    // delta2 and delta3 -- plus one short row (x0..x6 | delta4_0..6: 14 words); a3 and delta1, which only the lane that wrote
    // them reads (the 15 slots of the W1 / b1 / W4 blocks), live in a thread-strided HBM workspace behind the mu columns
    // (GFAC words per thread).  With that FOUR wavefronts fit next to the 74 KB of weights -- 4 x (8 x 2.2 KB + 1 KB) + 74 KB +
    // stage storage = 151 KB of the CU's 160 KB: every SIMD has its wavefront (the kernels use all 512 registers of a SIMD lane;
    // round 2 kept six rows in LDS and ran three wavefronts per CU)
    static constexpr bool COMPACT_STAGES = true;
    static constexpr int NSTG = 10, NSTC = 8, WPB = 4, RX = 16;   // tableau stages, stored stages, wavefronts per block, short row
    static constexpr int NROW = 4;                                 // LDS rows per stored stage: a1 a2 delta2 delta3
    static constexpr int STG = NROW * H + RX;                      // doubles of one stored stage
    static constexpr int GFAC = 2 * NSTC;                          // HBM words per thread: (a3, delta1) of every stored stage
    static constexpr int LD = 65;
    static constexpr int OFF_W1 = 0, OFF_B1 = NIN * H, OFF_W2 = OFF_B1 + H, OFF_B2 = OFF_W2 + H * H, OFF_W3 = OFF_B2 + H,
                         OFF_B3 = OFF_W3 + H * H, OFF_W4 = OFF_B3 + H, OFF_B4 = OFF_W4 + NOUT * H, NPARAM = OFF_B4 + NOUT;
    static_assert(NPARAM == 9287, "7-64-64-64-7");
    static constexpr int SCRATCH = WPB * (NSTC * STG + 2 * H);  // stage factors + 2 broadcast rows per wavefront
    static constexpr int FWD_BLOCK_THREADS = 192;               // forward / rhs kernels: three wavefronts per block ...
    static constexpr int SCRATCH_FWD = (FWD_BLOCK_THREADS / 64) * 2 * H;   // ... their broadcast rows only
    static constexpr int FWD_BLOCKS = 2;                        // ... and two blocks per CU: 2 x 80 KB of LDS, 256 registers (six wavefronts per CU)
    typedef __attribute__((address_space(3))) double lds_t;
    struct Ctx {
        double b1, b2, b3, b4[NOUT];
        const lds_t *W2p, *W3p;  // LDS, ld = 65
        const lds_t* wx;         // LDS: this lane's column of the narrow layers, W1[j, m] at wx[m*H] (m < 7), W4[i, j] at wx[(7+i)*H]
                                 // (28 registers less per lane than a register copy; the adjoint kernels are at the 512-register limit)
        lds_t* bc;               // two wave-private broadcast rows: lane j writes, every lane reads all 64
        lds_t* fac;              // stage factors of this wavefront: row f < 4 (a1 a2 delta2 delta3) of stored stage q at fac[q*STG + f*H + lane], the short row at fac[q*STG + 4*H]
        double* gfac;            // HBM: a3 of stored stage q at gfac[(2q) * gms], delta1 at gfac[(2q + 1) * gms] (this thread's words)
        int gms;
        double mu_c, sg;
        int j, r;
    };
    static constexpr int NWX = (NIN + NOUT) * H;  // narrow-layer table
    static __host__ __device__ constexpr int theta_lds(int) { return 2 * H * LD + NWX; }
    static __device__ __forceinline__ void stage_theta(double* th, const double* theta, int, int tid, int nthreads) {
        for (int i = tid; i < H * H; i += nthreads) {
            th[(i % H) + (i / H) * LD] = theta[OFF_W2 + i];
            th[H * LD + (i % H) + (i / H) * LD] = theta[OFF_W3 + i];
        }
        for (int i = tid; i < NWX; i += nthreads) {
            const int q = i / H, j = i % H;
            th[2 * H * LD + i] = q < NIN ? theta[OFF_W1 + j + q * H] : theta[OFF_W4 + (q - NIN) + j * NOUT];
        }
    }
    static __device__ __forceinline__ void init(Ctx& c, double* th, double* scratch, double*, int, const ModelConsts& mc, int r,
                                                const double* theta_g) {
        const int j = r & 63;
        c.j = j; c.r = r;
        c.W2p = (const lds_t*)th;
        c.W3p = (const lds_t*)th + H * LD;
        const int wv = (threadIdx.x >> 6) % WPB;
        c.bc = (lds_t*)scratch + wv * 2 * H;  // (the broadcast rows first: all the forward kernels need of the scratch)
        c.fac = (lds_t*)scratch + WPB * 2 * H + wv * (NSTC * STG);
        c.wx = (const lds_t*)th + 2 * H * LD + j;
        c.b1 = theta_g[OFF_B1 + j]; c.b2 = theta_g[OFF_B2 + j]; c.b3 = theta_g[OFF_B3 + j];
        static_for<0, NOUT>([&](auto i) { c.b4[i] = uniform_real(theta_g[OFF_B4 + decltype(i)::value]); });
        c.mu_c = mc.consts[4]; c.sg = mc.consts[5];
    }
    // 64-term hidden dot of neuron j (row j of W, or column j when TRANSPOSED): 4 blocks of 16, left to right
    template <bool TRANSPOSED>
    static __device__ __forceinline__ double hidden_dot(const Ctx& c, const lds_t* W, double v) {
        double tot = 0.0;
        const lds_t* wp = TRANSPOSED ? W + c.j * LD : W + c.j;
        lds_t* row = c.bc + (TRANSPOSED ? H : 0);
        row[c.j] = v;
#pragma unroll 1
        for (int b = 0; b < NBLK; ++b) {
            double acc = 0.0;
            static_for<0, 16>([&](auto ic) {
                const int k = b * 16 + decltype(ic)::value;
                const double x = row[k];
                const double wv = TRANSPOSED ? wp[k] : wp[k * LD];
                acc = __builtin_fma(wv, x, acc);
            });
            tot = b == 0 ? acc : tot + acc;
        }
        return tot;
    }
    struct Act {
        double x[NIN], a1, a2, a3;
    };
    // forward network: out[0..4] (the outputs the script uses), replicated
    static __device__ __forceinline__ void net(const Ctx& c, Act& q, double* out) {
        double z1 = 0.0;
        static_for<0, NIN>([&](auto k) { z1 = __builtin_fma(c.wx[(int)decltype(k)::value * H], q.x[k], z1); });
        z1 += c.b1;
        q.a1 = dtanh(z1);
        q.a2 = dtanh(hidden_dot<false>(c, c.W2p, q.a1) + c.b2);
        q.a3 = dtanh(hidden_dot<false>(c, c.W3p, q.a2) + c.b3);
        static_for<0, 5>([&](auto i) { out[i] = wave_tree_sum(c.wx[(NIN + (int)decltype(i)::value) * H] * q.a3) + c.b4[i]; });
    }
    static __device__ __forceinline__ void inputs(const double* u, Act& q) {
        const double S = u[0], N = u[4], D = u[5];
        q.x[0] = S / N; q.x[1] = u[1]; q.x[2] = u[2]; q.x[3] = u[3]; q.x[4] = N; q.x[5] = D / N; q.x[6] = u[6];
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        Act q;
        double o[5];
        inputs(u, q);
        net(c, q, o);
        du[0] = o[0]; du[1] = o[1]; du[2] = o[2]; du[3] = o[3];
        du[4] = -c.mu_c * u[4];
        du[5] = o[4];
        du[6] = c.sg * u[1];
    }
    struct Bwd {
        Act f;
        double d1, d2, d3, d4[NOUT];
    };
    static __device__ __forceinline__ void sweep(const Ctx& c, const double* u, const double* lam, double* dlam, Bwd& q) {
        const double S = u[0], N = u[4], D = u[5];
        double o[5];
        inputs(u, q.f);
        net(c, q.f, o);
        q.d4[0] = lam[0]; q.d4[1] = lam[1]; q.d4[2] = lam[2]; q.d4[3] = lam[3]; q.d4[4] = lam[5]; q.d4[5] = 0.0; q.d4[6] = 0.0;
        double s3 = 0.0;  // column j of the 7-row output layer against delta4 (7-term chain, the zero rows included)
        static_for<0, NOUT>([&](auto i) { s3 = __builtin_fma(c.wx[(NIN + (int)decltype(i)::value) * H], q.d4[i], s3); });
        q.d3 = s3 * __builtin_fma(-q.f.a3, q.f.a3, 1.0);
        q.d2 = hidden_dot<true>(c, c.W3p, q.d3) * __builtin_fma(-q.f.a2, q.f.a2, 1.0);
        q.d1 = hidden_dot<true>(c, c.W2p, q.d2) * __builtin_fma(-q.f.a1, q.f.a1, 1.0);
        double gx[NIN];
        static_for<0, NIN>([&](auto m) { gx[m] = wave_tree_sum(c.wx[(int)decltype(m)::value * H] * q.d1); });
        dlam[0] = gx[0] / N;
        dlam[1] = __builtin_fma(c.sg, lam[6], gx[1]);
        dlam[2] = gx[2];
        dlam[3] = gx[3];
        dlam[4] = ((gx[4] - gx[0] * S / (N * N)) - gx[5] * D / (N * N)) - c.mu_c * lam[4];
        dlam[5] = gx[5] / N;
        dlam[6] = gx[6];
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const double* u, const double* lam, double* dlam, double*) {
        static_assert(!WANT_PARAM, "deferred / fused parameter cotangent only");
        Bwd q;
        sweep(c, u, lam, dlam, q);
    }
    // extra slot e (0..17) of lane j at one stage: the cotangent (df/dtheta)^T lam (positive sign)
    template <int E>
    static __device__ __forceinline__ double extra_value(const Ctx& c, const Bwd& q) {
        if constexpr (E < NIN) return q.d1 * q.f.x[E];
        else if constexpr (E == NIN) return q.d1;
        else if constexpr (E == NIN + 1) return q.d2;
        else if constexpr (E == NIN + 2) return q.d3;
        else if constexpr (E < NIN + 3 + NOUT) return q.d4[E - NIN - 3] * q.f.a3;
        else {
            double v = 0.0;
            static_for<0, NOUT>([&](auto i) { v = (c.j == (int)decltype(i)::value) ? q.d4[i] : v; });
            return v;
        }
    }
    static __device__ __forceinline__ int extra_index(int e, int j) {
        if (e < NIN) return OFF_W1 + j + e * H;
        if (e == NIN) return OFF_B1 + j;
        if (e == NIN + 1) return OFF_B2 + j;
        if (e == NIN + 2) return OFF_B3 + j;
        if (e < NIN + 3 + NOUT) return OFF_W4 + (e - NIN - 3) + j * NOUT;
        return j < NOUT ? OFF_B4 + j : -1;
    }
    // fused accumulation (the discrete reverse sweep): ab[s] = fma(bs, -g_s, ab[s]) over this lane's 146 parameters
    template <bool WANT_E>
    static __device__ __forceinline__ void vjp_acc(const Ctx& c, const double* u, const double* lam, double* dlam, double* ab,
                                                   double* ae, double bs, double es) {
        Bwd q;
        sweep(c, u, lam, dlam, q);
        auto upd = [&](auto sc, double gpos) {
            constexpr int s = sc;
            ab[s] = __builtin_fma(bs, -gpos, ab[s]);
            if constexpr (WANT_E) ae[s] = __builtin_fma(es, -gpos, ae[s]);
        };
        static_for<0, H>([&](auto k) { upd(k, q.d2 * readlane_real(q.f.a1, decltype(k)::value)); });
        static_for<0, H>([&](auto k) { upd(std::integral_constant<int, H + decltype(k)::value>{}, q.d3 * readlane_real(q.f.a2, decltype(k)::value)); });
        static_for<0, NEXTRA>([&](auto e) { upd(std::integral_constant<int, 2 * H + decltype(e)::value>{}, extra_value<decltype(e)::value>(c, q)); });
    }
    // ---- deferred parameter cotangent ----
    // stage s: state cotangent out; factors a1 a2 a3 delta1 delta2 delta3 per lane, row 6 = x0..x6 | delta4_0..6 (lanes 0..13)
    static __device__ __forceinline__ void vjp_store(const Ctx& c, const double* u, const double* lam, double* dlam, int s) {
        Bwd q;
        sweep(c, u, lam, dlam, q);
        lds_t* f = c.fac + s * STG + c.j;  // (s: compacted slot)
        f[0] = q.f.a1; f[H] = q.f.a2; f[2 * H] = q.d2; f[3 * H] = q.d3;
        c.gfac[(size_t)(2 * s) * c.gms] = q.f.a3;
        c.gfac[(size_t)(2 * s + 1) * c.gms] = q.d1;
        double sh = 0.0;
        static_for<0, NIN>([&](auto m) { sh = (c.j == (int)decltype(m)::value) ? q.f.x[m] : sh; });
        static_for<0, NOUT>([&](auto i) { sh = (c.j == NIN + (int)decltype(i)::value) ? q.d4[i] : sh; });
        c.fac[s * STG + NROW * H + (c.j < RX ? c.j : RX - 1)] = sh;  // (lanes >= 14 all carry 0.0: one word, one value)
    }
    // this lane's delta of ONE layer at all stages (registers): the 64 slots of a weight block share it; the 18 extras
    // read their factors straight from the lane's own LDS words (one use each -- keeping all four per-lane factor sets
    // of ten stages in registers next to the chunked mu prefetch spilled 221 VGPRs)
    struct Fac {
        double d[NSTG];
    };
    template <unsigned MASK>
    static constexpr int slot_of(int s) {  // compacted slot of stage s: kept stages below it
        int q = 0;
        for (int i = 0; i < s; ++i) q += (MASK >> i) & 1u;
        return q;
    }
    template <int NST, unsigned MASK, int FIELD>
    static __device__ __forceinline__ void load_factor(const Ctx& c, Fac& f) {
        static_for<0, NST>([&](auto s) {
            if constexpr ((MASK >> decltype(s)::value) & 1u) f.d[s] = c.fac[slot_of<MASK>(decltype(s)::value) * STG + FIELD * H + c.j];
        });
    }
    // g_s (all stored stages) of slot k of the W2 block (LAYER = 0: -(delta2_j a1_k)) / the W3 block (1: -(delta3_j a2_k))
    template <int NST, unsigned MASK, int LAYER>
    static __device__ __forceinline__ void g_w(const Ctx& c, const Fac& f, int k, double* g) {
        static_for<0, NST>([&](auto s) {
            if constexpr ((MASK >> decltype(s)::value) & 1u) g[s] = -(f.d[s] * c.fac[slot_of<MASK>(decltype(s)::value) * STG + LAYER * H + k]);
        });
    }
    // a3 / delta1 of all stored stages (this lane's own words in the HBM workspace)
    template <int NST, unsigned MASK, int WHICH>
    static __device__ __forceinline__ void load_gfac(const Ctx& c, Fac& f) {
        static_for<0, NST>([&](auto s) {
            if constexpr ((MASK >> decltype(s)::value) & 1u) f.d[s] = c.gfac[(size_t)(2 * slot_of<MASK>(decltype(s)::value) + WHICH) * c.gms];
        });
    }
    template <int NST, unsigned MASK, int E>
    static __device__ __forceinline__ void g_extra(const Ctx& c, const Fac& a3, const Fac& d1, double* g) {
        static_for<0, NST>([&](auto s) {
            if constexpr ((MASK >> decltype(s)::value) & 1u) {
                const lds_t* own = c.fac + slot_of<MASK>(decltype(s)::value) * STG + c.j;       // a1 a2 delta2 delta3 of this lane
                const lds_t* p = c.fac + slot_of<MASK>(decltype(s)::value) * STG + NROW * H;   // x0..x6 | delta4_0..6
                double v;
                if constexpr (E < NIN) v = -(d1.d[s] * p[E]);
                else if constexpr (E == NIN) v = -d1.d[s];
                else if constexpr (E == NIN + 1) v = -own[2 * H];
                else if constexpr (E == NIN + 2) v = -own[3 * H];
                else if constexpr (E < NIN + 3 + NOUT) v = -(p[NIN + (E - NIN - 3)] * a3.d[s]);
                else v = c.j < NOUT ? -p[NIN + c.j] : -0.0;
                g[s] = v;
            }
        });
    }
    // slots in order 0..145, mu read in chunks of CH (next chunk in flight while this one is processed)
    template <int NST, unsigned MASK, class Body>
    static __device__ __forceinline__ void for_each_slot(const Ctx& c, const double* mu, int ms, Body body) {
#ifndef NODE_CH
#define NODE_CH 8
#endif
        constexpr int CH = NODE_CH;  // slots in flight per chunk
        constexpr int NCHX = (NEXTRA + CH - 1) / CH;
        double mcur[CH], mnext[CH];
        static_for<0, CH>([&](auto i) { mcur[i] = mu[(size_t)decltype(i)::value * ms]; });
        auto fetch = [&](int first) {
            static_for<0, CH>([&](auto i) {
                const int sl = first + decltype(i)::value;
                mnext[i] = sl < NSL ? mu[(size_t)sl * ms] : 0.0;
            });
        };
        auto roll = [&]() { static_for<0, CH>([&](auto i) { mcur[i] = mnext[i]; }); };
        static_for<0, 2>([&](auto layer) {
            constexpr int LAYER = decltype(layer)::value;
            Fac f;
            load_factor<NST, MASK, 2 + LAYER>(c, f);  // delta2 / delta3
#pragma unroll 1
            for (int k0 = 0; k0 < H; k0 += CH) {
                fetch(LAYER * H + k0 + CH);
                static_for<0, CH>([&](auto i) {
                    double g[NST];
                    g_w<NST, MASK, LAYER>(c, f, k0 + decltype(i)::value, g);
                    body(LAYER * H + k0 + decltype(i)::value, g, mcur[i]);
                });
                roll();
            }
        });
        Fac fa3, fd1;
        load_gfac<NST, MASK, 0>(c, fa3);
        load_gfac<NST, MASK, 1>(c, fd1);
        static_for<0, NCHX>([&](auto chunk) {
            constexpr int e0 = decltype(chunk)::value * CH;
            if constexpr (e0 + CH < NEXTRA) fetch(2 * H + e0 + CH);
            static_for<0, CH>([&](auto i) {
                constexpr int e = e0 + decltype(i)::value;
                if constexpr (e < NEXTRA) {
                    double g[NST];
                    g_extra<NST, MASK, e>(c, fa3, fd1, g);
                    body(2 * H + e, g, mcur[i]);
                    NODE_BARRIER;
                }
            });
            if constexpr (e0 + CH < NEXTRA) roll();
        });
    }
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ double step_slots(const Ctx& c, const double* B, const double* BT, double dt, double abstol,
                                                        double reltol, const double* mu, double* mu_new, int ms) {
        static_assert(MASK & 1u, "the first stage starts the chains");
        double bb[NST], bt[NST];
        static_for<0, NST>([&](auto s) { bb[s] = uniform_real(B[s]); bt[s] = uniform_real(BT[s]); });
        double ps = 0.0;
        for_each_slot<NST, MASK>(c, mu, ms, [&](int slot, const double* g, double m0) {
            double ab = bb[0] * g[0], ae = bt[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) {
                    ab = __builtin_fma(bb[s], g[s], ab);
                    ae = __builtin_fma(bt[s], g[s], ae);
                }
            });
            const double m1 = __builtin_fma(dt, ab, m0);
            mu_new[(size_t)slot * ms] = m1;
            const double a0 = fabs(m0), a1 = fabs(m1);
            const double res = (dt * ae) / __builtin_fma((a0 > a1 ? a0 : a1), reltol, abstol);
            ps = __builtin_fma(res, res, ps);
        });
        return ps;
    }
    // fast adjoint mode: mu += dt * sum_s B_s g_s in place, on accepted steps only (no error half, no division)
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ void commit_slots(const Ctx& c, const double* B, double dt, double* mu, int ms) {
        static_assert(MASK & 1u, "the first stage starts the chains");
        double bb[NST];
        static_for<0, NST>([&](auto s) { bb[s] = uniform_real(B[s]); });
        for_each_slot<NST, MASK>(c, mu, ms, [&](int slot, const double* g, double m0) {
            double ab = bb[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) ab = __builtin_fma(bb[s], g[s], ab);
            });
            mu[(size_t)slot * ms] = __builtin_fma(dt, ab, m0);
        });
    }
    static __device__ __forceinline__ void init_norm01(const Ctx& c, double abstol, double reltol, const double* mu, int ms,
                                                       double& h0, double& l0, double& h1, double& l1) {
        for_each_slot<1, 1u>(c, mu, ms, [&](int, const double* g, double m) {
            const double sk = __builtin_fma(fabs(m), reltol, abstol);
            const double q0 = m / sk, q1 = g[0] / sk;
            dd_acc(h0, l0, q0 * q0);
            dd_acc(h1, l1, q1 * q1);
        });
    }
    static __device__ __forceinline__ void init_norm2(const Ctx& c, double abstol, double reltol, const double* mu, int ms,
                                                      double& h2, double& l2) {
        for_each_slot<2, 3u>(c, mu, ms, [&](int, const double* g, double m) {
            const double sk = __builtin_fma(fabs(m), reltol, abstol);
            const double q = (g[1] - g[0]) / sk;
            dd_acc(h2, l2, q * q);
        });
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts&, int r, int s) {
        const int j = r & 63;
        if (s < H) return OFF_W2 + j + s * H;
        if (s < 2 * H) return OFF_W3 + j + (s - H) * H;
        return extra_index(s - 2 * H, j);
    }
};

}  // namespace ude
