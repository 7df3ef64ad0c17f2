// ude_model_kpp_vec.h -- nn_ode on LARGE grids (BASELINE configs[3]: 1024 points), round 6: the pointwise network on the VECTOR
// unit, one grid point per lane, weights broadcast out of registers (DPP); only the parameter contraction stays on the FP64 matrix cores.
//
// Why (tools/probe/mfma_valu_overlap_probe.hip, measured on the MI355X): v_mfma_f64_16x16x4 and vector instructions do NOT overlap
// on gfx950 -- neither inside a wavefront nor between the two wavefronts of a SIMD (4 MFMA = 256 clocks, 64 v_fma_f64 = 285, both
// interleaved = 573; one wavefront of each kind on a SIMD = 549; v_add_u32 instead of v_fma_f64: the same).  The FP64 matrix
// instruction occupies the vector issue port for its 64 cycles, and its peak (78.6 TF) IS the vector peak: a matrix product buys no
// arithmetic, only operand sharing between lanes.  The round 1-5 kernel (KppUdeW, ude_models.h) ran the four Dense layers
// 1-10-20-10-1 as zero-padded 16x16x4 tiles -- 30 + 24 matrix instructions per 16 points, 2.7x the algorithmic flop, every cycle of
// it in front of, not under, the `tanh` work.  A pointwise network needs no operand sharing between points at all:
//
//   forward / reverse sweep   lane = grid point; z_i = fma(W[i][k], a_k, z_i) as v_fmac_f64_dpp with W[i][k] BROADCAST FROM A REGISTER:
//                             the network's 461 numbers live in 29 VGPR pairs (theta index f in lane f % 16 of every 16-lane row of
//                             pair f / 16), and `row_newbcast:k` -- the one DPP control the FP64 vector unit has -- hands lane k of
//                             its row to every lane as the multiplier, at the plain instruction's rate (tools/probe/dpp_f64_probe.hip:
//                             4.56 clocks, v_fma_f64 with three register sources: 5.96).  420 + 420 fused multiply-adds per point, no
//                             padding, no transposes, no loads: weights, activations and deltas never leave the registers
//   contraction               dW_l = sum over points of delta_l [a_l ; 1]^T is the one product that DOES share operands between
//                             lanes: v_mfma_f64_16x16x4 over a wave-private [16 points][row] LDS tile, the outer products of all
//                             four layers AND the four stencil sums PACKED into 4 output tiles (KppUdeW: 6 tiles + 1 for the stencil
//                             sums): a tile is any 16 delta rows x any 16 [a ; 1] columns; every output element is its own chain, the
//                             cross terms nobody asked for are never written out
//
// ARITH-SPEC is untouched: every chain is the chain the matrix instruction executed (fma from +0 in ascending index order; a zero
// padding term was an exact no-op), the parameter sums are fused chains over the 256 consecutive points of a wavefront in ascending
// order, block sums added left to right.  Same bits as KppUdeW and the oracle (tests/test_gpu_parity.py, tests/test_gpu_fuzz.py).
// Reference: FisherKPP/Fisher-KPP-CNN.jl:92-126 (rx_nn, nn_ode).
#pragma once

namespace ude {

// what an adjoint evaluation does with its hidden activations (AdjSys::ACT_CACHE, ude_kernels.h): nothing, store them for the next evaluation
// at the same time, or load the ones the previous evaluation at this time stored
enum { ACT_NONE = 0, ACT_STORE = 1, ACT_LOAD = 2 };

#ifndef UDE_F32
// greedy first-fit-decreasing packing of the layers' outer-product rectangles into 16 x 16 output tiles (compile time)
template <int MAXT>
struct KppTilePack {
    int nt = 0;
    int nr[MAXT] = {}, nc[MAXT] = {};
    int rows[MAXT][16] = {}, cols[MAXT][16] = {};
};

template <class Net>
struct KppVecLayout {
    static constexpr int L = Net::L;
    // rows of the [point][row] transposition tile: a_0 = u | a_1 .. a_{L-1} | delta_0 .. delta_{L-1} | 1 | u_{i-1} | u_{i+1} | stencil sum
    static constexpr int a_row(int l) { int o = 0; for (int i = 0; i < l; ++i) o += Net::dim(i); return o; }
    static constexpr int ROWS_A = a_row(L);
    static constexpr int d_rel(int l) { int o = 0; for (int i = 0; i < l; ++i) o += Net::dim(i + 1); return o; }
    static constexpr int ROWS_D = d_rel(L);
    static constexpr int d_row(int l) { return ROWS_A + d_rel(l); }
    static constexpr int ROW_ONE = ROWS_A + ROWS_D, ROW_UM = ROW_ONE + 1, ROW_UP = ROW_ONE + 2, ROW_COMB = ROW_ONE + 3, NROW = ROW_ONE + 4;
    static constexpr int RS = NROW | 1;   // odd row stride
    static constexpr int TILE = 16 * RS;

    // ---- the outer products the gradient needs, as rectangles (row chunk of delta_l) x (column chunk of [a_l ; 1 (; stencil columns)])
    static constexpr int MAXT = 16;
    static constexpr int ncols_of(int l) { return Net::dim(l) + 1 + (l == L - 1 ? (L - 1 == 0 ? 3 : 4) : 0); }
    static constexpr int col_of(int l, int j) {   // LDS row of column j of layer l's rectangle
        if (j < Net::dim(l)) return a_row(l) + j;
        if (j == Net::dim(l)) return ROW_ONE;
        // the last layer's delta row is lambda itself: lambda x (u_{i-1}, u_i, u_{i+1}, w1 u_{i-1} + w2 u_i + w3 u_{i+1}) are the stencil sums
        const int e = j - Net::dim(l) - 1;
        if (L - 1 == 0) return e == 0 ? ROW_UM : e == 1 ? ROW_UP : ROW_COMB;   // (u itself is a_0)
        return e == 0 ? ROW_UM : e == 1 ? 0 : e == 2 ? ROW_UP : ROW_COMB;
    }
    static constexpr KppTilePack<MAXT> make_pack() {
        struct Rect { int l, r0, nr, c0, nc; };
        Rect rc[64] = {};
        int n = 0;
        for (int l = 0; l < L; ++l)
            for (int r0 = 0; r0 < Net::dim(l + 1); r0 += 16)
                for (int c0 = 0; c0 < ncols_of(l); c0 += 16) {
                    const int nr = Net::dim(l + 1) - r0 < 16 ? Net::dim(l + 1) - r0 : 16, nc = ncols_of(l) - c0 < 16 ? ncols_of(l) - c0 : 16;
                    rc[n++] = Rect{l, r0, nr, c0, nc};
                }
        for (int i = 1; i < n; ++i)   // insertion sort by area, descending (stable)
            for (int j = i; j > 0 && rc[j].nr * rc[j].nc > rc[j - 1].nr * rc[j - 1].nc; --j) { const Rect t = rc[j]; rc[j] = rc[j - 1]; rc[j - 1] = t; }
        KppTilePack<MAXT> pk;
        for (int i = 0; i < n; ++i) {
            const Rect& q = rc[i];
            int placed = -1;
            for (int t = 0; t <= pk.nt && placed < 0 && t < MAXT; ++t) {
                // rows / columns of the rectangle the tile does not hold yet
                int addr = 0, addc = 0;
                for (int a = 0; a < q.nr; ++a) {
                    bool have = false;
                    for (int b = 0; b < pk.nr[t]; ++b) have = have || pk.rows[t][b] == d_row(q.l) + q.r0 + a;
                    addr += have ? 0 : 1;
                }
                for (int a = 0; a < q.nc; ++a) {
                    bool have = false;
                    for (int b = 0; b < pk.nc[t]; ++b) have = have || pk.cols[t][b] == col_of(q.l, q.c0 + a);
                    addc += have ? 0 : 1;
                }
                if (pk.nr[t] + addr <= 16 && pk.nc[t] + addc <= 16) placed = t;
            }
            const int t = placed;
            if (t == pk.nt) pk.nt += 1;
            for (int a = 0; a < q.nr; ++a) {
                bool have = false;
                for (int b = 0; b < pk.nr[t]; ++b) have = have || pk.rows[t][b] == d_row(q.l) + q.r0 + a;
                if (!have) pk.rows[t][pk.nr[t]++] = d_row(q.l) + q.r0 + a;
            }
            for (int a = 0; a < q.nc; ++a) {
                bool have = false;
                for (int b = 0; b < pk.nc[t]; ++b) have = have || pk.cols[t][b] == col_of(q.l, q.c0 + a);
                if (!have) pk.cols[t][pk.nc[t]++] = col_of(q.l, q.c0 + a);
            }
        }
        return pk;
    }
    // theta index (relative to theta[0]; nn_offset / stencil_offset / d0_offset are run-time) of the output element (delta row, column):
    //   >= 0: nn-relative index; -2 .. -5: the stencil sums um, u, up, comb; -1: a cross term nobody asked for
    static constexpr int pair_index(int drow, int col) {
        for (int l = 0; l < L; ++l) {
            const int in = Net::dim(l), out = Net::dim(l + 1);
            if (drow >= d_row(l) && drow < d_row(l) + out) {
                const int j = drow - d_row(l);
                if (col >= a_row(l) && col < a_row(l) + in) return Net::off(l) + j + (col - a_row(l)) * out;
                if (col == ROW_ONE) return Net::off(l) + in * out + j;
                if (l == L - 1) {
                    if (col == ROW_UM) return -2;
                    if (col == 0) return -3;
                    if (col == ROW_UP) return -4;
                    if (col == ROW_COMB) return -5;
                }
                return -1;
            }
        }
        return -1;
    }
    static constexpr bool pack_covers(const KppTilePack<MAXT>& pk) {   // every parameter of the network and the four stencil sums are the output of some tile
        bool cov[NROW][NROW] = {};
        for (int t = 0; t < pk.nt; ++t)
            for (int a = 0; a < pk.nr[t]; ++a)
                for (int b = 0; b < pk.nc[t]; ++b) cov[pk.rows[t][a]][pk.cols[t][b]] = true;
        for (int l = 0; l < L; ++l) {
            const int dr = d_row(l), nc = ncols_of(l);
            for (int cj = 0; cj < nc; ++cj) {
                const int col = col_of(l, cj);
                for (int j = 0; j < Net::dim(l + 1); ++j)
                    if (!cov[dr + j][col]) return false;
            }
        }
        return true;
    }
};

// TPTS_: points of a pass the transposition tile holds at once (64: the whole pass; 32: two halves -- Vern7's ten stage rows leave the LDS
// no room for the larger tile)
template <class Net, int NWV_ = 4, int TPTS_ = 64>
struct KppUdeV : LinearTheta {
    static constexpr bool RECOMPUTE_OK = true;
    static constexpr bool PER_MEMBER_THETA = true;   // (weights are streamed from the member's own column)
    static constexpr int NWV = NWV_, G = 64 * NWV, PPL = 1024 / G, TP = 64, BLK = TP * PPL;
    static_assert(NWV == 4 || NWV == 8, "1024 points on four or eight wavefronts");
    static constexpr int TPTS = TPTS_, NH = 64 / TPTS;
    static_assert(TPTS == 64 || TPTS == 32, "tile of 64 or 32 points");
    static constexpr int FWD_BLOCKS = NWV == 8 ? 2 : 1;
    using FwdModel = KppUdeV<Net, 8, TPTS_>;
#ifndef UDE_KPPV_KS_STREAM
#define UDE_KPPV_KS_STREAM 1
#endif
    static constexpr bool KS_STREAM_ALWAYS = UDE_KPPV_KS_STREAM != 0;   // (the interval's stage derivatives are read from the dense store at every evaluation: AdjSys::KS_STREAM)
    static constexpr bool DADJ_K_FROM_DENSE = true;
    static __host__ __device__ constexpr int point(int c, int r) { return (r >> 6) * BLK + c * TP + (r & 63); }
    static_assert(!Net::RT, "compile-time shapes (a run-time shape stays on KppUdeW)");
    static_assert(Net::dim(0) == 1 && Net::dim(Net::L) == 1, "pointwise reaction network R -> R");
    static constexpr int NS = PPL;
    static constexpr int NP = Net::nparam + 5;
    static constexpr int NSL = (NP + G - 1) / G;
    static constexpr bool STATE_DISTRIBUTED = true;
    static constexpr int L = Net::L;
    static constexpr int NPT = G * PPL;
    static constexpr bool acts_ok() {  // the reverse sweep rebuilds act' from the activation VALUE (tanh only)
        for (int l = 0; l + 1 < L; ++l)
            if (Net::act(l) != ACT_TANH) return false;
        return Net::act(L - 1) == ACT_IDENTITY;
    }
    static_assert(acts_ok(), "KppUdeV: tanh hidden layers, linear output");
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) double lds_t;
    static constexpr int NWR = (Net::nparam + 15) / 16;

    using LY = KppVecLayout<Net>;
    static constexpr int ROWS_A = LY::ROWS_A, ROWS_D = LY::ROWS_D, ROW_ONE = LY::ROW_ONE, ROW_UM = LY::ROW_UM, ROW_UP = LY::ROW_UP, ROW_COMB = LY::ROW_COMB, NROW = LY::NROW, MAXT = LY::MAXT;
    static constexpr int a_row(int l) { return LY::a_row(l); }
    static constexpr int d_rel(int l) { return LY::d_rel(l); }
    static constexpr int d_row(int l) { return LY::d_row(l); }
    static constexpr int pair_index(int drow, int col) { return LY::pair_index(drow, col); }
    static constexpr KppTilePack<MAXT> PK = LY::make_pack();
    static constexpr int NT = PK.nt;
    static constexpr int max_slots() { int m = 0; for (int t = 0; t < PK.nt; ++t) m = PK.nr[t] + PK.nc[t] > m ? PK.nr[t] + PK.nc[t] : m; return m; }
    static constexpr int ST = max_slots() | 1;          // odd slot stride of the [64 points][slot] tile of ONE output tile's operands
    static constexpr int TILE = (TPTS * ST + 1) & ~1;
    static_assert(LY::pack_covers(PK), "tile packing must cover every required outer-product element");
    static constexpr int NPP = (NP + 1) & ~1;  // block-sum row (aliases the tile once the points are consumed)
    static_assert(NPP <= TILE, "block sums must fit the tile they alias");
    static constexpr int SCRATCH = 3 * (NPT + 2) + NWV * TILE;  // u, lambda, result rows + tiles
    // round 6: the hidden activations of an adjoint evaluation (ROWS_A - 1 numbers per point) can be handed to the next evaluation at the
    // same time through an HBM row of the block (AdjSys::ACT_CACHE in ude_kernels.h): the second of the two stages at t + dt, and the
    // evaluation after a save-time jump, then skip the forward pass -- 40 `tanh` of the 1-10-20-10-1 net per point -- and read what the
    // stage before them computed from the same u.  Row r of point i at acache[(r - 1) * NPT + i] (lane = point: coalesced).
    static constexpr int ACT_CACHE_WORDS = (ROWS_A - 1) * PPL;
    static constexpr int SCRATCH_FWD = 3 * (NPT + 2);
    struct Ctx {
        double wr[NWR];               // the network's parameters: theta index f in lane f % 16 (of every row) of wr[f / 16]
        double one;                   // 1.0 in a register (the bias is added as fma(b, 1, z) = RN(z + b): the DPP form of the add)
        lds_t *urow, *lrow, *orow, *tile, *part;
        double* acache;               // this block's activation row in HBM (ACT_CACHE_WORDS x block threads doubles; set by adj_kernel)
        double w1, w2, w3, D0;
        int r, lane, w, l16, kq, n, so, d0o, nno;
        int arow[NT], bcol[NT];       // this lane's operand slots of the contraction: delta row of output row l16, [a ; 1] column of output column l16
        int pidx[NT][4];              // theta index of this lane's four output elements of every tile (-1: not an output)
    };
    static __device__ __forceinline__ void init(Ctx& c, double* th_lds, double* scratch, double*, int, const ModelConsts& mc, int r, const double* = nullptr) {
        const double* th = th_lds;   // the block's LDS copy of theta, or -- UDE_PT_THETA -- the member's own column in HBM
        lds_t* sc = (lds_t*)scratch;
        c.urow = sc; c.lrow = sc + NPT + 2; c.orow = sc + 2 * (NPT + 2);
        c.part = sc + 3 * (NPT + 2);
        c.r = r; c.lane = r & 63; c.w = r >> 6; c.l16 = c.lane & 15; c.kq = c.lane >> 4;
        c.tile = c.part + c.w * TILE;
        c.n = mc.n_state; c.so = mc.stencil_offset; c.d0o = mc.d0_offset; c.nno = mc.nn_offset;
        c.w1 = th[c.so]; c.w2 = th[c.so + 1]; c.w3 = th[c.so + 2]; c.D0 = th[c.d0o];
        static_for<0, NWR>([&](auto m) {
            const int f = 16 * decltype(m)::value + c.l16;
            c.wr[m] = f < Net::nparam ? th[mc.nn_offset + f] : 0.0;
        });
        c.one = 1.0;
        asm volatile("" : "+v"(c.one));   // (stays a register: an immediate cannot be the DPP instruction's second source)
        static_for<0, NT>([&](auto tc) {
            constexpr int t = tc;
            // slot of output row / column l16 in the tile's operand rows (rows / columns beyond the tile's: any finite slot -- their
            // outputs are never written out)
            const int ar = c.l16 < PK.nr[t] ? c.l16 : 0, bc = PK.nr[t] + (c.l16 < PK.nc[t] ? c.l16 : 0);
            c.arow[t] = ar; c.bcol[t] = bc;
            static_for<0, 4>([&](auto rr) {
                // output element of register rr: row kq + 4 rr, column l16 (selects, not branches: one pass over the tile's 16 x 4 elements)
                int idx = -1;
                static_for<0, 4>([&](auto kk) {
                    constexpr int ri = decltype(kk)::value + 4 * decltype(rr)::value;
                    if constexpr (ri < PK.nr[t]) {
                        static_for<0, 16>([&](auto e) {
                            if constexpr (decltype(e)::value < PK.nc[t]) {
                                constexpr int pi = pair_index(PK.rows[t][ri], PK.cols[t][decltype(e)::value]);
                                if constexpr (pi != -1) {
                                    const int val = pi >= 0 ? c.nno + pi : pi == -2 ? c.so : pi == -3 ? c.so + 1 : pi == -4 ? c.so + 2 : c.d0o;
                                    idx = (c.kq == decltype(kk)::value && c.l16 == decltype(e)::value) ? val : idx;
                                }
                            }
                        });
                    }
                });
                c.pidx[t][rr] = idx;
            });
        });
    }
    static __device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // the transposition tile is private to the wavefront, and the LDS executes one wavefront's instructions in order: a read behind a store
    // (and a store behind the reads of the rows it overwrites) needs no wait -- only the COMPILER must keep the order.  UDE_KPPV_SYNC=1: the
    // conservative form of the first version, s_waitcnt lgkmcnt(0) on both sides of every tile (6.96 k instead of 6.74 k clocks per pass).
    // Measured and NOT kept (tools/exp/kpp_harness.hip): two half-size buffers with the next half's rows stored between the products of the
    // current one -- LDS stores do not proceed under a running v_mfma_f64 either, and every masked store costs a full one: 8.9 k clocks.
    static __device__ __forceinline__ void wave_sync() {
#if defined(UDE_KPPV_SYNC) && UDE_KPPV_SYNC
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("" ::: "memory");
#endif
        __builtin_amdgcn_wave_barrier();
    }

    // z += W * a with W = lane K of the row of `w` (row_newbcast): ONE v_fmac_f64_dpp.  Hazards the assembler does not see inside an
    // asm statement (a VALU write of `w` in the two instructions in front of it; v_cmpx in the five) are closed by the build's
    // assembly gate (tools/isa_endcf_fix.py: dpp pass), which also fails the build on any DPP control other than row_newbcast.
    template <int K>
    static __device__ __forceinline__ void fmac_bc(double& z, double w, double a) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(w), "v"(a), "n"(K));   // (volatile: the chains of a layer stay interleaved as written -- left alone the scheduler runs each chain to its end, a dependent instruction every slot)
    }
    template <int F>
    static __device__ __forceinline__ void fmac_w(const Ctx& c, double& z, double a) { fmac_bc<(F & 15)>(z, c.wr[F >> 4], a); }

    // forward pass of P points per lane: a[p][a_row(l) + k] = input k of layer l (a[p][0] = u); returns the network output y[p].
    // Chains: z_i = fma(W[i][k], a_k, z_i), k ascending from z_i = +0 -- what the matrix instruction computed -- then z_i + b_i.
    template <int P>
    static __device__ __forceinline__ void net_forward(const Ctx& c, double (&a)[P][ROWS_A], double (&y)[P]) {
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = Net::dim(l), out = Net::dim(l + 1), off = Net::off(l);
            double z[P][out];
            static_for<0, out>([&](auto i) { static_for<0, P>([&](auto p) { z[p][i] = 0.0; }); });
            static_for<0, in>([&](auto kc) {
                constexpr int k = kc;
                static_for<0, out>([&](auto ic) {
                    constexpr int i = ic;
                    static_for<0, P>([&](auto p) { fmac_w<off + i + k * out>(c, z[p][i], a[p][a_row(l) + k]); });
                });
            });
            static_for<0, out>([&](auto ic) {
                constexpr int i = ic;
                static_for<0, P>([&](auto p) {
                    fmac_w<off + in * out + i>(c, z[p][i], c.one);
                    if constexpr (l + 1 < L) a[p][a_row(l + 1) + i] = act_fwd<ACT_TANH>(z[p][i]);
                    else y[p] = z[p][i];
                });
            });
        });
    }
    // reverse sweep: d[p][d_rel(l) + j] = delta of layer l (cotangent of its pre-activation), gx[p] = cotangent of the input.
    // Chains: g_k = fma(W[j][k], delta_j, g_k), j ascending from +0.
    template <int P>
    static __device__ __forceinline__ void net_backward(const Ctx& c, const double (&a)[P][ROWS_A], const double (&lam)[P], double (&d)[P][ROWS_D], double (&gx)[P]) {
        static_for<0, P>([&](auto p) { d[p][d_rel(L - 1)] = lam[p] * 1.0; });
        static_for<0, L>([&](auto lr) {
            constexpr int l = L - 1 - lr;
            constexpr int in = Net::dim(l), out = Net::dim(l + 1), off = Net::off(l);
            double g[P][in];
            static_for<0, in>([&](auto k) { static_for<0, P>([&](auto p) { g[p][k] = 0.0; }); });
            static_for<0, out>([&](auto jc) {
                constexpr int j = jc;
                static_for<0, in>([&](auto kc) {
                    constexpr int k = kc;
                    static_for<0, P>([&](auto p) { fmac_w<off + j + k * out>(c, g[p][k], d[p][d_rel(l) + j]); });
                });
            });
            static_for<0, in>([&](auto kc) {
                constexpr int k = kc;
                static_for<0, P>([&](auto p) {
                    if constexpr (l > 0) d[p][d_rel(l - 1) + k] = g[p][k] * act_bwd<ACT_TANH>(0.0, a[p][a_row(l) + k]);
                    else gx[p] = g[p][k];
                });
            });
        });
    }

#ifndef UDE_KPPV_CLK
#define UDE_KPPV_CLK(i)   // (tools/exp/kpp_harness.hip defines it to stamp the phases of a pass)
#endif
#ifndef UDE_KPPV_PF
#define UDE_KPPV_PF 1   // points per lane and pass of the forward-only evaluation (2: the same time, 244 B of scratch in fwd_kernel at two wavefronts per SIMD)
#endif
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) { const int i = point(cc, c.r); if (i < n) c.urow[i] = u[cc]; });
        __syncthreads();
        constexpr int P = (PPL % UDE_KPPV_PF == 0) ? UDE_KPPV_PF : 1;
#pragma unroll 1
        for (int cc = 0; cc < PPL; cc += P) {
            double a[P][ROWS_A], y[P];
            int ip_[P];
            bool on[P];
            static_for<0, P>([&](auto p) {
                const int i = point(cc + (int)decltype(p)::value, c.r);
                on[p] = i < n; ip_[p] = i;
                a[p][0] = on[p] ? c.urow[i] : 0.0;
            });
            net_forward<P>(c, a, y);
            static_for<0, P>([&](auto p) {
                if (on[p]) {
                    const int i = ip_[p];
                    const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                    const double cnn = c.w1 * c.urow[im] + c.w2 * a[p][0] + c.w3 * c.urow[ip];
                    c.orow[i] = y[p] + c.D0 * cnn;
                }
            });
        }
        __syncthreads();
        static_for<0, PPL>([&](auto cc) { const int i = point(cc, c.r); du[cc] = i < n ? c.orow[i] : 0.0; });
    }

    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const double* u, const double* lam, double* dlam, double* g, int mode = ACT_NONE) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) {
            const int i = point(cc, c.r);
            if (i < n) { c.urow[i] = u[cc]; c.lrow[i] = lam[cc]; }
        });
        __syncthreads();
        v4d acc[NT];   // output tiles of the wavefront's block sums (the fused chains run on across its 256 points)
        static_for<0, NT>([&](auto t) { acc[t] = v4d{0.0, 0.0, 0.0, 0.0}; });
        const int l16 = c.l16, kq = c.kq;
#pragma unroll 1
        for (int cc = 0; cc < PPL; ++cc) {
            const int i = point(cc, c.r);
            const bool on = i < n;
            const int ic = on ? i : 0;
            const int im = wrap_prev(ic, n), ip = wrap_next(ic, n);
            double a[1][ROWS_A], y[1], d[1][ROWS_D], gx[1], li[1];
            a[0][0] = on ? c.urow[ic] : 0.0;
            li[0] = on ? c.lrow[ic] : 0.0;
            UDE_KPPV_CLK(0);
            if (mode == ACT_LOAD) {   // (wave-uniform: a scalar branch) the activations the previous evaluation at this very time stored
                static_for<1, ROWS_A>([&](auto rc) { a[0][rc] = c.acache[(size_t)(decltype(rc)::value - 1) * NPT + ic]; });
            } else {
                net_forward<1>(c, a, y);
                if (mode == ACT_STORE && on) static_for<1, ROWS_A>([&](auto rc) { c.acache[(size_t)(decltype(rc)::value - 1) * NPT + ic] = a[0][rc]; });   // (a lane beyond the grid has ic = 0: not its slot)
            }
            UDE_KPPV_CLK(1);
            net_backward<1>(c, a, li, d, gx);
            UDE_KPPV_CLK(2);
            if (on)   // transpose of the periodic stencil (the oracle's expression)
                c.orow[i] = gx[0] + c.D0 * (c.w1 * c.lrow[ip] + c.w2 * c.lrow[i] + c.w3 * c.lrow[im]);
            if constexpr (WANT_PARAM) {
                // the stencil columns: u_{i-1}, u_{i+1} and the stencil sum, zeros for a point beyond the grid
                const double um_ = c.urow[im], u0_ = c.urow[ic], up_ = c.urow[ip];
                const double comb_ = c.w1 * um_ + c.w2 * u0_ + c.w3 * up_;
                const double um = on ? um_ : 0.0, up = on ? up_ : 0.0, comb = on ? comb_ : 0.0;
                // the contraction over the 64 points of this pass, one output tile after the other: ALL 64 lanes store the <= 16 delta rows and
                // <= 16 [a ; 1] columns the tile multiplies into the wave's [64 points][slot] LDS tile (an EXEC-masked store costs the LDS the
                // same cycles as a full one: sixteen points at a time -- the first version -- spent more time storing than multiplying), then
                // 16 k-steps of four points; every output element is one fused chain over the points in ascending order
                auto rowval = [&](auto e) -> double {
                    constexpr int r = decltype(e)::value;
                    if constexpr (r < ROWS_A) return a[0][r];
                    else if constexpr (r < ROWS_A + ROWS_D) return d[0][r - ROWS_A];
                    else if constexpr (r == ROW_ONE) return 1.0;
                    else if constexpr (r == ROW_UM) return um;
                    else if constexpr (r == ROW_UP) return up;
                    else return comb;
                };
                // (the address goes through an opaque register: left visible, the tile's offset inside the block's LDS -- a five-digit
                //  constant -- is folded into every store, does not fit ds_write2_b64's 8-bit offsets, and the compiler keeps one address
                //  register per pair of rows)
                unsigned ra = (unsigned)(unsigned long long)(c.tile + (c.lane & (TPTS - 1)) * ST);
                asm volatile("" : "+v"(ra));
                lds_t* row = (lds_t*)(unsigned long long)ra;
                static_for<0, NT>([&](auto tc) {
                    constexpr int t = tc;
                    static_for<0, NH>([&](auto hc) {
                        constexpr int h = hc;
                        if (NH == 1 || (c.lane >> 5) == h) {
                            static_for<0, PK.nr[t]>([&](auto e) { row[e] = rowval(std::integral_constant<int, PK.rows[t][decltype(e)::value]>{}); });
                            static_for<0, PK.nc[t]>([&](auto e) { row[PK.nr[t] + e] = rowval(std::integral_constant<int, PK.cols[t][decltype(e)::value]>{}); });
                        }
                        wave_sync();
                        // operands of k-step s + PFD are requested before the product of step s is issued
                        constexpr int PFD = 4, KST = TPTS / 4;
                        double av[PFD], bv[PFD];
                        const lds_t* pa = c.tile + kq * ST + c.arow[t];
                        const lds_t* pb = c.tile + kq * ST + c.bcol[t];
                        static_for<0, PFD>([&](auto s4) { av[s4] = pa[4 * decltype(s4)::value * ST]; bv[s4] = pb[4 * decltype(s4)::value * ST]; });
                        static_for<0, KST>([&](auto s4) {
                            constexpr int sl = decltype(s4)::value % PFD;
                            const double a_ = av[sl], b_ = bv[sl];
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (decltype(s4)::value + PFD < KST) {
                                av[sl] = pa[4 * (decltype(s4)::value + PFD) * ST];
                                bv[sl] = pb[4 * (decltype(s4)::value + PFD) * ST];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#if defined(UDE_KPPV_EXP) && (UDE_KPPV_EXP & 1)   // timing experiment: no matrix instructions
                            acc[t][0] += a_ * b_;
#else
                            acc[t] = mfma(a_, b_, acc[t]);
#endif
                        });
                        wave_sync();   // (the next rows overwrite the tile)
                    });
                });
            }
            UDE_KPPV_CLK(3);
        }
        __syncthreads();  // (results were produced by other lanes than the ones that own the points: no -- but the rows are shared with the next evaluation)
        static_for<0, PPL>([&](auto cc) { const int i = point(cc, c.r); dlam[cc] = i < n ? c.orow[i] : 0.0; });
        if constexpr (WANT_PARAM) {
            // block sums -> this wavefront's row (aliases its tile)
            lds_t* prow = c.tile;
            static_for<0, NT>([&](auto t) {
                static_for<0, 4>([&](auto rr) {
                    const int idx = c.pidx[t][rr];
                    if (idx >= 0) prow[idx] = acc[t][decltype(rr)::value];
                });
            });
            if (c.lane == 0) prow[c.so + 3] = 0.0;   // the unused conv bias
            __syncthreads();
            static_for<0, NSL>([&](auto s) {
                const int p = c.r + G * decltype(s)::value;
                double v = 0.0;
                if (p < NP) {
                    v = c.part[p];
                    static_for<1, NWV>([&](auto w) { v += c.part[decltype(w)::value * TILE + p]; });
                    if (p >= c.so && p < c.so + 3) v = c.D0 * v;
                }
                g[s] = v;
            });
        }
        __syncthreads();
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts& mc, int r, int s) {
        const int p = r + G * s;
        if (p >= mc.n_param) return -1;
        if (p == mc.stencil_offset + 3) return -1;  // the unused conv bias (Fisher-KPP-CNN.jl:100-109): gradient stays 0
        return p;
    }
};
#endif  // UDE_F32

}  // namespace ude
