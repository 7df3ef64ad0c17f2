// ude_coop.h -- lane-group cooperation primitives and the cooperative small-MLP (a2).
//
// Execution layout: G consecutive lanes of a 64-wide wavefront own ONE trajectory.  The small ODE state
// (and the adjoint lambda) is replicated in the registers of all G lanes; the neurons of every Dense
// layer -- and with them the rows of the parameter gradient mu -- are dealt round-robin to the G lanes
// (neuron j lives on lane j % G).  One all-gather per layer hands the new activations to every lane.
// G = 1 degenerates to "one trajectory per lane", G = 64 to "one wavefront per trajectory".
#pragma once
#include <type_traits>

#include "ude_math.h"

namespace ude {

// ---- DPP (data-parallel primitives): cross-lane moves inside the VALU, no LDS round trip (dpp_mov: ude_real.h) -----
constexpr int DPP_QUAD(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <- lane 7-i inside each 8 lanes

// Lane groups need not be a power of two: G = 5 puts 12 trajectories in a wavefront (lanes 60-63 idle) with
// every lane of a 5-wide layer busy.  Groups start at lane multiples of G; for such G the cross-lane moves go
// through ds_bpermute with a computed source lane and sums are formed in a fixed sequential order.
template <int G>
constexpr bool pow2_group() { return (G & (G - 1)) == 0; }

// value of x held by lane `src` of this lane's group (src < G is a compile-time constant)
template <int G, int SRC>
__device__ __forceinline__ real group_bcast(real x) {
    if constexpr (G == 1) {
        return x;
    } else if constexpr (G == 4) {
        return dpp_mov<DPP_QUAD(SRC, SRC, SRC, SRC)>(x);  // one DPP move per 32-bit half
    } else if constexpr (pow2_group<G>()) {
        return __shfl(x, SRC, G);  // ds_bpermute
    } else {
        const int lane = threadIdx.x & 63;
        return bpermute_real((lane - lane % G + SRC) << 2, x);
    }
}

// sum over the G lanes of a group; every lane receives the same bits (xor butterfly of commutative adds, or
// the same sequential order on every lane)
constexpr int DPP_ROW_MIRROR = 0x140;  // lane i <- lane 15-i inside each row of 16
// scratch of the cross-wave reductions (trajectories spanning several wavefronts of one block)
__device__ __forceinline__ real* xwave_buf() {
    __shared__ real buf[16];
    return buf;
}
// component-per-lane helpers (replicated small states, one wavefront per trajectory): lane c keeps component c
template <int NR>
__device__ __forceinline__ real own_of(const real (&v)[NR]) {
    const int lane = threadIdx.x & 63;
    real r = 0.0;
    static_for<0, NR>([&](auto c) { r = (lane == (int)decltype(c)::value) ? v[c] : r; });
    return r;
}
template <int NR>
__device__ __forceinline__ void bcast_all(real own, real (&out)[NR]) {
    static_for<0, NR>([&](auto c) { out[c] = readlane_real(own, decltype(c)::value); });
}
// ARITH-SPEC tree sum over the 64 lanes of a wavefront: binary tree over adjacent index pairs (every lane gets
// the total; at each level both partners add the same two values, so all lanes hold identical bits)
__device__ __forceinline__ real wave_tree_sum(real x) {
    x += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(x);
    x += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(x);
    x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
    x += dpp_mov<DPP_ROW_MIRROR>(x);
#ifndef UDE_TREE_BUTTERFLY
    // rows 0+1 and 2+3 (row_bcast15 into rows 1 and 3), then the two halves (row_bcast31 into rows 2 and 3): lane 63 holds
    // ((r0 + r1) + (r2 + r3)), the same tree as the xor butterfly, and the total leaves through a scalar register --
    // two DPP moves and a v_readlane pair instead of two ds_bpermute round trips (tools/probe/tree_sum_probe.hip)
    if constexpr (sizeof(real) == 8) {
        x += dpp_mov_rows<0x142, 0xA>(x);
        x += dpp_mov_rows<0x143, 0xC>(x);
        return readlane_real(x, 63);
    }
#endif
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
template <int G, class T = real>
__device__ __forceinline__ T group_sum(T x) {
    if constexpr (G > 64) {
        // trajectory spans G/64 wavefronts: butterfly inside each, then the wave sums in ascending order through LDS
        static_assert(G <= 64 || sizeof(T) == sizeof(real), "multi-wavefront groups exist for Float64 only");
        x = group_sum<64>(x);
        T* buf = (T*)xwave_buf();
        if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = x;
        __syncthreads();
        T s = buf[0];
#pragma unroll
        for (int i = 1; i < G / 64; ++i) s += buf[i];
        __syncthreads();
        return s;
    } else if constexpr (!pow2_group<G>()) {
        const int lane = threadIdx.x & 63;
        const int base = (lane - lane % G) << 2;
        T s = bpermute_real(base, x);
#pragma unroll 1
        for (int i = 1; i < G; ++i) s += bpermute_real(base + (i << 2), x);
        return s;
    } else if constexpr (G == 32 && sizeof(T) == 8) {
        // two groups per wavefront: four DPP levels inside the 16-lane rows, then rows 1 and 3 add lane 15 of the row below
        // (row_bcast15) -- every lane of rows 1 / 3 holds its group's total, which leaves through a scalar register; the same tree as
        // the xor butterfly (commutative adds), 10 DPP moves + 4 v_readlane instead of 10 ds_bpermute round trips
        x += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(x);
        x += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(x);
        x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
        x += dpp_mov<DPP_ROW_MIRROR>(x);
        x += dpp_mov_rows<0x142, 0xA>(x);
        const T lo = readlane_real(x, 31), hi = readlane_real(x, 63);
        return (threadIdx.x & 32) ? hi : lo;
    } else {
        constexpr bool DPPG = (G == 4 || G == 8 || G == 16);   // groups inside a 16-lane row: every level is a DPP move
        if constexpr (G >= 2) x += DPPG ? dpp_mov<DPP_QUAD(1, 0, 3, 2)>(x) : __shfl_xor(x, 1, G);
        if constexpr (G >= 4) x += DPPG ? dpp_mov<DPP_QUAD(2, 3, 0, 1)>(x) : __shfl_xor(x, 2, G);
        if constexpr (G >= 8) x += DPPG ? dpp_mov<DPP_ROW_HALF_MIRROR>(x) : __shfl_xor(x, 4, G);
        if constexpr (G >= 16) x += DPPG ? dpp_mov<DPP_ROW_MIRROR>(x) : __shfl_xor(x, 8, G);
#pragma unroll
        for (int m = 16; m < G; m <<= 1) x += __shfl_xor(x, m, G);
        return x;
    }
}

// real-real accumulation (two-sum): order-independent sums for the initial-dt norms (ARITH-SPEC)
__device__ __forceinline__ void dd_acc(real& hi, real& lo, real x) {
    const real s = hi + x;
    const real bb = s - hi;
    const real e = (hi - (s - bb)) + (x - bb);
    hi = s;
    lo += e;
}
// combine the (hi, lo) pairs of the G lanes of a group; every lane ends with the same pair
template <int G>
__device__ __forceinline__ void group_dd_sum(real& hi, real& lo) {
    if constexpr (G > 64) {
        group_dd_sum<64>(hi, lo);
        real* buf = xwave_buf();
        if ((threadIdx.x & 63) == 0) {
            buf[2 * (threadIdx.x >> 6)] = hi;
            buf[2 * (threadIdx.x >> 6) + 1] = lo;
        }
        __syncthreads();
        real h = buf[0], l = buf[1];
#pragma unroll
        for (int i = 1; i < G / 64; ++i) {
            const real h2 = buf[2 * i], l2 = buf[2 * i + 1];
            const real s = h + h2;
            const real bb = s - h;
            const real e = (h - (s - bb)) + (h2 - bb);
            l = (l + l2) + e;
            h = s;
        }
        __syncthreads();
        hi = h;
        lo = l;
        return;
    }
    if constexpr (!pow2_group<G>()) {
        const int lane = threadIdx.x & 63;
        const int base = (lane - lane % G) << 2;
        real h = bpermute_real(base, hi), l = bpermute_real(base, lo);
#pragma unroll 1
        for (int i = 1; i < G; ++i) {
            const real h2 = bpermute_real(base + (i << 2), hi), l2 = bpermute_real(base + (i << 2), lo);
            const real s = h + h2;
            const real bb = s - h;
            const real e = (h - (s - bb)) + (h2 - bb);
            l = (l + l2) + e;
            h = s;
        }
        hi = h;
        lo = l;
        return;
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
        const real h2 = __shfl_xor(hi, m, G), l2 = __shfl_xor(lo, m, G);
        const real s = hi + h2;
        const real bb = s - hi;
        const real e = (hi - (s - bb)) + (h2 - bb);
        lo = (lo + l2) + e;
        hi = s;
    }
}

template <int... V>
struct IntList {
    static constexpr int n = sizeof...(V);
    static constexpr int at(int i) {
        constexpr int v[] = {V...};
        return v[i];
    }
};

// Dense-layer stack: dims D0 -> D1 -> ... -> DL, activation per layer.
// theta layout per layer: [vec(W) column-major (out x in); b(out)]  (Lux / FastChain / Flux.destructure)
enum { ACT_RT = 100 };   // an activation chosen at run time (NetCfgRt below)

template <class DIMS, class ACTS>
struct NetCfg {
    static constexpr bool RT = false;
    static constexpr int L = DIMS::n - 1;
    static_assert(ACTS::n == L, "one activation per Dense layer");
    static constexpr int dim(int l) { return DIMS::at(l); }
    static constexpr int act(int l) { return ACTS::at(l); }
    static constexpr int off(int l) {  // theta offset of layer l relative to the first NN parameter
        int o = 0;
        for (int i = 0; i < l; ++i) o += dim(i) * dim(i + 1) + dim(i + 1);
        return o;
    }
    static constexpr int nparam = off(L);
    static constexpr int maxdim() {
        int m = 0;
        for (int i = 0; i <= L; ++i) m = dim(i) > m ? dim(i) : m;
        return m;
    }
};

// A chain IN -> (L_ - 1 hidden layers of width <= W_) -> OUT whose hidden widths and activations are RUN-TIME values (round 5: an edited
// LV network -- `Lux.Chain(Dense(2,5,rbf), ...)` is a script variable -- on the lane-group kernels instead of the wavefront-per-trajectory
// fallback).  The compile-time shape is the PADDED one: the register copy of the weights (CoopMlp::WReg) is loaded with zeros beyond the
// true widths, a padded neuron's activation is forced to 0, so every fma chain of the true chain is followed by exact no-op terms
// (fma(0, 0, acc)) and every padded parameter slot holds 0: the bits of the true shape, whatever it is.  Linear last layer.
template <int IN_, int OUT_, int L_, int W_>
struct NetCfgRt {
    static constexpr bool RT = true;
    static constexpr int L = L_;
    static constexpr int dim(int l) { return l == 0 ? IN_ : l == L_ ? OUT_ : W_; }
    static constexpr int act(int l) { return l == L_ - 1 ? ACT_IDENTITY : ACT_RT; }
    static constexpr int off(int l) {   // (offsets of the PADDED shape: never used to address theta -- WReg::rt holds the true ones)
        int o = 0;
        for (int i = 0; i < l; ++i) o += dim(i) * dim(i + 1) + dim(i + 1);
        return o;
    }
    static constexpr int nparam = off(L_);
    static constexpr int maxdim() { return W_ > IN_ ? (W_ > OUT_ ? W_ : OUT_) : (IN_ > OUT_ ? IN_ : OUT_); }
};

template <class N, int G>
struct CoopMlp {
    static constexpr int L = N::L;
    static constexpr int own(int l) { return (N::dim(l + 1) + G - 1) / G; }  // neurons of layer l per lane
    // Parameter-cotangent slots of a layer normally follow its neurons (lane r: the rows of its own(l) neurons).  The narrow
    // LAST layer (out = 2 of 5 lanes) would leave most lanes with padding slots, so when its inputs are spread one per lane
    // (in <= G, one neuron of the layer below per lane) its slots go by INPUT instead: lane r owns W[:, r] (out slots) and,
    // for r < out, the bias b[r] -- out + 1 slots per lane instead of in + 1 (LV 2-5-5-5-2 on 5 lanes: 21 -> 18 slots).
    // Every slot value is the same product as before (delta_j * a_r with delta replicated, a_r the lane's own activation).
    // (round 4: also when every lane holds SEVERAL inputs of the last layer, dealt evenly -- 2-32-2 on 8 lanes: lane r owns W[:, r + 8m],
    //  m < 4: 4 * 2 + 1 = 9 slots instead of 33, and nothing reads the replicated copy of the 32 activations any more)
    static constexpr int OM = own(L - 2 >= 0 ? L - 2 : 0);   // inputs of the last layer per lane
    static constexpr bool KMAJ = (G > 1) && (L >= 2) && (N::dim(L) <= G) && (OM == 1 ? N::dim(L - 1) <= G : N::dim(L - 1) == OM * G);
    static constexpr int layer_slots(int l) { return (KMAJ && l == L - 1) ? OM * N::dim(L) + 1 : own(l) * (N::dim(l) + 1); }
    static constexpr int slot_off(int l) {
        int o = 0;
        for (int i = 0; i < l; ++i) o += layer_slots(i);
        return o;
    }
    static constexpr int NSLOT = slot_off(L);
    static constexpr int maxown() {
        int m = 0;
        for (int i = 0; i < L; ++i) m = own(i) > m ? own(i) : m;
        return m;
    }
    static constexpr int MAXD = N::maxdim();
    static constexpr int MAXOWN = maxown();
    // ARITH-SPEC wide-dot rule, tree case (oracle: wide_dot): a dot of n = 32 or 64 terms that reduces to fewer than 16 replicated
    // scalars is rounded products under the binary tree over adjacent index pairs.  With term j on lane j % G (register j / G) and
    // n a multiple of the power-of-two G, the first log2(G) levels of that tree are the xor butterfly of the group (per register),
    // the remaining levels pair the registers in place: no all-gather of the n terms, every lane ends with the same bits.
    static constexpr bool tree_dot(int n, int nres) { return (n == 32 || n == 64) && nres < 16; }
    static constexpr bool tree_ok(int n) { return G > 1 && pow2_group<G>() && G <= 64 && n % G == 0; }
    template <int CNT>
    static __device__ __forceinline__ real tree_reduce(real (&p)[CNT]) {   // p[m] = this lane's product of term r + m * G
        static_for<0, CNT>([&](auto m) { p[m] = group_sum<G>(p[m]); });
        static_for<0, 6>([&](auto lv) {
            constexpr int step = 1 << decltype(lv)::value;       // pair registers m, m + step (m a multiple of 2 * step)
            if constexpr (step < CNT)
                static_for<0, CNT>([&](auto m) {
                    if constexpr (decltype(m)::value % (2 * step) == 0 && decltype(m)::value + step < CNT) p[m] = p[m] + p[decltype(m)::value + step];
                });
        });
        return p[0];
    }

    // A narrow linear LAST layer (LV: 5 -> 2) is computed REPLICATED: its inputs are on every lane already (the gather below it), so
    // every lane runs the `out` chains itself -- the same fma chains, hence the same bits -- instead of one chain on `out` lanes
    // followed by a second gather: one LDS / DPP round trip less in the latency chain of every evaluation.
    static constexpr bool rep_last(int l) {
        return G > 1 && L >= 2 && l == L - 1 && N::dim(L) <= G && N::dim(L) * N::dim(L - 1) <= 16 && !(tree_dot(N::dim(L - 1), N::dim(L)) && tree_ok(N::dim(L - 1)));
    }
    static constexpr int REP_OUT = rep_last(L - 1) ? N::dim(L) : 1;

    typedef __attribute__((address_space(3))) real lds_t;
    // all-gather of one value per lane inside a group.  Power-of-two groups: DPP / bpermute broadcasts.  Other groups
    // (G = 5): through the group's words of a wave-private LDS row -- one ds_write + three ds_read2 per gather instead
    // of ten ds_bpermute (LDS is in order per wavefront: no barrier; `gb` = this GROUP's first word)
    // (round 4: wide power-of-two groups too -- G = 32 gathers 32 values: one ds_write_b64 + 16 broadcast ds_read_b128 instead of 64
    //  ds_bpermute_b32; the group's words of the wave-private row are written and read in order by one wavefront: no barrier)
#ifndef UDE_LDS_GATHER_MIN
#define UDE_LDS_GATHER_MIN 32
#endif
    static constexpr bool LDS_GATHER = !pow2_group<G>() || (G >= UDE_LDS_GATHER_MIN && G <= 64);
    template <int CNT>
    static __device__ __forceinline__ void allgather(lds_t* gb, int r, const real* own, real* out) {
        if constexpr (LDS_GATHER) {
            static_assert(CNT <= G, "LDS gather: one value per lane");
            gb[r] = own[0];
            static_for<0, CNT>([&](auto j) { out[j] = gb[decltype(j)::value]; });
        } else {
            static_for<0, CNT>([&](auto jc) {
                constexpr int j = jc;
                out[j] = group_bcast<G, j % G>(own[j / G]);
            });
        }
    }
    struct Cache {
        lds_t* gb;                // LDS gather row of this group (LDS_GATHER)
        real a[L + 1][MAXD];    // replicated activations (a[0] = input)
        real z[L][MAXOWN];      // pre-activations of the neurons this lane owns
        real ao[L][MAXOWN];     // their activations
    };

    // Register-resident copy of the weights THIS lane touches: the rows of its neurons (forward), the columns of
    // the next layer at its neurons (backward) and the first layer (input cotangent).  Loaded once per kernel; takes
    // the LDS round trips of the weight fetches out of the per-evaluation latency chain.
    struct RtInfo { int dim[L + 1], act[L], off[L]; };   // run-time shape of a NetCfgRt chain: true widths, activations, theta offsets
    struct NoRt {};
    struct WReg {
        real row[L][MAXOWN][MAXD + 1];  // [l][m][k], bias at k = dim(l)
        real col[L][MAXOWN][MAXD];      // [l][m][i] = W_{l+1}[i, j]
        real w0[MAXD * MAXD];           // W_0[j + k*out]
        real last[REP_OUT][MAXD + 1];   // replicated last layer (rep_last): every row of W_{L-1}, bias at k = dim(L-1)
        std::conditional_t<N::RT, RtInfo, NoRt> rt;
    };
    // the register copy of a run-time shape (NetCfgRt): zeros beyond the true widths
    template <class P, class MC>
    static __device__ __forceinline__ void load_weights_rt(const P* th, int r, WReg& w, const MC& mc) {
        int o = 0;
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            w.rt.dim[l] = mc.dims[l]; w.rt.act[l] = mc.act[l]; w.rt.off[l] = o;
            o += mc.dims[l] * mc.dims[l + 1] + mc.dims[l + 1];
        });
        w.rt.dim[L] = mc.dims[L];
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = N::dim(l), out = N::dim(l + 1);
            const int inr = w.rt.dim[l], outr = w.rt.dim[l + 1];
            const P* W = th + w.rt.off[l];
            static_for<0, own(l)>([&](auto mc_) {
                constexpr int m = mc_;
                const int j = r + m * G;
                const bool jv = j < outr;
                const int jj = jv ? j : 0;
                static_for<0, in>([&](auto k) { w.row[l][m][k] = (jv && (int)decltype(k)::value < inr) ? (real)W[jj + (int)decltype(k)::value * outr] : real(0); });
                w.row[l][m][in] = jv ? (real)W[inr * outr + jj] : real(0);
                if constexpr (l + 1 < L) {
                    constexpr int out2 = N::dim(l + 2);
                    const int out2r = w.rt.dim[l + 2];
                    const P* W2 = th + w.rt.off[l + 1];
                    static_for<0, out2>([&](auto i) { w.col[l][m][i] = (jv && (int)decltype(i)::value < out2r) ? (real)W2[(int)decltype(i)::value + jj * out2r] : real(0); });
                }
            });
        });
        {
            constexpr int in = N::dim(0), out = N::dim(1);
            const int outr = w.rt.dim[1];
            static_for<0, in * out>([&](auto ic) {
                constexpr int i = ic, j = i % out, k = i / out;
                w.w0[i] = j < outr ? (real)th[w.rt.off[0] + j + k * outr] : real(0);
            });
        }
        if constexpr (rep_last(L - 1)) {
            constexpr int in = N::dim(L - 1), out = N::dim(L);
            const int inr = w.rt.dim[L - 1];
            const P* WL = th + w.rt.off[L - 1];
            static_for<0, out>([&](auto i) {
                static_for<0, in>([&](auto k) { w.last[i][k] = (int)decltype(k)::value < inr ? (real)WL[(int)decltype(i)::value + (int)decltype(k)::value * out] : real(0); });
                w.last[i][in] = (real)WL[inr * out + (int)decltype(i)::value];
            });
        }
    }
    // activation of layer l and its derivative: compile-time, or (ACT_RT) the chain's run-time choice -- a wave-uniform branch
    template <int l, class WS>
    static __device__ __forceinline__ real actF(const WS& th, real z) {
        if constexpr (N::act(l) == ACT_RT) {
            const int a = rt_of(th).act[l];
            if (a == ACT_TANH) return rtanh(z);
            if (a == ACT_RBF) return rexp(-(z * z));
            if (a == ACT_RELU) return z > real(0) ? z : real(0);
            return z;
        } else { (void)th; return act_fwd<N::act(l)>(z); }
    }
    template <int l, class WS>
    static __device__ __forceinline__ real actB(const WS& th, real z, real a_) {
        if constexpr (N::act(l) == ACT_RT) {
            const int a = rt_of(th).act[l];
            if (a == ACT_TANH) return rfma(-a_, a_, real(1));
            if (a == ACT_RBF) return (real(-2) * z) * a_;
            if (a == ACT_RELU) return z > real(0) ? real(1) : real(0);
            return real(1);
        } else { (void)th; return act_bwd<N::act(l)>(z, a_); }
    }
    // (run-time shapes: is neuron j of layer l one of the true chain's?)
    template <int l, class WS>
    static __device__ __forceinline__ bool rt_valid(const WS& th, int j) {
        if constexpr (N::RT) return j < rt_of(th).dim[l + 1];
        else { (void)th; (void)j; return true; }
    }
    template <class P>
    static __device__ __forceinline__ void load_weights(const P* th, int r, WReg& w) {
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = N::dim(l), out = N::dim(l + 1);
            const P* W = th + N::off(l);
            static_for<0, own(l)>([&](auto mc) {
                constexpr int m = mc;
                const int j = r + m * G;
                const int jj = ((own(l) * G == out) || (j < out)) ? j : 0;
                static_for<0, in>([&](auto k) { w.row[l][m][k] = (real)W[jj + k * out]; });
                w.row[l][m][in] = (real)W[in * out + jj];
                if constexpr (l + 1 < L) {
                    constexpr int out2 = N::dim(l + 2);
                    const P* W2 = th + N::off(l + 1);
                    static_for<0, out2>([&](auto i) { w.col[l][m][i] = (real)W2[i + jj * out2]; });
                }
            });
        });
        static_for<0, N::dim(0) * N::dim(1)>([&](auto i) { w.w0[i] = (real)th[N::off(0) + i]; });
        if constexpr (rep_last(L - 1)) {
            constexpr int in = N::dim(L - 1), out = N::dim(L);
            static_for<0, out>([&](auto i) {
                static_for<0, in + 1>([&](auto k) { w.last[i][k] = (real)th[N::off(L - 1) + (int)decltype(i)::value + (int)decltype(k)::value * out]; });
            });
        }
    }

    // WS = const P* (weights read from LDS/global at every use) or WReg (register-resident copy)
    // a run-time shape whose weights are read where they lie at every use (widths > 8: no register copy): the pointer and the true shape
    struct PtrRt {
        const real* p;
        const RtInfo* rtp;
    };
    template <class MC>
    static __device__ __forceinline__ void load_rt_info(RtInfo& rt, const MC& mc) {
        int o = 0;
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            rt.dim[l] = mc.dims[l]; rt.act[l] = mc.act[l]; rt.off[l] = o;
            o += mc.dims[l] * mc.dims[l + 1] + mc.dims[l + 1];
        });
        rt.dim[L] = mc.dims[L];
    }
    template <class WS>
    static constexpr bool ws_is_reg = std::is_same<WS, WReg>::value;
    template <class WS>
    static constexpr bool ws_is_rtptr = std::is_same<WS, PtrRt>::value;
    template <class WS>
    static __device__ __forceinline__ const RtInfo& rt_of(const WS& th) {
        if constexpr (ws_is_rtptr<WS>) return *th.rtp;
        else return th.rt;
    }

    // parameter `idx` of a pointer-like weight source (the tree layers read theta in place; they are compiled for pointer sources
    // only -- the register copy holds rows, a tree layer wants the columns at the lane's inputs)
    template <class WS>
    static __device__ __forceinline__ real th_at(const WS& th, int idx) {
        if constexpr (ws_is_reg<WS> || ws_is_rtptr<WS>) { (void)th; (void)idx; return real(0); }
        else return (real)th[idx];
    }

    template <int I, int K, class WS>
    static __device__ __forceinline__ real last_at(const WS& th) {   // W_{L-1}[I, K]; K = dim(L-1): the bias of output I
        if constexpr (ws_is_reg<WS>) return th.last[I][K];
        else if constexpr (ws_is_rtptr<WS>) {
            const RtInfo& rt = *th.rtp;
            const int inr = rt.dim[L - 1];
            return K == N::dim(L - 1) ? th.p[rt.off[L - 1] + inr * N::dim(L) + I] : (K < inr ? th.p[rt.off[L - 1] + I + K * N::dim(L)] : real(0));
        }
        else return (real)th[N::off(L - 1) + I + K * N::dim(L)];
    }

    // th: NN parameters (LDS or global) or a WReg; r: lane index inside the group
    template <class WS>
    static __device__ __forceinline__ void forward(const WS& th, int r, const real* x, Cache& c, real* y) {
        static_for<0, N::dim(0)>([&](auto k) { c.a[0][k] = x[k]; });
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = N::dim(l), out = N::dim(l + 1);
            if constexpr (l > 0 && tree_dot(in, out) && tree_ok(in) && !ws_is_reg<WS> && !N::RT) {
                // a tree layer (2-32-2's output layer): every lane multiplies ITS inputs (the activations it produced in the layer
                // below) with its weights, the group's butterfly does the rest: all `out` results arrive replicated
                static_assert(own(l - 1) * G == in && own(l) == 1 && out <= G, "tree layer: inputs dealt evenly, one output neuron per lane at most");
                real zall[out];
                static_for<0, out>([&](auto ic) {
                    constexpr int i = ic;
                    real p[own(l - 1)];
                    static_for<0, own(l - 1)>([&](auto m) {
                        const int k = r + (int)decltype(m)::value * G;
                        p[m] = th_at(th, N::off(l) + i + k * out) * c.ao[l - 1][m];
                    });
                    zall[i] = tree_reduce(p) + th_at(th, N::off(l) + in * out + i);
                });
                real zr = zall[0];
                static_for<1, out>([&](auto i) { zr = (r == (int)decltype(i)::value) ? zall[i] : zr; });
                c.z[l][0] = zr;
                c.ao[l][0] = r < out ? act_fwd<N::act(l)>(zr) : real(0);
                static_for<0, out>([&](auto i) { c.a[l + 1][i] = act_fwd<N::act(l)>(zall[i]); });
                static_assert(!N::RT, "run-time shapes: no tree layers");
            } else if constexpr (rep_last(l)) {
                real zall[out];
                static_for<0, out>([&](auto ic) {
                    constexpr int i = ic;
                    real acc = 0.0;
                    static_for<0, in>([&](auto k) { acc = rfma(last_at<i, decltype(k)::value>(th), c.a[l][k], acc); });
                    zall[i] = acc + last_at<i, in>(th);
                });
                real zr = zall[0];
                static_for<1, out>([&](auto i) { zr = (r == (int)decltype(i)::value) ? zall[i] : zr; });
                c.z[l][0] = zr;
                c.ao[l][0] = r < out ? act_fwd<N::act(l)>(zr) : real(0);
                static_for<0, out>([&](auto i) { c.a[l + 1][i] = act_fwd<N::act(l)>(zall[i]); });
            } else {
            static_for<0, own(l)>([&](auto mc) {
                constexpr int m = mc;
                const int j = r + m * G;
                const bool valid = (own(l) * G == out) || (j < out);
                const int jj = valid ? j : 0;
                real acc = 0.0;
                if constexpr (ws_is_reg<WS>) {
                    static_for<0, in>([&](auto k) { acc = rfma(th.row[l][m][k], c.a[l][k], acc); });
                    acc += th.row[l][m][in];
                } else if constexpr (ws_is_rtptr<WS>) {
                    const RtInfo& rt = *th.rtp;
                    const int inr = rt.dim[l], outr = rt.dim[l + 1];
                    const int jr = j < outr ? j : 0;
                    const real* W = th.p + rt.off[l];
                    static_for<0, in>([&](auto k) { acc = rfma((int)decltype(k)::value < inr ? W[jr + (int)decltype(k)::value * outr] : real(0), c.a[l][k], acc); });
                    acc += W[inr * outr + jr];
                } else {
                    static_for<0, in>([&](auto k) { acc = rfma((real)th[N::off(l) + jj + k * out], c.a[l][k], acc); });
                    acc += (real)th[N::off(l) + in * out + jj];
                }
                c.z[l][m] = acc;
                c.ao[l][m] = (valid && rt_valid<l>(th, j)) ? actF<l>(th, acc) : 0.0;
            });
            // (the activations feed a tree layer above: it reads them where they are -- no gather)
            // (... unless its parameter slots go by input, KMAJ: then nothing else reads the replicated copy)
            if constexpr (!(KMAJ && l + 2 == L && tree_dot(out, N::dim(L)) && tree_ok(out) && !ws_is_reg<WS> && !N::RT)) allgather<out>(c.gb, r, c.ao[l], c.a[l + 1]);
            }
        });
        static_for<0, N::dim(L)>([&](auto k) { y[k] = c.a[L][k]; });
    }

    // gy: cotangent of the output (replicated).  gx: cotangent of the input (replicated).
    // g[NSLOT]: this lane's slice of (dNN/dtheta)^T gy; slot (l, m, k) = slot_off(l) + m*(in+1) + k, bias at k = in.
    struct NoSink {
        __device__ __forceinline__ void operator()(int, int, real) const {}
    };
    template <bool WANT_PARAM, class WS>
    static __device__ __forceinline__ void vjp(const WS& th, int r, const Cache& c, const real* gy, real* gx,
                                               real* g) {
        vjp_sink<WANT_PARAM>(th, r, c, gy, gx, g, NoSink{});
    }
    // sink(l, m, d): receives the delta of this lane's m-th neuron of layer l (pointwise networks export it)
    template <bool WANT_PARAM, class WS, class Sink>
    static __device__ __forceinline__ void vjp_sink(const WS& th, int r, const Cache& c, const real* gy, real* gx,
                                                    real* g, Sink sink) {
        static_assert(N::act(L - 1) == ACT_IDENTITY, "output layer must be linear");
        real dall[MAXD];  // replicated delta of the layer above
        static_for<0, N::dim(L)>([&](auto k) { dall[k] = gy[k]; });
        static_for<0, L>([&](auto lr) {
            constexpr int l = L - 1 - lr;
            constexpr int in = N::dim(l), out = N::dim(l + 1);
            real down[MAXOWN];
            static_for<0, own(l)>([&](auto mc) {
                constexpr int m = mc;
                const int j = r + m * G;
                const bool valid = (own(l) * G == out) || (j < out);
                const int jj = valid ? j : 0;
                real gp;
                if constexpr (l == L - 1) {
                    gp = dall[0];  // pick element j of the replicated output cotangent
                    static_for<1, out>([&](auto i) { gp = (jj == i) ? dall[i] : gp; });
                } else {
                    constexpr int out2 = N::dim(l + 2);
                    gp = 0.0;  // column jj of the next layer's W (contiguous in theta)
                    if constexpr (ws_is_reg<WS>) {
                        static_for<0, out2>([&](auto i) { gp = rfma(th.col[l][m][i], dall[i], gp); });
                    } else if constexpr (ws_is_rtptr<WS>) {
                        const RtInfo& rt = *th.rtp;
                        const int outr = rt.dim[l + 1], out2r = rt.dim[l + 2];
                        const int jr = j < outr ? j : 0;
                        const real* W2 = th.p + rt.off[l + 1];
                        static_for<0, out2>([&](auto i) { gp = rfma((int)decltype(i)::value < out2r ? W2[(int)decltype(i)::value + jr * out2r] : real(0), dall[i], gp); });
                    } else {
                        static_for<0, out2>([&](auto i) {
                            gp = rfma((real)th[N::off(l + 1) + i + jj * out2], dall[i], gp);
                        });
                    }
                }
                const real d = (valid && rt_valid<l>(th, j)) ? gp * actB<l>(th, c.z[l][m], c.ao[l][m]) : 0.0;
                down[m] = d;
                sink(l, m, d);
                if constexpr (WANT_PARAM && !(KMAJ && l == L - 1)) {
                    constexpr int s0 = slot_off(l) + m * (in + 1);
                    static_for<0, in>([&](auto k) { g[s0 + k] = d * c.a[l][k]; });
                    g[s0 + in] = d;
                }
            });
            if constexpr (WANT_PARAM && KMAJ && l == L - 1) {
                // slots by input: dall still holds the (replicated) deltas of this linear layer = gy
                constexpr int s0 = slot_off(L - 1);
                static_for<0, OM>([&](auto m) {
                    const real ak = c.ao[L - 2][m];  // this lane's own activation below = a_{L-1}[r + m G] (0 on lanes without a neuron)
                    static_for<0, out>([&](auto j) { g[s0 + (int)decltype(m)::value * out + j] = dall[j] * ak; });
                });
                real gb = dall[0];
                static_for<1, out>([&](auto i) { gb = (r == i) ? dall[i] : gb; });
                g[s0 + OM * out] = r < out ? gb : 0.0;
            }
            if constexpr (l > 0) {
                // (the output layer is linear: its deltas ARE the replicated output cotangent, gy[j] * 1 -- dall holds them already)
                if constexpr (l < L - 1) allgather<out>(c.gb, r, down, dall);
            } else if constexpr (tree_dot(out, in) && tree_ok(out) && !ws_is_reg<WS> && !N::RT) {
                // input cotangent as a tree (wide-dot rule: `out` = 32 or 64 terms, `in` < 16 results): products where the deltas are
                static_assert(own(0) * G == out, "tree: the first layer's neurons are dealt evenly");
                static_for<0, in>([&](auto k) {
                    real p[own(0)];
                    static_for<0, own(0)>([&](auto m) {
                        const int j = r + (int)decltype(m)::value * G;
                        p[m] = th_at(th, N::off(0) + j + (int)decltype(k)::value * out) * down[m];
                    });
                    gx[k] = tree_reduce(p);
                });
            } else {
                // input cotangent: gx[k] = sum_j W0[j,k] delta0[j]; every lane needs it, so gather delta0 too
                real d0[MAXD];
                allgather<out>(c.gb, r, down, d0);
                static_for<0, in>([&](auto k) {
                    real s = 0.0;
                    if constexpr (ws_is_reg<WS>) {
                        static_for<0, out>([&](auto j) { s = rfma(th.w0[j + k * out], d0[j], s); });
                    } else if constexpr (ws_is_rtptr<WS>) {
                        const RtInfo& rt = *th.rtp;
                        const int outr = rt.dim[1];
                        const real* W0 = th.p + rt.off[0];
                        static_for<0, out>([&](auto j) { s = rfma((int)decltype(j)::value < outr ? W0[(int)decltype(j)::value + (int)decltype(k)::value * outr] : real(0), d0[j], s); });
                    } else {
                        static_for<0, out>([&](auto j) { s = rfma((real)th[N::off(0) + j + k * out], d0[j], s); });
                    }
                    gx[k] = s;
                });
            }
        });
    }

    // theta index (relative to the first NN parameter) of lane r's slot s, or -1 if the slot is padding
    static __device__ __forceinline__ int slot_index(int r, int s) {
        int res = -1;
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = N::dim(l), out = N::dim(l + 1);
            constexpr int lo = slot_off(l), hi = slot_off(l) + layer_slots(l);
            if constexpr (KMAJ && l == L - 1) {
                if (s >= lo && s < hi) {
                    const int q = s - lo;
                    if (q < OM * out) { const int k = r + (q / out) * G; if (k < in) res = N::off(l) + q % out + k * out; }
                    else if (r < out) res = N::off(l) + in * out + r;
                }
            } else
            if (s >= lo && s < hi) {
                const int m = (s - lo) / (in + 1), k = (s - lo) % (in + 1);
                const int j = r + m * G;
                if (j < out) res = N::off(l) + (k < in ? j + k * out : in * out + j);
            }
        });
        return res;
    }
    // ... of a run-time shape (NetCfgRt): the slots are those of the padded shape, the theta indices those of the true one
    template <class MC>
    static __device__ __forceinline__ int slot_index_rt(const MC& mc, int r, int s) {
        int res = -1, offr = 0;
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = N::dim(l), out = N::dim(l + 1);
            constexpr int lo = slot_off(l), hi = slot_off(l) + layer_slots(l);
            const int inr = mc.dims[l], outr = mc.dims[l + 1];
            if constexpr (KMAJ && l == L - 1) {
                if (s >= lo && s < hi) {
                    const int q = s - lo;
                    if (q < OM * out) { const int k = r + (q / out) * G; if (k < inr) res = offr + q % out + k * outr; }
                    else if (r < outr) res = offr + inr * outr + r;
                }
            } else
            if (s >= lo && s < hi) {
                const int m = (s - lo) / (in + 1), k = (s - lo) % (in + 1);
                const int j = r + m * G;
                if (j < outr) {
                    if (k < inr) res = offr + j + k * outr;
                    else if (k == in) res = offr + inr * outr + j;
                }
            }
            offr += inr * outr + outr;
        });
        return res;
    }
};

}  // namespace ude
