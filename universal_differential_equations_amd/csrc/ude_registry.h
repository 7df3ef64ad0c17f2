// ude_registry.h -- kernel instance registry shared by udecore.hip and the per-instance translation units.
#pragma once
#include "ude_kernels.h"

namespace ude {

struct Launch {
    void (*fwd)(const KParams);
    void (*adj)(const KParams);
    void (*dadj)(const KParams);  // discretise-then-optimise reverse sweep (a9)
    void (*rhs)(const KParams);   // one right-hand-side evaluation per state
    void (*fwd_pt)(const KParams), (*adj_pt)(const KParams), (*dadj_pt)(const KParams);  // per-trajectory tspan / saveat
    void (*adj_fast)(const KParams);  // UDE_SENSE_FAST: lambda-only error control (shared time grid only)
    void (*adj_ckpt)(const KParams);  // checkpointed adjoint: store u only, recompute the stages (null: no such instance)
    void (*adj_sorted)(const KParams);  // cost-ordered launch of a multi-round ensemble (KParams::perm; lane-group models only, else null)
    void (*fwd_sorted)(const KParams);  // ... and its forward kernel: the same order, workspace columns = lane-group positions
    int nf;  // dense fields per step
    int G, block;
    int block_fwd;  // threads per block of the forward / rhs kernels (Model::FWD_BLOCK_THREADS or Model::FwdModel; else = block)
    int G_fwd;      // lanes per trajectory of the forward / rhs kernels (Model::FwdModel; else = G)
    // dynamic LDS (doubles): theta copy (<0: (np+1)&~1) + scratch + k [+ adjoint: slot columns (mu, FSAL hand-over) + interval cache]
    int theta_lds, scratch, scratch_fwd, k_doubles, k_doubles_d, slots_reg, k_doubles_fwd, k_doubles_adj;
    bool dadj_k_dense;  // the reverse sweep reads k from the dense store (HBM) instead of an LDS copy
    int slot_glob;  // > 0: slot state in HBM, this many elements per thread (SLOTS_GLOBAL models)
    int elem;       // bytes of the instance's scalar type (8: Float64, 4: Float32)
    bool per_member;  // the per-trajectory kernel variants take per-member parameters (UDE_PT_THETA)
    size_t lds_bytes(int np, bool adjoint, bool discrete = false) const {
        const size_t np_pad = (size_t)((np + 1) & ~1);
        size_t d = (theta_lds < 0 ? np_pad : (size_t)theta_lds) + ((adjoint || discrete) ? scratch : scratch_fwd) + (discrete ? k_doubles : adjoint ? k_doubles_adj : k_doubles_fwd);
        if (discrete) d += (dadj_k_dense ? 1 : 2) * (size_t)k_doubles_d - k_doubles;  // [k and] kbar in the reverse sweep's own layout
        if (adjoint) d += (size_t)slots_reg;
        return d * (size_t)elem + 16;
    }
};

// models without a discretise-then-optimise sweep (Model::NO_DADJ: the runtime-shape fallback) leave those entries null
template <class M, class = void> struct no_dadj { static constexpr bool v = false; };
template <class M> struct no_dadj<M, std::void_t<decltype(M::NO_DADJ)>> { static constexpr bool v = M::NO_DADJ; };

// a model whose forward / rhs kernels are a different instantiation (Model::FwdModel, with FwdModel::G lanes per trajectory = threads per block)
template <class M, class = void> struct fwd_model { using type = M; static constexpr int G = 0; };
template <class M> struct fwd_model<M, std::void_t<typename M::FwdModel>> {
    using type = std::conditional_t<std::is_same<typename M::FwdModel, M>::value, M, typename M::FwdModel>;
    static constexpr int G = M::FwdModel::G;
};
// a model whose forward / rhs kernels run with fewer threads per block than its adjoint (Model::FWD_BLOCK_THREADS: LDS per block)
template <class M, int BLOCK, class = void> struct fwd_block_threads { static constexpr int v = BLOCK; };
template <class M, int BLOCK> struct fwd_block_threads<M, BLOCK, std::void_t<decltype(M::FWD_BLOCK_THREADS)>> { static constexpr int v = M::FWD_BLOCK_THREADS; };
// per-thread HBM words behind the two mu columns that a model keeps stage factors in (Model::GFAC: SeirNode's a3 / delta1 rows)
template <class M, class = void> struct gfac_words { static constexpr int v = 0; };
template <class M> struct gfac_words<M, std::void_t<decltype(M::GFAC)>> { static constexpr int v = M::GFAC; };
// models that declare Model::RECOMPUTE_OK get the checkpointed-adjoint kernel (store u only, recompute the stages) for FSAL tableaux
template <class M, class = void> struct recompute_ok { static constexpr bool v = false; };
template <class M> struct recompute_ok<M, std::void_t<decltype(M::RECOMPUTE_OK)>> { static constexpr bool v = M::RECOMPUTE_OK; };

template <class Model, class Tab, int G, int BLOCK = 64, int VAR = 1, class RTag = real>
inline Launch make_launch() {
    Launch l;
    // forward / rhs kernels: the model itself, with its own block size where it declares one, or a different instantiation of the
    // model altogether (Model::FwdModel: Fisher-KPP's eight-wavefront forward next to the four-wavefront adjoint)
    using FM = typename fwd_model<Model>::type;
    constexpr bool OWN_FWD = !std::is_same<FM, Model>::value;
    constexpr int GF = OWN_FWD ? fwd_model<Model>::G : G;
    constexpr int FB = OWN_FWD ? fwd_model<Model>::G : fwd_block_threads<Model, BLOCK>::v;
    l.fwd = fwd_kernel<FM, Tab, GF, FB>;
    l.adj = adj_kernel<Model, Tab, G, BLOCK, false, VAR>;
    if constexpr (no_dadj<Model>::v) { l.dadj = nullptr; l.dadj_pt = nullptr; }
    else { l.dadj = dadj_kernel<Model, Tab, G, BLOCK>; l.dadj_pt = dadj_kernel<Model, Tab, G, BLOCK, true>; }
    l.rhs = rhs_kernel<FM, Tab, GF, FB>;
    l.fwd_pt = fwd_kernel<FM, Tab, GF, FB, true>;
    l.adj_pt = adj_kernel<Model, Tab, G, BLOCK, true, VAR>;
    l.adj_fast = adj_kernel<Model, Tab, G, BLOCK, false, 3>;
    // checkpointed adjoint (store u only, recompute the stages): every replicated-state model; distributed states where the NK
    // recomputed stage vectors fit the registers (AdjSys::RECOMPUTE) -- Model::RECOMPUTE_OK marks the distributed models that have it
    if constexpr (!Model::STATE_DISTRIBUTED || (recompute_ok<Model>::v && (Tab::NK * Model::NS <= 32 || (Tab::FSAL && Tab::NK == Tab::S))))
        l.adj_ckpt = adj_kernel<Model, Tab, G, BLOCK, false, 5>;
    else l.adj_ckpt = nullptr;
    // cost-ordered launch: models with several members per wavefront and their parameter slots in registers (the LV family on 4 .. 32 lanes)
    if constexpr (G < 64 && BLOCK == 64 && !Model::STATE_DISTRIBUTED && !Model::SLOTS_GLOBAL && !Model::DEFERRED && VAR == 1)
        l.adj_sorted = adj_kernel<Model, Tab, G, BLOCK, false, 6>;
    else l.adj_sorted = nullptr;
    if constexpr (G < 64 && BLOCK == 64 && !Model::STATE_DISTRIBUTED && !Model::SLOTS_GLOBAL && !Model::DEFERRED && VAR == 1 && !OWN_FWD)
        l.fwd_sorted = fwd_kernel<FM, Tab, GF, FB, false, real, true>;
    else { l.fwd_sorted = nullptr; l.adj_sorted = nullptr; }
    l.nf = Tab::NK;  // dense fields per step = 2 + n_state + NK * n_state (host adds the state size)
    l.G = G;
    l.block = BLOCK;
    l.block_fwd = FB;
    l.G_fwd = GF;
    l.k_doubles_fwd = Layout<FM, Tab, GF, FB>::K_DOUBLES;
    l.theta_lds = Model::theta_lds(7) == 8 ? -1 : Model::theta_lds(0);
    l.scratch = Model::SCRATCH;
    l.scratch_fwd = scratch_fwd<FM>::v;
    l.k_doubles = Layout<Model, Tab, G, BLOCK>::K_DOUBLES;
    l.k_doubles_adj = Layout<Model, Tab, G, BLOCK>::K_DOUBLES_ADJ;
    l.k_doubles_d = Layout<Model, Tab, G, BLOCK, false>::K_DOUBLES;
    l.dadj_k_dense = Model::DADJ_K_FROM_DENSE;
    l.slots_reg = (Model::SLOTS_GLOBAL ? 0 : (Tab::FSAL ? 3 : 2) * (Model::NSL > 0 ? Model::NSL : 1) * BLOCK) + Layout<Model, Tab, G, BLOCK>::IC_DOUBLES;
    l.slot_glob = Model::SLOTS_GLOBAL ? (Model::DEFERRED ? 2 : 1) * Model::NSL + gfac_words<Model>::v : act_cache<Model>::v;   // (or the model's activation row: AdjSys::ACT_CACHE)
    l.elem = (int)sizeof(real);
    l.per_member = per_member<Model>::v && !Model::SLOTS_GLOBAL;
    return l;
}

// network configurations of the reference scripts
using NetS1 = NetCfg<IntList<2, 5, 5, 5, 2>, IntList<ACT_RBF, ACT_RBF, ACT_RBF, ACT_IDENTITY>>;       // scenario_1/2.jl:62-64
using NetHudson = NetCfg<IntList<2, 5, 5, 5, 2>, IntList<ACT_RBF, ACT_RBF, ACT_TANH, ACT_IDENTITY>>;  // hudson_bay.jl:77-79
using NetTanh32 = NetCfg<IntList<2, 32, 2>, IntList<ACT_TANH, ACT_IDENTITY>>;                         // BASELINE C2 "2-layer tanh"
// run-time shapes of the LV kind on the lane-group kernels: 2 -> (two / three hidden layers of width <= 8, any activation) -> 2
using NetLvRt2 = NetCfgRt<2, 2, 2, 8>;      // (one hidden layer: BASELINE's "2-layer MLP" with an edited width)
using NetLvRt2W16 = NetCfgRt<2, 2, 2, 16>;
using NetLvRt3 = NetCfgRt<2, 2, 3, 8>;
using NetLvRt4 = NetCfgRt<2, 2, 4, 8>;
// ... and of width <= 5 on FIVE lanes (the headline instance's layout: twelve trajectories per wavefront, the 10k ensemble in one round) --
// what an edit of the activations alone needs
// ... and of width <= 16 on SIXTEEN lanes (four trajectories per wavefront), the weights read from the block's LDS copy of theta at every use
using NetLvRt3W16 = NetCfgRt<2, 2, 3, 16>;
using NetLvRt4W16 = NetCfgRt<2, 2, 4, 16>;
using NetLvRt3W5 = NetCfgRt<2, 2, 3, 5>;
using NetLvRt4W5 = NetCfgRt<2, 2, 4, 5>;

enum { MID_NONE = -1, MID_LV_TRUE = 0, MID_LV_S1, MID_LV_HUDSON, MID_LV_TANH32, MID_SEIR_TRUE, MID_SEIR_UDE,
       MID_KPP_TRUE_32, MID_KPP_TRUE_1024, MID_KPP_UDE_32, MID_KPP_UDE_1024, MID_KPP_S3_32, MID_KPP_SMALL_32,
       MID_LV_S1N /* scenario_1's chain with CONSTANT diagonal coefficients: no slots for them */,
       // Float32 problems (ude_model_desc.dtype = 1): hudson_bay.jl:77-104, scenario_3.jl:26-57 (true Fisher-KPP) and :83-126 (its UDE)
       MID_LV_HUDSON_F32, MID_KPP_TRUE_32_F32, MID_KPP_S3_32_F32,
       MID_SEIR_NODE /* the pure neural ODE 7-64-64-64-7 of seir_exposure.jl:53-73 */,
       // runtime-shape fallback (ude_model_generic.h): any chain of <= 8 Dense layers of width <= 64 for the replicated-state kinds
       MID_GENERIC_2 /* UDE_KIND_LV_UDE */, MID_GENERIC_7 /* UDE_KIND_SEIR_UDE, UDE_KIND_SEIR_NODE */,
       MID_KPP_SMALL1_32, MID_KPP_SMALL2_32 /* Fisher-KPP-CNN-Small.jl:88 with n_weights = 1, 2 */,
       MID_GENERIC_2_L4, MID_GENERIC_7_L4 /* the runtime-shape fallback for chains of <= 4 layers: half the LDS, twice the wavefronts per CU */,
       MID_GENERIC_2_F32, MID_GENERIC_2_L4_F32 /* Float32 LV-kind problems with any chain (hudson_bay.jl:77-79) */,
       MID_KPP_GENERIC_32 /* nn_ode with any pointwise reaction chain of <= 4 layers, width <= 32 (ude_model_kpp_generic.h) */,
       MID_LV_RT3, MID_LV_RT4 /* LV kind, run-time shape 2 -> (2 / 3 hidden layers of width <= 8, any activation) -> 2 on 8-lane groups (NetCfgRt) */,
       MID_LV_RT3_W5, MID_LV_RT4_W5 /* ... of width <= 5 on 5-lane groups */,
       MID_LV_RT3_F32, MID_LV_RT4_F32, MID_LV_RT3_W5_F32, MID_LV_RT4_W5_F32 /* ... as Float32 problems (hudson_bay.jl:77-104 with an edited FastChain) */,
       MID_LV_RT3_W16, MID_LV_RT4_W16 /* ... of width <= 16 on 16-lane groups, weights read from the LDS copy of theta (Float64) */,
       MID_LV_RT2, MID_LV_RT2_W16 /* LV kind, ONE hidden layer of width <= 8 / <= 16 (run-time shape) */,
       MID_KPP_RT_1024 /* nn_ode on 33 .. 1024 points with any reaction chain 1 -> a -> b -> c -> 1, tanh hidden layers of width <= 16 (KppUdeW over NetCfgRt) */ };

using NetKpp = NetCfg<IntList<1, 10, 20, 10, 1>, IntList<ACT_TANH, ACT_TANH, ACT_TANH, ACT_IDENTITY>>;  // Fisher-KPP-CNN.jl:92-96
using NetKppS3 = NetCfg<IntList<1, 5, 5, 5, 1>, IntList<ACT_RBF, ACT_RBF, ACT_RBF, ACT_IDENTITY>>;      // scenario_3.jl:83-88
using NetKppRt16 = NetCfgRt<1, 1, 4, 16>;   // run-time shape of the reaction network on the large grids: three tanh layers of width <= 16
using NetKppSmall = NetCfg<IntList<1, 3, 1>, IntList<ACT_TANH, ACT_IDENTITY>>;                        // Fisher-KPP-CNN-Small.jl:89-94 (15 parameters)
using NetKppSmall1 = NetCfg<IntList<1, 1, 1>, IntList<ACT_TANH, ACT_IDENTITY>>;                       // n_weights = 1 (:88, timing log :343-391)
using NetKppSmall2 = NetCfg<IntList<1, 2, 1>, IntList<ACT_TANH, ACT_IDENTITY>>;                       // n_weights = 2

}  // namespace ude
