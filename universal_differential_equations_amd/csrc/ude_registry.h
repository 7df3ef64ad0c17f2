// ude_registry.h -- kernel instance registry shared by udecore.hip and the per-instance translation units.
#pragma once
#include "ude_kernels.h"

namespace ude {

struct Launch {
    void (*fwd)(const KParams);
    void (*adj)(const KParams);
    int nf;  // dense fields per step
    int G, block;
    int lds_fwd, lds_adj;  // dynamic LDS doubles per thread besides theta (stage derivatives [+ slot state])
};

constexpr int BLOCK = 64;

template <class Model, class Tab, int G>
inline Launch make_launch() {
    Launch l;
    l.fwd = fwd_kernel<Model, Tab, G, BLOCK>;
    l.adj = adj_kernel<Model, Tab, G, BLOCK>;
    l.nf = 2 + Model::NS + Tab::NK * Model::NS;
    l.G = G;
    l.block = BLOCK;
    l.lds_fwd = Tab::NK * Model::NS;
    l.lds_adj = Tab::NK * Model::NS + Model::NSL;
    return l;
}

// network configurations of the reference scripts
using NetS1 = NetCfg<IntList<2, 5, 5, 5, 2>, IntList<ACT_RBF, ACT_RBF, ACT_RBF, ACT_IDENTITY>>;       // scenario_1/2.jl:62-64
using NetHudson = NetCfg<IntList<2, 5, 5, 5, 2>, IntList<ACT_RBF, ACT_RBF, ACT_TANH, ACT_IDENTITY>>;  // hudson_bay.jl:77-79
using NetTanh32 = NetCfg<IntList<2, 32, 2>, IntList<ACT_TANH, ACT_IDENTITY>>;                         // BASELINE C2 "2-layer tanh"

enum { MID_NONE = -1, MID_LV_TRUE = 0, MID_LV_S1, MID_LV_HUDSON, MID_LV_TANH32 };

}  // namespace ude
