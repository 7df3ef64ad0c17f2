// ude_node_ls.hip -- translation unit of the lock-step matrix-core adjoint of the SEIR neural ODE (ude_node_ls.h).
#include <hip/hip_runtime.h>
#include "ude_node_ls.h"
// the parity-mode backward kernel (second generation, round 5: ude_node_ls2.h; ude_node_ls.h keeps what it shares with the fast mode)
#include "ude_node_ls2.h"
// two blocks of the forward kernel per compute unit (60 KB of LDS each, 256 registers per lane)
#ifndef UDE_LS_FWD_PER_CU
#define UDE_LS_FWD_PER_CU 2
#endif
#include "ude_node_ls_fwd.h"
using namespace ude;
extern "C" void ude_node_ls_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, size_t* fac_doubles_per_block) {
    *fac_doubles_per_block = alg == 1 ? nodels::fac_doubles_per_block<Vern7Tab>() : nodels::fac_doubles_per_block<Tsit5Tab>();
    *kern = alg == 1 ? nodels2::node_ls2_adj_kernel<Vern7Tab> : nodels2::node_ls2_adj_kernel<Tsit5Tab>;
    *lds_bytes = sizeof(double) * (alg == 1 ? nodels2::lds_doubles<Vern7Tab>() : nodels2::lds_doubles<Tsit5Tab>()) + 16;
}

// the forward solve on the same architecture (ude_node_ls_fwd.h)
extern "C" void ude_node_ls_get_fwd(int alg, void (**kern)(const KParams, int*), size_t* lds_bytes, int* blocks_per_cu) {
    *blocks_per_cu = UDE_LS_FWD_PER_CU;
    *kern = alg == 1 ? nodels::node_ls_fwd_kernel<Vern7Tab> : nodels::node_ls_fwd_kernel<Tsit5Tab>;
    *lds_bytes = sizeof(double) * (alg == 1 ? nodels::fwd_lds_doubles<Vern7Tab>() : nodels::fwd_lds_doubles<Tsit5Tab>()) + 16;
}
