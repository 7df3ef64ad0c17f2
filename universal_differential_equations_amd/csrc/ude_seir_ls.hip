// ude_seir_ls.hip -- translation unit of the lock-step matrix-core adjoint of the SEIR exposure UDE (ude_seir_ls.h).
#include <hip/hip_runtime.h>

#include "ude_seir_ls.h"
// the parity-mode backward kernel (second generation, round 5: ude_seir_ls2.h; ude_seir_ls.h keeps what it shares with the fast mode)
#include "ude_seir_ls2.h"
// two blocks of the forward kernel per compute unit (256 registers per lane: 132 B of scratch in cold paths): 2.7 -> 1.8 ms
#ifndef UDE_LS_FWD_PER_CU
#define UDE_LS_FWD_PER_CU 2
#endif
#include "ude_seir_ls_fwd.h"

using namespace ude;

// kernel entry points for udecore.hip: alg 0 = Tsit5, 1 = Vern7
extern "C" void ude_seir_ls_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, size_t* fac_doubles_per_block) {
    *fac_doubles_per_block = alg == 1 ? seirls::fac_doubles_per_block<Vern7Tab>() : seirls::fac_doubles_per_block<Tsit5Tab>();
    *kern = alg == 1 ? seirls2::seir_ls2_adj_kernel<Vern7Tab> : seirls2::seir_ls2_adj_kernel<Tsit5Tab>;
    *lds_bytes = sizeof(double) * (alg == 1 ? seirls2::lds_doubles<Vern7Tab>() : seirls2::lds_doubles<Tsit5Tab>()) + 16;
}

// the runtime-shape instance of the second-generation kernel: any exposure-UDE chain 3 -> H1 -> H2 -> 1 (tanh, tanh, identity), H1, H2 <= 64,
// that udecore.hip's seir_gen_ls_shape admits (ude_seir_ls2.h, GEN = true)
extern "C" void ude_seir_ls_get_gen(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, size_t* fac_doubles_per_block) {
    *fac_doubles_per_block = alg == 1 ? seirls::fac_doubles_per_block<Vern7Tab>() : seirls::fac_doubles_per_block<Tsit5Tab>();
    *kern = alg == 1 ? seirls2::seir_ls2_adj_kernel<Vern7Tab, true> : seirls2::seir_ls2_adj_kernel<Tsit5Tab, true>;
    *lds_bytes = sizeof(double) * (alg == 1 ? seirls2::lds_doubles<Vern7Tab>() : seirls2::lds_doubles<Tsit5Tab>()) + 16;
}

// ... and the runtime-shape instance of the FORWARD kernel (additionally H2 != 32: a 32-term output layer is the oracle's tree case)
extern "C" void ude_seir_ls_get_fwd_gen(int alg, void (**kern)(const KParams, int*), size_t* lds_bytes, int* blocks_per_cu) {
    *blocks_per_cu = UDE_LS_FWD_PER_CU;
    *kern = alg == 1 ? seirls::seir_ls_fwd_kernel<Vern7Tab, true> : seirls::seir_ls_fwd_kernel<Tsit5Tab, true>;
    *lds_bytes = sizeof(double) * (alg == 1 ? seirls::fwd_lds_doubles<Vern7Tab>() : seirls::fwd_lds_doubles<Tsit5Tab>()) + 16;
}

// the forward solve on the same architecture (ude_seir_ls_fwd.h)
extern "C" void ude_seir_ls_get_fwd(int alg, void (**kern)(const KParams, int*), size_t* lds_bytes, int* blocks_per_cu) {
    *blocks_per_cu = UDE_LS_FWD_PER_CU;
    *kern = alg == 1 ? seirls::seir_ls_fwd_kernel<Vern7Tab> : seirls::seir_ls_fwd_kernel<Tsit5Tab>;
    *lds_bytes = sizeof(double) * (alg == 1 ? seirls::fwd_lds_doubles<Vern7Tab>() : seirls::fwd_lds_doubles<Tsit5Tab>()) + 16;
}
