// ude_seir_lsf.hip -- translation unit of the `fast` lock-step adjoint of the SEIR exposure UDE with the parameter cotangent as a
// block-level matrix-core accumulation (ude_seir_lsf.h).
#include <hip/hip_runtime.h>

#include "ude_seir_lsf.h"

using namespace ude;

// kernel entry point for udecore.hip: alg 0 = Tsit5, 1 = Vern7
extern "C" void ude_seir_lsf_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, int* blocks_per_cu) {
    *blocks_per_cu = 1;
    if (alg == 1) {
        *kern = seirlf::seir_lsf_adj_kernel<Vern7Tab>;
        *lds_bytes = sizeof(double) * seirlf::lds_doubles<Vern7Tab>() + 16;
    } else {
        *kern = seirlf::seir_lsf_adj_kernel<Tsit5Tab>;
        *lds_bytes = sizeof(double) * seirlf::lds_doubles<Tsit5Tab>() + 16;
    }
}

// ... and its runtime-shape instance (exposure chains 3 -> H1 -> H2 -> 1 without a compiled instance: udecore.hip, seir_gen_ls_shape)
extern "C" void ude_seir_lsf_get_gen(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, int* blocks_per_cu) {
    *blocks_per_cu = 1;
    if (alg == 1) {
        *kern = seirlf::seir_lsf_adj_kernel<Vern7Tab, true>;
        *lds_bytes = sizeof(double) * seirlf::lds_doubles<Vern7Tab>() + 16;
    } else {
        *kern = seirlf::seir_lsf_adj_kernel<Tsit5Tab, true>;
        *lds_bytes = sizeof(double) * seirlf::lds_doubles<Tsit5Tab>() + 16;
    }
}
