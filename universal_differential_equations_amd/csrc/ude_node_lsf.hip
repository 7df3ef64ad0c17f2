// ude_node_lsf.hip -- translation unit of the `fast` lock-step adjoint of the SEIR script's neural ODE with the parameter cotangent as a
// block-level matrix-core accumulation (ude_node_lsf.h).
#include <hip/hip_runtime.h>

#include "ude_node_lsf.h"

using namespace ude;

// kernel entry point for udecore.hip: alg 0 = Tsit5, 1 = Vern7
extern "C" void ude_node_lsf_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, int* blocks_per_cu) {
    *blocks_per_cu = 1;
    if (alg == 1) {
        *kern = nodelf::node_lsf_adj_kernel<Vern7Tab>;
        *lds_bytes = sizeof(double) * nodelf::lds_doubles<Vern7Tab>() + 16;
    } else {
        *kern = nodelf::node_lsf_adj_kernel<Tsit5Tab>;
        *lds_bytes = sizeof(double) * nodelf::lds_doubles<Tsit5Tab>() + 16;
    }
}
