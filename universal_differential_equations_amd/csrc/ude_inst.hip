// ude_inst.hip -- one kernel instance per translation unit (compiled many times by build.py with
// -DINST_NAME=... -DINST_MODEL=... -DINST_TAB=... -DINST_G=...), so the heavy templates build in parallel.
#include <hip/hip_runtime.h>

#include "ude_registry.h"

using namespace ude;
using InstModel = INST_MODEL;

#ifndef INST_BLOCK
#define INST_BLOCK 64
#endif
#ifndef INST_VAR
#define INST_VAR 1
#endif
extern "C" void INST_NAME(Launch* out) { *out = make_launch<InstModel, INST_TAB, INST_G, INST_BLOCK, INST_VAR>(); }
