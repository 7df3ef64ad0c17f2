#!/bin/bash
# Reproduces the code-generation defect of DESIGN.md 2a from this repository's own sources (no GPU needed: hipcc cross-compiles):
# compiles one kernel instance to device assembly with the PLAIN compiler (no repair) and lets tools/isa_endcf_fix.py --audit list
# every join block whose register copies stand in front of the EXEC restore.  Which instances contain sites moves with every change
# of the sources and of the compiler; at the time of writing (ROCm 7.2.0, round 5): adj_kernel<LvUde<NetS1, 1>, Tsit5Tab> (the
# checkpointed variant) with 104 v_accvgpr_write in front of `s_or_b64 exec, exec, s[12:13]`.
#   usage: tools/probe/endcf_repro.sh [model] [lanes] [tab]      default: "LvUde<NetS1,1>" 1 Tsit5Tab
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
MODEL=${1:-"LvUde<NetS1,1>"}; G=${2:-1}; TAB=${3:-Tsit5Tab}
OUT=${TMPDIR:-/tmp}/endcf_repro.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -S \
    -DINST_NAME=repro "-DINST_MODEL=$MODEL" -DINST_TAB=$TAB -DINST_G=$G -DINST_VAR=1 -DINST_BLOCK=64 \
    "$R/universal_differential_equations_amd/csrc/ude_inst.hip" -o "$OUT"
echo "device assembly: $OUT"
python3 "$R/tools/isa_endcf_fix.py" --audit "$OUT" && echo "no site in this instance with this compiler"
