#!/bin/bash
# usage (GPU box, from the repo root): scripts/pmc_valu.sh <tag> [ANIREC_LIB_PATH]
# Three SEPARATE rocprofv3 --pmc passes (never combined with each other or with another trace domain than the
# kernel trace) over `bench.py --no-also --no-cpu-baseline --steps 96 --warmup 32`:
#   a  SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
#   b  SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_SCA
#   c  GRBM_GUI_ACTIVE
# Output: gpurun_out/pmc_valu_<tag>_{a,b,c}/ ; scripts/pmc_valu_summary.py <tag> turns them into one JSON.
set -u
TAG=$1
ROOT=$(pwd)
if [ $# -ge 2 ]; then export ANIREC_LIB_PATH=$2; fi
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/bench.py --no-also --no-cpu-baseline --steps 96 --warmup 32"
run() {
  local sub=$1; shift
  local out=$ROOT/gpurun_out/pmc_valu_${TAG}_$sub
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o pmc -- python3 $CMD > $out.log 2>&1
  echo "pmc_valu $TAG $sub rc=$?"
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run b SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_SCA && \
run c GRBM_GUI_ACTIVE
