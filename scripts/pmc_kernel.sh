#!/bin/bash
# usage (GPU box, from the repo root): scripts/pmc_kernel.sh <tag> script.py [args...]
# Two SEPARATE rocprofv3 --pmc passes (kernel trace only beside them) over the script; scripts/pmc_kernel_summary.py
# <tag> <kernel substring> prints the per-launch means and the ratios (see scripts/pmc_valu_summary.py for the units).
set -u
TAG=$1; shift
ROOT=$(pwd)
CMD="$ROOT/$1"; shift
cd /tmp && export TMPDIR=/tmp
run() {
  local sub=$1; shift
  local out=$ROOT/gpurun_out/pmc_kernel_${TAG}_$sub
  rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $out -o pmc -- python3 $CMD "$@" > $out.log 2>&1
  echo "pmc_kernel $TAG $sub rc=$?"
}
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" run a "$@" && \
PMC="SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" run b "$@"
