#!/bin/bash
# PMC passes (SQ, instruction mix, FETCH, WRITE: each its own run) + a kernel trace of ONE command, tables for kernels matching <filter>:
#   tools/prof_kernel.sh <tag> <kernel-name-filter> -- <command...>       (run from the repo root on the MI355X box; output under gpurun_out/<tag>/)
set -u
TAG=$1; FLT=$2; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd /tmp
run() { local name=$1; shift; rm -rf $OUT/$name; timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "[$name] rc=$?"; }
run sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/sq -- "$@"
run mix --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d $OUT/mix -- "$@"
run fetch --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- "$@"
run write --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- "$@"
cd $ROOT
{ echo "## SQ"; python tools/rocpd_pmc_table.py $OUT/sq "$FLT"; echo "## instruction mix"; python tools/rocpd_pmc_table.py $OUT/mix "$FLT";
  echo "## FETCH_SIZE (KB; x2 for bytes on gfx950)"; python tools/rocpd_pmc_table.py $OUT/fetch "$FLT"; echo "## WRITE_SIZE (KB)"; python tools/rocpd_pmc_table.py $OUT/write "$FLT"; } > $OUT/${TAG}_pmc.txt 2>&1
cat $OUT/${TAG}_pmc.txt
rm -rf $OUT/sq $OUT/mix $OUT/fetch $OUT/write        # the raw databases (tens of MB) stay on the box: gpurun_out/ is capped at 64 MiB
