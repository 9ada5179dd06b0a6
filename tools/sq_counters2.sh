#!/bin/bash
# SQ counter passes incl. LDS counters for one command: usage tools/sq_counters2.sh OUTDIR -- python3 script.py
out=$1; shift; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  ( cd $root && timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -o p -- "$@" > /dev/null 2>&1 )
  echo "pass $i ($set) rc=$?"
  i=$((i+1))
done
( cd $root && timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/pass9 -o p -- "$@" > /dev/null 2>&1 ); echo "pass coexec rc=$?"
