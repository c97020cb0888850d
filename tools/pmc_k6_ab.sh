#!/bin/bash
# TCP / SQ counters of the chain-backward kernel for the shipped library and one variant library (tools/variant.sh)
#   tools/pmc_k6_ab.sh VARIANT OUTDIR
cd "$(dirname "$0")/.."
V=$1; O=$2
tools/pmc_kernel.sh "iter_chain_bwd" $O/base "TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU"
TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$V.so tools/pmc_kernel.sh "iter_chain_bwd" $O/$V "TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU"
