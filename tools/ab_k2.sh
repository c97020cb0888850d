#!/bin/bash
# Ablation variants of the two scatter kernels (timing only; built from patched COPIES of tef_loss.hip by tools/variant.sh):
#   noatom   the eight LDS accumulations of an event replaced by a register sink (conversions kept)
#   oneline  every event load of a batch reads the same 128 bytes (cache hits: no memory system behind the sweep)
cd "$(dirname "$0")/.."
SINK='__device__ __forceinline__ void tef_sink(unsigned long long *p, unsigned long long v) { asm volatile("" :: "v"(p), "v"(v)); }'
tools/variant.sh noatom "s|^typedef float f32x2_e .*|&\n$SINK|; s/atomicAdd\((c0|q0)/tef_sink(\1/"
tools/variant.sh oneline 's/const uint32_t off = \(uint32_t\)lane_list\[s_ \+ 4 \* k\] \* 128u \+ lane_off;/const uint32_t off = lane_off + 0u * (uint32_t)lane_list[s_ + 4 * k];/; s/const uint32_t voff = \(en \& 0xffffffu\) \* 128u \+ lane_off;/const uint32_t voff = lane_off + 0u * en;/'
tools/ab.sh noatom oneline
