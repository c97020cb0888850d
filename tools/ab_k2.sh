#!/bin/bash
# Ablation variants of the IWE scatter kernel K2 (timing only; built from patched COPIES of tef_loss.hip by tools/variant.sh):
#   noatom    the eight LDS accumulations of an event replaced by a register sink (weight math and conversions kept)
#   noproc    hit-row FIFO and event loads as they are, but no event is processed (no weight math, no atomics)
#   nosweep   no sweep at all: queue, zero fill, run tables, barriers, statistics + write-out only
# Together: what the kernel costs with every instruction of the splat removed (the ceiling of the design).
cd "$(dirname "$0")/.."
SINK='__device__ __forceinline__ void tef_sink(unsigned long long *p, unsigned long long v) { asm volatile("" :: "v"(p), "v"(v)); }'
tools/variant.sh noatom "s|^typedef float f32x2_e .*|&\n$SINK|; /^__device__ __forceinline__ void splat_fixed/,/^}/ s/atomicAdd\((c0|q0)/tef_sink(\1/"
tools/variant.sh noproc '/const int rr = \(int\)floorf\(qd\[k\]\.y\) - r0m1;/,/if \(take\) \{/ s/if \(take\) \{/if (take \&\& im.tref < -1.0f) {/'
tools/variant.sh nosweep '/void splat_stats_kernel/,/void loss_reduce_kernel/ { s/^        if \(fixed\) \{$/        if (fixed \&\& im.tref < -1.0f) {/; s/for \(int li = 0; li < nlists; \+\+li\)/for (int li = 0; li < 0; ++li)/ }'
tools/ab.sh noatom noproc nosweep
