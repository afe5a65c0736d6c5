// Magnitude bound of one operand of a two-piece fp16 split GEMM (gemm_split.h, NP == 2), kept on the device:
//     |x| <= bits_as_float(amax[set * stride]) * (gain ? *gain : 1) * hgain      for every element x of the set's rows.
// amax slots hold fp32 BIT PATTERNS of non-negative values (they order as unsigned integers: producers use atomicMax on zeroed
// slots); gain is the aggregate's row gain of the batch (gm_batch::d_gain) when the operand is an aggregate of the bounded tensor;
// hgain a host-side factor (the head-room of weights whose planes are written before their own maximum is known).
#pragma once
#include <stdint.h>
// Slots of different sets sit GM_BOUND_PAD words (256 B) apart: device-scope atomics / coherent loads on one cache line are served one at a
// time by that line's memory channel (thousands of them per launch were measured as 10-250 us), different lines go to different channels.
#define GM_BOUND_PAD 64
// viol (optional): a device word that receives GM_VIOL_* bits when a producer finds the bound broken -- a weight that outgrew the head-room of
// its bound (its fp16 pieces would be inf), a recorded maximum beyond the scale range.  gm_meta_step reports the word in the last float of `out`;
// the reference's fp32 path has no such limits, so the host re-runs such a step with the three-piece kernels instead of returning its numbers.
#define GM_VIOL_WEIGHT 1u      // |w| * scale > fp16 max: a fast weight outgrew GM_W_HEADROOM x max |theta_W|
#define GM_VIOL_RANGE 2u       // a recorded operand maximum above 2^50 (scales are clamped to [2^-40, 2^40])
struct gm_bound { const unsigned* amax; int64_t stride; const float* gain; float hgain; unsigned* viol; };
static inline gm_bound gm_no_bound() { gm_bound b; b.amax = nullptr; b.stride = 0; b.gain = nullptr; b.hgain = 1.f; b.viol = nullptr; return b; }
