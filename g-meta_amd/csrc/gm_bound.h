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
struct gm_bound { const unsigned* amax; int64_t stride; const float* gain; float hgain; };
static inline gm_bound gm_no_bound() { gm_bound b; b.amax = nullptr; b.stride = 0; b.gain = nullptr; b.hgain = 1.f; return b; }
