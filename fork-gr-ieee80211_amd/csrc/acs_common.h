// Pieces of the 64-state add-compare-select recursion shared by the decoder kernels.
#pragma once

#include "rx_common.h"

namespace gr80211 {

template <int KIND, int OP>
__device__ __forceinline__ float add_tab(float m, float t0, float t1, float t01)
{
    // KIND 0: both soft bits present; 1: second erased (t1 = 0); 2: first erased (t0 = 0).
    // Adding the table entry 0.0f leaves a metric unchanged, so those adds are dropped.
    if constexpr (KIND == 0) {
        if constexpr (OP == 0) return m;
        else if constexpr (OP == 1) return m + t1;
        else if constexpr (OP == 2) return m + t0;
        else return m + t01;
    } else if constexpr (KIND == 1) {
        if constexpr (OP == 0 || OP == 1) return m;
        else return m + t0;
    } else {
        if constexpr (OP == 0 || OP == 2) return m;
        else return m + t1;
    }
}

// The larger of two path metrics as one v_max_f32.  fmaxf would first quiet a possible signalling NaN in every operand
// whose origin the compiler cannot see -- a v_max_f32 of its own for each metric that enters a step unchanged from the
// loop's previous round, 32 per pair of puncture periods.  No signalling NaN gets here: a metric is a constant, a sum
// (an add quiets what it is given) or the result of this instruction.
__device__ __forceinline__ float metric_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Puncture patterns (lib/cloud80211phy.cc:1857-1860): P trellis steps consume COLS soft bits.
template <int CR> struct Rate;
template <> struct Rate<CR_12> { static constexpr int P = 1, COLS = 2, NPER = 32; };
template <> struct Rate<CR_23> { static constexpr int P = 2, COLS = 3, NPER = 20; };
template <> struct Rate<CR_34> { static constexpr int P = 3, COLS = 4, NPER = 16; };
template <> struct Rate<CR_56> { static constexpr int P = 5, COLS = 6, NPER = 10; };
// Step ps of a period: kind 0 = both soft bits present, 1 = second erased, 2 = first erased.
constexpr int step_kind(int cr, int ps) { return (cr == CR_12 || ps == 0) ? 0 : (ps & 1) ? 1 : 2; }
constexpr int step_col(int ps) { return ps == 0 ? 0 : ps + 1; }

constexpr int TILE_STRIDE = 65;  // floats per row of the 64 x 64 soft-bit tile in LDS (odd: conflict-free columns)

}  // namespace gr80211
