// Device code of the Viterbi forward passes that share a group of 64 frames among several waves of one workgroup
// (kernels: vit_split.hip, rx_kernels.hip).
#pragma once

#include <utility>

#include "acs_common.h"
#include "rx_kernels.h"

namespace gr80211 {

constexpr int XCH_FLOATS = 64 * 64;  // metrics of one exchange buffer: [state][lane]
constexpr int XCH_BUF = XCH_FLOATS + 128;  // + decision nibbles [2][lane]

// Four butterflies JB .. JB+3: old states 2j, 2j+1 (held in X, whose first entry is state XO) -> new states j
// (input bit 0, to ol) and j + 32 (input bit 1, to oh).  Written phase by phase like the one-wave kernel's state
// quads; the decisions (sign of a0 - a1, set when the odd predecessor wins) go into one accumulator per half, first
// state in bit 3.
template <int KIND, int JB, int XO, int NX>
__device__ __forceinline__ void bfly4(const float (&X)[NX], float *ol, float *oh, float t0, float t1, float t01,
                                      uint32_t &accl, uint32_t &acch)
{
    float l0[4], l1[4], h0[4], h1[4], dl[4], dh[4];
#define GR_BF_IN(K)                                                      \
    {                                                                    \
        constexpr int j = JB + K, p0 = 2 * j, p1 = 2 * j + 1;            \
        l0[K] = add_tab<KIND, sv_out(p0, 0)>(X[p0 - XO], t0, t1, t01);   \
        l1[K] = add_tab<KIND, sv_out(p1, 0)>(X[p1 - XO], t0, t1, t01);   \
        h0[K] = add_tab<KIND, sv_out(p0, 1)>(X[p0 - XO], t0, t1, t01);   \
        h1[K] = add_tab<KIND, sv_out(p1, 1)>(X[p1 - XO], t0, t1, t01);   \
    }
    GR_BF_IN(0) GR_BF_IN(1) GR_BF_IN(2) GR_BF_IN(3)
#undef GR_BF_IN
#pragma unroll
    for (int k = 0; k < 4; k++) {
        dl[k] = l0[k] - l1[k];
        dh[k] = h0[k] - h1[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ol[k] = metric_max(l0[k], l1[k]);
        oh[k] = metric_max(h0[k], h1[k]);
    }
    accl = 0;
    acch = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        accl = __builtin_amdgcn_alignbit(accl, __float_as_uint(dl[k]), 31);
        acch = __builtin_amdgcn_alignbit(acch, __float_as_uint(dh[k]), 31);
    }
}

// Steps 2U and 2U + 1 of a pair of puncture periods for wave WI.  A holds states 16 WI .. 16 WI + 15 on entry and
// on return.  xbuf: this pair's exchange buffer.
template <int CR, int WI, int U>
__device__ __forceinline__ void split_pair(float (&A)[16], const float *c, uint32_t *dp, float *xbuf, int lane)
{
    constexpr int P = Rate<CR>::P;
    uint32_t *nibs = reinterpret_cast<uint32_t *>(xbuf + XCH_FLOATS);
    float nl[8], nh[8];  // after the first step: states 8 WI + k and 32 + 8 WI + k
    uint32_t e_l, e_h;
    {
        constexpr int S = 2 * U, ps = S % P, kind = step_kind(CR, ps), col = (S / P) * Rate<CR>::COLS + step_col(ps);
        const float t0 = kind == 2 ? 0.0f : c[col];
        const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
        const float t01 = t1 + t0;
        uint32_t a0, a1, b0, b1;
        bfly4<kind, 8 * WI, 16 * WI>(A, nl, nh, t0, t1, t01, a0, b0);
        bfly4<kind, 8 * WI + 4, 16 * WI>(A, nl + 4, nh + 4, t0, t1, t01, a1, b1);
        e_l = (a0 << 4) | a1;  // states 8 WI .. 8 WI + 7: byte WI of w0
        e_h = (b0 << 4) | b1;  // states 32 + 8 WI ..   : byte WI of w1
    }
    float r[4][4];  // after the second step: r[d][k] = state 16 d + 4 WI + k, held next by wave d
    uint32_t n0, n1, n2, n3;
    {
        constexpr int S = 2 * U + 1, ps = S % P, kind = step_kind(CR, ps), col = (S / P) * Rate<CR>::COLS + step_col(ps);
        const float t0 = kind == 2 ? 0.0f : c[col];
        const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
        const float t01 = t1 + t0;
        bfly4<kind, 4 * WI, 8 * WI>(nl, r[0], r[2], t0, t1, t01, n0, n2);              // j = 4 WI + k      -> j, j + 32
        bfly4<kind, 16 + 4 * WI, 32 + 8 * WI>(nh, r[1], r[3], t0, t1, t01, n1, n3);  // j = 16 + 4 WI + k -> j, j + 32
    }
    // hand over: wave d holds states 16 d .. 16 d + 15 next
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (d == WI) A[4 * WI + k] = r[d][k];
            else xbuf[(16 * d + 4 * WI + k) * 64 + lane] = r[d][k];
        }
    // second-step decisions are four nibbles per wave (states 16 d + 4 WI ..): waves 2k and 2k+1 complete each
    // other's bytes, the odd wave passes its nibbles to the even one
    const uint32_t mine = n0 | (n1 << 4) | (n2 << 8) | (n3 << 12);
    if constexpr ((WI & 1) != 0) nibs[(WI >> 1) * 64 + lane] = mine;
    // workgroup barrier that waits for this wave's LDS traffic only (the survivor stores still in flight concern
    // no other wave)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (q / 4 != WI) A[q] = xbuf[(16 * WI + q) * 64 + lane];
    uint8_t *rec = reinterpret_cast<uint8_t *>(dp + U * 256);
    rec[WI] = (uint8_t)e_l;
    rec[4 + WI] = (uint8_t)e_h;
    if constexpr ((WI & 1) == 0) {
        const uint32_t other = nibs[(WI >> 1) * 64 + lane];
        // bytes WI/2 and 2 + WI/2 of w0 (states 4 WI.., 16 + 4 WI..) and of w1 (32 + .., 48 + ..); the even wave's
        // states come first (top nibble)
        rec[8 + (WI >> 1)] = (uint8_t)(((mine & 0xf) << 4) | (other & 0xf));
        rec[8 + 2 + (WI >> 1)] = (uint8_t)((mine & 0xf0) | ((other >> 4) & 0xf));
        rec[12 + (WI >> 1)] = (uint8_t)(((mine >> 4) & 0xf0) | ((other >> 8) & 0xf));
        rec[12 + 2 + (WI >> 1)] = (uint8_t)(((mine >> 8) & 0xf0) | ((other >> 12) & 0xf));
    }
}

template <int CR, int WI, int... U>
__device__ __forceinline__ void split_periods(float (&A)[16], const float *c, uint32_t *dp, float *xch, int &flip, int lane,
                                              std::integer_sequence<int, U...>)
{
    ((split_pair<CR, WI, U>(A, c, dp, xch + (flip ? XCH_BUF : 0), lane), flip ^= 1), ...);
}

template <int CR, int WI>
__device__ __forceinline__ void viterbi_split(const float *__restrict__ llr, unsigned long long mybase, int myn, int nchunks,
                                              uint32_t *__restrict__ dec, float *tile, float *xch, int lane)
{
    constexpr int P = Rate<CR>::P, COLS = Rate<CR>::COLS, NPER = Rate<CR>::NPER;
    constexpr int CH = COLS * NPER, STEPS = P * NPER;
    float A[16];
#pragma unroll
    for (int i = 0; i < 16; i++) A[i] = (16 * WI + i) == 0 ? 0.0f : NEG_INIT;
    const float *row = tile + lane * TILE_STRIDE;
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float *lt = (lds_float *)tile;
    // soft bits the shortest frame of the group has (lanes without a frame do not count): tiles below that mark
    // need no clamping, and a row's address is a wave-uniform base plus the lane's column
    int least = myn > 0 ? myn : 0x7fffffff;
    for (int d = 32; d > 0; d >>= 1) least = min(least, __shfl_xor(least, d));
    least = __builtin_amdgcn_readfirstlane(least);
    const unsigned long long mine = (unsigned long long)(llr + mybase);  // this lane's row address in bytes
    int flip = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        __syncthreads();
        const int c0 = ch * CH + lane;
        // the 64 x 64 soft-bit tile is shared; each wave fetches every fourth row (LDS-DMA, one row per load)
        if (ch * CH + 64 <= least) {
            const unsigned long long step = (unsigned long long)(ch * CH) * sizeof(float);
#pragma unroll
            for (int r = WI; r < 64; r += 4) {
                const unsigned lo = __builtin_amdgcn_readlane((int)(mine & 0xffffffffull), r);
                const unsigned hi = __builtin_amdgcn_readlane((int)(mine >> 32), r);
                const float *rowp = (const float *)((((unsigned long long)hi << 32) | lo) + step);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp + lane),
                                                 (__attribute__((address_space(3))) void *)(lt + r * TILE_STRIDE), 4, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = WI; r < 64; r += 4) {
                const unsigned lo = __builtin_amdgcn_readlane((int)(mybase & 0xffffffffull), r);
                const unsigned hi = __builtin_amdgcn_readlane((int)(mybase >> 32), r);
                const int nr = __builtin_amdgcn_readlane(myn, r);
                const unsigned long long base = ((unsigned long long)hi << 32) | lo;
                int idx = c0 < nr ? c0 : nr - 1;
                idx = idx < 0 ? 0 : idx;
                const float *src = llr + base + (unsigned long long)idx;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(lt + r * TILE_STRIDE), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *dp = dec + (size_t)ch * STEPS * 128 + lane * 4;
#pragma unroll 1
        for (int pp = 0; pp < NPER; pp += 2)
            split_periods<CR, WI>(A, row + pp * COLS, dp + (size_t)pp * P * 128, xch, flip, lane,
                                  std::make_integer_sequence<int, P>{});
    }
}

template <int CR>
__device__ __forceinline__ void split_dispatch(int wi, const float *llr, unsigned long long mybase, int myn, int nchunks,
                                               uint32_t *dec, float *tile, float *xch, int lane)
{
    // every wave of the workgroup runs the same number of chunks, step pairs and barriers; only the states differ
    if (wi == 0) viterbi_split<CR, 0>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
    else if (wi == 1) viterbi_split<CR, 1>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
    else if (wi == 2) viterbi_split<CR, 2>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
    else viterbi_split<CR, 3>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
}

// ---- two waves per group: wave w keeps states [32w, 32w + 32).  The same two-steps-then-exchange scheme: the 32 states
// are the predecessors of the butterflies 16w .. 16w+15 of one step, whose results (states 16w.. and 32+16w..) are
// the predecessors of the butterflies 8w.. and 16+8w.. of the next; after that each wave holds eight states of every
// quarter and hands the two quarters of the other wave's half over through LDS (16 of its 32 metrics; the four-wave
// form moves 12 of 16).  Every decision byte belongs to one wave, so nothing but metrics is exchanged.
constexpr int XCH2_BUF = 2 * 16 * 64 + 2 * 64;  // one exchange buffer: [sending wave][16 metrics][lane], then wave 1's
                                                // decision bytes of the pair [2 words][lane]

template <int CR, int WI, int U>
__device__ __forceinline__ void split2_pair(float (&A)[32], const float *c, uint32_t *dp, float *xbuf, int lane)
{
    constexpr int P = Rate<CR>::P;
    float nl[16], nh[16];  // after the first step: states 16 WI + k and 32 + 16 WI + k
    uint32_t e_l, e_h;
    {
        constexpr int S = 2 * U, ps = S % P, kind = step_kind(CR, ps), col = (S / P) * Rate<CR>::COLS + step_col(ps);
        const float t0 = kind == 2 ? 0.0f : c[col];
        const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
        const float t01 = t1 + t0;
        uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
        bfly4<kind, 16 * WI, 32 * WI>(A, nl, nh, t0, t1, t01, a0, b0);
        bfly4<kind, 16 * WI + 4, 32 * WI>(A, nl + 4, nh + 4, t0, t1, t01, a1, b1);
        bfly4<kind, 16 * WI + 8, 32 * WI>(A, nl + 8, nh + 8, t0, t1, t01, a2, b2);
        bfly4<kind, 16 * WI + 12, 32 * WI>(A, nl + 12, nh + 12, t0, t1, t01, a3, b3);
        // states 16 WI .. 16 WI + 15: bytes 2 WI and 2 WI + 1 of w0; states 32 + 16 WI ..: the same bytes of w1
        e_l = ((a0 << 4) | a1) | (((a2 << 4) | a3) << 8);
        e_h = ((b0 << 4) | b1) | (((b2 << 4) | b3) << 8);
    }
    float ra[8], rb[8], rc[8], rd[8];  // after the second step: states 8 WI + k, 16 + 8 WI + k, 32 + 8 WI + k, 48 + 8 WI + k
    uint32_t o_a, o_b, o_c, o_d;
    {
        constexpr int S = 2 * U + 1, ps = S % P, kind = step_kind(CR, ps), col = (S / P) * Rate<CR>::COLS + step_col(ps);
        const float t0 = kind == 2 ? 0.0f : c[col];
        const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
        const float t01 = t1 + t0;
        uint32_t n0, n1, m0, m1, p0, p1, q0, q1;
        bfly4<kind, 8 * WI, 16 * WI>(nl, ra, rc, t0, t1, t01, n0, p0);                        // j = 8 WI + k       -> j, j + 32
        bfly4<kind, 8 * WI + 4, 16 * WI>(nl, ra + 4, rc + 4, t0, t1, t01, n1, p1);
        bfly4<kind, 16 + 8 * WI, 32 + 16 * WI>(nh, rb, rd, t0, t1, t01, m0, q0);            // j = 16 + 8 WI + k  -> j, j + 32
        bfly4<kind, 16 + 8 * WI + 4, 32 + 16 * WI>(nh, rb + 4, rd + 4, t0, t1, t01, m1, q1);
        o_a = (n0 << 4) | n1;  // states 8 WI ..      : byte WI of w0
        o_b = (m0 << 4) | m1;  // states 16 + 8 WI .. : byte 2 + WI of w0
        o_c = (p0 << 4) | p1;  // states 32 + 8 WI .. : byte WI of w1
        o_d = (q0 << 4) | q1;  // states 48 + 8 WI .. : byte 2 + WI of w1
    }
    // hand over: wave 0 holds states 0 .. 31 next, wave 1 states 32 .. 63
    float *mine = xbuf + WI * (16 * 64) + lane;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if constexpr (WI == 0) {
            A[k] = ra[k];
            A[16 + k] = rb[k];
            mine[k * 64] = rc[k];         // state 32 + k      -> wave 1's A[k]
            mine[(8 + k) * 64] = rd[k];   // state 48 + k      -> wave 1's A[16 + k]
        } else {
            A[8 + k] = rc[k];
            A[24 + k] = rd[k];
            mine[k * 64] = ra[k];         // state 8 + k       -> wave 0's A[8 + k]
            mine[(8 + k) * 64] = rb[k];   // state 24 + k      -> wave 0's A[24 + k]
        }
    }
    // Wave 1's decision bytes travel with the metrics, so that wave 0 stores the pair's whole 16-byte record at once
    // (narrow stores from both waves cost twelve store instructions per pair, each over sixteen cache lines).
    uint32_t *dbytes = reinterpret_cast<uint32_t *>(xbuf + 2 * 16 * 64) + lane;
    if constexpr (WI == 1) {
        dbytes[0] = e_l | (e_h << 16);
        dbytes[64] = o_a | (o_b << 8) | (o_c << 16) | (o_d << 24);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float *theirs = xbuf + (1 - WI) * (16 * 64) + lane;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if constexpr (WI == 0) {
            A[8 + k] = theirs[k * 64];
            A[24 + k] = theirs[(8 + k) * 64];
        } else {
            A[k] = theirs[k * 64];
            A[16 + k] = theirs[(8 + k) * 64];
        }
    }
    if constexpr (WI == 0) {
        const uint32_t x = dbytes[0], y = dbytes[64];
        const uint32_t z = o_a | (o_b << 8) | (o_c << 16) | (o_d << 24);
        uint4 rec;
        rec.x = e_l | (x << 16);                        // even step, states 0-31: bytes 0, 1 from wave 0, bytes 2, 3 from wave 1
        rec.y = e_h | (x & 0xffff0000u);                // even step, states 32-63
        rec.z = __builtin_amdgcn_perm(y, z, 0x05010400u);  // odd step, w0: bytes [z0 y0 z1 y1] (selector 0-3 = z, 4-7 = y)
        rec.w = __builtin_amdgcn_perm(y, z, 0x07030602u);  // odd step, w1: bytes [z2 y2 z3 y3]
        *reinterpret_cast<uint4 *>(dp + U * 256) = rec;
    }
}

template <int CR, int WI, int... U>
__device__ __forceinline__ void split2_periods(float (&A)[32], const float *c, uint32_t *dp, float *xch, int &flip, int lane,
                                               std::integer_sequence<int, U...>)
{
    ((split2_pair<CR, WI, U>(A, c, dp, xch + (flip ? XCH2_BUF : 0), lane), flip ^= 1), ...);
}

template <int CR, int WI>
__device__ __forceinline__ void viterbi_split2(const float *__restrict__ llr, unsigned long long mybase, int myn, int nchunks,
                                               uint32_t *__restrict__ dec, float *tile, float *xch, int lane)
{
    constexpr int P = Rate<CR>::P, COLS = Rate<CR>::COLS, NPER = Rate<CR>::NPER;
    constexpr int CH = COLS * NPER, STEPS = P * NPER;
    float A[32];
#pragma unroll
    for (int i = 0; i < 32; i++) A[i] = (32 * WI + i) == 0 ? 0.0f : NEG_INIT;
    const float *row = tile + lane * TILE_STRIDE;
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float *lt = (lds_float *)tile;
    int least = myn > 0 ? myn : 0x7fffffff;
    for (int d = 32; d > 0; d >>= 1) least = min(least, __shfl_xor(least, d));
    least = __builtin_amdgcn_readfirstlane(least);
    const unsigned long long mine = (unsigned long long)(llr + mybase);  // this lane's row address in bytes
    int flip = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        __syncthreads();
        const int c0 = ch * CH + lane;
        // the 64 x 64 soft-bit tile is shared; each wave fetches every second row (LDS-DMA, one row per load)
        if (ch * CH + 64 <= least) {
            const unsigned long long step = (unsigned long long)(ch * CH) * sizeof(float);
#pragma unroll
            for (int r = WI; r < 64; r += 2) {
                const unsigned lo = __builtin_amdgcn_readlane((int)(mine & 0xffffffffull), r);
                const unsigned hi = __builtin_amdgcn_readlane((int)(mine >> 32), r);
                const float *rowp = (const float *)((((unsigned long long)hi << 32) | lo) + step);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp + lane),
                                                 (__attribute__((address_space(3))) void *)(lt + r * TILE_STRIDE), 4, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = WI; r < 64; r += 2) {
                const unsigned lo = __builtin_amdgcn_readlane((int)(mybase & 0xffffffffull), r);
                const unsigned hi = __builtin_amdgcn_readlane((int)(mybase >> 32), r);
                const int nr = __builtin_amdgcn_readlane(myn, r);
                const unsigned long long base = ((unsigned long long)hi << 32) | lo;
                int idx = c0 < nr ? c0 : nr - 1;
                idx = idx < 0 ? 0 : idx;
                const float *src = llr + base + (unsigned long long)idx;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(lt + r * TILE_STRIDE), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *dp = dec + (size_t)ch * STEPS * 128 + lane * 4;
#pragma unroll 1
        for (int pp = 0; pp < NPER; pp += 2)
            split2_periods<CR, WI>(A, row + pp * COLS, dp + (size_t)pp * P * 128, xch, flip, lane,
                                   std::make_integer_sequence<int, P>{});
    }
}

template <int CR>
__device__ __forceinline__ void split2_dispatch(int wi, const float *llr, unsigned long long mybase, int myn, int nchunks,
                                                uint32_t *dec, float *tile, float *xch, int lane)
{
    if (wi == 0) viterbi_split2<CR, 0>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
    else viterbi_split2<CR, 1>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
}

}  // namespace gr80211
