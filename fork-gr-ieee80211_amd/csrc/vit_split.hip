// Viterbi forward pass for small batches: W wavefronts (W = 2 or 4, one workgroup) share one group of 64 frames.
// Measured on MI355X (VHT 1500-byte frames): 157 groups 7.6 -> 4.0 ms and 500 groups 8.1 -> 5.4 ms with W = 4
// (W = 2: 5.6 and 7.2 ms); from about 600 groups on one wave per group is as fast and keeps the fused traceback.
// A variant with four lanes per frame and the exchange as DPP operands of the adds was no faster (4.2 ms at 157
// groups: DPP-modified adds issue at about a quarter of the plain rate) and is not kept.
//
// With lane = frame a group's trellis is one wave's serial work; a batch of fewer groups than the chip has SIMDs
// leaves most of it idle for the whole length of the longest trellis.  Here the 64 states of a group are divided
// among W waves (each on its own SIMD of one CU): wave w keeps the path metrics of states [64w/W, 64(w+1)/W),
// which are exactly the predecessors of the butterflies j in [32w/W, 32(w+1)/W) it computes (new states j and
// j + 32).  After every trellis step the new metrics a wave does not need itself go to the wave that does, through
// LDS (double-buffered, one workgroup barrier per step).  Same float operations in the same order as the
// one-wave kernel, same survivor layout ([step pair][lane][w0e w1e w0o w1o], state n at bit n ^ 7): each wave
// stores its own bytes of the four words, and the traceback kernel reads them unchanged.
#include <utility>

#include "acs_common.h"
#include "rx_kernels.h"

namespace gr80211 {

namespace {

constexpr int XCH_FLOATS = 64 * 64;  // one exchange buffer: [state][lane]

template <int W, int WI> struct Geo {
    static constexpr int S = 64 / W;   // states whose metrics this wave holds
    static constexpr int H = S / 2;    // butterflies it computes
    static constexpr int O0 = S * WI;  // first held state
    static constexpr int J0 = H * WI;  // first butterfly
    static constexpr bool holds(int s) { return s >= O0 && s < O0 + S; }
    static constexpr bool makes(int s) { return (s & 31) >= J0 && (s & 31) < J0 + H; }
};

// Four butterflies JB .. JB+3: old states 2j, 2j+1 -> new states j (input bit 0) and j + 32 (input bit 1).
// Written phase by phase like the one-wave kernel's state quads; decisions (sign of a0 - a1, set when the odd
// predecessor wins) are shifted into one accumulator per half, first state in bit 3.
template <int KIND, int W, int WI, int JB>
__device__ __forceinline__ void bfly4(const float (&X)[64 / W], float (&nl)[32 / W], float (&nh)[32 / W], float t0, float t1,
                                      float t01, uint32_t &accl, uint32_t &acch)
{
    using G = Geo<W, WI>;
    float l0[4], l1[4], h0[4], h1[4], dl[4], dh[4];
#define GR_BF_IN(K)                                                          \
    {                                                                        \
        constexpr int j = JB + K, p0 = 2 * j, p1 = 2 * j + 1;                \
        l0[K] = add_tab<KIND, sv_out(p0, 0)>(X[p0 - G::O0], t0, t1, t01);    \
        l1[K] = add_tab<KIND, sv_out(p1, 0)>(X[p1 - G::O0], t0, t1, t01);    \
        h0[K] = add_tab<KIND, sv_out(p0, 1)>(X[p0 - G::O0], t0, t1, t01);    \
        h1[K] = add_tab<KIND, sv_out(p1, 1)>(X[p1 - G::O0], t0, t1, t01);    \
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
        nl[JB - G::J0 + k] = fmaxf(l0[k], l1[k]);
        nh[JB - G::J0 + k] = fmaxf(h0[k], h1[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        accl = __builtin_amdgcn_alignbit(accl, __float_as_uint(dl[k]), 31);
        acch = __builtin_amdgcn_alignbit(acch, __float_as_uint(dh[k]), 31);
    }
}

template <int KIND, int W, int WI, int... Q>
__device__ __forceinline__ void bfly_all(const float (&X)[64 / W], float (&nl)[32 / W], float (&nh)[32 / W], float t0, float t1,
                                         float t01, uint32_t (&al)[8 / W], uint32_t (&ah)[8 / W],
                                         std::integer_sequence<int, Q...>)
{
    (bfly4<KIND, W, WI, Geo<W, WI>::J0 + 4 * Q>(X, nl, nh, t0, t1, t01, al[Q], ah[Q]), ...);
}

template <int W, int WI, int... K>
__device__ __forceinline__ void hand_over(const float (&nl)[32 / W], const float (&nh)[32 / W], float (&Y)[64 / W], float *xbuf,
                                          int lane, std::integer_sequence<int, K...>)
{
    using G = Geo<W, WI>;
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value, sl = G::J0 + k, sh = G::J0 + k + 32;
        if constexpr (G::holds(sl)) Y[sl - G::O0] = nl[k];
        else xbuf[sl * 64 + lane] = nl[k];
        if constexpr (G::holds(sh)) Y[sh - G::O0] = nh[k];
        else xbuf[sh * 64 + lane] = nh[k];
    };
    (one(std::integral_constant<int, K>{}), ...);
}

template <int W, int WI, int... R>
__device__ __forceinline__ void fetch(float (&Y)[64 / W], const float *xbuf, int lane, std::integer_sequence<int, R...>)
{
    using G = Geo<W, WI>;
    auto one = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (!G::makes(G::O0 + r)) Y[r] = xbuf[(G::O0 + r) * 64 + lane];
    };
    (one(std::integral_constant<int, R>{}), ...);
}

// One trellis step of this wave's part: X (held old metrics) -> Y (held metrics of the next step).  xbuf is this
// step's exchange buffer.  wl / wh receive the decision bits of the wave's low (j) and high (j + 32) new states,
// H bits each, in the order of the survivor words (first state of each byte in its top bit).
template <int KIND, int W, int WI>
__device__ __forceinline__ void split_step(const float (&X)[64 / W], float (&Y)[64 / W], float t0, float t1, float *xbuf,
                                           int lane, uint32_t &wl, uint32_t &wh)
{
    using G = Geo<W, WI>;
    constexpr int NQ = G::H / 4;  // groups of four butterflies
    const float t01 = t1 + t0;
    float nl[G::H], nh[G::H];
    uint32_t al[NQ], ah[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) al[q] = ah[q] = 0;
    bfly_all<KIND, W, WI>(X, nl, nh, t0, t1, t01, al, ah, std::make_integer_sequence<int, NQ>{});
    // nibbles -> bytes -> the wave's bits of w0 / w1
    wl = 0;
    wh = 0;
#pragma unroll
    for (int q = 0; q < NQ; q += 2) {
        wl |= ((al[q] << 4) | al[q + 1]) << (4 * q);
        wh |= ((ah[q] << 4) | ah[q + 1]) << (4 * q);
    }
    // hand over what other waves hold next, keep the rest
    hand_over<W, WI>(nl, nh, Y, xbuf, lane, std::make_integer_sequence<int, G::H>{});
    __syncthreads();
    fetch<W, WI>(Y, xbuf, lane, std::make_integer_sequence<int, G::S>{});
}

// Steps 2U and 2U + 1 of a pair of puncture periods; the wave's bytes of the pair's four survivor words leave as
// four short (W = 2) or byte (W = 4) stores per lane.
template <int CR, int W, int WI, int U>
__device__ __forceinline__ void split_pair(float (&A)[64 / W], float (&B)[64 / W], const float *c, uint32_t *dp, float *xch,
                                           int lane)
{
    using G = Geo<W, WI>;
    constexpr int P = Rate<CR>::P;
    uint32_t le, he, lo, ho;
    {
        constexpr int S = 2 * U, ps = S % P, kind = step_kind(CR, ps), col = (S / P) * Rate<CR>::COLS + step_col(ps);
        const float t0 = kind == 2 ? 0.0f : c[col];
        const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
        split_step<kind, W, WI>(A, B, t0, t1, xch, lane, le, he);
    }
    {
        constexpr int S = 2 * U + 1, ps = S % P, kind = step_kind(CR, ps), col = (S / P) * Rate<CR>::COLS + step_col(ps);
        const float t0 = kind == 2 ? 0.0f : c[col];
        const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
        split_step<kind, W, WI>(B, A, t0, t1, xch + XCH_FLOATS, lane, lo, ho);
    }
    // the wave's butterflies J0 .. J0+H-1 are states J0.. of w0 and J0+32.. of w1: bytes J0/8 .. of each word
    uint8_t *rec = reinterpret_cast<uint8_t *>(dp + U * 256) + G::J0 / 8;
    if constexpr (W == 2) {
        *reinterpret_cast<uint16_t *>(rec) = (uint16_t)le;
        *reinterpret_cast<uint16_t *>(rec + 4) = (uint16_t)he;
        *reinterpret_cast<uint16_t *>(rec + 8) = (uint16_t)lo;
        *reinterpret_cast<uint16_t *>(rec + 12) = (uint16_t)ho;
    } else {
        rec[0] = (uint8_t)le;
        rec[4] = (uint8_t)he;
        rec[8] = (uint8_t)lo;
        rec[12] = (uint8_t)ho;
    }
}

template <int CR, int W, int WI, int... U>
__device__ __forceinline__ void split_periods(float (&A)[64 / W], float (&B)[64 / W], const float *c, uint32_t *dp, float *xch,
                                              int lane, std::integer_sequence<int, U...>)
{
    (split_pair<CR, W, WI, U>(A, B, c, dp, xch, lane), ...);
}

template <int CR, int W, int WI>
__device__ __forceinline__ void viterbi_split(const float *__restrict__ llr, unsigned long long mybase, int myn, int nchunks,
                                              uint32_t *__restrict__ dec, float *tile, float *xch, int lane)
{
    using G = Geo<W, WI>;
    constexpr int P = Rate<CR>::P, COLS = Rate<CR>::COLS, NPER = Rate<CR>::NPER;
    constexpr int CH = COLS * NPER, STEPS = P * NPER;
    float A[G::S], B[G::S];
#pragma unroll
    for (int i = 0; i < G::S; i++) A[i] = (G::O0 + i) == 0 ? 0.0f : NEG_INIT;
    const float *row = tile + lane * TILE_STRIDE;
    for (int ch = 0; ch < nchunks; ch++) {
        __syncthreads();
        const int c0 = ch * CH + lane;
        // the 64 x 64 soft-bit tile is shared; each wave fetches every W-th row (LDS-DMA, one row per load)
#pragma unroll
        for (int r = WI; r < 64; r += W) {
            const unsigned lo = __builtin_amdgcn_readlane((int)(mybase & 0xffffffffull), r);
            const unsigned hi = __builtin_amdgcn_readlane((int)(mybase >> 32), r);
            const int nr = __builtin_amdgcn_readlane(myn, r);
            const unsigned long long base = ((unsigned long long)hi << 32) | lo;
            int idx = c0 < nr ? c0 : nr - 1;
            idx = idx < 0 ? 0 : idx;
            const float *src = llr + base + (unsigned long long)idx;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(tile + r * TILE_STRIDE), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *dp = dec + (size_t)ch * STEPS * 128 + lane * 4;
#pragma unroll 1
        for (int pp = 0; pp < NPER; pp += 2)
            split_periods<CR, W, WI>(A, B, row + pp * COLS, dp + (size_t)pp * P * 128, xch, lane,
                                     std::make_integer_sequence<int, P>{});
    }
}

template <int CR, int W>
__device__ __forceinline__ void split_dispatch(int wi, const float *llr, unsigned long long mybase, int myn, int nchunks,
                                               uint32_t *dec, float *tile, float *xch, int lane)
{
    // every wave of the workgroup runs the same number of chunks, steps and barriers; only the states differ
    if constexpr (W == 2) {
        if (wi == 0) viterbi_split<CR, 2, 0>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
        else viterbi_split<CR, 2, 1>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
    } else {
        if (wi == 0) viterbi_split<CR, 4, 0>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
        else if (wi == 1) viterbi_split<CR, 4, 1>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
        else if (wi == 2) viterbi_split<CR, 4, 2>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
        else viterbi_split<CR, 4, 3>(llr, mybase, myn, nchunks, dec, tile, xch, lane);
    }
}

template <int W>
__global__ __launch_bounds__(64 * W) void k_viterbi_split(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                                          const int32_t *__restrict__ slot_frame,
                                                          const GroupInfo *__restrict__ groups, int n_groups,
                                                          uint32_t *__restrict__ dec)
{
    const int g = blockIdx.x;
    if (g >= n_groups) return;  // whole workgroup
    const int lane = threadIdx.x & 63;
    const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float tile[64 * TILE_STRIDE];
    __shared__ float xch[2 * XCH_FLOATS];
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    const int myn = fr >= 0 ? info[fr].total : 0;
    const unsigned long long mybase = myn > 0 ? (unsigned long long)info[fr].llr_off : 0ull;
    uint32_t *d = dec + gi.dec_off;
    switch (gi.cr) {
    case CR_12: split_dispatch<CR_12, W>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_23: split_dispatch<CR_23, W>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_34: split_dispatch<CR_34, W>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    default: split_dispatch<CR_56, W>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    }
}

}  // namespace

void launch_viterbi_split(int waves, const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                          const GroupInfo *groups, int n_groups, uint32_t *dec, hipStream_t s)
{
    if (waves == 2)
        hipLaunchKernelGGL(k_viterbi_split<2>, dim3(n_groups), dim3(128), 0, s, llr, info, slot_frame, groups, n_groups, dec);
    else
        hipLaunchKernelGGL(k_viterbi_split<4>, dim3(n_groups), dim3(256), 0, s, llr, info, slot_frame, groups, n_groups, dec);
}

}  // namespace gr80211
