// Decoder for small batches: one frame per wavefront, one trellis state per lane.
//
// With lane = frame (k_decode and its two- and four-wave forms) a frame's trellis is serial work of a wave that
// carries 63 other frames with it: a batch of a few frames, or a few thousand, leaves the chip idle for the length of
// one trellis whatever its size.  Here a wave decodes one frame; a batch of n frames is n waves, and its latency is
// one trellis at a lone wave's issue rate.
//
// Forward pass (decode_impl::vstb_update, lib/decode_impl.cc:205-281).  The trellis of the code is a shuffle-exchange
// graph: states 2j and 2j+1 lead to states j and j+32.  Keeping state s of time t in lane rol6^t(s) (its six index
// bits rotated left t times) puts the two results of a butterfly into the two lanes that held its inputs: at step t
// every lane pairs with the lane whose index differs in bit t mod 6, and nothing else moves.  The pair's metrics
// reach both lanes by DPP moves (distances 1, 2, 4, 8) or by the row and half swaps of gfx950 (16, 32): vector
// instructions only, no LDS round trip in the dependent chain.  The soft bits of a frame are the same for every
// lane, so they arrive by scalar loads and cost no vector instruction; a lane's branch metric is a0*t0 + a1*t1 with
// its own constants a0, a1 in {0, 1} per phase (one multiply and one fused multiply-add whose product is exact, so
// the sum is rounded once, like the reference's t1 + t0).  The decision of a step is the compare mask of the whole
// wave (64 states = 64 bits); it is parked in lane (t mod 60) of a register pair and leaves every 60 steps as one
// contiguous 480-byte store: survivor word t of the frame at 8*t bytes, bit l = lane l's decision.
//
// Walk back (decode_impl::vstb_end, lib/decode_impl.cc:282-302).  In lane coordinates one step back replaces bit
// t mod 6 of the lane index by the decision bit, and the bit it replaces is decoded bit t.  The walk is cut into up
// to 64 segments, one per lane; every lane but the topmost starts 96 steps above its segment in an arbitrary state,
// and the state it arrives in at the top of its segment is compared with the state the lane above ends in.  Survivor
// paths merge within a few constraint lengths, so this nearly always holds for every lane at once; where it does
// not, the pass is repeated with every lane starting from its upper neighbour's end state until nothing changes
// (the topmost lane is exact from the start, so each pass fixes at least one more segment).  The result is the
// reference's one walk from state 0 at the last step, bit for bit, whatever the soft bits were.
//
// Then the wave descrambles, assembles and checks the packet: what assemble.h does with one lane per frame, with the
// frame's words, the CRC-32 of an MPDU and the copy into the PDU slot spread over the 64 lanes.
#include <cstdlib>
#include <utility>

#include "acs_common.h"
#include "rx_kernels.h"
#include "assemble.h"

namespace gr80211 {

namespace {

constexpr int LS_BLOCK = 60;   // steps per survivor block: a multiple of 6 (phases) and of every puncture period
constexpr int LS_WAVES = 4;    // frames per workgroup, one per wave (the four SIMDs of a CU); the waves share nothing
constexpr int LS_WARM_AHEAD = 1024;  // soft bits between the scalar loads and the lines being brought into L2
constexpr int TB_WARM = 96;    // steps a lane walks above its segment before its bits count
constexpr int TB_UNIT = 96;    // segment lengths are multiples of lcm(6 phases, 32 bits per word)

typedef __attribute__((address_space(4))) const float *cfloatp;  // constant address space: uniform loads are scalar loads

template <int CR> struct LsRate;
template <> struct LsRate<CR_12> { static constexpr int SUB = 12; };
template <> struct LsRate<CR_23> { static constexpr int SUB = 12; };
template <> struct LsRate<CR_34> { static constexpr int SUB = 12; };
template <> struct LsRate<CR_56> { static constexpr int SUB = 10; };

__device__ __forceinline__ int ls_clamp(int c, int n) { c = c < n ? c : n - 1; return c < 0 ? 0 : c; }
// A load whose only purpose is to bring its cache line into L2.  As inline assembly it stays where it is written (the
// compiler moves an ordinary load down to its first use); its register is held until ls_touched has waited for it.
__device__ __forceinline__ float ls_touch(const float *p)
{
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void ls_touched(float v) { asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory"); }
__device__ __forceinline__ int ror6(int v, int k) { return ((v >> k) | (v << (6 - k))) & 63; }

// Per-lane constants of the six phases: coefficient of the first and second soft bit in the branch metric along
// the even (E) and odd (O) predecessor.
struct LsCoef {
    float e0[6], e1[6], o0[6], o1[6];
};

__device__ __forceinline__ void ls_coefficients(int lane, LsCoef &c)
{
#pragma unroll
    for (int ph = 0; ph < 6; ph++) {
        const int p0 = ror6(lane & ~(1 << ph), ph);  // the even state of this lane's pair at phase ph
        const int b = (lane >> ph) & 1;              // the input bit that leads to this lane's new state
        const int oe = sv_out(p0, b), oo = sv_out(p0 + 1, b);
        c.e0[ph] = (float)(oe >> 1);
        c.e1[ph] = (float)(oe & 1);
        c.o0[ph] = (float)(oo >> 1);
        c.o1[ph] = (float)(oo & 1);
    }
}

// v_writelane_b32: `value` (wave-uniform) into lane `lane` of `old`.  This compiler has no builtin of that name; the
// intrinsic is reached by its assembler label.
extern "C" __device__ int ls_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_mov(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, BANKS, false));
}

// A permutation that writes every lane: no previous value to keep, so source and destination may differ.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float src)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), CTRL, 0xF, 0xF, true));
}

// The metrics of the pair this lane belongs to at phase PH: x from the lane whose bit PH is clear (the even state),
// y from the lane whose bit PH is set (the odd state).
template <int PH>
__device__ __forceinline__ void pair_metrics(float m, float &x, float &y)
{
    if constexpr (PH == 0) {
        x = dpp_perm<0xA0>(m);  // quad_perm [0,0,2,2]
        y = dpp_perm<0xF5>(m);  // quad_perm [1,1,3,3]
    } else if constexpr (PH == 1) {
        x = dpp_perm<0x44>(m);  // quad_perm [0,1,0,1]
        y = dpp_perm<0xEE>(m);  // quad_perm [2,3,2,3]
    } else if constexpr (PH == 2) {
        x = dpp_mov<0x114, 0xA>(m, m);  // row_shr:4 into lanes 4-7, 12-15 of a row, the others keep their own
        y = dpp_mov<0x104, 0x5>(m, m);  // row_shl:4 into lanes 0-3, 8-11
    } else if constexpr (PH == 3) {
        x = dpp_mov<0x128, 0xC>(m, m);  // row_ror:8 into lanes 8-15
        y = dpp_mov<0x128, 0x3>(m, m);  // row_ror:8 into lanes 0-7
    } else if constexpr (PH == 4) {
        // odd rows of the first operand change places with even rows of the second: [r0 r0 r2 r2], [r1 r1 r3 r3]
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        x = __uint_as_float(r[0]);
        y = __uint_as_float(r[1]);
    } else {
        // upper half of the first operand changes places with the lower half of the second
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        x = __uint_as_float(r[0]);
        y = __uint_as_float(r[1]);
    }
}

// One trellis step.  KIND 0: both soft bits present; 1: the second erased; 2: the first erased (t1 is the one
// present).  The decision mask is parked in lane IDX of (plo, phi).
template <int KIND, int PH, int IDX>
__device__ __forceinline__ void ls_step(float &m, float t0, float t1, const LsCoef &c, uint32_t &plo, uint32_t &phi)
{
    float x, y;
    pair_metrics<PH>(m, x, y);
    float e, o;
    if constexpr (KIND == 0) {
        e = x + __builtin_fmaf(c.e0[PH], t0, c.e1[PH] * t1);
        o = y + __builtin_fmaf(c.o0[PH], t0, c.o1[PH] * t1);
    } else if constexpr (KIND == 1) {
        e = __builtin_fmaf(c.e0[PH], t0, x);
        o = __builtin_fmaf(c.o0[PH], t0, y);
    } else {
        e = __builtin_fmaf(c.e1[PH], t1, x);
        o = __builtin_fmaf(c.o1[PH], t1, y);
    }
    // the odd predecessor wins only when strictly larger (lib/decode_impl.cc:248-258)
    const unsigned long long won = __builtin_amdgcn_ballot_w64(o > e);
    m = __builtin_fmaxf(e, o);  // both are sums: nothing to quiet, one v_max_f32 that does not wait for the compare
    // through the compiler's own instruction (not inline assembly) so that it keeps the wait states this target needs
    // between a vector compare's mask and a vector instruction that reads it
    plo = (uint32_t)ls_writelane((int)(uint32_t)won, IDX, (int)plo);
    phi = (uint32_t)ls_writelane((int)(uint32_t)(won >> 32), IDX, (int)phi);
}

// Step J of sub-body SB of a block, soft bits of the sub-body in c[].
template <int CR, int SB, int J>
__device__ __forceinline__ void ls_sub_step(float &m, const float *c, const LsCoef &k, uint32_t &plo, uint32_t &phi)
{
    constexpr int P = Rate<CR>::P, SUB = LsRate<CR>::SUB, I = SB * SUB + J, ps = J % P, kind = step_kind(CR, ps);
    constexpr int col = (J / P) * Rate<CR>::COLS + step_col(ps);
    const float t0 = kind == 2 ? 0.0f : c[col];
    const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
    ls_step<kind, I % 6, I>(m, t0, t1, k, plo, phi);
}

template <int CR, int SB, int... J>
__device__ __forceinline__ void ls_sub_body(float &m, const float *c, const LsCoef &k, uint32_t &plo, uint32_t &phi,
                                            std::integer_sequence<int, J...>)
{
    (ls_sub_step<CR, SB, J>(m, c, k, plo, phi), ...);
}

// NC soft bits from column col0 of a frame of n soft bits; columns past the end repeat the last one (they only feed
// steps past the frame's last).
template <int NC, bool INSIDE>
__device__ __forceinline__ void ls_load(float (&dst)[NC], cfloatp base, int col0, int n)
{
    if constexpr (INSIDE) {
        // the caller knows that the columns lie inside the frame: straight-line scalar loads the scheduler can issue
        // long before their first use
#pragma unroll
        for (int j = 0; j < NC; j++) dst[j] = base[col0 + j];
    } else {
#pragma unroll
        for (int j = 0; j < NC; j++) {
            int c = col0 + j;
            c = c < n ? c : n - 1;
            dst[j] = base[c < 0 ? 0 : c];
        }
    }
}

// Sub-body SB of a block: the next sub-body's soft bits are on their way while this one runs.
template <int CR, bool INSIDE, int SB>
__device__ __forceinline__ void ls_sub(float &m, float (&cur)[LsRate<CR>::SUB / Rate<CR>::P * Rate<CR>::COLS], cfloatp base, int &col,
                                       int n, const LsCoef &k, uint32_t &plo, uint32_t &phi)
{
    constexpr int SUB = LsRate<CR>::SUB, NC = SUB / Rate<CR>::P * Rate<CR>::COLS;
    float nxt[NC];
    ls_load<NC, INSIDE>(nxt, base, col + NC, n);
    ls_sub_body<CR, SB>(m, cur, k, plo, phi, std::make_integer_sequence<int, SUB>{});
#pragma unroll
    for (int j = 0; j < NC; j++) cur[j] = nxt[j];
    col += NC;
}

template <int CR, bool INSIDE, int... SB>
__device__ __forceinline__ void ls_block(float &m, float (&cur)[LsRate<CR>::SUB / Rate<CR>::P * Rate<CR>::COLS], cfloatp base,
                                         int &col, int n, const LsCoef &k, uint32_t &plo, uint32_t &phi,
                                         std::integer_sequence<int, SB...>)
{
    (ls_sub<CR, INSIDE, SB>(m, cur, base, col, n, k, plo, phi), ...);
}

// Forward pass of one frame: T steps, survivor words to w64[0 .. T).
template <int CR>
__device__ __forceinline__ void ls_forward(cfloatp base, int n, int T, unsigned long long *__restrict__ w64, int lane)
{
    constexpr int SUB = LsRate<CR>::SUB, NC = SUB / Rate<CR>::P * Rate<CR>::COLS;
    LsCoef k;
    ls_coefficients(lane, k);
    float m = lane == 0 ? 0.0f : NEG_INIT;
    float cur[NC];
    int col = 0;
    ls_load<NC, false>(cur, base, 0, n);
    uint32_t plo = 0, phi = 0;
    const int nblk = (T + LS_BLOCK - 1) / LS_BLOCK;
    // The soft bits were written by another kernel, in general on another XCD, so they are in memory and not in this
    // XCD's L2; a scalar load that has to go there returns after more steps than one sub-body lasts.  A vector load
    // per block, whose value nobody needs, touches the cache lines LS_WARM_AHEAD soft bits ahead of the scalar loads
    // (a block reads at most 120 soft bits, the eight lanes' addresses span 128), so that these find them in L2.
    const float *gl = (const float *)(unsigned long long)base;
    const float warm0 = ls_touch(gl + ls_clamp(lane * 16, n));
    // a block reads its own soft bits and, ahead of time, the first sub-body of the next block: blocks for which all
    // of that lies inside the frame run without index clamping
    constexpr int BLOCK_COLS = LS_BLOCK / Rate<CR>::P * Rate<CR>::COLS;
    int ninside = (n - NC) / BLOCK_COLS;
    ninside = ninside < 0 ? 0 : ninside > nblk ? nblk : ninside;
    int b = 0;
#pragma unroll 1
    for (; b < ninside; b++) {
        const float warm = ls_touch(gl + ls_clamp(col + LS_WARM_AHEAD + (lane & 7) * 16, n));
        ls_block<CR, true>(m, cur, base, col, n, k, plo, phi, std::make_integer_sequence<int, LS_BLOCK / SUB>{});
        ls_touched(warm);
        const int t = b * LS_BLOCK + lane;
        if (lane < LS_BLOCK && t < T) w64[t] = ((unsigned long long)phi << 32) | plo;
    }
#pragma unroll 1
    for (; b < nblk; b++) {
        const float warm = ls_touch(gl + ls_clamp(col + LS_WARM_AHEAD + (lane & 7) * 16, n));
        ls_block<CR, false>(m, cur, base, col, n, k, plo, phi, std::make_integer_sequence<int, LS_BLOCK / SUB>{});
        ls_touched(warm);
        const int t = b * LS_BLOCK + lane;
        if (lane < LS_BLOCK && t < T) w64[t] = ((unsigned long long)phi << 32) | plo;
    }
    ls_touched(warm0);
}

// ---- the same forward pass with the soft bits travelling through LDS (batches of about 3000 frames and more) -------
// Scalar loads can only be waited for all at once, so the look-ahead above is one sub-body and no more, and the scalar
// registers are full.  Here the wave fetches a block's soft bits with two vector loads two blocks ahead of their use
// (one soft bit per lane and load), parks them in a ring of four blocks in LDS when the block in between is done, and
// every sub-body reads its soft bits from there (all lanes the same address: a broadcast); the loads are counted, so
// nothing of this waits.  Columns past the frame's end repeat its last soft bit, as above.  A few per cent faster with
// several waves per SIMD, slower for a lone wave (the LDS reads cost it more issue slots than the waits they remove):
// the launcher takes it from LS_RING_FROM_GROUPS groups on.
constexpr int LS_RING_BLOCKS = 4;
constexpr int LS_RING_FROM_GROUPS = 48;  // batches of about 3000 frames and more take this form
constexpr int LS_RING_MAX = LS_RING_BLOCKS * LS_BLOCK * 2;  // rate 1/2 reads two soft bits per step

template <int NC>
__device__ __forceinline__ void lsr_load(float (&dst)[NC], const float *ring, int q)
{
#pragma unroll
    for (int j = 0; j < NC; j++) dst[j] = ring[q + j];
}

template <int CR, int SB>
__device__ __forceinline__ void lsr_sub(float &m, float (&cur)[LsRate<CR>::SUB / Rate<CR>::P * Rate<CR>::COLS], const float *ring, int &rpos,
                                        const LsCoef &k, uint32_t &plo, uint32_t &phi)
{
    constexpr int SUB = LsRate<CR>::SUB, NC = SUB / Rate<CR>::P * Rate<CR>::COLS;
    constexpr int RING = LS_RING_BLOCKS * (LS_BLOCK / Rate<CR>::P * Rate<CR>::COLS);
    float nxt[NC];
    int q = rpos + NC;  // the ring is a whole number of sub-bodies long: a sub-body never wraps
    q = q >= RING ? q - RING : q;
    lsr_load<NC>(nxt, ring, q);
    __builtin_amdgcn_sched_barrier(0);  // the reads set out here, a sub-body ahead, and not where the scheduler would sink them to
    ls_sub_body<CR, SB>(m, cur, k, plo, phi, std::make_integer_sequence<int, SUB>{});
#pragma unroll
    for (int j = 0; j < NC; j++) cur[j] = nxt[j];
    rpos = q;
}

template <int CR, int... SB>
__device__ __forceinline__ void lsr_block(float &m, float (&cur)[LsRate<CR>::SUB / Rate<CR>::P * Rate<CR>::COLS], const float *ring, int &rpos,
                                          const LsCoef &k, uint32_t &plo, uint32_t &phi, std::integer_sequence<int, SB...>)
{
    (lsr_sub<CR, SB>(m, cur, ring, rpos, k, plo, phi), ...);
}

template <int CR>
__device__ __forceinline__ void lsr_forward(const float *__restrict__ gl, int n, int T, unsigned long long *__restrict__ w64, int lane,
                                            float *ring)
{
    constexpr int SUB = LsRate<CR>::SUB, NC = SUB / Rate<CR>::P * Rate<CR>::COLS;
    constexpr int BLOCK_COLS = LS_BLOCK / Rate<CR>::P * Rate<CR>::COLS;  // 72 ... 120: two loads of 64 lanes
    static_assert(BLOCK_COLS > 64 && BLOCK_COLS <= 128 && BLOCK_COLS % NC == 0 && LS_RING_BLOCKS * BLOCK_COLS <= LS_RING_MAX, "ring layout");
    LsCoef k;
    ls_coefficients(lane, k);
    float m = lane == 0 ? 0.0f : NEG_INIT;
    uint32_t plo = 0, phi = 0;
    const int nblk = (T + LS_BLOCK - 1) / LS_BLOCK;
    const bool second = lane + 64 < BLOCK_COLS;  // lanes that carry a soft bit of the block's second load
    // blocks 0 and 1 before the first step
    {
        const float a0 = gl[ls_clamp(lane, n)], a1 = gl[ls_clamp(64 + lane, n)];
        const float b0 = gl[ls_clamp(BLOCK_COLS + lane, n)], b1 = gl[ls_clamp(BLOCK_COLS + 64 + lane, n)];
        ring[lane] = a0;
        ring[BLOCK_COLS + lane] = b0;
        if (second) {
            ring[64 + lane] = a1;
            ring[BLOCK_COLS + 64 + lane] = b1;
        }
    }
    float cur[NC];
    int rpos = 0;
    lsr_load<NC>(cur, ring, 0);
#pragma unroll 1
    for (int b = 0; b < nblk; b++) {
        // block b + 2 sets out now and is parked when this block is done; its place in the ring is neither this block's
        // nor the next one's, whose first sub-body this block reads ahead
        const int c2 = (b + 2) * BLOCK_COLS;
        const float a0 = gl[ls_clamp(c2 + lane, n)], a1 = gl[ls_clamp(c2 + 64 + lane, n)];
        lsr_block<CR>(m, cur, ring, rpos, k, plo, phi, std::make_integer_sequence<int, LS_BLOCK / SUB>{});
        const int slot = ((b + 2) % LS_RING_BLOCKS) * BLOCK_COLS;
        ring[slot + lane] = a0;
        if (second) ring[slot + 64 + lane] = a1;
        const int t = b * LS_BLOCK + lane;
        if (lane < LS_BLOCK && t < T) w64[t] = ((unsigned long long)phi << 32) | plo;
    }
}

// One pass of the walk back.  Lane k walks warm + seg steps down from time (k + 1) * seg + warm, starting in lane
// index l (steps at or past T are skipped, so a lane that starts above the trellis in state 0 arrives at T in state
// 0); its decoded bits count from time (k + 1) * seg on and leave as whole words.  seen = the lane index it holds at
// time (k + 1) * seg, the return value the one at time k * seg.  seg and warm are multiples of 96 (6 phases, 32 bits).
__device__ __forceinline__ uint32_t ls_walk(const unsigned long long *__restrict__ w64, int T, int wmax, int seg, int warm, int k,
                                           uint32_t l, uint32_t *__restrict__ bits, int nwords, uint32_t &seen)
{
    const int top = (k + 1) * seg;
    seen = l;
    uint32_t word = 0;
#pragma unroll 1
    for (int r0 = 0; r0 < warm + seg; r0 += 96) {
        if (r0 == warm) seen = l;
        const bool emit = r0 >= warm;
        const int tb = top + warm - 1 - r0;  // the batch's first step: 95 mod 96, so 5 mod 6, 31 mod 32 and odd
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
            uint4 v[12];  // 24 steps as 12 pairs (even step, odd step)
#pragma unroll
            for (int q = 0; q < 12; q++) {
                int te = tb - 24 * sb - 1 - 2 * q;
                te = te > wmax ? wmax : te;  // past the survivor words: the pair is not used, any address inside will do
                v[q] = *reinterpret_cast<const uint4 *>(w64 + te);
            }
#pragma unroll
            for (int j = 0; j < 24; j++) {
                const int jj = 24 * sb + j, t = tb - jj, ph = 5 - (jj % 6);
                const uint4 q = v[j >> 1];
                const unsigned long long w =
                    (j & 1) ? (((unsigned long long)q.y << 32) | q.x) : (((unsigned long long)q.w << 32) | q.z);
                const uint32_t d = (uint32_t)(w >> (l & 63u)) & 1u;
                const uint32_t bit = (l >> ph) & 1u;         // decoded bit t: the index bit this step replaces
                const uint32_t nl = (l & ~(1u << ph)) | (d << ph);
                const bool live = t < T;
                l = live ? nl : l;
                word = (word << 1) | (live ? bit : 0u);
                if ((jj & 31) == 31 && emit) {
                    const int wi = t >> 5;  // t is a multiple of 32 here
                    if (wi < nwords) bits[wi] = word;
                }
            }
        }
    }
    return l;
}


// ---- packet assembly by the whole wave (what assemble_frame does with one lane) --------------------------------
// CRC-32 arithmetic on the reflected polynomial (bit 31 = x^0), as zlib's crc32_combine does it: the CRC of a message
// cut into pieces is the sum of the pieces' CRCs, each multiplied by x^(8 * bytes that follow it).
constexpr uint32_t CRC_POLY = 0xEDB88320u;
__host__ __device__ constexpr uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 31; i >= 0; i--) {
        p ^= (0u - ((a >> i) & 1u)) & b;
        b = (b >> 1) ^ ((0u - (b & 1u)) & CRC_POLY);
    }
    return p;
}
struct X2nTable {
    uint32_t v[16];  // x^(2^k) for k = 0 .. 15
};
constexpr X2nTable make_x2n()
{
    X2nTable t = {};
    uint32_t p = 1u << 30;  // x^1
    for (int k = 0; k < 16; k++) {
        t.v[k] = p;
        p = gf2_mul(p, p);
    }
    return t;
}
__device__ constexpr X2nTable c_x2n = make_x2n();

// x^(8 n), n < 8192
__device__ __forceinline__ uint32_t gf2_x8n(uint32_t n)
{
    uint32_t p = 1u << 31;
#pragma unroll 1
    for (int b = 0; b < 13; b++) {
        const uint32_t q = gf2_mul(c_x2n.v[3 + b], p);
        p = ((n >> b) & 1u) ? q : p;
    }
    return p;
}

constexpr int LS_DW = 1032;  // descrambled words a wave keeps in LDS (32 782 steps are 1025)

// CRC-32 (standard: preset and final inversion) of bytes [0, mlen) of the PSDU from dword q0 on, by all lanes: every
// lane takes a run of dwords, the runs' CRCs are combined.  dw: descrambled frame words in LDS (PSDU dword q is bits
// 16 + 32 q ...).  The same value in every lane.
__device__ __forceinline__ uint32_t ls_psdu(const uint32_t *dw, int q) { return (dw[q] >> 16) | (dw[q + 1] << 16); }

__device__ __forceinline__ uint32_t ls_crc(const uint32_t *dw, int q0, int mlen, const uint32_t *s_crc, int lane)
{
    const int nd = (mlen + 3) >> 2, per = (nd + 63) >> 6;
    const int d0 = lane * per, d1 = min(d0 + per, nd);
    const int end = min(4 * d1, mlen);  // bytes of the message up to the end of this lane's run
    uint32_t c = 0xFFFFFFFFu;
    for (int d = d0; d < d1; d++) {
        const uint32_t w = ls_psdu(dw, q0 + d);
        const int nb = min(4, mlen - 4 * d);
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (b < nb) c = crc_byte(c, (w >> (8 * b)) & 0xffu, s_crc);
    }
    c ^= 0xFFFFFFFFu;  // a lane without bytes holds 0, which adds nothing
    uint32_t part = d0 < d1 ? gf2_mul(gf2_x8n((uint32_t)(mlen - end)), c) : 0u;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) part ^= __shfl_xor(part, m);
    return part;
}

// assemble_frame (assemble.h) with the frame's words spread over the lanes of a wave: descrambling, the CRC-32 of an
// MPDU and the copy into the PDU slot run in parallel, the walk over the (A-)MPDU structure is wave-uniform.
// bits: the frame's decoded words in memory (contiguous); dw: LS_DW words of LDS of this wave.
__device__ __forceinline__ void ls_assemble(const uint32_t *__restrict__ bits, uint32_t *dw, const gr80211_frame_info &fi, const int fr,
                                            const int nwords, const uint32_t *s_crc, uint8_t *__restrict__ out, int out_stride,
                                            int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu,
                                            uint8_t *__restrict__ bits_out, int bits_stride, int strict, int lane)
{
    const int T = fi.trellis;
    const int T32 = (T + 31) / 32;
    const int K = T32 < nwords ? T32 : nwords;  // PSDU dwords 0 .. K - 1 exist (dword q needs frame words q and q + 1)
    // ---- descrambler sequence (lib/decode_impl.cc:304-323): S[0..6] = the first seven decoded bits, S[i] = S[i-7] ^ S[i-4];
    // its first 160 bits as five words, word k of the frame starts at bit 32 k mod 127 of the sequence
    uint32_t W[5];
    {
        const uint32_t first = bits[0];
        uint32_t hist = 0, w0 = 0, w1 = 0;
        for (int i = 0; i < 64; i++) {
            const uint32_t sbit = i < 7 ? (first >> i) & 1u : ((hist >> 6) ^ (hist >> 3)) & 1u;
            hist = (hist << 1) | sbit;
            if (i < 32) w0 |= sbit << i;
            else w1 |= sbit << (i - 32);
        }
        W[0] = w0;
        W[1] = w1;
#pragma unroll
        for (int k = 2; k < 5; k++) W[k] = W[k - 1] ^ ((W[k - 2] >> 8) | (W[k - 1] << 24));
    }
    uint8_t *bo = bits_out ? bits_out + (size_t)fr * bits_stride : nullptr;
    for (int k = lane; k <= K && k < LS_DW; k += 64) {
        const uint32_t o = (32u * (uint32_t)k) % 127u, i = o >> 5, sh = o & 31u;
        const uint32_t lo = i == 0 ? W[0] : i == 1 ? W[1] : i == 2 ? W[2] : W[3];
        const uint32_t hi = i == 0 ? W[1] : i == 1 ? W[2] : i == 2 ? W[3] : W[4];
        const uint32_t sk = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        const uint32_t d = (k < nwords ? bits[k] : 0u) ^ sk;
        dw[k] = d;
        if (bo && k < T32 && (k + 1) * 4 <= bits_stride) *(uint32_t *)(bo + (size_t)k * 4) = d;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- the walk over the PSDU (lib/decode_impl.cc:325-520); every lane holds the same values
    uint8_t *o = out + (size_t)fr * out_stride;
    const bool vht = fi.format == GR80211_F_VHT;
    const bool walk = strict && (vht || fi.ampdu);
    const int plen = fi.len < (T - 16) / 8 ? fi.len : (T - 16) / 8;
    int wpos = 0, npdu = 0, q = 0, tl = 0, procd = 16;
    bool done = false;
    if (!walk && !vht && fi.ampdu) done = true;   // HT A-MPDU: nothing published (lib/decode_impl.cc:432-435)
    if (!walk && vht && procd >= T) done = true;
    while (!done && q < K) {
        int mlen, mpad, eof = 0;
        if (walk) {
            // the standard's rule: the next delimiter with a good CRC-8 and signature, its own length
            const int at = 4 * q;
            if (at + 4 > plen) break;
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)ls_psdu(dw, q));
            q++;
            if (!delimiter_ok(d)) continue;
            const int l = (int)((d >> 4) & 0xfffu) | (vht ? (int)((d >> 2) & 3u) << 12 : 0);
            if (l == 0 || at + 4 + l > plen) continue;
            mlen = l;
            mpad = (l + 3) & ~3;
        } else if (vht) {
            // the reference's rule: delimiters are trusted, the length is OR-accumulated (lib/decode_impl.cc:336-358)
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)ls_psdu(dw, q));
            q++;
            procd += 32;
            if (procd > T) break;
            eof = (int)(d & 1u);
            tl |= (int)((d >> 2) & 1u) << 12;
            tl |= (int)((d >> 3) & 1u) << 13;
            tl |= (int)((d >> 4) & 0xfffu);
            const int padded = (tl / 4 + ((tl % 4) != 0)) * 4;
            procd += padded * 8;
            if (procd > T) break;
            mlen = tl;
            mpad = padded;
            if (mpad == 0) {
                if (eof) break;
                continue;
            }
        } else {
            mlen = fi.len;
            mpad = fi.len;
        }
        const int nd = (mpad + 3) >> 2;       // dwords the subframe takes (a zero-length legacy frame takes none)
        if (q + nd > K) break;                // the frame ends inside it: never completed, nothing more follows
        const bool store = (mlen <= GR80211_MAX_PSDU) && (wpos + 3 + mlen + 1 <= out_stride);
        if (store && nd > 0 && ls_crc(dw, q, mlen, s_crc, lane) == 558161692u) {  // lib/decode_impl.cc:374,453
            for (int j = lane; j < mlen; j += 64) o[wpos + 3 + j] = (uint8_t)(ls_psdu(dw, q + (j >> 2)) >> (8 * (j & 3)));
            if (lane == 0) {
                o[wpos] = (uint8_t)fi.format;
                o[wpos + 1] = (uint8_t)(mlen % 256);
                o[wpos + 2] = (uint8_t)(mlen / 256);
                o[wpos + 3 + mlen] = (uint8_t)fi.mcs;
            }
            wpos += mlen + 4;
            npdu++;
        }
        q += nd;
        if (!walk && (!vht || eof)) done = true;
    }
    if (lane == 0) {
        out_len[fr] = wpos;
        n_pdu[fr] = npdu;
    }
}

template <bool RING>
__global__ __launch_bounds__(64 * LS_WAVES) void k_decode_lane_state(
    const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info, const int32_t *__restrict__ slot_frame,
    const GroupInfo *__restrict__ groups, int n_groups, uint32_t *__restrict__ dec, uint32_t *__restrict__ bitsw,
    uint8_t *__restrict__ out, int out_stride, int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu,
    uint8_t *__restrict__ bits_out, int bits_stride, int strict, const SchedTotals *__restrict__ totals)
{
    __shared__ uint32_t s_crc[256];
    __shared__ uint32_t s_dw[LS_WAVES][LS_DW];
    __shared__ float s_ring[RING ? LS_WAVES : 1][RING ? LS_RING_MAX : 1];
    {
        uint32_t c = threadIdx.x & 255u;
        for (int i = 0; i < 8; i++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        s_crc[threadIdx.x & 255u] = c;
    }
    __syncthreads();  // the only meeting of the waves
    n_groups = groups_to_run(n_groups, totals);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot0 = blockIdx.x * LS_WAVES + wv;
    if (slot0 >= n_groups * 64) return;
    // groups are numbered by rising trellis length within a code rate: the longest start first
    const int g = n_groups - 1 - slot0 / 64, sl = slot0 % 64;
    const int lane = threadIdx.x & 63;
    const int fr = __builtin_amdgcn_readfirstlane(slot_frame[g * 64 + sl]);
    if (fr < 0) return;
    const GroupInfo gi = groups[g];
    const gr80211_frame_info fi = info[fr];
    const int T = __builtin_amdgcn_readfirstlane(fi.trellis);
    const int n = __builtin_amdgcn_readfirstlane(fi.total);
    const unsigned long long off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(fi.llr_off >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(fi.llr_off & 0xffffffffll));
    const int steps = __builtin_amdgcn_readfirstlane(gi.steps), words = __builtin_amdgcn_readfirstlane(gi.words);
    // the group's survivor and bit areas hold 64 frames of `steps` steps: this frame's share of each is contiguous
    unsigned long long *w64 = reinterpret_cast<unsigned long long *>(dec + gi.dec_off) + (size_t)sl * steps;
    uint32_t *bits = bitsw + gi.bits_off + (size_t)sl * words;
    const cfloatp base = (cfloatp)(unsigned long long)(llr + off);
    if constexpr (RING) {
        const float *gl = llr + off;
        switch (__builtin_amdgcn_readfirstlane(gi.cr)) {
        case CR_12: lsr_forward<CR_12>(gl, n, T, w64, lane, s_ring[wv]); break;
        case CR_23: lsr_forward<CR_23>(gl, n, T, w64, lane, s_ring[wv]); break;
        case CR_34: lsr_forward<CR_34>(gl, n, T, w64, lane, s_ring[wv]); break;
        default: lsr_forward<CR_56>(gl, n, T, w64, lane, s_ring[wv]); break;
        }
    } else {
        switch (__builtin_amdgcn_readfirstlane(gi.cr)) {
        case CR_12: ls_forward<CR_12>(base, n, T, w64, lane); break;
        case CR_23: ls_forward<CR_23>(base, n, T, w64, lane); break;
        case CR_34: ls_forward<CR_34>(base, n, T, w64, lane); break;
        default: ls_forward<CR_56>(base, n, T, w64, lane); break;
        }
    }
    // the walk reads survivor words other lanes of this wave stored: they have reached memory when the counter is zero
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

    const int seg = TB_UNIT * ((T + TB_UNIT * 64 - 1) / (TB_UNIT * 64));
    const int nseg = (T + seg - 1) / seg;
    const int wmax = ((T - 1) & ~1);  // last pair of survivor words that exists (steps is even and >= T)
    const int k = lane < nseg ? lane : nseg - 1;  // idle lanes shadow the top segment (same loads, same stores)
    uint32_t seen = 0;
    uint32_t end = ls_walk(w64, T, wmax, seg, TB_WARM, k, 0u, bits, words, seen);
    for (int pass = 0; pass < 64; pass++) {
        const uint32_t above = __shfl_down(end, 1);
        const bool ok = lane >= nseg - 1 || seen == above;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        // every lane again, from where its upper neighbour ended; the topmost from state 0 at the last step
        const uint32_t start = lane >= nseg - 1 ? 0u : above;
        end = ls_walk(w64, T, wmax, seg, 0, k, start, bits, words, seen);
        seen = start;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    ls_assemble(bits, s_dw[wv], fi, fr, words, s_crc, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict, lane);
}

}  // namespace

void launch_decode_lane_state(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame, const GroupInfo *groups,
                              int n_groups, uint32_t *dec, uint32_t *bitsw, uint8_t *out, int out_stride, int32_t *out_len,
                              int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict, const SchedTotals *totals, hipStream_t s)
{
    const int waves = n_groups * 64;
    // Soft bits through LDS pay from a few waves per SIMD on (measured: level at 2000 frames, 2-5 % faster from 4000 on,
    // 20 % slower for a lone wave); GR80211_LS_RING=1 / 0 forces the form (read per launch, so that a test can switch it).
    const char *e = getenv("GR80211_LS_RING");
    const bool ring = e && (e[0] == '0' || e[0] == '1') ? e[0] == '1' : n_groups >= LS_RING_FROM_GROUPS;
    if (ring)
        hipLaunchKernelGGL(k_decode_lane_state<true>, dim3((waves + LS_WAVES - 1) / LS_WAVES), dim3(64 * LS_WAVES), 0, s, llr, info, slot_frame,
                           groups, n_groups, dec, bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict, totals);
    else
        hipLaunchKernelGGL(k_decode_lane_state<false>, dim3((waves + LS_WAVES - 1) / LS_WAVES), dim3(64 * LS_WAVES), 0, s, llr, info, slot_frame,
                           groups, n_groups, dec, bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict, totals);
}

}  // namespace gr80211
