// Viterbi forward pass for small batches: four wavefronts (one workgroup, one wave per SIMD of a CU) share one
// group of 64 frames.
//
// With lane = frame a group's trellis is one wave's serial work; a batch of fewer groups than the chip has SIMDs
// leaves most of it idle for the whole length of the longest trellis.  Here wave w keeps the path metrics of the
// 16 states [16w, 16w + 16).  Those are the predecessors of the eight butterflies j in [8w, 8w + 8) of one
// trellis step, and the 16 states those produce (j and j + 32) are in turn the predecessors of the eight
// butterflies 4w .. 4w+3 and 16+4w .. 16+4w+3 of the next step: two steps run inside a wave before anything has to
// move.  After the second step each wave owns four states of every other wave's next 16; they change hands through
// LDS (double-buffered, one workgroup barrier per step pair).  Same float operations in the same order as the
// one-wave kernel and the same survivor words ([step pair][lane][w0e w1e w0o w1o], state n at bit n ^ 7), so the
// traceback kernel reads them unchanged.
//
// Variants that were measured and not kept: an exchange after every step (states split in halves or quarters by
// range), and four lanes per frame with the exchange as DPP operands of the adds (DPP-modified adds issue at about
// a quarter of the plain rate).
#include "split_forward.h"

namespace gr80211 {

namespace {

__global__ __launch_bounds__(256) void k_viterbi_split(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                                       const int32_t *__restrict__ slot_frame,
                                                       const GroupInfo *__restrict__ groups, int n_groups,
                                                       uint32_t *__restrict__ dec, const SchedTotals *__restrict__ totals)
{
    const int g = blockIdx.x;
    if (g >= groups_to_run(n_groups, totals)) return;  // whole workgroup
    const int lane = threadIdx.x & 63;
    const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float tile[64 * TILE_STRIDE];
    __shared__ float xch[2 * XCH_BUF];
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    const int myn = fr >= 0 ? info[fr].total : 0;
    const unsigned long long mybase = myn > 0 ? (unsigned long long)info[fr].llr_off : 0ull;
    uint32_t *d = dec + gi.dec_off;
    switch (gi.cr) {
    case CR_12: split_dispatch<CR_12>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_23: split_dispatch<CR_23>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_34: split_dispatch<CR_34>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    default: split_dispatch<CR_56>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    }
}

__global__ __launch_bounds__(128) void k_viterbi_split2(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                                        const int32_t *__restrict__ slot_frame,
                                                        const GroupInfo *__restrict__ groups, int n_groups,
                                                        uint32_t *__restrict__ dec, const SchedTotals *__restrict__ totals)
{
    const int g = blockIdx.x;
    if (g >= groups_to_run(n_groups, totals)) return;  // whole workgroup
    const int lane = threadIdx.x & 63;
    const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float tile[64 * TILE_STRIDE];
    __shared__ float xch[2 * XCH2_BUF];
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    const int myn = fr >= 0 ? info[fr].total : 0;
    const unsigned long long mybase = myn > 0 ? (unsigned long long)info[fr].llr_off : 0ull;
    uint32_t *d = dec + gi.dec_off;
    switch (gi.cr) {
    case CR_12: split2_dispatch<CR_12>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_23: split2_dispatch<CR_23>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_34: split2_dispatch<CR_34>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    default: split2_dispatch<CR_56>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    }
}

}  // namespace

void launch_viterbi_split2(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                           const GroupInfo *groups, int n_groups, uint32_t *dec, const SchedTotals *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_viterbi_split2, dim3(n_groups), dim3(128), 0, s, llr, info, slot_frame, groups, n_groups, dec, totals);
}

void launch_viterbi_split(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                          const GroupInfo *groups, int n_groups, uint32_t *dec, const SchedTotals *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_viterbi_split, dim3(n_groups), dim3(256), 0, s, llr, info, slot_frame, groups, n_groups, dec, totals);
}

}  // namespace gr80211
