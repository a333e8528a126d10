// Kernel launchers of the RX hot path (internal to the library).
#pragma once

#include "rx_common.h"

namespace gr80211 {

// One group = 64 frames decoded by one wavefront (one frame per lane), all of one code rate.
struct GroupInfo {
    int32_t cr;       // code rate of every frame in the group
    int32_t steps;    // trellis steps run (max over lanes, rounded up to whole chunks)
    int32_t nchunks;  // LLR tile refills
    int32_t words;    // 32-bit words of decoded bits per lane
    uint64_t dec_off; // offset of the group's survivor words in the decision arena (uint32 units)
    uint64_t bits_off;// offset of the group's decoded-bit words in the bit arena (uint32 units)
};

// Totals the host needs to size the arenas (copied back once per batch).
struct SchedTotals {
    uint64_t llr_floats;
    uint64_t dec_words;
    uint64_t bits_words;
    int32_t n_groups;
    int32_t n_valid;
};

// Room the caller has for what the scheduler lays out (soft bits, survivor words, decoded-bit words).  A batch that
// needs more is not run: the scheduler reports n_groups = -1 and every later kernel of the batch leaves at once.
struct SchedCaps {
    uint64_t llr_floats, dec_words, bits_words;
};
constexpr SchedCaps SCHED_NO_CAPS = {~0ull, ~0ull, ~0ull};

constexpr int SCHED_NB = 513;  // trellis-length bins of 64 steps
constexpr int SCHED_WS_WORDS = 4 * SCHED_NB + 8;  // scheduler scratch: bin counters, rate starts, valid count

// steps per LLR tile for each code rate (whole puncture periods)
__host__ __device__ constexpr int chunk_steps(int cr) { return cr == CR_12 ? 32 : cr == CR_23 ? 40 : cr == CR_34 ? 48 : 50; }

int upload_tables(const Tables &t);

void launch_preamble(const void *iq0, const void *iq1, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                     gr80211_frame_info *info, void *chan1, void *chan2, const gr80211_sig_desc *sig, const void *chan_in,
                     void *chanl_out, hipStream_t s);
void launch_sig_to_desc(const gr80211_sig_desc *sig, int n, gr80211_frame_desc *desc, hipStream_t s);
void launch_signal_copy(const void *iq0, const void *iq1, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                        const gr80211_frame_info *info, const uint64_t *out_off, void *out0, void *out1, uint64_t out_len,
                        hipStream_t s);
void launch_desc_to_info(const gr80211_dec_desc *desc, int n, uint64_t llr_floats, gr80211_frame_info *info, hipStream_t s);
void launch_schedule(gr80211_frame_info *info, int n, int assign_llr, int32_t *slot_frame, int slot_cap,
                     GroupInfo *groups, int group_cap, SchedTotals *totals, uint32_t *ws, unsigned long long *blk_llr,
                     SchedCaps caps, hipStream_t s);
void launch_pre_gather(const int32_t *path, int n, const gr80211_frame_info *info_in, const void *chan1_in,
                       const void *chan2_in, gr80211_frame_info *info_out, void *chan1_out, void *chan2_out, hipStream_t s);
// `totals` (the scheduler's record on the device) lets the kernels of a batch that was launched without waiting for
// the scheduler see what it decided: a batch it refused (n_groups < 0) is not run, and no group beyond its count is.
void launch_demod(const void *iq0, const void *iq1, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                  const gr80211_frame_info *info, const void *chan1, const void *chan2, float *llr,
                  const SchedTotals *totals, hipStream_t s, bool axis = false);
void launch_viterbi(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                    const GroupInfo *groups, int n_groups, uint32_t *dec, const SchedTotals *totals, hipStream_t s);
void launch_traceback(const uint32_t *dec, const gr80211_frame_info *info, const int32_t *slot_frame,
                      const GroupInfo *groups, int n_groups, uint32_t *bitsw, uint8_t *out, int out_stride,
                      int32_t *out_len, int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict,
                      const SchedTotals *totals, hipStream_t s);
// Forward pass with four wavefronts per group, for batches too small to fill the chip with one wave per group;
// same survivor words as launch_viterbi, to be followed by launch_traceback.
void launch_viterbi_split(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                          const GroupInfo *groups, int n_groups, uint32_t *dec, const SchedTotals *totals, hipStream_t s);
// The same with two waves per group (states 0-31 / 32-63).
void launch_viterbi_split2(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                          const GroupInfo *groups, int n_groups, uint32_t *dec, const SchedTotals *totals, hipStream_t s);
void launch_decode(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                   const GroupInfo *groups, int n_groups, uint32_t *dec, uint32_t *bitsw, uint8_t *out, int out_stride,
                   int32_t *out_len, int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict,
                   const SchedTotals *totals, bool large_form, hipStream_t s, bool axis = false);
// Groups a decoder kernel may run: the launch's count, cut to what the scheduler reports on the device.
__device__ __forceinline__ int groups_to_run(int n_groups, const SchedTotals *totals)
{
    if (!totals) return n_groups;
    const int t = totals->n_groups;
    return t < n_groups ? (t < 0 ? 0 : t) : n_groups;
}
#ifdef GR80211_TIMELINE
int read_timeline(unsigned long long *dst, int n_groups);
#endif
// Two waves per group, forward pass and walk back in one kernel (k_decode_split2).
void launch_decode_split2(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame, const GroupInfo *groups,
                          int n_groups, uint32_t *dec, uint32_t *bitsw, uint8_t *out, int out_stride, int32_t *out_len,
                          int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict, const SchedTotals *totals, hipStream_t s);
// One frame per wavefront, one trellis state per lane (lane_state.hip): forward pass, walk back and packet assembly
// in one kernel, for batches of so few frames that the forms above leave most of the chip idle.  Uses the same
// group layout: a frame's survivor words and decoded bits take its 64th of its group's areas.
void launch_decode_lane_state(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame, const GroupInfo *groups,
                              int n_groups, uint32_t *dec, uint32_t *bitsw, uint8_t *out, int out_stride, int32_t *out_len,
                              int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict, const SchedTotals *totals, hipStream_t s);
void launch_ndp_emit(const void *iq0, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                     const gr80211_frame_info *info, uint8_t *out, int out_stride, int32_t *out_len, int32_t *n_pdu,
                     const SchedTotals *totals, hipStream_t s);
void launch_clear_outputs(int32_t *out_len, int32_t *n_pdu, int n, hipStream_t s);

}  // namespace gr80211
