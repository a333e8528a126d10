// Launchers of the frame-detection kernels (internal to the library).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gr80211_rx.h"

namespace gr80211 {

long long det_mask_words(long long n);  // uint32 words of the plateau bit stream of n samples
int det_edge_blocks(long long n);       // entries of the per-block edge counters

void launch_det_ac(const void *iq, long long n, uint32_t *mask, hipStream_t s);
// totals[0], totals[1] = number of rising / falling edges; blk_* become exclusive offsets
void launch_det_edge_count(const uint32_t *mask, long long n, uint32_t *blk_rise, uint32_t *blk_fall, uint32_t *totals,
                           hipStream_t s);
void launch_det_edge_emit(const uint32_t *mask, long long n, const uint32_t *blk_rise, const uint32_t *blk_fall,
                          long long cap, long long *rise, long long *fall, hipStream_t s);
// totals[2] = number of sync bytes emitted; blk becomes exclusive offsets over blocks of 256 runs
void launch_det_trigger_sync(const void *iq, long long n, const long long *rise, const long long *fall, long long nruns,
                             long long *seen_at, long long *scan_lo, long long *prev_seen, long long *pos, float *rad,
                             float *snr, float *rssi, uint32_t *emitted, uint32_t *blk, uint32_t *totals, hipStream_t s);
void launch_det_emit(const uint32_t *emitted, long long nruns, const uint32_t *blk, const long long *pos, const float *rad,
                     const float *snr, const float *rssi, long long n, gr80211_frame_desc *desc, int max_desc, uint32_t mu_pos,
                     hipStream_t s);

// `signal`'s trigger search over nd sync bytes whose L-SIG results are in info: out[0 .. *count) = the bytes it deals
// with, in order, n_avail cut to the frame's own samples.  jump_a, jump_b, path: nd + 1 entries each.
void launch_span_filter(const gr80211_frame_desc *desc, const gr80211_frame_info *info, int nd, long long n,
                        int32_t *jump_a, int32_t *jump_b, int32_t *path, gr80211_frame_desc *out, uint32_t *count,
                        hipStream_t s);

}  // namespace gr80211
