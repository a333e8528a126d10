/*
 * C-ABI of the MI355X-native 802.11 OFDM RX hot path (signal -> demod/demod2 -> decode).
 *
 * Plain C: opaque context handle, plain pointers and sizes, integer return codes.  This is
 * the boundary a GNU Radio out-of-tree block (or any FFI: ctypes, cgo, JNI) binds to.  Each
 * entry point names the reference interface it stands in for; paths are relative to the
 * reference tree.
 *
 * Conventions
 *  - Pointers named d_* are DEVICE pointers (HBM of the context's GPU); h_* are host pointers.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the context's own
 *    stream).  Calls return after the work is enqueued and the small bookkeeping copy the
 *    pipeline needs has completed; results in device buffers are ready after the stream is
 *    synchronised (gr80211_sync).
 *  - No exceptions, no global state: malformed frames are dropped in-band (status field), as
 *    the reference blocks do; API misuse returns a negative code and sets gr80211_last_error.
 *  - A context is not thread-safe; distinct contexts are independent.
 */
#ifndef GR80211_RX_H
#define GR80211_RX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR80211_OK 0
#define GR80211_ERR_ARG (-1)
#define GR80211_ERR_HIP (-2)
#define GR80211_ERR_NOMEM (-3)

/* frame formats: C8P_F_L / C8P_F_HT / C8P_F_VHT (lib/cloud80211phy.h:35-37) */
#define GR80211_F_L 0
#define GR80211_F_HT 1
#define GR80211_F_VHT 2

/* per-frame status */
#define GR80211_ST_OK 0
#define GR80211_ST_DROP_LSIG 1        /* signal: L-SIG check failed (lib/signal_impl.cc:156-160) */
#define GR80211_ST_DROP_LEN_CHECK 2   /* demod: HT/VHT length sanity failed (lib/demod_impl.cc:159,186) */
#define GR80211_ST_DROP_UNSUPPORTED 3 /* nSS>2, MU group id, VHT MCS>9, two data streams on one antenna */
#define GR80211_ST_DROP_DECODE_LIMIT 4/* decode: len>4095 or trellis>32782 (lib/decode_impl.cc:93) */
#define GR80211_ST_NDP 5              /* VHT sounding NDP seen by the one-antenna chain: no soft bits; its PDU is the
                                         channel message [20][len lo][len hi][128 complex64 raw VHT-LTF samples]
                                         (lib/demod_impl.cc:238-249,397-398, lib/decode_impl.cc:100-121) */

#define GR80211_MAX_TRELLIS 32782 /* DECODE_T_MAX, lib/decode_impl.h:36 */
#define GR80211_MAX_PSDU 4095     /* DECODE_B_MAX, lib/decode_impl.h:35 */

typedef struct gr80211_ctx gr80211_ctx;

/*
 * One captured frame = what `sync` tells `signal` with its tagged trigger byte
 * (lib/sync_impl.cc:122-136; read at lib/signal_impl.cc:83-92): where the frame is and the
 * tags rad/snr/rssi.  sync_off is the sample index of the trigger byte (L-LTF start + 16) in
 * the IQ arena(s); n_avail the number of samples of the capture available from there.
 */
typedef struct {
    uint64_t sync_off;
    uint32_t n_avail;
    float rad;  /* CFO in rad/sample */
    float snr;
    float rssi;
    uint32_t mu_pos;   /* this station's position (0 or 1) in a VHT MU group: the mupos argument of
                          ieee80211::demod::make(mupos, mugid) (lib/demod_impl.cc:28-39); only read for MU frames */
    uint32_t reserved; /* 0 */
} gr80211_frame_desc; /* 32 bytes */

/*
 * Per-frame result record = the stream tags of the reference blocks:
 * signal -> demod: cfo,snr,rssi,mcs,len,nsamp (lib/signal_impl.cc:135-152);
 * demod -> decode: format,mcs,len,cr,ampdu,trellis,total,sssnr0/1 (lib/demod_impl.cc:224-253,
 * lib/demod2_impl.cc:230-250); plus llr_off = offset of the frame's LLR run in the LLR arena.
 */
typedef struct {
    int32_t status;
    int32_t lsig_mcs, lsig_len, nsamp;
    int32_t format, mcs, len, cr, ampdu, trellis, total;
    int32_t nSym, nSS, nLTF, nCBPS, nDBPS, nBPSCS, mod;
    float sssnr0, sssnr1;
    float cfo_hz, snr, rssi;
    int32_t slot;      /* internal: decoder lane assignment */
    int64_t llr_off;   /* floats */
    int32_t data_off;  /* first data symbol, samples after L-SIG */
    int32_t nSymSamp;
} gr80211_frame_info; /* 112 bytes */

/*
 * What `decode` reads from the tags on the first LLR of a packet (lib/decode_impl.cc:70-90).
 */
typedef struct {
    int64_t llr_off; /* floats, into d_llr */
    int32_t n_llr;   /* tag "total" */
    int32_t format, len, cr, mcs, ampdu, trellis;
    int32_t reserved;
} gr80211_dec_desc; /* 40 bytes */

/*
 * What `signal` tells `demod` with the tags on the first item of a frame (lib/signal_impl.cc:135-152, read at
 * lib/demod_impl.cc:75-100): the frame's items start right after L-SIG and have the CFO taken out.
 */
typedef struct {
    uint64_t off;     /* index of the frame's first item in the arena(s) */
    uint32_t n_avail; /* items of the frame the arena holds from there (nsamp, or fewer at the end of a stream) */
    int32_t mcs;      /* tag "mcs": L-SIG rate 0..7 */
    int32_t len;      /* tag "len": L-SIG length */
    int32_t nsamp;    /* tag "nsamp": 80 * symbols the L-SIG announces */
    float cfo;        /* tags "cfo", "snr", "rssi": handed on unchanged */
    float snr;
    float rssi;
    uint32_t mu_pos;  /* as in gr80211_frame_desc */
    uint32_t reserved;
} gr80211_sig_desc; /* 48 bytes */

#define GR80211_FLAG_DESC_ON_DEVICE 1u /* the descriptor array argument is a device pointer */
/*
 * gr80211_rx_batch only: do not wait for the scheduler's totals in the middle of the batch.  The work buffers stay as
 * large as the batches before have made them (a first batch without the flag sizes them), the kernels are launched
 * for the most groups the frame count allows and take the real count from the device, and the call returns with
 * everything queued.  A batch that needs more room than there is is not run: after the stream has passed it,
 * gr80211_last_batch_refused says so, no PDU has been written, and the caller repeats the batch without the flag.
 */
#define GR80211_FLAG_NO_WAIT 2u

/*
 * One two-user VHT MU-MIMO frame to transmit: the arguments of the reference's Python generator
 * phy80211.genAmpduMu(nUser=2, bfQ, groupId, ampdu0, mod0, ampdu1, mod1) (tools/phy80211.py:183-224), one
 * space-time stream per user.
 */
typedef struct {
    uint64_t out_off;     /* sample index in both output arenas where the frame starts */
    uint32_t psdu_off[2]; /* byte offsets of the users' A-MPDUs in d_psdu */
    int32_t psdu_len[2];  /* their lengths in bytes: multiples of 4 (as mac80211.genAmpduVHT builds them) */
    int32_t mcs[2];       /* VHT MCS 0..8 of each user */
    int32_t group_id;     /* 1..62 */
    int32_t q_index;      /* which steering matrix set of d_q the frame uses */
    int32_t scrambler;    /* initial scrambler state 1..127 */
    float scale;          /* genFinalSig's multiplier */
    float cfo_rad;        /* rotation per sample */
    uint32_t reserved;
} gr80211_tx_mu_desc; /* 56 bytes */

/*
 * One frame to transmit: what the reference's Python generator takes in
 * phy80211.genFromMpdu(mpdu, modulation(format, mcs, bw20, nSTS, shortGi=False)) / genFromAmpdu(ampdu, ...)
 * (tools/phy80211.py:80-181) and genFinalSig(multiplier, cfoHz, ...) (tools/phy80211.py:1077-1094).
 */
typedef struct {
    uint64_t out_off;  /* sample index in the output arena(s) where the frame (its L-STF) starts */
    uint32_t psdu_off; /* byte offset of the PSDU in d_psdu: the MPDU (legacy, HT) or the A-MPDU with its
                          delimiters (VHT), FCS included */
    int32_t psdu_len;  /* bytes, 1..4095 (VHT: APEP length; padding to the symbol boundary is added; VHT with 0 =
                          null data packet, the sounding frame without a data field) */
    int32_t format;    /* GR80211_F_* */
    int32_t mcs;       /* legacy 0..7, HT 0..15 (8..15 = two streams), VHT 0..8 */
    int32_t nss;       /* VHT: 1 or 2 space-time streams; ignored otherwise */
    int32_t scrambler; /* initial scrambler state 1..127 (the reference generator uses 93) */
    float scale;       /* genFinalSig's multiplier */
    float cfo_rad;     /* sample i of the frame is rotated by exp(j*i*cfo_rad): genFinalSig's cfoHz*2*pi/20e6 */
} gr80211_tx_desc; /* 40 bytes */

/* Library/device set-up.  device = HIP device ordinal. */
int gr80211_create(gr80211_ctx **out, int device);
void gr80211_destroy(gr80211_ctx *ctx);
const char *gr80211_last_error(gr80211_ctx *ctx);
int gr80211_sync(gr80211_ctx *ctx, void *stream);
/*
 * The HIP stream the context created for itself (what every call uses when it is handed stream == NULL), as a
 * hipStream_t: for a caller that wants other work (a collective, its own kernels) ordered on it.  Streams a context
 * creates get hardware queues of their own, which streams taken from another library's pool may not.
 */
void *gr80211_get_stream(gr80211_ctx *ctx);

/*
 * Whole RX path for a batch of independent frames: the work of
 *   ieee80211::signal/signal2::general_work (lib/signal_impl.cc:62-206, lib/signal2_impl.cc:63-212)
 *   ieee80211::demod/demod2::general_work  (lib/demod_impl.cc:59-342, lib/demod2_impl.cc:58-349)
 *   ieee80211::decode::general_work         (lib/decode_impl.cc:60-162)
 * d_iq0/d_iq1: complex64 IQ arenas of antenna 1/2 (d_iq1 NULL = one-antenna `demod` path).
 * d_out: n_frames slots of out_stride bytes; each receives the frame's PDU messages
 *        concatenated in the layout `decode` publishes: [format][len lo][len hi][MPDU][mcs]
 *        (lib/decode_impl.cc:359-361,414-419,512-516), only MPDUs whose FCS passes.
 * d_out_len[i] = bytes written for frame i, d_n_pdu[i] = number of PDUs.
 * d_info (nullable) receives the tag record of each frame.
 * The call waits once, after the preamble stage, for the scheduler's totals (they size the work buffers) unless
 * flags carries GR80211_FLAG_NO_WAIT; what follows that point is queued and the call returns.
 */
int gr80211_rx_batch(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                     const gr80211_frame_desc *desc, int n_frames, unsigned flags,
                     uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu,
                     gr80211_frame_info *d_info, void *stream);

/*
 * gr80211_rx_batch in two calls, for a caller that keeps several contexts busy from one thread (no counterpart in
 * the reference, where GNU Radio's scheduler runs every block on a thread of its own).  gr80211_rx_submit launches the
 * preamble and scheduling kernels and returns without waiting; gr80211_rx_finish waits for the scheduler's totals, which
 * size the work buffers, launches the demodulator and the decoder and returns without waiting for them.  Between the
 * two calls the context refuses every other batch call, and the arena and descriptors must stay as they are until the
 * stream has run the work of gr80211_rx_finish.  With two contexts: submit(A), submit(B), finish(A), finish(B), and
 * while A decodes, B's preamble and demodulator fill the SIMDs A leaves idle.  Same results as gr80211_rx_batch.
 */
int gr80211_rx_submit(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                      const gr80211_frame_desc *desc, int n_frames, unsigned flags, void *stream);
int gr80211_rx_finish(gr80211_ctx *ctx, uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu,
                      gr80211_frame_info *d_info);

/*
 * The receive path in the two steps a streaming caller takes anyway, with no wait for the GPU inside either.  Such a
 * caller (a block with the input contract of ieee80211::signal) first asks what every tagged sync byte starts --
 * gr80211_signal_batch: L-SIG decision and frame span per sync byte, the work of `signal`'s S_DEMOD state
 * (lib/signal_impl.cc:108-162) -- reads the records, replays `signal`'s trigger search over them on the host (a frame
 * hides the sync bytes inside its 224 + nsamp samples, a failed L-SIG 80: lib/signal_impl.cc:75-106,153-160) and then
 * continues with the frames that search keeps: gr80211_rx_resume = demod/demod2 + decode for frames whose preamble
 * results the context still holds from that gr80211_signal_batch call (no other batch call in between).
 * h_desc[i]: descriptor of the i-th kept frame (as given to gr80211_signal_batch, n_avail possibly cut to the frame's
 * span); h_index[i]: its position in that call; h_info: the caller's host copy of all records that call returned.
 * The records tell the host what the scheduler kernels will lay out, so the work buffers are sized before anything
 * is launched and nothing is waited for; outputs as for gr80211_rx_batch.  Should h_info not be what the device holds,
 * the batch may find less room than it needs: it then publishes nothing (every out_len 0) and
 * gr80211_last_batch_refused, asked after the stream has run the batch, returns 1.
 */
int gr80211_rx_resume(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                      const gr80211_frame_desc *h_desc, const int32_t *h_index, const gr80211_frame_info *h_info,
                      int n_frames, uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu,
                      gr80211_frame_info *d_info, void *stream);
int gr80211_last_batch_refused(gr80211_ctx *ctx);

/*
 * Frame detection over a whole capture: the work of the blocks upstream of `signal` in the reference's receive
 * flowgraph (examples/rx.grc: presiso -> trigger -> sync; examples/presiso.grc:68-137,218-229,
 * lib/trigger_impl.cc:59-117, lib/sync_impl.cc:61-196).  d_iq: complex64 samples of the (first) antenna.
 * Writes one descriptor per sync byte `sync` would emit, in stream order: sync_off = its position
 * (L-LTF start + 16), rad/snr/rssi = its tags, n_avail = samples from there to the end of the capture.  The
 * descriptors are ready for gr80211_signal_batch / gr80211_rx_batch with GR80211_FLAG_DESC_ON_DEVICE.
 * *n_desc (host) = number of sync bytes; more than max_desc is an error (the first max_desc are written).
 */
int gr80211_detect(gr80211_ctx *ctx, const void *d_iq, uint64_t n_samples, gr80211_frame_desc *d_desc, int max_desc,
                   int *n_desc, void *stream);

/*
 * Raw capture in, PDUs out: presiso -> trigger -> sync -> signal(2) -> demod(2) -> decode (examples/rx.grc,
 * examples/rx2.grc) over one capture.  Detection runs on antenna 1 (d_iq0) as in the reference flowgraphs.  Of the
 * sync bytes found, the frames are the ones `signal` deals with: after a frame's 224 + nsamp samples (80 after a
 * failed L-SIG check) it resumes its search, sync bytes in between are ignored (lib/signal_impl.cc:75-106,153-202).
 * Slot i of d_out / d_out_len / d_n_pdu belongs to the i-th such sync byte in stream order; d_desc_out (nullable,
 * max_frames entries) receives their descriptors.  *n_frames (host) = their number; more than max_frames is an
 * error.
 */
int gr80211_rx_capture(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t n_samples, uint8_t *d_out,
                       int out_stride, int32_t *d_out_len, int32_t *d_n_pdu, int max_frames,
                       gr80211_frame_desc *d_desc_out, int *n_frames, void *stream);

/*
 * Work memory shared by the contexts of a GPU.  A batch's large work buffers (soft bits; survivor words and decoded
 * bits: 6.5 + 9.7 + 0.15 GB for 100 000 frames of 1500 bytes at VHT MCS8) are in use for part of it only -- the soft
 * bits from its demodulator to the end of its decoder, the others for the length of its decoder -- so contexts that
 * take turns (one stream each, batches of neighbouring steps overlapping) need fewer of them than there are contexts.
 * After this call the context takes a soft-bit buffer and a survivor buffer from its GPU's pool for the length of every
 * batch call that decodes (gr80211_rx_batch, _rx_finish, _rx_capture, _rx_resume, _decode_batch) instead of keeping
 * its own; the pool creates at most max_sets survivor buffers and max_sets + 2 soft-bit buffers (max_sets: the largest
 * value any context of the GPU asked for), grows them like a context grows its buffers, and when every buffer of a
 * kind is still in use by an earlier batch the new batch's kernels that touch it wait, on the device, for the one that
 * was handed back first -- the host never blocks.  GR80211_FLAG_NO_WAIT batches are sized by the buffers they happen
 * to get: warm the pool with max_sets + 2 batches without the flag.  max_sets = 0 leaves the pool (the last context to
 * leave frees the buffers).  The environment variable GR80211_WORK_POOL=<n> makes every new context join with
 * max_sets = n.  Entry points that hand out pointers into the work buffers (gr80211_demod_batch,
 * gr80211_demod_tagged_batch) and gr80211_decode_batch_noalloc keep to the context's own memory.  No counterpart in
 * the reference, whose blocks hold fixed arrays (lib/decode_impl.h:73).
 */
int gr80211_use_work_pool(gr80211_ctx *ctx, int max_sets);
/* Bytes of the three large work buffers the context holds itself, bytes and number of sets of its GPU's pool (each nullable). */
int gr80211_work_memory(gr80211_ctx *ctx, uint64_t *own_bytes, uint64_t *pool_bytes, int *pool_sets);

/*
 * Position of this station in VHT MU groups (0 or 1; ieee80211::demod::make(mupos, mugid), lib/demod_impl.cc:28-39)
 * for the descriptors gr80211_detect / gr80211_rx_capture make themselves; callers that build descriptors set
 * gr80211_frame_desc.mu_pos.  Default 0.
 */
int gr80211_set_mu_pos(gr80211_ctx *ctx, int mu_pos);

/*
 * A-MPDU de-aggregation rule of this context.  0 (default) = the reference's: VHT delimiters are trusted unchecked
 * and their length field is OR-accumulated from subframe to subframe, aggregated HT PSDUs publish nothing
 * (lib/decode_impl.cc:329-435).  1 = IEEE 802.11-2016 10.13.3: every delimiter's CRC-8 and signature are checked, each
 * carries its own length, after a bad delimiter the search resumes on the next 4-byte boundary, zero-length
 * delimiters are padding; applies to VHT PSDUs and to HT PSDUs whose HT-SIG aggregation bit is set.  No counterpart
 * in the reference (it lists the fix as open work).
 */
int gr80211_set_ampdu_strict(gr80211_ctx *ctx, int on);

/*
 * Which decoder form small batches take.  A batch of at most max_frames frames is decoded with one wave per frame and
 * one trellis state per lane (k_decode_lane_state): the whole chip works on the batch at once, which is the shortest
 * way to its results when it runs alone (4000 VHT MCS8 frames of 1500 bytes: 1.1 ms against 3.4 ms), at 1.3 times the
 * vector instructions per frame of the forms that put a frame in a lane.  A caller that keeps several batches in
 * flight and is after their combined rate sets 0; max_frames < 0 goes back to the default (18 000, or
 * GR80211_LANE_STATE_MAX).  All forms decode the same bits (lib/decode_impl.cc:237-302).  No counterpart in the
 * reference, whose decoder has one form.
 */
int gr80211_set_lane_state_max(gr80211_ctx *ctx, int max_frames);

/*
 * signal/signal2 (and the format detection of demod/demod2) only: L-SIG decode, format detection,
 * HT-SIG / VHT-SIG-A / VHT-SIG-B decode and channel estimation of every frame, no data symbols.
 * d_info[i] receives status, L-SIG mcs/len, nsamp (the number of samples `signal` forwards after
 * L-SIG, lib/signal_impl.cc:126-131) and the frame's modulation record.  A streaming caller uses it
 * the way the reference chain uses `signal`: to learn how many samples a triggered frame spans before
 * it looks for the next trigger (lib/signal_impl.cc:75-106,153-160).
 */
int gr80211_signal_batch(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                         const gr80211_frame_desc *desc, int n_frames, unsigned flags,
                         gr80211_frame_info *d_info, void *stream);

/*
 * signal + demod only: IQ -> soft bits in the layout of the demod->decode stream, i.e. for each
 * frame `total` floats in decoder order starting at info.llr_off of an internal LLR arena whose
 * device address is returned in *d_llr (valid until the next call on this context).
 * Stands for signal/signal2 + demod/demod2 ::general_work as above.
 */
int gr80211_demod_batch(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                        const gr80211_frame_desc *desc, int n_frames, unsigned flags,
                        gr80211_frame_info *d_info, const float **d_llr, uint64_t *llr_floats,
                        void *stream);

/*
 * decode only: soft bits -> PDUs.  Stands for ieee80211::decode::general_work + vstb_init /
 * vstb_update / vstb_end / descramble / packetAssemble (lib/decode_impl.cc:60-520).
 * d_bits (nullable): n_frames slots of bits_stride bytes receiving the descrambled bit array
 * v_unCodedBits packed LSB-first (bit t of the frame at byte t/8, bit t%8).
 * With the descriptors on the host the call waits for nothing: the work buffers are sized from the descriptors, the
 * descriptors travel through page-locked memory of the context (the caller's array is free on return), and the
 * results are ready when the stream has run what the call gave it.  With GR80211_FLAG_DESC_ON_DEVICE the call waits
 * once, for the scheduler's totals.
 */
int gr80211_decode_batch(gr80211_ctx *ctx, const float *d_llr, uint64_t llr_floats,
                         const gr80211_dec_desc *desc, int n_frames, unsigned flags,
                         uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu,
                         uint8_t *d_bits, int bits_stride, void *stream);

/*
 * gr80211_decode_batch for callers that cannot allocate device memory once they run: serving loops and test
 * harnesses that forbid hipMalloc / hipFree after start-up.  No counterpart in the reference.  Same inputs, outputs and
 * results as gr80211_decode_batch, with these differences: the descriptors are a HOST array (the limits below are
 * checked on them before anything is launched); every work buffer (frame records, decoder lanes, survivor words,
 * decoded bits) lies in one 48-MiB array that is part of the library's code object, one copy per GPU, held until the
 * process ends; the call allocates and frees nothing, on the device or through any library.  Limits: at most 1024
 * frames per call, and the survivor words of the batch -- 512 bytes per trellis step of every group of up to 64
 * frames of one code rate, a group running as many steps as its longest frame -- must fit into the array's 46.8 MiB
 * left for them (e.g. 448 frames of 1500 bytes at one rate, or 1024 frames of up to 700 bytes); a batch beyond
 * either limit is refused with GR80211_ERR_ARG and gr80211_last_error says what it needed.  The array is shared by
 * every context and stream of the process on that GPU and keeps the last call's contents: calls that overlap on one
 * GPU race, so a caller runs them one after the other (the next call may start once the stream of the one before has
 * been synchronised).  Like gr80211_decode_batch it waits once, inside the call, for the scheduler's totals.
 */
int gr80211_decode_batch_noalloc(gr80211_ctx *ctx, const float *d_llr, uint64_t llr_floats,
                                 const gr80211_dec_desc *h_desc, int n_frames, uint8_t *d_out, int out_stride,
                                 int32_t *d_out_len, int32_t *d_n_pdu, uint8_t *d_bits, int bits_stride, void *stream);

/*
 * Synthetic transmitter: PSDU bytes -> 20-MHz baseband frames, the waveform of the reference's Python generator
 * (tools/phy80211.py genFromMpdu 80-117, genFromAmpdu 119-181, genFinalSig 1077-1094; the C++ blocks
 * encode / modulation / pad of lib/ implement the same standard clauses for the GNU Radio flowgraph).
 * Frame i is written to d_iq0[desc[i].out_off ...] (first chain) and, for two-stream frames, d_iq1 (second
 * chain; may be NULL when no frame has two streams); samples outside the frames are left untouched, so the
 * caller zeroes the arena and chooses the gaps.  d_n_samples (nullable) receives the sample count of each
 * frame, or -1 for a descriptor that was rejected (unsupported modulation or length, PSDU outside d_psdu, frame
 * outside the arena) -- such frames write nothing.  gr80211_tx_samples is the same count on the host (-1 if
 * the combination is not supported).
 */
int gr80211_tx_samples(int format, int mcs, int nss, int psdu_len);
int gr80211_tx_batch(gr80211_ctx *ctx, const uint8_t *d_psdu, uint64_t psdu_bytes, const gr80211_tx_desc *desc,
                     int n_frames, unsigned flags, void *d_iq0, void *d_iq1, uint64_t arena_len,
                     int32_t *d_n_samples, void *stream);

/*
 * Two-user MU-MIMO transmitter (tools/phy80211.py genAmpduMu 183-224, __genVhtSignalBMu 596-662, __genOfdmSignalMu
 * 815-838).  d_q: n_q sets of steering matrices, complex64 [set][FFT bin 0..63][transmit chain 0..1][stream 0..1]
 * (bfQ of the generator, in FFT-bin order); from the VHT-STF on every tone is mapped through its matrix, the legacy
 * part and VHT-SIG-A are sent unsteered on both chains.  Output and d_n_samples as for gr80211_tx_batch; both arenas
 * are required.  Both users are padded to the longer one's symbol count; a frame whose padded PSDU would exceed
 * 4098 bytes for either user (very unequal rates) is rejected.
 */
int gr80211_tx_mu_batch(gr80211_ctx *ctx, const uint8_t *d_psdu, uint64_t psdu_bytes, const gr80211_tx_mu_desc *desc,
                        int n_frames, unsigned flags, const void *d_q, int n_q, void *d_iq0, void *d_iq1,
                        uint64_t arena_len, int32_t *d_n_samples, void *stream);

/*
 * `signal` / `signal2` alone (lib/signal_impl.cc:108-202, lib/signal2_impl.cc): for every descriptor the L-SIG
 * decision and tags (d_info: status, lsig_mcs, lsig_len, nsamp, cfo_hz, snr, rssi), the L-LTF channel estimate of
 * the first antenna that travels as tag "chan" (d_chan: 64 complex64 per frame, FFT bin order), and the items the
 * block writes: nsamp samples after L-SIG with the CFO taken out, then 320 items of padding (zeros here; the
 * reference leaves them unwritten), at item out_off[i] of d_out0 (and d_out1 for the second antenna).  out_off is
 * a host array unless GR80211_FLAG_DESC_ON_DEVICE is set (then both arrays are device arrays); frames whose L-SIG
 * check fails write nothing.  A caller learns nsamp with gr80211_signal_batch first and lays the output out.
 */
int gr80211_signal_emit_batch(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                              const gr80211_frame_desc *desc, int n_frames, unsigned flags, const uint64_t *out_off,
                              void *d_out0, void *d_out1, uint64_t out_len, gr80211_frame_info *d_info, void *d_chan,
                              void *stream);

/*
 * `demod` / `demod2` alone (lib/demod_impl.cc:59-342, lib/demod2_impl.cc:58-349): frames as `signal` hands them
 * on -- items after L-SIG in d_iq0 (d_iq1), the tags in desc, the "chan" vectors in d_chan (64 complex64 per frame)
 * -- to soft bits and the tags `demod` attaches to them, exactly as gr80211_demod_batch returns them.  Tags a
 * correct `signal` cannot have produced (mcs outside 0..7, len outside 1..4095, nsamp not matching them) give
 * status GR80211_ST_DROP_LSIG.
 */
int gr80211_demod_tagged_batch(gr80211_ctx *ctx, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                               const gr80211_sig_desc *desc, int n_frames, unsigned flags, const void *d_chan,
                               gr80211_frame_info *d_info, const float **d_llr, uint64_t *llr_floats, void *stream);

/*
 * The PDU slots of a batch as one contiguous run of bytes, for the one exchange step of a multi-GPU receiver (decoded
 * MPDUs gathered on one rank): frame i's blob (d_out_len[i] bytes of slot i, the layout `decode` publishes,
 * lib/decode_impl.cc:512-516) lies at d_packed + off[i], off = exclusive prefix sum of the lengths rounded up to 16
 * bytes (the bytes between a blob's end and the boundary are zeros); frames without a PDU take no room.  The receiver
 * rebuilds off from the gathered d_out_len.  *d_total (device, required) = bytes of the run; blobs that would reach
 * past packed_cap are not written (d_total then exceeds packed_cap).  d_off (nullable, device, n entries) receives off.
 * d_out and d_packed 16-byte aligned, out_stride a multiple of 16.  The reference has no counterpart: its `decode`
 * publishes one message per MPDU to a message port (lib/decode_impl.cc:414-419).
 */
int gr80211_pack_pdus(gr80211_ctx *ctx, const uint8_t *d_out, int out_stride, const int32_t *d_out_len, int n_frames,
                      uint8_t *d_packed, uint64_t packed_cap, uint64_t *d_total, uint64_t *d_off, void *stream);

/* Device memory helpers so that a C / FFI caller needs no HIP binding of its own. */
void *gr80211_dev_alloc(gr80211_ctx *ctx, size_t bytes);
void gr80211_dev_free(gr80211_ctx *ctx, void *p);
int gr80211_copy_h2d(gr80211_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, void *stream);
int gr80211_copy_d2h(gr80211_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream);
int gr80211_copy_d2d(gr80211_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream); /* no overlap */
/*
 * The same copies without the wait at the end, and what a single host thread needs to keep several contexts' copies
 * and kernels in flight (GNU Radio gives a block one thread; no counterpart in the reference, whose blocks compute on
 * that thread).  Host memory of an asynchronous copy must be page-locked (gr80211_host_alloc, gr80211_host_register)
 * and stay untouched until the copy has run.
 *   gr80211_pull_h2d       host-to-device copy done by a kernel that reads page-locked host memory over the link (memory
 *                          from gr80211_host_alloc; fastest with both ends 16-byte aligned, any alignment works): the
 *                          host pays one kernel launch whatever the size, all pull copies of a context go to ONE stream
 *                          (the stream of its first pull; another stream is refused with GR80211_ERR_ARG: the copies
 *                          share one completion counter and flag; use one context per upload stream), and pieces below about 1 MB move faster than
 *                          by the runtime's copy engine (256 KiB: 32 GB/s against 18).  Returns the copy's number
 *                          (>= 1) or a negative code.
 *   gr80211_pull_done      1 if copy `number` has read all of its source (the source may be overwritten), 0 if not;
 *                          wait != 0: spin until it has.  Numbers arrive in order when a context's pulls all go to one
 *                          stream; the destination is ready for work enqueued on that stream after the pull, as usual.
 *   gr80211_push_d2h       device-to-host copy done by a kernel that writes page-locked host memory over the link; the
 *                          data is there when the stream has passed it (gr80211_stream_idle, a marker, gr80211_sync).
 *                          For results that follow a long kernel: the runtime's copy engines take their copies in the
 *                          order they were handed over, so a copy that waits for a decoder holds back the uploads of
 *                          other streams handed over after it (measured: every upload stood still for the length of
 *                          the decoder); a kernel does not.
 *   gr80211_stream_idle    1 if the stream has run everything enqueued on it, 0 if not (no wait).
 *   gr80211_stream_wait    work enqueued on (ctx, stream) from now on starts after what (other, other_stream) holds now.
 *   gr80211_marker_record  marks the current end of the stream's queue; returns the marker's number (>= 0).
 *   gr80211_marker_done    1 if the stream has passed marker `id`, 0 if not; wait != 0: wait for it first.  A context
 *                          keeps 256 markers: recording one more waits for the marker 256 records back.
 *   gr80211_host_register  page-locks memory the caller owns (a scheduler's stream buffer) for as long as it is
 *                          registered, so that copies from it run asynchronously at the rate of the link.
 */
int gr80211_copy_h2d_async(gr80211_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, void *stream);
long long gr80211_pull_h2d(gr80211_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, void *stream);
int gr80211_pull_done(gr80211_ctx *ctx, long long number, int wait);
int gr80211_push_d2h(gr80211_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream);
int gr80211_copy_d2h_async(gr80211_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream);
int gr80211_copy_d2d_async(gr80211_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream); /* no overlap */
int gr80211_stream_idle(gr80211_ctx *ctx, void *stream);
int gr80211_stream_wait(gr80211_ctx *ctx, void *stream, gr80211_ctx *other, void *other_stream);
long long gr80211_marker_record(gr80211_ctx *ctx, void *stream);
int gr80211_marker_done(gr80211_ctx *ctx, long long id, int wait);
int gr80211_host_register(gr80211_ctx *ctx, void *p, size_t bytes);
int gr80211_host_unregister(gr80211_ctx *ctx, void *p);

/*
 * Page-locked host memory: copies to and from it run at the full rate of the link, copies from ordinary memory at
 * a fraction of it.  NULL when the allocation fails.
 */
void *gr80211_host_alloc(gr80211_ctx *ctx, size_t bytes);
void gr80211_host_free(gr80211_ctx *ctx, void *p);

/*
 * Timing of the last batch call, measured with HIP events on the stream the kernels ran on.
 * ms[0]=preamble, [1]=schedule, [2]=demod, [3]=decode (Viterbi forward pass + traceback + assembly, one kernel),
 * [4]=0, [5]=whole call.  Batches of at most 512 decoder groups (about 33 000 frames of one code rate) run the
 * forward pass with four wavefronts per group and the traceback as a second kernel: [3] is then the forward pass
 * and [4] the traceback + assembly; up to 1024 groups two wavefronts share a group, beyond that one wavefront runs
 * it (the environment variables GR80211_GROUP_WAVES=1|2|4, or 10 for the one-wavefront form of the largest batches
 * at any size, and GR80211_SPLIT_DECODE=1 force a form for experiments).
 * Only recorded when profiling is switched on (costs one stream synchronisation per call).
 */
int gr80211_set_profiling(gr80211_ctx *ctx, int on);
int gr80211_last_timing(gr80211_ctx *ctx, float *ms, int n);

/* Library self-description: build arch and table checksum (for load tests without a GPU). */
const char *gr80211_version(void);
uint32_t gr80211_table_checksum(void);

#ifdef __cplusplus
}
#endif
#endif /* GR80211_RX_H */
