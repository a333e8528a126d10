/*
 * Streaming C-ABI of the host blocks (libgr80211_host.so): one entry that mirrors gr::block::general_work, with
 * stream tags as plain records and published messages through a callback, so that a GNU Radio block wrapper (or
 * any FFI caller) only translates pmt <-> these records.  Block kinds and what they stand for:
 *
 *   GR80211_BLOCK_DECODE    ieee80211::decode (include/gnuradio/ieee80211/decode.h:22-36, lib/decode_impl.cc:60-162):
 *                           in {float soft bits}; tags format, len, total, cr, mcs, ampdu, trellis (and mu2x1chan)
 *                           on a packet's first item (lib/decode_impl.cc:70-90)
 *   GR80211_BLOCK_RX        ieee80211::signal -> demod -> decode as one block: in {u8 sync, c64 samples}
 *                           (lib/signal_impl.cc:35-37); tags rad, snr, rssi on a set sync byte (lib/signal_impl.cc:83-92)
 *   GR80211_BLOCK_RX2       ieee80211::signal2 -> demod2 -> decode: in {u8 sync, c64, c64} (lib/signal2_impl.cc:35-37)
 *   GR80211_BLOCK_CAPTURE   presiso -> trigger -> sync -> signal -> demod -> decode (examples/rx.grc): in {c64}
 *   GR80211_BLOCK_CAPTURE2  the same with two antennas (examples/rx2.grc): in {c64, c64}
 *   GR80211_BLOCK_SIGNAL    ieee80211::signal alone (lib/signal_impl.cc:34-206): in {u8 sync, c64}, out {c64} with the
 *                           tags cfo, snr, rssi, seq, mcs, len, nsamp, chan on a frame's first item
 *   GR80211_BLOCK_SIGNAL2   ieee80211::signal2 (lib/signal2_impl.cc): in {u8 sync, c64, c64}, out {c64, c64}
 *   GR80211_BLOCK_DEMOD     ieee80211::demod alone (lib/demod_impl.cc:59-342): in {c64} with those tags, out {f32 soft
 *                           bits} with the tags decode reads (plus cfo, snr, rssi, sssnr0)
 *   GR80211_BLOCK_DEMOD2    ieee80211::demod2 (lib/demod2_impl.cc:58-349): in {c64, c64}, out {f32}
 *
 * The first five kinds have no stream output; messages of port "out" are the blobs `decode` publishes
 * (lib/decode_impl.cc:512-516).  The others are driven with gr80211_block_work_out.  A block is driven by one
 * thread at a time, like a GNU Radio block.
 */
#ifndef GR80211_HOST_H
#define GR80211_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR80211_BLOCK_DECODE 0
#define GR80211_BLOCK_RX 1
#define GR80211_BLOCK_RX2 2
#define GR80211_BLOCK_CAPTURE 3
#define GR80211_BLOCK_CAPTURE2 4
#define GR80211_BLOCK_SIGNAL 5
#define GR80211_BLOCK_SIGNAL2 6
#define GR80211_BLOCK_DEMOD 7
#define GR80211_BLOCK_DEMOD2 8

/* tag keys (the pmt symbols of the reference's tags) */
#define GR80211_TAG_RAD 0
#define GR80211_TAG_SNR 1
#define GR80211_TAG_RSSI 2
#define GR80211_TAG_FORMAT 3
#define GR80211_TAG_LEN 4
#define GR80211_TAG_TOTAL 5
#define GR80211_TAG_CR 6
#define GR80211_TAG_MCS 7
#define GR80211_TAG_AMPDU 8
#define GR80211_TAG_TRELLIS 9
#define GR80211_TAG_MU2X1CHAN 10 /* c32vector of 128: cvec points at 256 floats, re/im interleaved */
#define GR80211_TAG_CFO 11
#define GR80211_TAG_SEQ 12
#define GR80211_TAG_NSAMP 13
#define GR80211_TAG_CHAN 14 /* c32vector of 64 */
#define GR80211_TAG_SSSNR0 15
#define GR80211_TAG_SSSNR1 16

typedef struct {
    uint64_t offset; /* absolute item index on the port, like gr::tag_t::offset */
    uint32_t port;   /* input port the tag sits on */
    uint32_t key;    /* GR80211_TAG_* */
    double value;    /* pmt long / float value */
    const float *cvec; /* vector value (GR80211_TAG_MU2X1CHAN) or NULL */
    uint32_t cvec_len; /* complex elements in cvec */
    uint32_t reserved;
} gr80211_tag;

#define GR80211_OPT_MU_POS 0       /* ieee80211::demod::make(mupos, mugid): position in VHT MU groups, 0 or 1 */
#define GR80211_OPT_AMPDU_STRICT 1 /* 1: the standard's A-MPDU de-aggregation instead of the reference's */
#define GR80211_OPT_DEBUG 2        /* ieee80211::decode::make(ifdebug) (DECODE only): 1 = one line per packet on stdout with the
                                      running per-MCS counts of packets whose FCS held, in the reference's wording
                                      (lib/decode_impl.cc:375-411,453-509: "ieee80211 decode, vht crc32 correct, total:...") */

#define GR80211_OPT_FINISH_IN_CALL 3 /* DECODE, DEMOD(2): 1 = what a call completes leaves in that call -- the packets (frames) whose last
                                      item the window brings form a round of their own, whatever batch_frames says, and the call
                                      waits for it; the reference's timing (lib/decode_impl.cc:128-160 publishes within the call,
                                      lib/demod_impl.cc:279-314 hands soft bits on symbol by symbol).  For live reception, where
                                      the next call may be long in coming; 0 = rounds of batch_frames, nothing waited for, for
                                      streams played at full speed.  Default: 1 for blocks created with batch_frames <= 0, else 0.
                                      DEMOD: set before the first window.  The other kinds take continuous input and publish
                                      within a window or two whatever this says. */
#define GR80211_OPT_MAX_HOLD 4       /* items a round holds at most before it is cut whatever the number of frames in it: soft bits for
                                      DECODE (at least 4096), samples from the first pending sync byte on for RX, SIGNAL and CAPTURE
                                      (at least four times the longest frame, 438 016 samples; CAPTURE cuts its rounds by this
                                      alone).  Bounds the time a frame waits for its round on a channel with little traffic. */
#define GR80211_OPT_MIN_ROUND 5      /* CAPTURE(2), live reception: a round may go once it holds this many samples the round before did
                                      not (at least 8192) while fewer than two rounds are on their way, instead of waiting for
                                      GR80211_OPT_MAX_HOLD samples; 0 (the default): off. */

typedef struct gr80211_block gr80211_block;
typedef void (*gr80211_msg_fn)(void *user, const uint8_t *blob, int len);

/* batch_frames: frames (packets) a round collects before it goes to the GPU -- the knob between latency and rate.
   <= 0, the default, is for live reception: a block made the way the reference makes its own never sits on a frame while
   the channel is quiet.  RX and SIGNAL then size their rounds by the traffic -- a frame goes at once while fewer than two rounds
   are on their way, and what arrives while two are goes together in the next (a message leaves two windows after the
   frame's last sample on a quiet channel; a recording played at full speed gets rounds of hundreds) --, DECODE and DEMOD(2) keep the reference's timing (GR80211_OPT_FINISH_IN_CALL = 1).  A stream
   played at full speed wants thousands (the rates quoted for these blocks are at 4096 and 16 384).  CAPTURE(2) cuts
   its rounds by samples (GR80211_OPT_MAX_HOLD) and ignores this. */
gr80211_block *gr80211_block_create(int kind, int batch_frames, int device);
void gr80211_block_destroy(gr80211_block *b);
void gr80211_block_set_msg_handler(gr80211_block *b, gr80211_msg_fn fn, void *user);
int gr80211_block_set_option(gr80211_block *b, int option, int value);
int gr80211_block_n_inputs(const gr80211_block *b);
/* gr::block::forecast: items needed on every input for noutput_items of progress. */
int gr80211_block_forecast(gr80211_block *b, int noutput_items, int *ninput_items_required, int n_inputs);
/*
 * gr::block::general_work.  ninput_items / input_items: one entry per input port; tags: the tags inside the
 * offered windows (in any order, cheapest by non-decreasing offset; all of them with every call, as get_tags_in_range over the
 * window returns them: the block keeps no tag between calls).  *consumed receives what consume_each() was called with.
 * Returns the items produced (always 0: there is no stream output) or a negative value on error.
 */
int gr80211_block_work(gr80211_block *b, int noutput_items, const int *ninput_items, const void *const *input_items,
                       int n_inputs, const gr80211_tag *tags, int n_tags, int *consumed);
/*
 * general_work of a block with stream outputs.  output_items: one buffer per output port with room for
 * noutput_items items; tags_out receives the tags of the items produced (their cvec pointers stay valid until the
 * block's next call).  A frame carries at most 16 tags and no fewer than 48 items, so cap_tags_out =
 * 16 * (noutput_items / 48 + 1) always suffices; tags that do not fit are an error (-3), not dropped silently.
 * Returns the items produced on every output port.  A block may hold
 * produced items back until there is room: gr80211_block_pending_output tells how many, and a call without
 * input (ninput_items all 0) fetches them.
 */
int gr80211_block_n_outputs(const gr80211_block *b);
int gr80211_block_work_out(gr80211_block *b, int noutput_items, const int *ninput_items, const void *const *input_items,
                           int n_inputs, const gr80211_tag *tags, int n_tags, void *const *output_items, int n_outputs,
                           gr80211_tag *tags_out, int cap_tags_out, int *n_tags_out, int *consumed);
long long gr80211_block_pending_output(const gr80211_block *b);
/*
 * Page-locked input buffers.  Windows that lie in memory from gr80211_block_alloc_input travel to the GPU
 * asynchronously at the rate of the link, and a call then consumes the items whose copies have run, which may be
 * fewer than it was offered (never none); windows in other memory are copied before the call returns.  A GNU Radio
 * 3.10 wrapper hands the memory out through a custom buffer type of the block's input ports; the scheduler of the
 * reference's GNU Radio allocates its own buffers, which gr80211_block_pin_input page-locks in place where the system
 * allows it (0 on success; ordinary copies are used otherwise).  Buffers are freed with the block at the latest.
 */
void *gr80211_block_alloc_input(gr80211_block *b, uint64_t bytes);
void gr80211_block_free_input(gr80211_block *b, void *p);
int gr80211_block_pin_input(gr80211_block *b, const void *p, uint64_t bytes);
/*
 * Page-locked output buffers.  Tells a block with stream outputs (SIGNAL, SIGNAL2, DEMOD, DEMOD2) that [p, p + bytes) --
 * typically the input buffer of the block downstream, from that block's gr80211_block_alloc_input -- is page-locked.
 * Into such a buffer, when it takes a megabyte or more at a time, produced items travel from the GPU directly: a call
 * sends the next stretch on its way behind the items it reports (inside the noutput_items it was offered, beyond what
 * it returns) and reports the stretch when called again, at the position the scheduler's write pointer has then
 * reached (lib/signal_impl.cc:164-202 and lib/demod_impl.cc:279-330 write their items inside general_work instead).
 * The caller must therefore offer output space that does not shrink or move by more than the items reported; a ring
 * buffer that wraps is handled (the stretch is moved).  0 on success.
 */
int gr80211_block_declare_pinned(gr80211_block *b, const void *p, uint64_t bytes);
/*
 * The UDP hand-off of the reference flowgraphs: decode's "out" port feeds a network.socket_pdu block of type
 * UDP_CLIENT (examples/cmu_v3/trxSta0.grc:301-313,1054; 127.0.0.1:9527), which sends every PDU's blob --
 * [format][len lo][len hi][payload][mcs] -- as one datagram.  gr80211_block_send_udp makes the block do the same
 * with every message it publishes, besides calling the handler (host == NULL: stop); 0 on success.
 * gr80211_udp_* is the sender on its own, for a caller that publishes blobs itself.
 */
int gr80211_block_send_udp(gr80211_block *b, const char *host, int port);
typedef struct gr80211_udp gr80211_udp;
gr80211_udp *gr80211_udp_open(const char *host, int port);
int gr80211_udp_send(gr80211_udp *u, const uint8_t *blob, int len); /* 0 when the datagram has left */
void gr80211_udp_close(gr80211_udp *u);
/* End of stream: decode what is buffered.  0 on success. */
int gr80211_block_flush(gr80211_block *b);
/* Text of the block's last error; with b == NULL, why the calling thread's last gr80211_block_create returned NULL. */
const char *gr80211_block_last_error(const gr80211_block *b);

#ifdef __cplusplus
}
#endif
#endif /* GR80211_HOST_H */
