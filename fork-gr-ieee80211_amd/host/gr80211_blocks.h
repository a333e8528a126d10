// Host side of the MI355X RX path: C++ blocks with the work()/forecast()/tag/message interface of the
// reference's GNU Radio blocks, calling the C-ABI of libgr80211_rx.so underneath.
//
//   gr80211::host::decode  -- drop-in for ieee80211::decode  (include/gnuradio/ieee80211/decode.h:22-36,
//                             lib/decode_impl.cc:33-162): float soft-bit stream + tags in, PDU messages out.
//   gr80211::host::rx      -- signal(2) -> demod(2) -> decode fused into one block with the input contract of
//                             ieee80211::signal/signal2 (lib/signal_impl.cc:34-106, lib/signal2_impl.cc:34-110:
//                             sync byte stream with tags rad/snr/rssi + one or two complex sample streams) and
//                             the output contract of ieee80211::decode (message port "out").
//   gr80211::host::signal  -- drop-in for ieee80211::signal / signal2 (lib/signal_impl.cc, lib/signal2_impl.cc):
//                             the same inputs, the frames' items after L-SIG with the tags cfo, snr, rssi, seq, mcs,
//                             len, nsamp, chan out.
//   gr80211::host::demod   -- drop-in for ieee80211::demod / demod2 (lib/demod_impl.cc, lib/demod2_impl.cc): those
//                             items and tags in, soft bits with the tags decode reads out.
//
// GNU Radio itself is not a dependency: `block` below carries the few scheduler services those blocks use
// (nitems_read, get_tags_in_range, consume_each, message_port_pub), so a gr::block wrapper only forwards.
// Blocks collect whole frames and hand them to the GPU in batches; a batch is cut when `batch_frames` frames
// are pending, when flush() is called, or when the sample hold reaches its limit.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <complex>
#include <cstdint>
#include <deque>
#include <functional>
#include <atomic>
#include <memory>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gr80211_rx.h"

namespace gr80211 {
namespace host {

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

// Stream tag with the value kinds of the RX chain's tags (pmt long / float / c32vector).
struct tag_t {
    uint64_t offset = 0;
    std::string key;
    double value = 0.0;
    std::vector<gr_complex> cvec;
};

// Message of port "out": pmt::cons(dict{"len": len}, blob)  (lib/decode_impl.cc:512-516).
struct pdu_t {
    long len = 0;
    std::vector<uint8_t> blob;
};

class block
{
public:
    virtual ~block();
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required) = 0;
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items) = 0;
    // End of stream: decode whatever is buffered.  Returns false on a GPU / API error (see last_error()).
    virtual bool flush() { return true; }

    // ---- scheduler side
    void add_item_tag_in(int port, const tag_t &t);  // any order; cheapest in non-decreasing offset order
    void add_item_tag_in(int port, tag_t &&t);
    void clear_item_tags_in() { for (auto &tv : d_tags) tv.clear(); }
    void set_msg_handler(std::function<void(const std::string &port, const pdu_t &msg)> f) { d_handler = std::move(f); }
    uint64_t nitems_read(int port) const { return d_read[port]; }
    uint64_t nitems_written(int port) const { return d_written[port]; }
    int n_outputs() const { return (int)d_written.size(); }
    int consumed() const { return d_consumed; }  // items consumed by the last general_work call (every port)
    // advance nitems_read by consumed() and nitems_written by `produced` (what general_work returned), forget
    // passed tags
    void commit(int produced = 0);
    // tags general_work attached to the items it has just produced on `port` (absolute offsets); the scheduler
    // takes them with every call
    std::vector<tag_t> take_output_tags(int port);
    const std::string &last_error() const { return d_err; }
    // Page-locked stream buffers.  Windows that lie in page-locked memory travel to the GPU by asynchronous DMA at the
    // rate of the link instead of through the runtime's staging copies; their items are consumed once their copy has
    // run, so a call may consume fewer items than it was offered (it always consumes some).  Two ways to get there:
    // alloc_input hands out such a buffer for the scheduler to use as the block's input buffer (what a GNU Radio 3.10
    // custom buffer type does; freed by free_input or with the block), pin_input page-locks a buffer the scheduler
    // already owns (not every system lets user pages be locked: false then, and ordinary copies are used).
    void *alloc_input(size_t bytes);
    void free_input(void *p);
    bool pin_input(const void *p, size_t bytes);
    // a buffer somebody else made page-locked (e.g. the input buffer of the block downstream, which is this block's
    // output buffer): items may then travel to it straight from the GPU
    void declare_pinned(const void *p, size_t bytes) { if (p && bytes) d_pinned.emplace_back(static_cast<const char *>(p), bytes); }

protected:
    explicit block(int n_in, int n_out = 0);
    virtual gr80211_ctx *pin_context() { return nullptr; }  // the context page-locking goes through
    bool is_pinned(const void *p, size_t bytes) const;
    void unpin_all();
    void consume_each(int n) { d_consumed = n; }
    void add_item_tag(int port, const tag_t &t) { d_out_tags[port].push_back(t); }
    void get_tags_in_range(std::vector<tag_t> &v, int port, uint64_t start, uint64_t end) const;
    // The values of the tags `keys` on item `offset` of `port` (the last tag of a key wins, as pmt::dict_add overwrites)
    // without copying tags; false if the item carries no tag at all.  out[k] keeps its value where key k is missing.
    bool tag_values_at(int port, uint64_t offset, const char *const *keys, double *out, int n_keys) const;
    // offset of the first tag of `port` at or after `from`, or `end` if there is none before it
    uint64_t next_tag_offset(int port, uint64_t from, uint64_t end) const;
    void message_port_pub(const std::string &port, const pdu_t &m);
    // Publishes the messages n frame slots hold (stride bytes apart, len[i] bytes used); returns how many.
    uint64_t publish_slots(const uint8_t *out, const int32_t *len, int n, int stride);
    std::string d_err;

private:
    std::vector<uint64_t> d_read, d_written;
    std::vector<std::vector<tag_t>> d_out_tags;
    std::vector<std::deque<tag_t>> d_tags;  // per port, sorted by offset; passed tags leave at the front
    int d_consumed = 0;
    std::function<void(const std::string &, const pdu_t &)> d_handler;
    pdu_t d_msg;  // publish_slots builds every message here: no allocation per message once it has grown
    std::vector<std::pair<const char *, size_t>> d_pinned;  // ranges known to be page-locked
    std::vector<void *> d_owned;                            // ... of which alloc_input made these
    std::vector<const char *> d_registered;                 // ... and pin_input locked these
};

// Owns a gr80211 context and a few growable device buffers.
class gpu_session
{
public:
    explicit gpu_session(int device);
    ~gpu_session();
    gpu_session(const gpu_session &) = delete;
    gpu_session &operator=(const gpu_session &) = delete;
    bool ok() const { return d_ctx != nullptr; }
    // A-MPDU rule of every batch this session decodes: false = the reference's, true = the standard's
    // (gr80211_set_ampdu_strict)
    void set_ampdu_strict(bool on) { if (d_ctx) gr80211_set_ampdu_strict(d_ctx, on ? 1 : 0); }
    gr80211_ctx *ctx() const { return d_ctx; }
    void *buffer(int slot, size_t bytes);  // device buffer `slot` of at least `bytes` (contents not preserved)
    void *staging(int slot, size_t bytes);  // page-locked host buffer `slot`, same rules
    bool idle() const { return d_ctx && gr80211_stream_idle(d_ctx, nullptr) == 1; }   // the context's stream has run dry
    bool sync() const { return d_ctx && gr80211_sync(d_ctx, nullptr) == GR80211_OK; }
    std::string error() const;

private:
    gr80211_ctx *d_ctx = nullptr;
    struct dbuf {
        void *p = nullptr;
        size_t cap = 0;
    };
    dbuf d_buf[8];
    dbuf d_stage[6];
};

// Input items whose copies to the GPU are still on their way: general_work consumes a window's items only once their
// copy has run (the scheduler may overwrite consumed items at once).  Entries are in stream order.
class upload_log
{
public:
    uint64_t enqueued() const { return d_enq; }  // absolute index up to which input has been taken
    void set_enqueued(uint64_t v) { d_enq = v; }
    // items up to `upto` are free once copy `id` of `g` has read its source: a gr80211_pull_h2d number (pull) or a
    // marker behind a runtime copy; g == nullptr: at once
    void add(gpu_session *g, long long id, uint64_t upto, bool pull = false) { d_q.push_back({g, id, upto, pull}); }
    // absolute index up to which items may be consumed, at least `from`; waits for the oldest copy if nothing beyond
    // `from` is free yet and something is on its way
    uint64_t settled(uint64_t from, bool may_wait = true);  // may_wait = false: only what has arrived already
    bool drain();  // waits for every copy
    bool empty() const { return d_q.empty(); }

private:
    struct entry {
        gpu_session *g;
        long long id;
        uint64_t upto;
        bool pull;
    };
    std::deque<entry> d_q;
    uint64_t d_enq = 0;
};

// The way of a block's input windows to the GPU.  A window in page-locked memory is cut into pieces that travel by a
// stream of their own, so that none queues behind a decoder; every piece has an entry in the log, and general_work
// consumes the items whose pieces have arrived.  Pieces below a megabyte go by gr80211_pull_h2d, larger ones by the
// runtime's copy engines.  Windows in ordinary memory are copied before send() returns, by the context the caller
// names.  One upload stream and one stream per round in flight is what a process has hardware queues for (four, of
// which the runtime keeps one for the default stream): streams that share a queue run one after the other, and a
// decoder ahead in the queue holds up whatever follows.  GR80211_UPLOAD_STREAMS=2 sends every other piece by a second
// stream (for a process that has been given more queues).
class uploader
{
public:
    explicit uploader(int device);
    ~uploader();
    uploader(const uploader &) = delete;
    uploader &operator=(const uploader &) = delete;
    bool ok() const { return d_a.ok() && (!d_b || d_b->ok()); }
    gpu_session &session() { return d_a; }
    upload_log &log() { return d_log; }
    // `items` items of `item` bytes from each of n_src sources (parallel streams: item k of all of them has stream
    // position pos0 + k) to dst[0..n_src).  false on an error (text in err).
    bool send(gpu_session &direct, int n_src, const char *const *src, char *const *dst, size_t items, size_t item,
              uint64_t pos0, bool pinned, std::string &err);
    // what c's stream is given from now on runs after every piece sent so far
    bool join(gr80211_ctx *c);
    bool sync() { return flush_staged() && d_a.sync() && (!d_b || d_b->sync()); }  // every piece sent so far has arrived
    void set_first_split(int pieces) { d_firstSplit = pieces; }
    // page-locked windows of up to `bytes` are copied by hand too (see below); the environment's limit, if set, stays
    void set_stage_max_pinned(size_t bytes) { if (!getenv("GR80211_STAGE_MAX_PINNED")) d_stageMaxPinned = bytes; }

private:
    gpu_session d_a;
    std::unique_ptr<gpu_session> d_b;  // the second stream, if asked for
    upload_log d_log;
    unsigned d_nPiece = 0;  // pieces since the last join
    int d_firstSplit = 0;   // pieces a window that finds nothing on its way is cut into (0: by its size)
    // Small windows (GNU Radio's default buffers hold 64 KB): a copy of their own costs a launch and, for the caller,
    // the wait until it has read its source.  They are copied by hand into page-locked memory of the uploader instead
    // -- their items are free at once -- and travel from there in pieces of STAGE_FLUSH bytes, as long as they follow
    // each other in device memory.
    // (from page-locked windows up to 96 KB: beyond that the copy that reads the window itself is the cheaper way; from
    // ordinary memory up to a megabyte, against a copy the caller has to wait for)
    static constexpr size_t STAGE_MAX_PINNED = (size_t)96 << 10, STAGE_MAX_PAGEABLE = (size_t)1 << 20,
                            STAGE_FLUSH = (size_t)256 << 10, STAGE_BYTES = (size_t)8 << 20;
    bool stage(int n_src, const char *const *src, char *const *dst, size_t bytes, std::string &err);
    bool flush_staged();
    char *d_stage[2] = {nullptr, nullptr};  // one ring per source
    size_t d_wpos = 0;                      // where the next bytes go
    size_t d_pendStart = 0, d_pendBytes = 0;  // copied in, not yet sent
    char *d_pendDst[2] = {nullptr, nullptr};
    int d_pendSrcs = 0;
    struct sent { size_t lo, hi; long long id; };  // ring bytes [lo, hi) are free once pull `id` has run
    std::deque<sent> d_sentRing;
    bool d_stageOn = true;
    size_t d_stageMaxPinned = STAGE_MAX_PINNED;  // GR80211_STAGE_MAX_PINNED=<bytes>: another limit (experiments)
    size_t d_stageBytes = STAGE_BYTES;           // GR80211_STAGE_BYTES=<bytes>: another ring size (tests: a ring that wraps at once)
};

// Items waiting for the GPU, kept in device memory: every window of the stream is uploaded when it arrives, so the
// host holds no copy.  Append at the back, release at the front.
class device_fifo
{
public:
    device_fifo(gpu_session &g, size_t item_bytes) : d_g(g), d_item(item_bytes) {}
    ~device_fifo();
    device_fifo(const device_fifo &) = delete;
    device_fifo &operator=(const device_fifo &) = delete;
    size_t size() const { return d_n; }
    bool empty() const { return d_n == 0; }
    const void *data() const { return d_p; }    // device pointer; 64 spare items follow the last one
    bool append(const void *host_src, size_t items);  // false on a GPU error (see gpu_session::error())
    bool release_front(size_t items);
    void truncate(size_t items) { if (items < d_n) d_n = items; }
    void clear() { d_n = 0; }

private:
    bool reserve(size_t items);
    gpu_session &d_g;
    size_t d_item;
    char *d_p = nullptr, *d_alt = nullptr;  // d_alt: same capacity, target of release_front's move (made on demand)
    size_t d_n = 0, d_cap = 0;              // in items
};

// Items a block has produced and the scheduler has not yet made room for, with their tags (offsets count the
// items ever pushed).  Items are copied in (push) or stay where they are (push_ref: memory the caller keeps
// untouched until the items have left, e.g. the page-locked area a batch's results were copied to; `left`, if
// given, counts the referenced items that have not left yet).
class out_stream
{
public:
    explicit out_stream(size_t item_bytes) : d_item(item_bytes) {}
    uint64_t pushed() const { return d_pushed; }
    size_t pending() const { return (size_t)(d_pushed - d_popped); }
    void push(const void *src, size_t items);
    void push_zeros(size_t items);
    void push_ref(const void *src, size_t items, size_t *left = nullptr);
    // items that stay in device memory of `g`'s context until they leave (kept untouched by the caller like push_ref's)
    void push_dev(gpu_session *g, const void *dev_src, size_t items, size_t *left = nullptr);
    // referenced items that lie in [lo, hi) are copied, so that the memory may be written again
    void release_refs(const void *lo, const void *hi);
    void tag(const tag_t &t) { d_tags.push_back(t); }  // t.offset = index among the pushed items
    // up to max_items into dst; their tags into `out`, re-based to the port's absolute offset `written` (items in
    // device memory are copied and waited for)
    int pop(void *dst, int max_items, uint64_t written, std::vector<tag_t> &out);
    // The same without the wait, for a dst in page-locked memory: items in device memory are sent on their way by the
    // stream of the context that holds them (contiguous runs as one copy), the others are copied at once.  `f` says
    // what to wait for (arrived) and what to do then (settle); until then dst belongs to the copies.
    struct flight {
        void *dst = nullptr;
        size_t items = 0;
        std::vector<std::pair<gpu_session *, long long>> markers;  // passed = the copies have landed
        std::vector<std::pair<size_t *, size_t>> lefts;            // referenced-item counters to lower afterwards
        std::vector<tag_t> tags;                                   // offsets relative to the first item
        bool arrived(bool wait);
        void settle();  // lowers the counters; the items may be reported
    };
    bool pop_launch(void *dst, int max_items, flight &f, std::string &err);

private:
    struct seg {
        const uint8_t *ext = nullptr;  // referenced items (nullptr: own holds them)
        gpu_session *dev = nullptr;    // ext is an address in this context's device memory
        std::vector<uint8_t> own;
        size_t bytes = 0, head = 0;    // size and what has left
        size_t *left = nullptr;
    };
    size_t d_item;
    std::deque<seg> d_segs;
    uint64_t d_pushed = 0, d_popped = 0;
    std::deque<tag_t> d_tags;
};

// Produced items on their way from device memory into output buffers in page-locked memory (one or two ports that
// always carry the same item counts).  A call reports the stretches that have arrived and sends the next one on its
// way behind those still travelling; a stretch lies where the scheduler's write pointer will be once everything before
// it has been reported (after the buffer wrapped: where it would have been, and is moved).  Up to two stretches travel
// at a time, so that the link is busy while the scheduler takes the one before.
constexpr size_t DIRECT_OUT_MIN_BYTES = (size_t)1 << 20;  // output windows from which on items travel directly
class direct_out
{
public:
    size_t in_flight() const { return d_items; }
    // q / dst / written: per port; tag(port, t): the block's add_item_tag.  must_progress: the call has taken no input,
    // so it waits for the oldest stretch rather than report nothing.  Returns items produced (every port the same), -1
    // on an error.
    int work(out_stream *const *q, void *const *dst, const uint64_t *written, int n_ports, size_t item, int noutput_items,
             bool must_progress, const std::function<void(int, const tag_t &)> &tag, std::string &err);

private:
    struct stretch {
        out_stream::flight f[2];
        size_t items = 0;
    };
    std::deque<stretch> d_sent;
    size_t d_items = 0;
};

// What the reference flowgraphs connect to decode's "out" port: a network.socket_pdu block of type UDP_CLIENT
// (examples/cmu_v3/trxSta0.grc:301-313,1054: host 127.0.0.1, port 9527, MTU 65535), which sends the blob of every
// PDU -- [format][len lo][len hi][payload][mcs] -- as one datagram.  Connect it with
//     b->set_msg_handler([&](const std::string &, const pdu_t &m) { sink.send(m); });
class udp_sink
{
public:
    udp_sink() = default;
    ~udp_sink();
    udp_sink(const udp_sink &) = delete;
    udp_sink &operator=(const udp_sink &) = delete;
    // host: dotted IPv4 address or a name the resolver knows; false if the socket cannot be made
    bool open(const std::string &host, int port);
    bool is_open() const { return d_fd >= 0; }
    // one datagram per message; blobs beyond the MTU are cut to it, as socket_pdu does.  false if the send failed
    bool send(const pdu_t &m);
    uint64_t datagrams_sent() const { return d_nSent; }
    static constexpr size_t MTU = 65507;  // the longest UDP payload IPv4 carries (socket_pdu's 65535 less the headers)

private:
    int d_fd = -1;
    uint64_t d_nSent = 0;
};

class decode : public block
{
public:
    typedef std::shared_ptr<decode> sptr;
    // ifdebug as in ieee80211::decode::make(bool) (the reference only prints with it); batch_frames and
    // device are the GPU side's knobs and default to values that keep a stream decoder busy.
    static sptr make(bool ifdebug = false, int batch_frames = 0, int device = 0);
    ~decode() override;

    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
    bool flush() override;

    void set_ampdu_strict(bool on);
    // the reference's ifdebug after construction, and where its lines go (stdout by default)
    void set_debug(bool on) { d_debug = on; }
    void set_debug_output(FILE *f) { d_debugOut = f; }
    // The reference publishes a packet in the call that brings its last soft bit (lib/decode_impl.cc:128-160).  With
    // this on, the block does too: the packets a call completes form a round of their own, whatever batch_frames says,
    // and the call waits for it.  For live reception, where the next call may be a long time coming; a stream played
    // at full speed does better without.  On by itself when the block was made without a batch size.
    void set_finish_in_call(bool on) { d_finish = on; }
    bool finish_in_call() const { return d_finish; }
    uint64_t packets_correct() const { return d_nPktCorrect; }
    // soft bits a round holds at most before it is cut (at the next packet boundary) whatever the packet count
    void set_max_hold(size_t items) { d_maxHold = items < 4096 ? 4096 : items; }
    uint64_t packets_seen() const { return d_nSeen; }
    uint64_t pdus_published() const { return d_nPdu; }
    uint64_t rounds_cut() const { return d_nRounds; }

protected:
    gr80211_ctx *pin_context() override { return d_up.session().ctx(); }

private:
    decode(bool ifdebug, int batch_frames, int device);
    // One stretch of the stream on its way through the GPU: its soft bits as they came (packets back to back, whatever
    // lies between them included), a descriptor per packet.  While one round is filled, the ones before it decode on
    // the streams of their own contexts and are published in order.
    struct debug_tags { float cfo, snr, rssi, sssnr0, sssnr1; };
    struct round {
        std::unique_ptr<gpu_session> gpu;
        float *llr = nullptr;              // device arena
        size_t cap = 0, n = 0;             // soft bits it can hold / holds
        uint64_t base = 0;                 // absolute index of item 0
        std::vector<gr80211_dec_desc> desc;
        std::vector<debug_tags> dbg;       // with ifdebug: the tags the debug lines quote, one per packet
        std::vector<std::pair<size_t, pdu_t>> ready;  // channel messages of sounding packets and how many packets of
                                                      // the round precede each
        enum { FREE, FILL, DECODE } state = FREE;
    };
    bool reserve(round &r, size_t items);
    bool send(const float *in, int from, int to);  // window items [from, to) to the round being filled
    bool cut();                                    // the round being filled goes to the decoder; another one is filled
    bool advance(bool wait_front, bool wait_all);
    void publish_ready(round &r, size_t upto_desc, size_t &at);
    void debug_lines(const gr80211_dec_desc &d, const debug_tags &t, uint64_t n_pub);

    bool d_debug;
    FILE *d_debugOut = nullptr;  // where the debug lines go (stdout unless set_debug_output was called)
    uint64_t d_nPktCorrect = 0, d_legacyMcsCount[8] = {}, d_htMcsCount[8] = {}, d_vhtMcsCount[10] = {};
    int d_batch;
    uploader d_up;
    static constexpr int N_ROUNDS = 3;
    round d_round[N_ROUNDS];
    round *d_fill = nullptr;
    std::deque<round *> d_flight;
    size_t d_maxHold = (size_t)1 << 25;
    bool d_strict = false;
    bool d_finish = false;
    enum { S_IDLE, S_COLLECT, S_CLEAN } d_state = S_IDLE;
    // tags of the packet being collected (lib/decode_impl.cc:77-88)
    long t_format = 0, t_len = 0, t_nTotal = 0, t_cr = 0, t_mcs = 0, t_ampdu = 0, v_trellis = 0;
    long t_nProcd = 0;
    uint64_t d_nSeen = 0, d_nPdu = 0, d_nRounds = 0, d_nCalls = 0;
    std::chrono::steady_clock::time_point d_lastPoll{};  // when the rounds in flight were last asked after
};

class rx : public block
{
public:
    typedef std::shared_ptr<rx> sptr;
    // n_rx = 1: inputs {u8 sync, c64} like ieee80211::signal; n_rx = 2: {u8 sync, c64, c64} like signal2.
    static sptr make(int n_rx = 1, int batch_frames = 0, int device = 0);  // no batch size: rounds sized by the traffic
    ~rx() override;

    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
    bool flush() override;

    // this station's position in VHT MU groups: the mupos argument of ieee80211::demod::make (lib/demod_impl.cc:28-39)
    void set_mu_pos(int pos) { d_muPos = pos ? 1 : 0; }
    void set_ampdu_strict(bool on);
    // samples a round holds at most before it is cut whatever the trigger count (at least four frame spans)
    void set_max_hold(size_t items);
    uint64_t triggers_seen() const { return d_nTrig; }
    uint64_t frames_decoded() const { return d_nFrames; }
    uint64_t pdus_published() const { return d_nPdu; }
    uint64_t rounds_cut() const { return d_nRounds; }

protected:
    // sync_port: input 0 is the sync byte stream (false: the antenna streams only, for capture_rx)
    rx(int n_rx, int batch_frames, int device, int n_out = 0, bool sync_port = true);
    gr80211_ctx *pin_context() override { return d_up.session().ctx(); }
    struct frame_ref {
        uint64_t pos;  // absolute item index of the trigger byte
        int mcs, len, nsamp;
        float rad, snr, rssi;
    };
    struct trigger {
        uint64_t pos;  // absolute item index of the sync byte
        float rad, snr, rssi;
    };
    // One stretch of the stream on its way through the GPU.  Rounds follow each other: while one is filled with the
    // windows that arrive, the ones before it are probed (L-SIG of every sync byte), searched (`signal`'s trigger
    // search, on the host), decoded and published, each on the stream of a context of its own.
    struct round {
        std::unique_ptr<gpu_session> gpu;
        char *iq[2] = {nullptr, nullptr};  // device arenas, one per antenna
        size_t cap = 0, n = 0;             // items they can hold / hold
        uint64_t base = 0;                 // absolute index of item 0
        std::vector<trigger> trig;         // sync bytes inside, in stream order
        size_t n_new = 0;                  // ... of which arrived while this round was filled (the rest were carried over)
        size_t n_carry = 0;                // samples at the round's start that the round before handed over
        enum { FREE, FILL, PROBE, DECODE } state = FREE;
        bool final = false;
        std::vector<gr80211_frame_desc> pd;  // probe: one per sync byte with its 224 samples
        std::vector<int32_t> p_of;           // sync byte -> place in pd, or -1
        gr80211_frame_info *info = nullptr;  // page-locked: the records of pd
        std::vector<gr80211_frame_desc> fd;  // the frames `signal` deals with
        std::vector<int32_t> fidx;           // their places in pd
        std::vector<frame_ref> fr;
        double t_cut = 0, t_replay = 0;      // profile: seconds since the block was made
    };
    // What becomes of the frames of a round: rx starts demodulator and decoder and the copy of the PDU slots back
    // (nothing waited for) and publishes them in finish_frames once the round's stream has run dry.
    virtual bool start_frames(round &r);
    virtual bool finish_frames(round &r);
    // Called by cut() once the round's samples are all behind its stream: a block that finds the round's sync bytes
    // itself adds them to r.trig here (capture_rx).  final: the stream ends with this round.
    virtual bool before_cut(round &, bool /*final*/) { return true; }
    int take_input(gr_vector_int &ninput_items, gr_vector_const_void_star &input_items);  // items consumed, -1 on error
    bool cut(bool final);
    bool advance(bool wait_front, bool wait_all);
    bool replay(round &r);
    bool reserve(round &r, size_t items);
    round *free_round();

    int d_nrx, d_batch;
    bool d_adaptive = false;            // rounds sized by the traffic (made without a batch size)
    int d_port0 = 1;                    // input port of the first antenna (1 behind a sync port, else 0)
    size_t d_carryBack = 0;             // samples before a round's end that the next round always gets again
    int d_muPos = 0;
    uploader d_up;                      // the samples' way to the arenas of the round being filled
    static constexpr int N_ROUNDS = 3;
    round d_round[N_ROUNDS];
    round *d_fill = nullptr;
    std::deque<round *> d_flight;       // rounds being probed / decoded, oldest first
    uint64_t d_cursor = 0;              // absolute index where the trigger search resumes (lib/signal_impl.cc:75-106)
    uint64_t d_wantEnd = 0;             // stream position at which the oldest carried-over frame is complete (0: none known)
    size_t d_maxHold;
    bool d_strict = false;
    uint64_t d_nTrig = 0, d_nFrames = 0, d_nPdu = 0, d_nRounds = 0;
    // where the host's time goes (seconds), printed when the block goes if GR80211_BLOCK_PROFILE is set
    double d_tScan = 0, d_tUpload = 0, d_tSettle = 0, d_tAdvance = 0, d_tCut = 0, d_tPublish = 0;
    uint64_t d_nCalls = 0;
    std::chrono::steady_clock::time_point d_lastPoll{};  // when the rounds in flight were last asked after
    bool d_profile = false;
    std::chrono::steady_clock::time_point d_t0 = std::chrono::steady_clock::now();
    double now() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - d_t0).count(); }
};

// ieee80211::signal (n_rx = 1: in {u8 sync, c64}, out {c64}) and signal2 (n_rx = 2: in {u8, c64, c64}, out {c64, c64}),
// include/gnuradio/ieee80211/signal.h:22-36, lib/signal_impl.cc:34-206: per frame nsamp items after L-SIG with the
// CFO taken out, then 320 items of padding, with the tags cfo, snr, rssi, seq, mcs, len, nsamp and chan on the first
// one.  The trigger search is the fused block's; the items leave as the scheduler makes room for them.
class signal : public rx
{
public:
    typedef std::shared_ptr<signal> sptr;
    static sptr make(int n_rx = 1, int batch_frames = 0, int device = 0);  // no batch size: rounds sized by the traffic
    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
    size_t pending_output() const { return d_out0.pending() + d_direct.in_flight(); }

protected:
    signal(int n_rx, int batch_frames, int device);
    // a round's items are written by the GPU and copied to page-locked memory of the round without the host waiting
    // (start_frames); once the round's stream has run dry they are handed to the output queues by reference, with
    // their tags (finish_frames), and leave for the scheduler's buffers from where they are
    bool start_frames(round &r) override;
    bool finish_frames(round &r) override;

private:
    out_stream d_out0, d_out1;
    long d_seq = 0;
    std::vector<uint64_t> d_off[N_ROUNDS];  // per round: item offset of every frame in the round's output
    size_t d_left[N_ROUNDS] = {};           // per round: items the output queues still reference
    struct span {
        const char *lo = nullptr, *hi = nullptr;
    } d_span[N_ROUNDS][2];                  // per round and antenna: the memory they lie in (page-locked, or the device's)
    // output buffers in page-locked memory (decided by the first one offered): a round's items stay on the device and
    // travel into the scheduler's buffers directly
    int d_outMode = 0;                      // 0 undecided, 1 through page-locked memory of the round, 2 direct
    bool d_devOut[N_ROUNDS] = {};           // per round: its items were left on the device
    direct_out d_direct;
};

// ieee80211::demod(mupos, mugid) (n_rx = 1: in {c64}) and demod2 (n_rx = 2: in {c64, c64}), out {f32},
// include/gnuradio/ieee80211/demod.h:22-36, lib/demod_impl.cc:59-342, lib/demod2_impl.cc:58-349: reads signal's tags
// on the first item of a frame, takes nsamp + 320 items, writes the frame's soft bits with the tags cfo, snr, rssi,
// format, mcs, len, cr, ampdu, trellis, total (and sssnr0 / sssnr1 for VHT) on the first one.  Frames it cannot
// demodulate leave nothing, as in the reference.
class demod : public block
{
public:
    typedef std::shared_ptr<demod> sptr;
    static sptr make(int n_rx = 1, int mupos = 0, int mugid = 2, int batch_frames = 0, int device = 0);
    ~demod() override;
    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
    bool flush() override;
    size_t pending_output() const { return d_out.pending() + d_direct.in_flight(); }
    uint64_t frames_seen() const { return d_nSeen; }
    uint64_t rounds_cut() const { return d_nRounds; }
    void set_mu_pos(int pos) { d_muPos = pos ? 1 : 0; }
    // The reference hands on a frame's soft bits symbol by symbol as its samples arrive (lib/demod_impl.cc:279-314).
    // With this on, the frames a call completes form a round of their own and the call waits for it, so their soft
    // bits leave in that call (as far as the output buffer has room).  For live reception; see decode.  Set before the
    // first window.  On by itself when the block was made without a batch size.
    void set_finish_in_call(bool on) { d_finish = on; }
    bool finish_in_call() const { return d_finish; }

protected:
    gr80211_ctx *pin_context() override { return !d_round.empty() && d_round[0]->gpu ? d_round[0]->gpu->ctx() : nullptr; }

private:
    demod(int n_rx, int mupos, int batch_frames, int device);
    // One batch of frames on its way through the GPU.  While one round collects the frames that arrive (their items
    // go to page-locked memory of the round), the rounds before it run on contexts and host threads of their own --
    // upload, demodulator, soft bits back to page-locked memory -- and are published in order; general_work waits
    // for none of them.  A round's soft bits leave for the scheduler's buffers straight from that memory, so the
    // round is free again once the last of them has left.
    struct round {
        std::unique_ptr<gpu_session> gpu;
        gr_complex *iq[2] = {nullptr, nullptr};  // page-locked: the frames' items back to back
        size_t cap = 0, n = 0;                   // items they can hold / hold
        // input windows in page-locked memory: the items go to the device arena as they arrive and the arena mirrors
        // the stream from `base` (the round's first frame) on
        bool direct = false;
        char *dev[2] = {nullptr, nullptr};
        size_t dev_cap = 0;                      // items the arena can hold
        uint64_t base = 0;
        std::vector<gr80211_sig_desc> desc;      // one per frame
        std::vector<gr_complex> chan;            // 64 per frame
        std::vector<gr80211_frame_info> info;    // results: one record per frame ...
        const float *llr = nullptr;              // ... and the soft bits, page-locked,
        const float *llr_dev = nullptr;          // or (output buffers in page-locked memory) still on the device
        std::vector<std::vector<gr_complex>> ndp;  // raw VHT-LTF windows of sounding frames, in frame order
        std::thread worker;
        std::atomic<int> state{0};               // FREE, FILL, RUN, DONE (results there), DRAIN (soft bits leaving)
        size_t left = 0;                         // soft bits of the round the scheduler has not taken yet
        std::string err;
    };
    enum { R_FREE = 0, R_FILL, R_RUN, R_DONE, R_DRAIN };
    bool grow(round &r, size_t items);
    int fill_direct(int n, gr_vector_const_void_star &input_items);
    int settle_direct();  // waits for the oldest uploaded piece and consumes what has arrived
    bool cut();                  // the round being filled goes to its worker; another one is filled (if one is free)
    void run(round &r);          // worker thread: everything that waits for the GPU
    bool harvest(bool wait);     // publish finished rounds in order; wait: for every round in flight
    bool harvest_front();        // the same after waiting for the oldest round still running
    round *free_round();
    int d_nrx, d_muPos, d_batch, d_device;
    bool d_finish = false;
    bool d_cutAhead = false;                // the round being cut is not waited for by this call
    // Three rounds to begin with (one filling, two on their way); more are made when the soft bits of finished rounds
    // leave slowly (small output buffers), up to MAX_ROUNDS, so that input is only refused for lack of output room.
    static constexpr int N_ROUNDS = 3, MAX_ROUNDS = 16;
    std::vector<std::unique_ptr<round>> d_round;
    round *d_fill = nullptr;
    std::deque<round *> d_flight;           // rounds running or draining, oldest first
    enum { S_IDLE, S_COLLECT, S_CLEAN } d_state = S_IDLE;
    long d_need = 0, d_skip = 0;            // items of the current frame still to collect / to pass over
    out_stream d_out;
    // Output buffers in page-locked memory: a round's soft bits stay on the device and travel into the scheduler's
    // buffer directly.  A call sends the next stretch on its way behind the items it reports and reports that stretch
    // when called again (the scheduler's write pointer has then moved to exactly where it lies).
    std::atomic<int> d_outMode{0};           // 0 undecided, 1 through page-locked memory of the round, 2 direct
    direct_out d_direct;                     // the stretches on their way
    std::unique_ptr<uploader> d_up;          // the way of page-locked input windows to the GPU
    enum { M_UNKNOWN, M_STAGED, M_DIRECT } d_mode = M_UNKNOWN;  // decided by the first window
    uint64_t d_nSeen = 0, d_nRounds = 0;
    double d_tHarvest = 0, d_tTags = 0, d_tCopyIn = 0, d_tPop = 0, d_tOther = 0;  // GR80211_BLOCK_PROFILE: where the host time goes
};

// Whole receive flowgraph of examples/rx.grc / rx2.grc: inputs are the antenna streams only ({c64} or {c64, c64});
// presiso -> trigger -> sync -> signal(2) -> demod(2) -> decode all run on the GPU.  It is the fused rx block with
// the sync bytes found by gr80211_detect instead of read from a port: the stream is cut into rounds of set_max_hold
// samples, every round's samples are searched for sync bytes when it is cut, and from there on a round goes the way
// of rx (probe, trigger search, decode, publish).  Detection keeps state for a few hundred samples (plateaus,
// countdowns, the 111 items sync skips, 64-sample correlations), so a round takes the sync bytes up to 1024 samples
// before its end and leaves the rest to the next round, which starts another 1024 samples earlier.
class capture_rx : public rx
{
public:
    typedef std::shared_ptr<capture_rx> sptr;
    static sptr make(int n_rx = 1, int device = 0);

    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
    bool flush() override;
    uint64_t sync_bytes_found() const { return d_nSync; }
    // Live reception: a round may go once it holds this many samples the round before did not (at least 8192) while
    // fewer than two rounds are on their way, instead of waiting for set_max_hold samples (22 ms of air at least): a
    // frame's message then follows its last sample by a round's length and the GPU's time.  0, the default: off.
    void set_min_round(size_t items) { d_minNew = items ? std::max<size_t>(items, 8192) : 0; }

protected:
    bool before_cut(round &r, bool final) override;

private:
    capture_rx(int n_rx, int device);
    static constexpr size_t DET_MARGIN = 1024;   // a round leaves the sync bytes of its last samples to the next round
    static constexpr size_t DET_HISTORY = 1024;  // samples the next round gets before the first one it searches
    uint64_t d_detFrom = 0;                      // absolute index from which sync bytes are still to be taken
    size_t d_minNew = 0;                         // set_min_round
    uint64_t d_nSync = 0;
};

}  // namespace host
}  // namespace gr80211
