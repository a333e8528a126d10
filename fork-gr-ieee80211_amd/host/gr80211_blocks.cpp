// Host blocks over the C-ABI of libgr80211_rx.so (see gr80211_blocks.h).
#include "gr80211_blocks.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <arpa/inet.h>
#include <netdb.h>
#include <sys/socket.h>
#include <unistd.h>

namespace gr80211 {
namespace host {

namespace {

struct stopwatch {  // adds the time it lives to a counter; with a name, says so when it lived for more than half a millisecond
    double &acc;
    const char *slow;
    std::chrono::steady_clock::time_point t0;
    explicit stopwatch(double &a, const char *say_if_slow = nullptr) : acc(a), slow(say_if_slow), t0(std::chrono::steady_clock::now()) {}
    ~stopwatch()
    {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        acc += dt;
        if (slow && dt > 0.0005) fprintf(stderr, "  host held %.2f ms in %s\n", dt * 1e3, slow);
    }
};

// True, and the time noted, if 100 us have passed since the time noted last.
bool poll_due(std::chrono::steady_clock::time_point &last)
{
    const auto t = std::chrono::steady_clock::now();
    if (t - last < std::chrono::microseconds(100)) return false;
    last = t;
    return true;
}

const int LSIG_SPAN = 224;       // samples from the trigger to the end of L-SIG (lib/signal_impl.cc:110,154)
const int F_VHT_CHAN = 20;       // format byte of the sounding channel message (lib/cloud80211phy.h:40)
const int LSIG_FAIL_SKIP = 80;   // samples dropped after a failed L-SIG check (lib/signal_impl.cc:156-160)
const int OUT_STRIDE = 4352;     // room for the PDUs of one frame: 4095 bytes + headers of a few subframes
const size_t OUT_QUEUE_LIMIT = (size_t)1 << 24;  // produced items a block holds before it stops taking input
const long DEFAULT_TAG = 99999;  // value `decode` assumes for a missing tag (lib/decode_impl.cc:77-83)

double tag_value(const std::vector<tag_t> &tags, const char *key, double dflt)
{
    // pmt::dict_add overwrites, so the last tag of a key wins (lib/decode_impl.cc:73-76)
    double v = dflt;
    for (const tag_t &t : tags)
        if (t.key == key) v = t.value;
    return v;
}

const tag_t *tag_find(const std::vector<tag_t> &tags, const char *key)
{
    const tag_t *r = nullptr;
    for (const tag_t &t : tags)
        if (t.key == key) r = &t;
    return r;
}

// Index of the first non-zero byte of s[i, n), or n: a sync stream is almost all zeros, so eight bytes go at a time.
int next_set(const uint8_t *s, int i, int n)
{
    while (i + 64 <= n) {  // a cache line at a time while nothing is set
        uint64_t w[8];
        memcpy(w, s + i, 64);
        if (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) break;
        i += 64;
    }
    while (i + 8 <= n) {
        uint64_t w;
        memcpy(&w, s + i, 8);
        if (w) break;
        i += 8;
    }
    while (i < n && !s[i]) i++;
    return i;
}

}  // namespace

static void copy_past_cache(void *dst, const void *src, size_t n);  // defined with the output queues below

// ------------------------------------------------------------------------------------ block
block::block(int n_in, int n_out) : d_read(n_in, 0), d_written(n_out, 0), d_out_tags(n_out), d_tags(n_in) {}
block::~block() {}

void block::add_item_tag_in(int port, const tag_t &t)
{
    // kept sorted by offset (tags of one offset in arrival order); the usual case appends
    std::deque<tag_t> &tv = d_tags[port];
    if (tv.empty() || tv.back().offset <= t.offset) {
        tv.push_back(t);
        return;
    }
    auto it = std::upper_bound(tv.begin(), tv.end(), t.offset, [](uint64_t o, const tag_t &x) { return o < x.offset; });
    tv.insert(it, t);
}

void block::add_item_tag_in(int port, tag_t &&t)
{
    std::deque<tag_t> &tv = d_tags[port];
    if (tv.empty() || tv.back().offset <= t.offset) {
        tv.push_back(std::move(t));
        return;
    }
    auto it = std::upper_bound(tv.begin(), tv.end(), t.offset, [](uint64_t o, const tag_t &x) { return o < x.offset; });
    tv.insert(it, std::move(t));
}

void block::get_tags_in_range(std::vector<tag_t> &v, int port, uint64_t start, uint64_t end) const
{
    v.clear();
    const std::deque<tag_t> &tv = d_tags[port];
    auto it = std::lower_bound(tv.begin(), tv.end(), start, [](const tag_t &t, uint64_t o) { return t.offset < o; });
    for (; it != tv.end() && it->offset < end; ++it) v.push_back(*it);
}

bool block::tag_values_at(int port, uint64_t offset, const char *const *keys, double *out, int n_keys) const
{
    const std::deque<tag_t> &tv = d_tags[port];
    auto it = std::lower_bound(tv.begin(), tv.end(), offset, [](const tag_t &t, uint64_t o) { return t.offset < o; });
    bool any = false;
    for (; it != tv.end() && it->offset == offset; ++it) {
        any = true;
        for (int k = 0; k < n_keys; k++)
            if (it->key == keys[k]) out[k] = it->value;
    }
    return any;
}

uint64_t block::next_tag_offset(int port, uint64_t from, uint64_t end) const
{
    const std::deque<tag_t> &tv = d_tags[port];
    auto it = std::lower_bound(tv.begin(), tv.end(), from, [](const tag_t &t, uint64_t o) { return t.offset < o; });
    return it != tv.end() && it->offset < end ? it->offset : end;
}

std::vector<tag_t> block::take_output_tags(int port)
{
    std::vector<tag_t> t;
    t.swap(d_out_tags[port]);
    return t;
}

void block::commit(int produced)
{
    for (uint64_t &w : d_written) w += (uint64_t)(produced > 0 ? produced : 0);
    for (size_t p = 0; p < d_read.size(); p++) {
        d_read[p] += (uint64_t)d_consumed;
        std::deque<tag_t> &tv = d_tags[p];
        while (!tv.empty() && tv.front().offset < d_read[p]) tv.pop_front();
    }
    d_consumed = 0;
}

uint64_t block::publish_slots(const uint8_t *out, const int32_t *len, int n, int stride)
{
    uint64_t n_pub = 0;
    for (int i = 0; i < n; i++) {
        // a slot holds the frame's messages back to back: [format][len lo][len hi][MPDU][mcs]
        const uint8_t *p = out + (size_t)i * (size_t)stride;
        int at = 0;
        while (at + 4 <= len[i]) {
            const int L = p[at + 1] | (p[at + 2] << 8);
            const int rec = L + (p[at] == F_VHT_CHAN ? 3 : 4);  // a channel message has no trailing mcs byte
            if (at + rec > len[i] || at + rec > stride) break;
            d_msg.blob.assign(p + at, p + at + rec);
            d_msg.len = rec;
            message_port_pub("out", d_msg);
            n_pub++;
            at += rec;
        }
    }
    return n_pub;
}

void block::message_port_pub(const std::string &port, const pdu_t &m)
{
    if (d_handler) d_handler(port, m);
}

void *block::alloc_input(size_t bytes)
{
    gr80211_ctx *c = pin_context();
    if (!c || !bytes) return nullptr;
    void *p = gr80211_host_alloc(c, bytes);
    if (!p) return nullptr;
    d_pinned.emplace_back(static_cast<const char *>(p), bytes);
    d_owned.push_back(p);
    return p;
}

void block::free_input(void *p)
{
    gr80211_ctx *c = pin_context();
    for (size_t i = 0; i < d_owned.size(); i++)
        if (d_owned[i] == p) {
            d_owned.erase(d_owned.begin() + (long)i);
            for (size_t k = 0; k < d_pinned.size(); k++)
                if (d_pinned[k].first == p) {
                    d_pinned.erase(d_pinned.begin() + (long)k);
                    break;
                }
            if (c) gr80211_host_free(c, p);
            return;
        }
}

bool block::pin_input(const void *p, size_t bytes)
{
    gr80211_ctx *c = pin_context();
    if (!c || !p || !bytes) return false;
    if (is_pinned(p, bytes)) return true;
    if (gr80211_host_register(c, const_cast<void *>(p), bytes) != GR80211_OK) return false;
    d_pinned.emplace_back(static_cast<const char *>(p), bytes);
    d_registered.push_back(static_cast<const char *>(p));
    return true;
}

bool block::is_pinned(const void *p, size_t bytes) const
{
    const char *q = static_cast<const char *>(p);
    for (const auto &r : d_pinned)
        if (q >= r.first && q + bytes <= r.first + r.second) return true;
    return false;
}

// Called by the destructor of every block that has a context, while that context still exists.
void block::unpin_all()
{
    gr80211_ctx *c = pin_context();
    if (c) {
        for (const char *p : d_registered) gr80211_host_unregister(c, const_cast<char *>(p));
        for (void *p : d_owned) gr80211_host_free(c, p);
    }
    d_registered.clear();
    d_owned.clear();
    d_pinned.clear();
}

uint64_t upload_log::settled(uint64_t from, bool may_wait)
{
    uint64_t upto = from;
    while (!d_q.empty()) {
        const entry &e = d_q.front();
        if (e.g) {
            // nothing free yet and a copy on its way: wait for it, so that every call consumes something
            const int wait = upto == from && may_wait ? 1 : 0;
            const auto t0 = std::chrono::steady_clock::now();
            const int done = e.pull ? gr80211_pull_done(e.g->ctx(), e.id, wait) : gr80211_marker_done(e.g->ctx(), e.id, wait);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > 0.0005 && getenv("GR80211_BLOCK_PROFILE"))
                fprintf(stderr, "  waited %.2f ms for %s %lld of session %p; %zu entries queued, this one ends at item %llu, enqueued up to %llu\n",
                        dt * 1e3, e.pull ? "pull" : "marker", e.id, (void *)e.g, d_q.size(), (unsigned long long)e.upto, (unsigned long long)d_enq);
            if (done != 1) break;
        }
        upto = std::max(upto, e.upto);
        d_q.pop_front();
    }
    return upto;
}

bool upload_log::drain()
{
    bool ok = true;
    for (const entry &e : d_q)
        if (e.g && (e.pull ? gr80211_pull_done(e.g->ctx(), e.id, 1) : gr80211_marker_done(e.g->ctx(), e.id, 1)) != 1) ok = false;
    d_q.clear();
    return ok;
}

// ------------------------------------------------------------------------------------ uploader
uploader::uploader(int device) : d_a(device)
{
    if (const char *e = getenv("GR80211_UPLOAD_STREAMS"))
        if (atoi(e) == 2) d_b.reset(new gpu_session(device));
    if (const char *e = getenv("GR80211_UPLOAD_SPLIT")) d_firstSplit = atoi(e);  // experiments
    if (const char *e = getenv("GR80211_UPLOAD_STAGE")) d_stageOn = e[0] != '0';  // experiments
    if (const char *e = getenv("GR80211_STAGE_MAX_PINNED")) d_stageMaxPinned = (size_t)atoll(e);
    if (const char *e = getenv("GR80211_STAGE_BYTES")) d_stageBytes = std::max<size_t>((size_t)atoll(e), 4096);
}

uploader::~uploader()
{
    if (!d_a.ok()) return;
    d_a.sync();
    for (char *p : d_stage)
        if (p) gr80211_host_free(d_a.ctx(), p);
}

// Ring bytes [d_pendStart, d_pendStart + d_pendBytes) of every source go on their way to d_pendDst.
bool uploader::flush_staged()
{
    if (!d_pendBytes) return true;
    long long id = 0;
    for (int a = 0; a < d_pendSrcs; a++) {
        id = gr80211_pull_h2d(d_a.ctx(), d_pendDst[a], d_stage[a] + d_pendStart, d_pendBytes, nullptr);
        if (id < 0) return false;
    }
    d_sentRing.push_back({d_pendStart, d_pendStart + d_pendBytes, id});
    d_nPiece++;
    d_pendBytes = 0;
    return true;
}

bool uploader::stage(int n_src, const char *const *src, char *const *dst, size_t bytes, std::string &err)
{
    if (bytes > d_stageBytes) return false;  // (only with a ring made small on purpose)
    for (int a = 0; a < n_src; a++)
        if (!d_stage[a]) {
            d_stage[a] = static_cast<char *>(gr80211_host_alloc(d_a.ctx(), d_stageBytes));
            if (!d_stage[a]) return false;  // the caller sends the window the other way
        }
    // what is waiting goes first unless these bytes follow it, in the ring and in device memory
    bool follows = d_pendBytes && d_pendSrcs == n_src && d_wpos + bytes <= d_stageBytes;
    for (int a = 0; a < n_src && follows; a++) follows = dst[a] == d_pendDst[a] + d_pendBytes;
    if (!follows && !flush_staged()) {
        err = "upload failed: " + d_a.error();
        return false;
    }
    if (d_wpos + bytes > d_stageBytes) d_wpos = 0;
    // room in the ring: pieces sent earlier that overlap [d_wpos, d_wpos + bytes) must have been read
    while (!d_sentRing.empty()) {
        const sent &o = d_sentRing.front();
        const bool overlaps = o.lo < d_wpos + bytes && d_wpos < o.hi;
        if (!overlaps && gr80211_pull_done(d_a.ctx(), o.id, 0) != 1) break;  // still travelling, but not in the way
        if (overlaps && gr80211_pull_done(d_a.ctx(), o.id, 1) != 1) {
            err = "upload failed: " + d_a.error();
            return false;
        }
        d_sentRing.pop_front();
    }
    for (const sent &o : d_sentRing)  // (pieces behind a piece that is not in the way)
        if (o.lo < d_wpos + bytes && d_wpos < o.hi && gr80211_pull_done(d_a.ctx(), o.id, 1) != 1) {
            err = "upload failed: " + d_a.error();
            return false;
        }
    for (int a = 0; a < n_src; a++) copy_past_cache(d_stage[a] + d_wpos, src[a], bytes);
    if (!d_pendBytes) {
        d_pendStart = d_wpos;
        d_pendSrcs = n_src;
        for (int a = 0; a < n_src; a++) d_pendDst[a] = dst[a];
    }
    d_pendBytes += bytes;
    d_wpos += bytes;
    if (d_pendBytes >= STAGE_FLUSH && !flush_staged()) {
        err = "upload failed: " + d_a.error();
        return false;
    }
    return true;
}

bool uploader::send(gpu_session &direct, int n_src, const char *const *src, char *const *dst, size_t items, size_t item,
                    uint64_t pos0, bool pinned, std::string &err)
{
    if (items == 0) return true;
    size_t src_skip = 0;  // items the staging path has taken already (it ran out of memory part-way)
    if (d_stageOn && n_src <= 2 && (!pinned || items * item <= d_stageMaxPinned)) {
        // (a window in ordinary memory of any size, in pieces the ring takes: copying by hand runs beside the pieces
        // already on their way, where the runtime's copy out of ordinary memory makes the caller wait for all of it)
        const size_t per = std::max<size_t>(STAGE_FLUSH / item, 1);
        std::string e2;
        size_t at = 0;
        while (at < items) {
            const size_t cnt = std::min(per, items - at);
            const char *s2[2];
            char *d2[2];
            for (int a = 0; a < n_src; a++) s2[a] = src[a] + at * item, d2[a] = dst[a] + at * item;
            if (!stage(n_src, s2, d2, cnt * item, e2)) break;
            at += cnt;
        }
        if (at == items) {
            d_log.add(nullptr, 0, pos0 + items);  // copied out of the window: free at once
            return true;
        }
        if (!e2.empty()) {
            err = e2;
            return false;
        }
        // no staging memory: what is left goes the other way
        src_skip = at;
    }
    const char *src_rest[2];
    char *dst_rest[2];
    if (src_skip) {
        for (int a = 0; a < n_src; a++) src_rest[a] = src[a] + src_skip * item, dst_rest[a] = dst[a] + src_skip * item;
        src = src_rest, dst = dst_rest;
        items -= src_skip;
        pos0 += src_skip;
    }
    // a larger window overtakes nothing that was staged before it
    if (!flush_staged()) {
        err = "upload failed: " + d_a.error();
        return false;
    }
    if (!pinned) {
        for (int a = 0; a < n_src; a++)
            if (gr80211_copy_h2d(direct.ctx(), dst[a], src[a], items * item, nullptr) != GR80211_OK) {
                err = "upload failed: " + direct.error();
                return false;
            }
        d_log.add(nullptr, 0, pos0 + items);  // an ordinary copy has run when it returns
        return true;
    }
    // Each piece has its own completion: this call's caller consumes the first piece's items and leaves the others on
    // their way, later calls add what they are newly offered as one more piece and consume what has arrived, so that
    // copies are queued ahead of the link while the host works.  Only a window that finds nothing on its way is cut
    // up: then the lag has to be built.
    const size_t bytes = items * item;
    size_t pieces = 1;
    if (d_log.empty())
        pieces = d_firstSplit > 0 ? (size_t)d_firstSplit : bytes >= ((size_t)4 << 20) ? 8 : bytes >= ((size_t)1 << 20) ? 4 : bytes >= ((size_t)256 << 10) ? 2 : 1;
    size_t per = items / pieces * item / 64 * 64 / item;  // whole 64-byte lines: the wide form of the copy kernel
    if (per == 0) pieces = 1;
    // small pieces go by the kernel that reads host memory itself (a launch, nothing else); from a megabyte on the
    // runtime's copy engines are as fast and leave the compute units alone
    static const char *mode = getenv("GR80211_UPLOAD");  // "dma" / "pull": one way for every size (experiments)
    for (size_t at = 0, k = 0; at < items; k++) {
        const size_t len = k + 1 < pieces ? per : items - at;
        const bool use_dma = mode ? mode[0] == 'd' : len * item >= ((size_t)1 << 20);
        gpu_session &via = (d_nPiece++ & 1) && d_b ? *d_b : d_a;
        gr80211_ctx *c = via.ctx();
        long long id = 0;
        for (int a = 0; a < n_src; a++) {
            const long long rc = use_dma ? gr80211_copy_h2d_async(c, dst[a] + at * item, src[a] + at * item, len * item, nullptr)
                                         : gr80211_pull_h2d(c, dst[a] + at * item, src[a] + at * item, len * item, nullptr);
            if (rc < 0) {
                err = "upload failed: " + via.error();
                return false;
            }
            id = rc;  // the last source's copy is the last to finish
        }
        at += len;
        if (use_dma) {
            id = gr80211_marker_record(c, nullptr);
            if (id < 0) {
                err = "marker: " + via.error();
                return false;
            }
        }
        d_log.add(&via, id, pos0 + at, !use_dma);
    }
    return true;
}

bool uploader::join(gr80211_ctx *c)
{
    if (!flush_staged()) return false;
    if (d_nPiece == 0) return true;
    d_nPiece = 0;
    return gr80211_stream_wait(c, nullptr, d_a.ctx(), nullptr) == GR80211_OK &&
           (!d_b || gr80211_stream_wait(c, nullptr, d_b->ctx(), nullptr) == GR80211_OK);
}

// ------------------------------------------------------------------------------------ gpu_session
gpu_session::gpu_session(int device)
{
    if (gr80211_create(&d_ctx, device) != GR80211_OK) d_ctx = nullptr;
}

gpu_session::~gpu_session()
{
    if (!d_ctx) return;
    for (dbuf &b : d_buf)
        if (b.p) gr80211_dev_free(d_ctx, b.p);
    for (dbuf &b : d_stage)
        if (b.p) gr80211_host_free(d_ctx, b.p);
    gr80211_destroy(d_ctx);
}

void *gpu_session::buffer(int slot, size_t bytes)
{
    dbuf &b = d_buf[slot];
    if (bytes <= b.cap) return b.p;
    if (b.p) gr80211_dev_free(d_ctx, b.p);
    b.cap = bytes + bytes / 4 + 4096;
    b.p = gr80211_dev_alloc(d_ctx, b.cap);
    if (!b.p) b.cap = 0;
    return b.p;
}

void *gpu_session::staging(int slot, size_t bytes)
{
    dbuf &b = d_stage[slot];
    if (bytes <= b.cap) return b.p;
    if (b.p) gr80211_host_free(d_ctx, b.p);
    b.cap = bytes + bytes / 4 + 4096;
    b.p = d_ctx ? gr80211_host_alloc(d_ctx, b.cap) : nullptr;
    if (!b.p) b.cap = 0;
    return b.p;
}

device_fifo::~device_fifo()
{
    if (!d_g.ok()) return;
    if (d_p) gr80211_dev_free(d_g.ctx(), d_p);
    if (d_alt) gr80211_dev_free(d_g.ctx(), d_alt);
}

bool device_fifo::reserve(size_t items)
{
    if (items <= d_cap) return true;
    if (!d_g.ok()) return false;
    size_t cap = std::max<size_t>(2 * d_cap, items);
    cap = std::max<size_t>(cap, (size_t)1 << 20);
    char *q = static_cast<char *>(gr80211_dev_alloc(d_g.ctx(), (cap + 64) * d_item));
    if (!q) return false;
    if (d_n && gr80211_copy_d2d(d_g.ctx(), q, d_p, d_n * d_item, nullptr) != GR80211_OK) {
        gr80211_dev_free(d_g.ctx(), q);
        return false;
    }
    if (d_p) gr80211_dev_free(d_g.ctx(), d_p);
    if (d_alt) gr80211_dev_free(d_g.ctx(), d_alt);
    d_alt = nullptr;
    d_p = q;
    d_cap = cap;
    return true;
}

bool device_fifo::append(const void *host_src, size_t items)
{
    if (items == 0) return true;
    if (!reserve(d_n + items)) return false;
    if (gr80211_copy_h2d(d_g.ctx(), d_p + d_n * d_item, host_src, items * d_item, nullptr) != GR80211_OK) return false;
    d_n += items;
    return true;
}

bool device_fifo::release_front(size_t items)
{
    if (items >= d_n) {
        d_n = 0;
        return true;
    }
    if (items == 0) return true;
    // the rest moves to the front of the other buffer (a copy inside one buffer could overlap)
    if (!d_alt) {
        d_alt = static_cast<char *>(gr80211_dev_alloc(d_g.ctx(), (d_cap + 64) * d_item));
        if (!d_alt) return false;
    }
    if (gr80211_copy_d2d(d_g.ctx(), d_alt, d_p + items * d_item, (d_n - items) * d_item, nullptr) != GR80211_OK) return false;
    std::swap(d_p, d_alt);
    d_n -= items;
    return true;
}

static void copy_past_cache(void *dst, const void *src, size_t n);

// GR80211_COPY_PAST_CACHE=0 switches the streaming stores of copy_past_cache off (A/B measurements).
static bool getenv_flag_copy_past_cache()
{
    static const bool on = [] {
        const char *e = getenv("GR80211_COPY_PAST_CACHE");
        return !e || e[0] != '0';
    }();
    return on;
}

void out_stream::push(const void *src, size_t items)
{
    if (items == 0) return;
    if (d_segs.empty() || d_segs.back().ext) d_segs.emplace_back();
    seg &g = d_segs.back();
    const uint8_t *p = static_cast<const uint8_t *>(src);
    g.own.insert(g.own.end(), p, p + items * d_item);
    g.bytes = g.own.size();
    d_pushed += items;
}

void out_stream::push_zeros(size_t items)
{
    if (items == 0) return;
    if (d_segs.empty() || d_segs.back().ext) d_segs.emplace_back();
    seg &g = d_segs.back();
    g.own.insert(g.own.end(), items * d_item, 0);
    g.bytes = g.own.size();
    d_pushed += items;
}

void out_stream::push_ref(const void *src, size_t items, size_t *left)
{
    if (items == 0) return;
    d_segs.emplace_back();
    seg &g = d_segs.back();
    g.ext = static_cast<const uint8_t *>(src);
    g.bytes = items * d_item;
    g.left = left;
    if (left) *left += items;
    d_pushed += items;
}

void out_stream::push_dev(gpu_session *g, const void *dev_src, size_t items, size_t *left)
{
    if (items == 0) return;
    push_ref(dev_src, items, left);
    d_segs.back().dev = g;
}

bool out_stream::flight::arrived(bool wait)
{
    for (auto &m : markers)
        if (gr80211_marker_done(m.first->ctx(), m.second, wait ? 1 : 0) != 1) return false;
    return true;
}

void out_stream::flight::settle()
{
    for (auto &l : lefts)
        if (l.first) *l.first -= l.second;
    lefts.clear();
    markers.clear();
}

bool out_stream::pop_launch(void *dst, int max_items, flight &f, std::string &err)
{
    f = flight();
    const size_t n = std::min<size_t>(pending(), max_items > 0 ? (size_t)max_items : 0);
    if (n == 0) return true;
    uint8_t *d = static_cast<uint8_t *>(dst);
    size_t want = n * d_item;
    // a run of device items that lie back to back travels as one copy
    gpu_session *run_g = nullptr;
    const uint8_t *run_src = nullptr;
    uint8_t *run_dst = nullptr;
    size_t run_bytes = 0;
    auto flush_run = [&]() -> bool {
        if (!run_g || !run_bytes) return true;
        if (gr80211_push_d2h(run_g->ctx(), run_dst, run_src, run_bytes, nullptr) != GR80211_OK) {
            err = "copy of produced items failed: " + run_g->error();
            return false;
        }
        if (f.markers.empty() || f.markers.back().first != run_g) f.markers.emplace_back(run_g, -1);
        run_bytes = 0;
        return true;
    };
    while (want > 0) {
        seg &g = d_segs.front();
        const size_t take = std::min(want, g.bytes - g.head);
        if (g.dev) {
            const uint8_t *src = g.ext + g.head;
            if (run_g != g.dev || src != run_src + run_bytes) {
                if (!flush_run()) return false;
                run_g = g.dev, run_src = src, run_dst = d;
            }
            run_bytes += take;
        } else if (g.ext) {
            copy_past_cache(d, g.ext + g.head, take);
        } else {
            memcpy(d, g.own.data() + g.head, take);
        }
        if (g.left) f.lefts.emplace_back(g.left, take / d_item);
        g.head += take;
        d += take;
        want -= take;
        if (g.head == g.bytes) d_segs.pop_front();
    }
    if (!flush_run()) return false;
    for (auto &m : f.markers) {
        m.second = gr80211_marker_record(m.first->ctx(), nullptr);
        if (m.second < 0) {
            err = "marker: " + m.first->error();
            return false;
        }
    }
    while (!d_tags.empty() && d_tags.front().offset < d_popped + n) {
        tag_t t = d_tags.front();
        t.offset -= d_popped;
        f.tags.push_back(t);
        d_tags.pop_front();
    }
    d_popped += n;
    f.dst = dst;
    f.items = n;
    return true;
}

int direct_out::work(out_stream *const *q, void *const *dst, const uint64_t *written, int n_ports, size_t item, int noutput_items,
                     bool must_progress, const std::function<void(int, const tag_t &)> &tag, std::string &err)
{
    int produced = 0;
    auto deliver = [&]() {
        stretch &st = d_sent.front();
        for (int p = 0; p < n_ports; p++) {
            out_stream::flight &f = st.f[p];
            f.arrived(true);
            uint8_t *here = static_cast<uint8_t *>(dst[p]) + (size_t)produced * item;
            if (f.dst != here) memcpy(here, f.dst, st.items * item);  // the buffer wrapped since the stretch left
            for (tag_t &t : f.tags) {
                t.offset += written[p] + (uint64_t)produced;
                tag(p, t);
            }
            f.settle();
        }
        produced += (int)st.items;
        d_items -= st.items;
        d_sent.pop_front();
    };
    auto arrived = [&](stretch &st) {
        for (int p = 0; p < n_ports; p++)
            if (!st.f[p].arrived(false)) return false;
        return true;
    };
    auto launch = [&]() -> bool {
        // half the room at a time while nothing else travels, so that the next call finds something to report while the
        // other half is on its way
        const size_t room = (size_t)noutput_items - (size_t)produced - d_items;
        size_t n = std::min(room, q[0]->pending());
        if (d_sent.empty() && n >= 16384) n = (n / 2 + 3) & ~(size_t)3;
        if (n == 0) return true;
        stretch st;
        for (int p = 0; p < n_ports; p++) {
            uint8_t *to = static_cast<uint8_t *>(dst[p]) + ((size_t)produced + d_items) * item;
            if (!q[p]->pop_launch(to, (int)n, st.f[p], err)) return false;
        }
        st.items = st.f[0].items;
        d_items += st.items;
        d_sent.push_back(std::move(st));
        return true;
    };
    while (!d_sent.empty() && (size_t)produced + d_sent.front().items <= (size_t)noutput_items && arrived(d_sent.front())) deliver();
    if (d_sent.size() < 2 && (size_t)produced + d_items < (size_t)noutput_items && q[0]->pending() && !launch()) return -1;
    if (produced == 0 && must_progress && !d_sent.empty() && d_sent.front().items <= (size_t)noutput_items) {
        deliver();  // waits for it
        if (d_sent.size() < 2 && (size_t)produced + d_items < (size_t)noutput_items && q[0]->pending() && !launch()) return -1;
    }
    return produced;
}

void out_stream::release_refs(const void *lo, const void *hi)
{
    for (seg &g : d_segs) {
        if (!g.ext || g.ext < static_cast<const uint8_t *>(lo) || g.ext >= static_cast<const uint8_t *>(hi)) continue;
        if (g.dev) {
            // device memory about to be rewritten: what has not left comes back now (waits)
            g.own.resize(g.bytes - g.head);
            (void)gr80211_copy_d2h(g.dev->ctx(), g.own.data(), g.ext + g.head, g.bytes - g.head, nullptr);
            g.dev = nullptr;
        } else {
            g.own.assign(g.ext + g.head, g.ext + g.bytes);
        }
        g.bytes -= g.head;
        g.head = 0;
        g.ext = nullptr;
        if (g.left) *g.left -= g.bytes / d_item;
        g.left = nullptr;
    }
}

// A copy for runs that nobody on this core reads again soon (items on their way from page-locked staging memory, which
// the GPU wrote, to a scheduler buffer, which another thread or the GPU reads): stores that go past the cache, so that
// the destination's lines are not fetched before they are overwritten and the copy does not push the thread's working
// set out.  Short runs and x86 machines without SSE2 take memcpy.
static void copy_past_cache(void *dst, const void *src, size_t n)
{
#if defined(__SSE2__)
    if (n >= 4096 && getenv_flag_copy_past_cache()) {
        uint8_t *d = static_cast<uint8_t *>(dst);
        const uint8_t *s = static_cast<const uint8_t *>(src);
        const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
        if (head) {
            memcpy(d, s, head);
            d += head, s += head, n -= head;
        }
        size_t k = 0;
        for (; k + 64 <= n; k += 64) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + k));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + k + 16));
            const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + k + 32));
            const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + k + 48));
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + k), a);
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + k + 16), b);
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + k + 32), c);
            _mm_stream_si128(reinterpret_cast<__m128i *>(d + k + 48), e);
        }
        _mm_sfence();
        if (k < n) memcpy(d + k, s + k, n - k);
        return;
    }
#endif
    memcpy(dst, src, n);
}

int out_stream::pop(void *dst, int max_items, uint64_t written, std::vector<tag_t> &out)
{
    const size_t n = std::min<size_t>(pending(), max_items > 0 ? (size_t)max_items : 0);
    if (n == 0) return 0;
    uint8_t *d = static_cast<uint8_t *>(dst);
    size_t want = n * d_item;
    while (want > 0) {
        seg &g = d_segs.front();
        const size_t take = std::min(want, g.bytes - g.head);
        if (g.dev) (void)gr80211_copy_d2h(g.dev->ctx(), d, g.ext + g.head, take, nullptr);  // waits
        else if (g.ext) copy_past_cache(d, g.ext + g.head, take);
        else memcpy(d, g.own.data() + g.head, take);
        g.head += take;
        d += take;
        want -= take;
        if (g.left) *g.left -= take / d_item;
        if (g.head == g.bytes) d_segs.pop_front();
    }
    while (!d_tags.empty() && d_tags.front().offset < d_popped + n) {
        tag_t t = d_tags.front();
        t.offset = written + (t.offset - d_popped);
        out.push_back(t);
        d_tags.pop_front();
    }
    d_popped += n;
    return (int)n;
}

std::string gpu_session::error() const { return d_ctx ? gr80211_last_error(d_ctx) : "no GPU context"; }

// ------------------------------------------------------------------------------------ udp_sink
udp_sink::~udp_sink()
{
    if (d_fd >= 0) ::close(d_fd);
}

bool udp_sink::open(const std::string &host, int port)
{
    if (d_fd >= 0) ::close(d_fd);
    d_fd = -1;
    if (port < 1 || port > 65535) return false;
    addrinfo hints;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_DGRAM;
    addrinfo *res = nullptr;
    if (getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res) return false;
    const int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
    // connected: send() needs no address, and a port nobody listens on shows as an error of a later send
    if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) d_fd = fd;
    else if (fd >= 0) ::close(fd);
    freeaddrinfo(res);
    return d_fd >= 0;
}

bool udp_sink::send(const pdu_t &m)
{
    if (d_fd < 0) return false;
    const size_t n = std::min(m.blob.size(), MTU);
    const ssize_t rc = ::send(d_fd, m.blob.data(), n, 0);
    if (rc != (ssize_t)n) return false;
    d_nSent++;
    return true;
}

// ------------------------------------------------------------------------------------ decode
decode::sptr decode::make(bool ifdebug, int batch_frames, int device)
{
    return sptr(new decode(ifdebug, batch_frames, device));
}

decode::decode(bool ifdebug, int batch_frames, int device)
    : block(1), d_debug(ifdebug), d_batch(batch_frames < 1 ? 4096 : batch_frames), d_up(device)
{
    d_finish = batch_frames < 1;  // made like the reference's block: timed like it
    if (!d_up.ok()) {
        d_err = "gr80211_create failed";
        return;
    }
    // soft bits: windows of 64 k items (256 KB) still do better copied by hand (1.10 -> 1.21 Gsamples/s-equivalent)
    d_up.set_stage_max_pinned((size_t)300 << 10);
    for (round &r : d_round) {
        r.gpu.reset(new gpu_session(device));
        if (!r.gpu->ok()) {
            d_err = "gr80211_create failed";
            return;
        }
    }
    d_fill = &d_round[0];
    d_fill->state = round::FILL;
    if (const char *e = getenv("GR80211_DECODE_MAX_HOLD")) set_max_hold((size_t)atoll(e));  // experiments
    // the arenas and result buffers of the first two rounds exist before the stream starts (making them takes
    // milliseconds); the third follows when a stream is fast enough to need it
    for (int k = 0; k < 2; k++) {
        round &r = d_round[k];
        const size_t nf = (size_t)std::min(d_batch, 8192);
        if (!reserve(r, 1) || !r.gpu->buffer(1, nf * OUT_STRIDE) || !r.gpu->buffer(2, nf * 4) || !r.gpu->buffer(3, nf * 4) ||
            !r.gpu->staging(0, nf * OUT_STRIDE) || !r.gpu->staging(1, nf * 4)) {
            d_err = "device memory for the held soft bits exhausted";
            return;
        }
    }
}

decode::~decode()
{
    // nothing of this block may still be on its way when its buffers go
    d_up.log().drain();
    if (d_up.ok()) d_up.sync();
    for (round &r : d_round) {
        if (!r.gpu || !r.gpu->ok()) continue;
        r.gpu->sync();
        if (r.llr) gr80211_dev_free(r.gpu->ctx(), r.llr);
    }
    unpin_all();
}

void decode::set_ampdu_strict(bool on)
{
    d_strict = on;
    for (round &r : d_round)
        if (r.gpu) r.gpu->set_ampdu_strict(on);
}

void decode::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    ninput_items_required[0] = noutput_items + 160;  // lib/decode_impl.cc:54-58
}

bool decode::reserve(round &r, size_t items)
{
    if (items <= r.cap) return true;
    gr80211_ctx *c = r.gpu->ctx();
    // an arena is made for a whole round at once (what a round may hold, and a window and the longest packet on
    // top): growing it later means a copy and a wait
    const size_t cap = std::max<size_t>(std::max<size_t>(2 * r.cap, items), std::min<size_t>(d_maxHold, (size_t)1 << 27) + ((size_t)1 << 22));
    float *q = static_cast<float *>(gr80211_dev_alloc(c, (cap + 64) * sizeof(float)));
    if (!q) return false;
    if (r.n) {
        // what the arena holds moves over once the copies that are filling it have run
        if (!d_up.sync() || gr80211_copy_d2d(c, q, r.llr, r.n * sizeof(float), nullptr) != GR80211_OK) {
            gr80211_dev_free(c, q);
            return false;
        }
    }
    if (r.llr) gr80211_dev_free(c, r.llr);
    r.llr = q;
    r.cap = cap;
    return true;
}

// Window items [from, to) go to the round being filled (it holds the stream from its first packet on, gaps and all);
// a round without a packet holds nothing, so the items are free at once.
bool decode::send(const float *in, int from, int to)
{
    if (to <= from) return true;
    round &f = *d_fill;
    const uint64_t r0 = nitems_read(0);
    if (f.desc.empty()) {
        d_up.log().add(nullptr, 0, r0 + (uint64_t)to);
        return true;
    }
    const size_t cnt = (size_t)(to - from);
    if (!reserve(f, f.n + cnt)) {
        d_err = "device memory for the held soft bits exhausted";
        return false;
    }
    const char *src = reinterpret_cast<const char *>(in + from);
    char *dst = reinterpret_cast<char *>(f.llr + f.n);
    if (!d_up.send(*f.gpu, 1, &src, &dst, cnt, sizeof(float), r0 + (uint64_t)from, is_pinned(src, cnt * sizeof(float)), d_err))
        return false;
    f.n += cnt;
    return true;
}

int decode::general_work(int, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    consume_each(0);
    if (!d_err.empty()) return 0;
    const float *in = static_cast<const float *>(input_items[0]);
    const int n = ninput_items[0];
    const uint64_t r0 = nitems_read(0);
    upload_log &log = d_up.log();
    if (log.enqueued() < r0) log.set_enqueued(r0);
    d_nCalls++;
    // asking the runtime about the rounds in flight costs a few us: at most every 100 us
    if (!d_flight.empty() && poll_due(d_lastPoll) && !advance(false, false)) return 0;
    // items [used, n) of the window are new; the ones before were taken by an earlier call and wait for their copies
    int used = (int)std::min<uint64_t>(log.enqueued() - r0, (uint64_t)n);
    int span = used;  // window items from here on have not been sent yet
    std::vector<tag_t> tags;
    while (used < n) {
        if (d_state == S_IDLE) {
            const uint64_t at = r0 + (uint64_t)used;
            // the packet's tags by value, without copying them (the last tag of a key wins, a missing one reads 99999)
            static const char *const keys[7] = {"format", "len", "total", "cr", "mcs", "ampdu", "trellis"};
            double v[7] = {DEFAULT_TAG, DEFAULT_TAG, DEFAULT_TAG, DEFAULT_TAG, DEFAULT_TAG, DEFAULT_TAG, DEFAULT_TAG};
            if (!tag_values_at(0, at, keys, v, 7)) {
                // The reference stays idle until an item carries tags (lib/decode_impl.cc:70-72,125-126);
                // untagged items between packets are passed over here.
                used = (int)(next_tag_offset(0, at + 1, r0 + (uint64_t)n) - r0);
                continue;
            }
            t_format = (long)v[0];
            t_len = (long)v[1];
            t_nTotal = (long)v[2];
            t_cr = (long)v[3];
            t_mcs = (long)v[4];
            t_ampdu = (long)v[5];
            v_trellis = (long)v[6];
            t_nProcd = 0;
            d_nSeen++;
            if (t_nTotal <= 0) {  // a packet without items cannot be passed over by counting items
                used++;
                continue;
            }
            if (t_len > GR80211_MAX_PSDU || v_trellis > GR80211_MAX_TRELLIS) {
                d_state = S_CLEAN;  // lib/decode_impl.cc:93-97
            } else if (v_trellis == 0) {
                // VHT NDP: the 2x1 channel dump travels as a message of its own (lib/decode_impl.cc:100-121)
                d_state = S_CLEAN;
                get_tags_in_range(tags, 0, at, at + 1);
                const tag_t *ch = tag_find(tags, "mu2x1chan");
                pdu_t m;
                const int nbytes = (int)sizeof(float) * 256;
                m.blob.assign(nbytes + 3, 0);
                m.blob[0] = F_VHT_CHAN;
                m.blob[1] = (uint8_t)(nbytes % 256);
                m.blob[2] = (uint8_t)(nbytes / 256);
                if (ch) {
                    float *f = reinterpret_cast<float *>(&m.blob[3]);
                    for (size_t i = 0; i < 128 && i < ch->cvec.size(); i++) {
                        const float re = ch->cvec[i].real(), im = ch->cvec[i].imag();
                        memcpy(f + 2 * i, &re, 4);
                        memcpy(f + 2 * i + 1, &im, 4);
                    }
                }
                m.len = nbytes + 3;
                if (d_flight.empty() && d_fill->desc.empty() && d_fill->ready.empty()) {
                    message_port_pub("out", m);
                } else {
                    // packets before it are still waiting for the GPU: it takes its turn behind them
                    d_fill->ready.emplace_back(d_fill->desc.size(), std::move(m));
                }
                d_nPdu++;
            } else {
                d_state = S_COLLECT;
                if (d_finish && !d_fill->desc.empty() && t_nTotal > (long)(n - used) && d_flight.size() < 2) {
                    // this packet ends in a later call: the ones before it do not wait for it (while two rounds are on
                    // their way they do wait, for company: a stream that comes faster than rounds return gets larger rounds)
                    if (!send(in, span, used)) return 0;
                    span = used;
                    if (!cut()) return 0;
                }
                round &f = *d_fill;
                if (f.desc.empty()) {
                    // the round's first packet: the round holds the stream from here on
                    if (!send(in, span, used)) return 0;
                    span = used;
                    f.base = at;
                    f.n = 0;
                }
                gr80211_dec_desc d;
                memset(&d, 0, sizeof(d));
                d.llr_off = (int64_t)(at - f.base);
                d.n_llr = (int32_t)t_nTotal;
                d.format = (int32_t)t_format;
                d.len = (int32_t)t_len;
                d.cr = (int32_t)t_cr;
                d.mcs = (int32_t)t_mcs;
                d.ampdu = (int32_t)t_ampdu;
                d.trellis = (int32_t)v_trellis;
                f.desc.push_back(d);
                if (d_debug) {
                    // what the reference's debug lines quote beside the counters (lib/decode_impl.cc:383-387)
                    static const char *const dkeys[5] = {"cfo", "snr", "rssi", "sssnr0", "sssnr1"};
                    double dv[5] = {0, 0, 0, 0, 0};
                    tag_values_at(0, at, dkeys, dv, 5);
                    f.dbg.push_back({(float)dv[0], (float)dv[1], (float)dv[2], (float)dv[3], (float)dv[4]});
                }
            }
        }
        if (d_state == S_COLLECT || d_state == S_CLEAN) {
            const long need = t_nTotal - t_nProcd;
            const int take = (int)std::min<long>(need < 0 ? 0 : need, n - used);
            t_nProcd += take;
            used += take;
            if (t_nProcd >= t_nTotal) {
                const bool collected = d_state == S_COLLECT;
                d_state = S_IDLE;
                round &f = *d_fill;
                if (collected && ((int)f.desc.size() >= d_batch || f.n + (size_t)(used - span) >= d_maxHold)) {
                    // the round ends with this packet
                    if (!send(in, span, used)) return 0;
                    span = used;
                    if (!cut()) return 0;
                }
            }
        }
    }
    if (!send(in, span, n)) return 0;
    log.set_enqueued(r0 + (uint64_t)n);
    if (d_finish) {
        // what this call completed is decoded and published before it returns -- unless the window ends inside a
        // packet: then the next call is certain and near, and rounds on their way need not hold this one up (which is
        // what keeps a stream played at full speed moving: its windows end anywhere)
        if (d_state != S_COLLECT && !d_fill->desc.empty() && !cut()) return 0;
        if (!d_flight.empty() && !advance(d_state != S_COLLECT, d_state != S_COLLECT)) return 0;
    }
    consume_each((int)(log.settled(r0) - r0));
    return 0;
}

// The round being filled goes to the decoder (descriptors, kernels and the copies of the results back are put on its
// stream, nothing is waited for); the packets that follow fill another round.
bool decode::cut()
{
    round &r = *d_fill;
    gr80211_ctx *c = r.gpu->ctx();
    const int nf = (int)r.desc.size();
    // the soft bits came by streams of their own
    if (!d_up.join(c)) {
        d_err = r.gpu->error();
        return false;
    }
    uint8_t *dout = static_cast<uint8_t *>(r.gpu->buffer(1, (size_t)nf * OUT_STRIDE));
    int32_t *dlen = static_cast<int32_t *>(r.gpu->buffer(2, (size_t)nf * 4));
    int32_t *dnp = static_cast<int32_t *>(r.gpu->buffer(3, (size_t)nf * 4));
    uint8_t *out = static_cast<uint8_t *>(r.gpu->staging(0, (size_t)nf * OUT_STRIDE));
    int32_t *len = static_cast<int32_t *>(r.gpu->staging(1, (size_t)nf * 4));
    if (!dout || !dlen || !dnp || !out || !len) {
        d_err = "allocation of the output buffers failed";
        return false;
    }
    if (gr80211_decode_batch(c, r.llr, r.n, r.desc.data(), nf, 0, dout, OUT_STRIDE, dlen, dnp, nullptr, 0, nullptr) != GR80211_OK ||
        gr80211_push_d2h(c, len, dlen, (size_t)nf * 4, nullptr) != GR80211_OK ||
        gr80211_push_d2h(c, out, dout, (size_t)nf * OUT_STRIDE, nullptr) != GR80211_OK) {
        d_err = r.gpu->error();
        return false;
    }
    r.state = round::DECODE;
    d_flight.push_back(&r);
    d_nRounds++;
    round *nx = nullptr;
    while (!nx) {
        for (round &q : d_round)
            if (q.state == round::FREE) nx = &q;
        if (!nx && !advance(true, false)) return false;  // the oldest round in flight gives its place
    }
    nx->n = 0;
    nx->desc.clear();
    nx->ready.clear();
    nx->state = round::FILL;
    d_fill = nx;
    return true;
}

void decode::publish_ready(round &r, size_t upto_desc, size_t &at)
{
    for (; at < r.ready.size() && r.ready[at].first <= upto_desc; at++) message_port_pub("out", r.ready[at].second);
}

// The per-MCS counters of packets with a good FCS that the reference prints when made with ifdebug
// (lib/decode_impl.cc:375-411,453-509; scraped by its tools/performance scripts: last line with "crc32" and the format's
// name, fields 3 ... 3 + nMcs).  One "correct" line per MPDU published, one "wrong" line for a packet that published
// none (the reference prints one per failed subframe of a VHT A-MPDU; the GPU returns the good ones only).  Aggregated HT
// packets print nothing there and nothing here.
void decode::debug_lines(const gr80211_dec_desc &d, const debug_tags &t, uint64_t n_pub)
{
    const bool vht = d.format == 2, legacy = d.format == 0;
    if (!vht && d.ampdu) return;
    const uint64_t lines = n_pub ? n_pub : 1;
    for (uint64_t k = 0; k < lines; k++) {
        if (n_pub) {
            d_nPktCorrect++;
            if (vht) {
                if (d.mcs >= 0 && d.mcs < 10) d_vhtMcsCount[d.mcs]++;
            } else if (legacy) {
                if (d.mcs >= 0 && d.mcs < 8) d_legacyMcsCount[d.mcs]++;
            } else if (d.mcs >= 0) {
                d_htMcsCount[d.mcs % 8]++;
            }
        }
        std::string line = std::string("ieee80211 decode, ") + (vht ? "vht" : legacy ? "legacy" : "ht") + " crc32 " +
                           (n_pub ? "correct" : "wrong") + ", total:" + std::to_string(d_nPktCorrect);
        const uint64_t *cnt = vht ? d_vhtMcsCount : legacy ? d_legacyMcsCount : d_htMcsCount;
        for (int i = 0; i < (vht ? 10 : 8); i++) line += "," + std::to_string(i) + ":" + std::to_string(cnt[i]);
        line += ",cfo:" + std::to_string(t.cfo) + ",snr:" + std::to_string(t.snr) + ",rssi:" + std::to_string(t.rssi);
        if (vht) line += ",sssnr0:" + std::to_string(t.sssnr0) + ",sssnr1:" + std::to_string(t.sssnr1);
        fprintf(d_debugOut ? d_debugOut : stdout, "%s\n", line.c_str());
    }
}

// The oldest rounds in flight are published as their streams run dry.  wait_front: wait for the oldest one;
// wait_all: for all.
bool decode::advance(bool wait_front, bool wait_all)
{
    while (!d_flight.empty()) {
        round *r = d_flight.front();
        if (!r->gpu->idle()) {
            if (!(wait_front || wait_all)) break;
            if (!r->gpu->sync()) {
                d_err = r->gpu->error();
                return false;
            }
        }
        if (gr80211_last_batch_refused(r->gpu->ctx()) != 0) {
            d_err = "the GPU refused a batch: the descriptors on the host do not describe it";
            return false;
        }
        const int nf = (int)r->desc.size();
        const uint8_t *out = static_cast<const uint8_t *>(r->gpu->staging(0, (size_t)nf * OUT_STRIDE));
        const int32_t *len = static_cast<const int32_t *>(r->gpu->staging(1, (size_t)nf * 4));
        size_t at = 0;
        for (int i = 0; i < nf; i++) {
            publish_ready(*r, (size_t)i, at);
            const uint64_t np = publish_slots(out + (size_t)i * OUT_STRIDE, len + i, 1, OUT_STRIDE);
            d_nPdu += np;
            if (d_debug && (size_t)i < r->dbg.size()) debug_lines(r->desc[(size_t)i], r->dbg[(size_t)i], np);
        }
        publish_ready(*r, (size_t)nf, at);
        r->ready.clear();
        r->desc.clear();
        r->dbg.clear();
        r->n = 0;
        r->state = round::FREE;
        d_flight.pop_front();
        wait_front = false;  // only the oldest is waited for
    }
    // channel messages that waited for rounds in flight only
    if (d_flight.empty() && d_fill && d_fill->desc.empty() && !d_fill->ready.empty()) {
        size_t at = 0;
        publish_ready(*d_fill, 0, at);
        d_fill->ready.clear();
    }
    return true;
}

bool decode::flush()
{
    if (!d_err.empty()) return false;
    if (!d_up.log().drain()) {
        d_err = "upload of the soft bits failed";
        return false;
    }
    // a packet cut off by the end of the stream is dropped, as the reference never finishes it either
    if (d_state == S_COLLECT && !d_fill->desc.empty()) d_fill->desc.pop_back();
    d_state = S_IDLE;
    if (!d_fill->desc.empty() && !cut()) return false;
    if (!advance(true, true)) return false;
    size_t at = 0;
    publish_ready(*d_fill, d_fill->desc.size(), at);
    d_fill->ready.clear();
    d_fill->n = 0;
    return true;
}

// ------------------------------------------------------------------------------------ rx
namespace {
// A frame's longest span: 224 samples to the end of L-SIG, then what the longest L-SIG length announces at the
// lowest rate (lib/signal_impl.cc:126-131): a sync byte further than this from the end of a round has its whole frame
// inside the round.
const size_t MAX_SPAN = 224 + 80 * ((4095 * 8 + 22 + 23) / 24);
}  // namespace

rx::sptr rx::make(int n_rx, int batch_frames, int device) { return sptr(new rx(n_rx, batch_frames, device)); }

rx::rx(int n_rx, int batch_frames, int device, int n_out, bool sync_port)
    : block((n_rx == 2 ? 2 : 1) + (sync_port ? 1 : 0), n_out), d_nrx(n_rx == 2 ? 2 : 1), d_batch(batch_frames < 1 ? 4096 : batch_frames),
      d_port0(sync_port ? 1 : 0), d_up(device), d_maxHold((size_t)1 << 24)
{
    d_adaptive = batch_frames < 1;
    if (!d_up.ok()) {
        d_err = "gr80211_create failed";
        return;
    }
    for (round &r : d_round) {
        r.gpu.reset(new gpu_session(device));
        if (!r.gpu->ok()) {
            d_err = "gr80211_create failed";
            return;
        }
    }
    d_fill = &d_round[0];
    d_fill->state = round::FILL;
    if (const char *e = getenv("GR80211_RX_MAX_HOLD")) set_max_hold((size_t)atoll(e));  // experiments
    d_profile = getenv("GR80211_BLOCK_PROFILE") != nullptr;
    // the arenas of the first two rounds exist before the stream starts (making one takes milliseconds); the others
    // follow when a stream is fast enough to need them
    for (int k = 0; k < 2; k++) {
        round &r = d_round[k];
        const size_t nf = (size_t)std::min(d_batch, 8192), np = nf + nf / 8;
        if (!reserve(r, 1) || !r.gpu->buffer(2, np * sizeof(gr80211_frame_info)) || !r.gpu->staging(2, np * sizeof(gr80211_frame_info)) ||
            !r.gpu->buffer(3, nf * OUT_STRIDE) || !r.gpu->buffer(4, nf * 4) || !r.gpu->buffer(5, nf * 4) ||
            !r.gpu->staging(0, nf * OUT_STRIDE) || !r.gpu->staging(1, nf * 4)) {
            d_err = "device memory for the held samples exhausted";
            return;
        }
    }
}

rx::~rx()
{
    if (getenv("GR80211_BLOCK_PROFILE"))
        fprintf(stderr, "rx block: %llu calls, %llu rounds; host seconds: scan %.4f upload %.4f settle %.4f advance %.4f "
                        "(of which cut %.4f publish %.4f)\n", (unsigned long long)d_nCalls, (unsigned long long)d_nRounds,
                d_tScan, d_tUpload, d_tSettle, d_tAdvance, d_tCut, d_tPublish);
    // nothing of this block may still be on its way when its buffers go
    d_up.log().drain();
    if (d_up.ok()) d_up.sync();
    for (round &r : d_round) {
        if (!r.gpu || !r.gpu->ok()) continue;
        r.gpu->sync();
        for (char *&q : r.iq)
            if (q) gr80211_dev_free(r.gpu->ctx(), q);
    }
    unpin_all();
}

void rx::set_ampdu_strict(bool on)
{
    d_strict = on;
    for (round &r : d_round)
        if (r.gpu) r.gpu->set_ampdu_strict(on);
}

void rx::set_max_hold(size_t items) { d_maxHold = std::max(items, 4 * MAX_SPAN); }

void rx::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    for (int &v : ninput_items_required) v = noutput_items;  // lib/signal_impl.cc:55-60
}

int rx::general_work(int, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    const int n = take_input(ninput_items, input_items);
    consume_each(n < 0 ? 0 : n);
    return 0;
}

bool rx::reserve(round &r, size_t items)
{
    if (items <= r.cap) return true;
    gr80211_ctx *c = r.gpu->ctx();
    // an arena is made for a whole round at once (what a round may hold, the longest carried-over stretch and a
    // window on top): growing it later means a copy and a wait
    size_t cap = std::max<size_t>(std::max<size_t>(2 * r.cap, items), std::min<size_t>(d_maxHold, (size_t)1 << 25) + 2 * MAX_SPAN + ((size_t)1 << 20));
    for (int a = 0; a < d_nrx; a++) {
        char *q = static_cast<char *>(gr80211_dev_alloc(c, (cap + 64) * sizeof(gr_complex)));
        if (!q) return false;
        // what the arena holds moves over once the copies that are filling it have run
        if (r.n && !d_up.sync()) return false;
        if (r.n && gr80211_copy_d2d_async(c, q, r.iq[a], r.n * sizeof(gr_complex), nullptr) != GR80211_OK) return false;
        if (r.iq[a]) {
            if (!r.gpu->sync()) return false;
            gr80211_dev_free(c, r.iq[a]);
        }
        r.iq[a] = q;
    }
    r.cap = cap;
    return true;
}

rx::round *rx::free_round()
{
    for (;;) {
        for (round &r : d_round)
            if (r.state == round::FREE) return &r;
        if (d_flight.empty() || !advance(true, false)) return nullptr;  // the oldest round in flight gives its place
    }
}

// The input half of general_work: the new part of the window goes to the round being filled (sync bytes to its list,
// samples to its arenas), rounds in flight move on as far as they can without waiting, and the items whose copies have
// run are consumed.
int rx::take_input(gr_vector_int &ninput_items, gr_vector_const_void_star &input_items)
{
    if (!d_err.empty()) return -1;
    const uint8_t *sync = static_cast<const uint8_t *>(input_items[0]);
    int n = std::min(ninput_items[0], ninput_items[1]);
    if (d_nrx == 2) n = std::min(n, ninput_items[2]);
    const uint64_t r0 = nitems_read(0);
    upload_log &log = d_up.log();
    if (log.enqueued() < r0) log.set_enqueued(r0);
    d_nCalls++;
    if (!d_flight.empty() && poll_due(d_lastPoll)) {  // asking the runtime about the rounds in flight costs a few us: at most every 100 us
        stopwatch sw(d_tAdvance, d_profile ? "advance" : nullptr);
        if (!advance(false, false)) return -1;
    }
    // items [lo, n) of the window are new; the ones before were taken by an earlier call and wait for their copies
    const int lo = (int)std::min<uint64_t>(log.enqueued() - r0, (uint64_t)n);
    if (lo < n) {
        round &f = *d_fill;
        // the new sync bytes with a tag become pending triggers
        auto scan = [&]() {
            stopwatch sw_scan(d_tScan, d_profile ? "scan" : nullptr);
            const size_t trig_before = f.trig.size();
            static const char *const keys[3] = {"rad", "snr", "rssi"};
            for (int i = next_set(sync, lo, n); i < n; i = next_set(sync, i + 1, n)) {
                const uint64_t at = r0 + (uint64_t)i;
                double v[3] = {0.0, 0.0, 0.0};
                if (!tag_values_at(0, at, keys, v, 3)) continue;  // "input sync with no tag" (lib/signal_impl.cc:96-100): not a trigger
                trigger t;
                t.pos = at;
                t.rad = (float)v[0];
                t.snr = (float)v[1];
                t.rssi = (float)v[2];
                f.trig.push_back(t);
                d_nTrig++;
            }
            f.n_new += f.trig.size() - trig_before;
        };
        // A round that holds samples takes all of the new ones, so they are sent on their way first and the sync bytes
        // are read while they travel; an empty round starts at its first sync byte, which has to be found first.
        const bool scan_first = f.n == 0;
        if (scan_first) scan();
        std::unique_ptr<stopwatch> sw_up(new stopwatch(d_tUpload, d_profile ? "upload" : nullptr));
        // samples are only held from the first pending sync byte on
        int from = lo;
        if (f.n == 0) {
            if (f.trig.empty()) {
                from = n;
            } else {
                const uint64_t first = std::max(f.trig[0].pos, r0 + (uint64_t)lo);
                from = (int)(first - r0);
            }
            f.base = r0 + (uint64_t)from;
        }
        if (from < n) {
            const size_t cnt = (size_t)(n - from);
            if (!reserve(f, f.n + cnt)) {
                d_err = "device memory for the held samples exhausted";
                return -1;
            }
            if (from > lo) log.add(nullptr, 0, r0 + (uint64_t)from);  // items passed over need no copy
            bool pinned = true;
            const char *src[2];
            char *dst[2];
            for (int a = 0; a < d_nrx; a++) {
                src[a] = reinterpret_cast<const char *>(static_cast<const gr_complex *>(input_items[d_port0 + a]) + from);
                dst[a] = f.iq[a] + f.n * sizeof(gr_complex);
                pinned = pinned && is_pinned(src[a], cnt * sizeof(gr_complex));
            }
            if (!d_up.send(*f.gpu, d_nrx, src, dst, cnt, sizeof(gr_complex), r0 + (uint64_t)from, pinned, d_err)) return -1;
            f.n += cnt;
        } else {
            log.add(nullptr, 0, r0 + (uint64_t)n);
        }
        sw_up.reset();
        if (!scan_first) scan();
        log.set_enqueued(r0 + (uint64_t)n);
    }
    {
        // a round is cut when enough sync bytes are pending (the frame a round carried over counts once its samples
        // have come in), or when it holds as many samples as it may
        round &f = *d_fill;
        const uint64_t end = f.base + f.n;
        // Made without a batch size, the block sizes its rounds by the traffic: a frame goes at once while fewer than two
        // rounds are on their way (a quiet channel: nothing ever waits for company), and what arrives while two are
        // goes together in the next one (a busy channel, or a recording played at full speed: rounds of hundreds) --
        // looked at in every call, also one that brings nothing new, since it is a round's return that lets it go.
        const size_t want = d_adaptive && d_flight.size() < 2 ? 1 : (size_t)d_batch;
        const bool enough = f.trig.size() >= want && (f.n_new > 0 || (d_wantEnd && end >= d_wantEnd));
        if (f.n > 0 && (enough || f.n >= d_maxHold)) {
            stopwatch sw_cut(d_tCut, d_profile ? "cut" : nullptr);
            if (!cut(false)) return -1;
        }
    }
    stopwatch sw_settle(d_tSettle, d_profile ? "settle" : nullptr);
    const uint64_t upto = log.settled(r0);
    return (int)(upto - r0);
}

// The round being filled goes on its way (probe of every sync byte's L-SIG, then the copy of the records to the
// host); the next round starts with the samples of the sync bytes whose frames may not have arrived in full.
bool rx::cut(bool final)
{
    round &r = *d_fill;
    gr80211_ctx *c = r.gpu->ctx();
    const uint64_t end = r.base + r.n;
    r.final = final;
    // whatever the round's stream does from here on comes after the uploads, which went by streams of their own
    if (!d_up.join(c)) {
        d_err = r.gpu->error();
        return false;
    }
    if (!before_cut(r, final)) return false;
    round *nx = nullptr;
    if (!final) {
        nx = free_round();
        if (!nx) {
            if (d_err.empty()) d_err = "no round free";
            return false;
        }
        // every sync byte within a frame's longest span of the end goes along: its frame may reach past the end
        size_t k = 0;
        while (k < r.trig.size() && r.trig[k].pos + MAX_SPAN <= end) k++;
        nx->trig.assign(r.trig.begin() + (long)k, r.trig.end());
        nx->n_new = 0;
        nx->n = 0;
        nx->n_carry = 0;
        nx->base = end;
        if (!nx->trig.empty() || d_carryBack) {
            uint64_t from = nx->trig.empty() ? end : std::max(nx->trig[0].pos, r.base);
            if (d_carryBack) from = std::min(from, end > r.base + d_carryBack ? end - d_carryBack : r.base);
            const size_t cnt = (size_t)(end - from);
            if (!reserve(*nx, cnt)) {
                d_err = "device memory for the held samples exhausted";
                return false;
            }
            for (int a = 0; a < d_nrx; a++)
                if (gr80211_copy_d2d_async(c, nx->iq[a], r.iq[a] + (size_t)(from - r.base) * sizeof(gr_complex),
                                           cnt * sizeof(gr_complex), nullptr) != GR80211_OK) {
                    d_err = r.gpu->error();
                    return false;
                }
            // the next round's stream starts behind this copy
            if (gr80211_stream_wait(nx->gpu->ctx(), nullptr, c, nullptr) != GR80211_OK) {
                d_err = nx->gpu->error();
                return false;
            }
            nx->base = from;
            nx->n = cnt;
            nx->n_carry = cnt;
        }
        nx->state = round::FILL;
    }
    // ---- probe: L-SIG of every sync byte that has its 224 samples (lib/signal_impl.cc:108-131)
    r.pd.clear();
    r.p_of.assign(r.trig.size(), -1);
    for (size_t j = 0; j < r.trig.size(); j++) {
        const trigger &t = r.trig[j];
        if (t.pos < r.base) continue;
        const uint64_t rel = t.pos - r.base;
        if (rel + LSIG_SPAN > r.n) continue;  // `signal` waits for 224 samples too
        gr80211_frame_desc d;
        memset(&d, 0, sizeof(d));
        d.sync_off = rel;
        d.n_avail = (uint32_t)std::min<uint64_t>(r.n - rel, 0xffffffffu);
        d.rad = t.rad;
        d.snr = t.snr;
        d.rssi = t.rssi;
        d.mu_pos = (uint32_t)d_muPos;
        r.p_of[j] = (int32_t)r.pd.size();
        r.pd.push_back(d);
    }
    if (!r.pd.empty()) {
        const int np = (int)r.pd.size();
        gr80211_frame_info *dinfo = static_cast<gr80211_frame_info *>(r.gpu->buffer(2, (size_t)np * sizeof(gr80211_frame_info)));
        r.info = static_cast<gr80211_frame_info *>(r.gpu->staging(2, (size_t)np * sizeof(gr80211_frame_info)));
        if (!dinfo || !r.info ||
            gr80211_signal_batch(c, r.iq[0], d_nrx == 2 ? r.iq[1] : nullptr, r.n, r.pd.data(), np, 0, dinfo, nullptr) != GR80211_OK ||
            gr80211_copy_d2h_async(c, r.info, dinfo, (size_t)np * sizeof(gr80211_frame_info), nullptr) != GR80211_OK) {
            d_err = "probe of the sync bytes failed: " + r.gpu->error();
            return false;
        }
    }
    r.state = round::PROBE;
    r.t_cut = now();
    if (d_profile) fprintf(stderr, "  cut of round %d at item %llu (%.4f s)\n", (int)(&r - d_round), (unsigned long long)end, r.t_cut);
    d_flight.push_back(&r);
    d_nRounds++;
    d_fill = nx;
    return final ? true : advance(false, false);
}

// `signal`'s trigger search over one round, in stream order (lib/signal_impl.cc:75-106,153-160,164-202): the frames it
// deals with go to r.fd.  Sync bytes it cannot settle yet (their frames reach past the round's end) were carried
// over to the next round, which meets them again.
bool rx::replay(round &r)
{
    const uint64_t end = r.base + r.n;
    r.fd.clear();
    r.fidx.clear();
    r.fr.clear();
    uint64_t want = 0;
    for (size_t j = 0; j < r.trig.size(); j++) {
        const trigger &t = r.trig[j];
        if (t.pos < d_cursor) continue;  // inside the previous frame's span: `signal` never looks at it
        const int32_t k = r.p_of[j];
        if (k < 0) {  // fewer than 224 samples in this round: the next one has it, or the stream ends here
            if (!r.final) want = t.pos + LSIG_SPAN;
            break;
        }
        const gr80211_frame_info &fi = r.info[k];
        if (fi.status == GR80211_ST_DROP_LSIG) {
            d_cursor = t.pos + LSIG_FAIL_SKIP;
            continue;
        }
        const int nsamp = fi.nsamp > 0 ? fi.nsamp : 0;
        const uint64_t span = (uint64_t)LSIG_SPAN + (uint64_t)nsamp;
        if (t.pos + span > end && !r.final) {  // its samples are still arriving
            want = t.pos + span;
            break;
        }
        gr80211_frame_desc d = r.pd[(size_t)k];
        d.n_avail = (uint32_t)std::min<uint64_t>(span, end - t.pos);
        r.fd.push_back(d);
        r.fidx.push_back(k);
        frame_ref fr;
        fr.pos = t.pos;
        fr.mcs = fi.lsig_mcs;
        fr.len = fi.lsig_len;
        fr.nsamp = nsamp;
        fr.rad = t.rad;
        fr.snr = t.snr;
        fr.rssi = t.rssi;
        r.fr.push_back(fr);
        d_cursor = t.pos + span;
    }
    d_wantEnd = want;
    r.t_replay = now();
    if (!r.fd.empty() && !start_frames(r)) return false;
    r.state = round::DECODE;
    return true;
}

// Moves the rounds in flight on: a probed round whose records have arrived is searched and its frames are started (in
// stream order: the search position carries over from round to round), the oldest round is published when its stream
// has run dry.  wait_front: wait for the oldest round and retire it; wait_all: retire all.
bool rx::advance(bool wait_front, bool wait_all)
{
    // 1. searches, oldest round first
    for (round *r : d_flight) {
        if (r->state != round::PROBE) continue;
        if (!r->gpu->idle()) {
            if (!(wait_all || (wait_front && r == d_flight.front()))) break;
            if (!r->gpu->sync()) {
                d_err = r->gpu->error();
                return false;
            }
        }
        if (!replay(*r)) return false;
    }
    // 2. the oldest rounds leave
    while (!d_flight.empty() && d_flight.front()->state == round::DECODE) {
        round *r = d_flight.front();
        if (!r->gpu->idle()) {
            if (!(wait_front || wait_all)) break;
            if (!r->gpu->sync()) {
                d_err = r->gpu->error();
                return false;
            }
        }
        if (!r->fd.empty() && !finish_frames(*r)) return false;
        if (d_profile)
            fprintf(stderr, "round %d: %zu items, %zu frames; cut %.4f searched %.4f published %.4f (in flight with %zu)\n",
                    (int)(r - d_round), r->n, r->fd.size(), r->t_cut, r->t_replay, now(), d_flight.size() - 1);
        r->trig.clear();
        r->n = 0;
        r->n_new = 0;
        r->state = round::FREE;
        d_flight.pop_front();
        wait_front = false;  // only the oldest is waited for
    }
    return true;
}

bool rx::flush()
{
    if (!d_err.empty()) return false;
    if (!d_up.log().drain()) {
        d_err = "upload of the samples failed";
        return false;
    }
    if (d_fill && (d_fill->n > 0 || !d_fill->trig.empty())) {
        if (!cut(true)) return false;
    }
    if (!advance(true, true)) return false;
    // ready for another stream
    if (!d_fill) {
        d_fill = &d_round[0];
        d_fill->state = round::FILL;
        d_fill->n = 0;
        d_fill->n_new = 0;
        d_fill->n_carry = 0;
        d_fill->trig.clear();
    }
    return true;
}

// rx: demodulator and decoder for the round's frames, then the PDU slots on their way back; nothing is waited for
bool rx::start_frames(round &r)
{
    gr80211_ctx *c = r.gpu->ctx();
    const int nf = (int)r.fd.size();
    uint8_t *dout = static_cast<uint8_t *>(r.gpu->buffer(3, (size_t)nf * OUT_STRIDE));
    int32_t *dlen = static_cast<int32_t *>(r.gpu->buffer(4, (size_t)nf * 4));
    int32_t *dnp = static_cast<int32_t *>(r.gpu->buffer(5, (size_t)nf * 4));
    uint8_t *out = static_cast<uint8_t *>(r.gpu->staging(0, (size_t)nf * OUT_STRIDE));
    int32_t *len = static_cast<int32_t *>(r.gpu->staging(1, (size_t)nf * 4));
    if (!dout || !dlen || !dnp || !out || !len) {
        d_err = "allocation of the output buffers failed";
        return false;
    }
    if (gr80211_rx_resume(c, r.iq[0], d_nrx == 2 ? r.iq[1] : nullptr, r.n, r.fd.data(), r.fidx.data(), r.info, nf, dout,
                          OUT_STRIDE, dlen, dnp, nullptr, nullptr) != GR80211_OK ||
        gr80211_push_d2h(c, len, dlen, (size_t)nf * 4, nullptr) != GR80211_OK ||
        gr80211_push_d2h(c, out, dout, (size_t)nf * OUT_STRIDE, nullptr) != GR80211_OK) {
        d_err = r.gpu->error();
        return false;
    }
    return true;
}

bool rx::finish_frames(round &r)
{
    const int nf = (int)r.fd.size();
    if (gr80211_last_batch_refused(r.gpu->ctx()) != 0) {
        d_err = "the GPU refused a batch: the frame records on the host do not describe it";
        return false;
    }
    const uint8_t *out = static_cast<const uint8_t *>(r.gpu->staging(0, (size_t)nf * OUT_STRIDE));
    const int32_t *len = static_cast<const int32_t *>(r.gpu->staging(1, (size_t)nf * 4));
    d_nFrames += (uint64_t)nf;
    stopwatch sw(d_tPublish);
    d_nPdu += publish_slots(out, len, nf, OUT_STRIDE);
    return true;
}

// ------------------------------------------------------------------------------------ signal
signal::sptr signal::make(int n_rx, int batch_frames, int device) { return sptr(new signal(n_rx, batch_frames, device)); }

signal::signal(int n_rx, int batch_frames, int device)
    : rx(n_rx, batch_frames, device, n_rx == 2 ? 2 : 1), d_out0(sizeof(gr_complex)), d_out1(sizeof(gr_complex))
{
}

void signal::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    // items held back for lack of room need no input to leave
    for (int &v : ninput_items_required) v = (d_out0.pending() || d_direct.in_flight()) ? 0 : noutput_items;
}

int signal::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                         gr_vector_void_star &output_items)
{
    // input is only taken while the items already produced find room downstream
    const int n = d_out0.pending() > OUT_QUEUE_LIMIT ? 0 : take_input(ninput_items, input_items);
    consume_each(n < 0 ? 0 : n);
    if ((int)output_items.size() < d_nrx) return 0;
    if (d_outMode == 0 && noutput_items > 0) {
        bool pinned = true;
        for (int a = 0; a < d_nrx; a++) pinned = pinned && is_pinned(output_items[(size_t)a], (size_t)noutput_items * sizeof(gr_complex));
        bool direct = pinned && (size_t)noutput_items * sizeof(gr_complex) >= 2 * DIRECT_OUT_MIN_BYTES;
        if (const char *e = getenv("GR80211_SIGNAL_DIRECT_OUT")) direct = e[0] != '0' && pinned;  // experiments
        d_outMode = direct ? 2 : 1;
    }
    if (d_outMode == 2) {
        out_stream *q[2] = {&d_out0, &d_out1};
        void *dst[2] = {output_items[0], d_nrx == 2 ? output_items[1] : nullptr};
        const uint64_t wr[2] = {nitems_written(0), d_nrx == 2 ? nitems_written(1) : 0};
        const int produced = d_direct.work(q, dst, wr, d_nrx, sizeof(gr_complex), noutput_items, n <= 0,
                                           [this](int port, const tag_t &t) { if (port == 0) add_item_tag(0, t); }, d_err);
        return produced < 0 ? 0 : produced;
    }
    std::vector<tag_t> tags;
    const int produced = d_out0.pop(output_items[0], noutput_items, nitems_written(0), tags);
    for (const tag_t &t : tags) add_item_tag(0, t);
    if (d_nrx == 2) {
        tags.clear();
        d_out1.pop(output_items[1], produced, nitems_written(1), tags);  // both queues always hold the same count
    }
    return produced;
}

// signal: the round's items are written and sent to page-locked memory of the round; nothing is waited for
bool signal::start_frames(round &r)
{
    // the round's own context runs the batch: its stream holds the samples' copies
    gpu_session &d_gpu = *r.gpu;
    const int ri = (int)(&r - d_round);
    const void *diq0 = r.iq[0], *diq1 = d_nrx == 2 ? r.iq[1] : nullptr;
    const size_t nheld = r.n;
    std::vector<gr80211_frame_desc> &fd = r.fd;
    const std::vector<frame_ref> &fr = r.fr;
    gr80211_ctx *c = d_gpu.ctx();
    const int nf = (int)fd.size();
    std::vector<uint64_t> &off = d_off[ri];
    off.assign((size_t)nf + 1, 0);
    for (int i = 0; i < nf; i++) off[(size_t)i + 1] = off[(size_t)i] + (uint64_t)fr[(size_t)i].nsamp + 320u;
    const uint64_t total = off[(size_t)nf];
    // items of this round's last turn that the scheduler has not taken yet move out of the memory about to be rewritten
    if (d_left[ri]) {
        d_out0.release_refs(d_span[ri][0].lo, d_span[ri][0].hi);
        d_out1.release_refs(d_span[ri][1].lo, d_span[ri][1].hi);
    }
    void *do0 = d_gpu.buffer(3, (size_t)(total + 64) * sizeof(gr_complex));
    void *do1 = d_nrx == 2 ? d_gpu.buffer(4, (size_t)(total + 64) * sizeof(gr_complex)) : nullptr;
    void *dch = d_gpu.buffer(5, (size_t)nf * 64 * sizeof(gr_complex));
    gr_complex *h0 = static_cast<gr_complex *>(d_gpu.staging(0, (size_t)total * sizeof(gr_complex) + 8));
    gr_complex *h1 = d_nrx == 2 ? static_cast<gr_complex *>(d_gpu.staging(1, (size_t)total * sizeof(gr_complex) + 8)) : nullptr;
    gr_complex *hch = static_cast<gr_complex *>(d_gpu.staging(3, (size_t)nf * 64 * sizeof(gr_complex)));
    if (!do0 || (d_nrx == 2 && (!do1 || !h1)) || !dch || !h0 || !hch) {
        d_err = "allocation of the output buffers failed";
        return false;
    }
    const bool dev_out = d_outMode == 2;  // the items stay on the device until the scheduler makes room
    d_devOut[ri] = dev_out;
    const gr_complex *s0 = dev_out ? static_cast<const gr_complex *>(do0) : h0, *s1 = dev_out ? static_cast<const gr_complex *>(do1) : h1;
    d_span[ri][0].lo = reinterpret_cast<const char *>(s0);
    d_span[ri][0].hi = reinterpret_cast<const char *>(s0 + total) + 8;
    d_span[ri][1].lo = reinterpret_cast<const char *>(s1);
    d_span[ri][1].hi = s1 ? reinterpret_cast<const char *>(s1 + total) + 8 : nullptr;
    if (gr80211_signal_emit_batch(c, diq0, diq1, nheld, fd.data(), nf, 0, off.data(), do0, do1, total, nullptr, dch,
                                  nullptr) != GR80211_OK ||
        (!dev_out && gr80211_push_d2h(c, h0, do0, (size_t)total * sizeof(gr_complex), nullptr) != GR80211_OK) ||
        (!dev_out && d_nrx == 2 && gr80211_push_d2h(c, h1, do1, (size_t)total * sizeof(gr_complex), nullptr) != GR80211_OK) ||
        gr80211_push_d2h(c, hch, dch, (size_t)nf * 64 * sizeof(gr_complex), nullptr) != GR80211_OK) {
        d_err = d_gpu.error();
        return false;
    }
    return true;
}

// signal: the round's stream has run dry, its items and channel estimates are in the round's page-locked memory
bool signal::finish_frames(round &r)
{
    gpu_session &d_gpu = *r.gpu;
    const int ri = (int)(&r - d_round);
    const std::vector<frame_ref> &fr = r.fr;
    const int nf = (int)r.fd.size();
    const std::vector<uint64_t> &off = d_off[ri];
    const uint64_t total = off[(size_t)nf];
    const gr_complex *h0 = static_cast<const gr_complex *>(d_gpu.staging(0, (size_t)total * sizeof(gr_complex) + 8));
    const gr_complex *h1 = d_nrx == 2 ? static_cast<const gr_complex *>(d_gpu.staging(1, (size_t)total * sizeof(gr_complex) + 8)) : nullptr;
    const gr_complex *hch = static_cast<const gr_complex *>(d_gpu.staging(3, (size_t)nf * 64 * sizeof(gr_complex)));
    const gr_complex *do0 = static_cast<const gr_complex *>(d_gpu.buffer(3, (size_t)(total + 64) * sizeof(gr_complex)));
    const gr_complex *do1 = d_nrx == 2 ? static_cast<const gr_complex *>(d_gpu.buffer(4, (size_t)(total + 64) * sizeof(gr_complex))) : nullptr;
    for (int i = 0; i < nf; i++) {
        const frame_ref &r = fr[(size_t)i];
        // lib/signal_impl.cc:133-152: the tags sit on the frame's first output item
        d_seq++;
        if (d_seq >= 1000000000) d_seq = 0;
        tag_t t;
        t.offset = d_out0.pushed();
        t.key = "cfo";   t.value = (double)(r.rad * 3183098.8618379068f); d_out0.tag(t);
        t.key = "snr";   t.value = (double)r.snr; d_out0.tag(t);
        t.key = "rssi";  t.value = (double)r.rssi; d_out0.tag(t);
        t.key = "seq";   t.value = (double)d_seq; d_out0.tag(t);
        t.key = "mcs";   t.value = (double)r.mcs; d_out0.tag(t);
        t.key = "len";   t.value = (double)r.len; d_out0.tag(t);
        t.key = "nsamp"; t.value = (double)r.nsamp; d_out0.tag(t);
        t.key = "chan";  t.value = 0.0;
        t.cvec.assign(hch + (size_t)i * 64, hch + (size_t)i * 64 + 64);
        d_out0.tag(t);
        const size_t cnt = (size_t)r.nsamp + 320;
        if (d_devOut[ri]) {
            d_out0.push_dev(&d_gpu, do0 + off[(size_t)i], cnt, &d_left[ri]);
            if (d_nrx == 2) d_out1.push_dev(&d_gpu, do1 + off[(size_t)i], cnt, &d_left[ri]);
            continue;
        }
        d_out0.push_ref(h0 + off[(size_t)i], cnt, &d_left[ri]);
        if (d_nrx == 2) d_out1.push_ref(h1 + off[(size_t)i], cnt, &d_left[ri]);
    }
    d_nFrames += (uint64_t)nf;
    return true;
}

// ------------------------------------------------------------------------------------ demod
demod::sptr demod::make(int n_rx, int mupos, int, int batch_frames, int device)
{
    return sptr(new demod(n_rx, mupos, batch_frames, device));
}

demod::demod(int n_rx, int mupos, int batch_frames, int device)
    : block(n_rx == 2 ? 2 : 1, 1), d_nrx(n_rx == 2 ? 2 : 1), d_muPos(mupos ? 1 : 0), d_batch(batch_frames < 1 ? 4096 : batch_frames),
      d_device(device), d_out(sizeof(float)), d_up(new uploader(device))
{
    d_finish = batch_frames < 1;  // made like the reference's block: timed like it
    if (!d_up->ok()) d_err = "gr80211_create failed";
    // this block's thread also hands soft bits on: windows of 64 k items (512 KB) copied by hand leave it free sooner
    // than pieces it has to see arrive (three-block chain 342 -> 385 Msamples/s)
    d_up->set_stage_max_pinned((size_t)600 << 10);
    for (int i = 0; i < N_ROUNDS; i++) {
        d_round.emplace_back(new round());
        d_round.back()->gpu.reset(new gpu_session(device));
        if (!d_round.back()->gpu->ok()) d_err = "gr80211_create failed";
    }
}

demod::~demod()
{
    if (getenv("GR80211_BLOCK_PROFILE"))
        fprintf(stderr, "demod block: %llu frames, %llu rounds; host seconds: harvest %.4f tags %.4f copy in %.4f soft bits out %.4f other %.4f\n",
                (unsigned long long)d_nSeen, (unsigned long long)d_nRounds, d_tHarvest, d_tTags, d_tCopyIn, d_tPop, d_tOther);
    // nothing of this block may still be on its way when its buffers go
    d_up->log().drain();
    if (d_up->ok()) d_up->sync();
    for (auto &r : d_round) {
        if (r->worker.joinable()) r->worker.join();
        for (int a = 0; a < 2; a++)
            if (r->iq[a] && r->gpu->ok()) gr80211_host_free(r->gpu->ctx(), r->iq[a]);
    }
    unpin_all();
}

void demod::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    // lib/demod_impl.cc:53-57; soft bits held back for lack of room need no input to leave
    for (int &v : ninput_items_required) v = (d_out.pending() || d_direct.in_flight()) ? 0 : noutput_items;
}

demod::round *demod::free_round()
{
    for (auto &r : d_round)
        if (r->state.load() == R_FREE) return r.get();
    if ((int)d_round.size() >= MAX_ROUNDS) return nullptr;
    d_round.emplace_back(new round());
    d_round.back()->gpu.reset(new gpu_session(d_device));
    if (!d_round.back()->gpu->ok()) {
        d_round.pop_back();
        return nullptr;
    }
    return d_round.back().get();
}

// Room for `items` items per antenna in the round's page-locked memory (what it holds moves along).
bool demod::grow(round &r, size_t items)
{
    if (items <= r.cap) return true;
    const size_t cap = std::max<size_t>(std::max<size_t>(2 * r.cap, items), (size_t)1 << 20);
    gr80211_ctx *c = r.gpu->ctx();
    for (int a = 0; a < d_nrx; a++) {
        gr_complex *q = static_cast<gr_complex *>(gr80211_host_alloc(c, cap * sizeof(gr_complex)));
        if (!q) {
            d_err = "page-locked host memory exhausted";
            return false;
        }
        if (r.n) memcpy(q, r.iq[a], r.n * sizeof(gr_complex));
        if (r.iq[a]) gr80211_host_free(c, r.iq[a]);
        r.iq[a] = q;
    }
    r.cap = cap;
    return true;
}

int demod::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                        gr_vector_void_star &output_items)
{
    int n = ninput_items[0];
    if (d_nrx == 2) n = std::min(n, ninput_items[1]);
    static const bool prof = getenv("GR80211_BLOCK_PROFILE") != nullptr;
    auto now = [] { return prof ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0; };
    double t0 = now();
    auto lap = [&](double &acc) { if (prof) { const double t = now(); acc += t - t0; t0 = t; } };
    const bool harvested = harvest(false);
    lap(d_tHarvest);
    if (!harvested) {
        consume_each(0);
        return 0;
    }
    if (d_mode == M_UNKNOWN && n > 0) {
        // windows in page-locked memory go to the GPU as they are; others are collected frame by frame first
        bool pinned = d_up->ok();
        for (int a = 0; a < d_nrx; a++) pinned = pinned && is_pinned(input_items[(size_t)a], (size_t)n * sizeof(gr_complex));
        if (const char *e = getenv("GR80211_DEMOD_DIRECT")) pinned = pinned && e[0] != '0';  // experiments
        // (a round that is to be finished within the call collects its frames by hand: it is a frame or two)
        d_mode = pinned && !d_finish ? M_DIRECT : M_STAGED;
    }
    if (d_out.pending() > OUT_QUEUE_LIMIT) n = 0;  // input is only taken while the soft bits produced find room
    int used = 0;
    std::vector<tag_t> tags;
    if (d_mode == M_DIRECT) {
        used = fill_direct(n, input_items);
        lap(d_tCopyIn);
        if (used < 0) {
            consume_each(0);
            return 0;
        }
        n = 0;  // the loop below is the other mode's
    }
    while (used < n) {
        if (!d_fill) {
            d_fill = free_round();
            if (!d_fill) break;  // every round is on its way: the input waits
            d_fill->state.store(R_FILL);
            d_fill->n = 0;
            d_fill->direct = false;
            d_fill->desc.clear();
            d_fill->chan.clear();
        }
        round &r = *d_fill;
        if (d_state == S_IDLE) {
            const uint64_t at = nitems_read(0) + (uint64_t)used;
            get_tags_in_range(tags, 0, at, at + 1);
            if (tags.empty()) {
                // the reference waits on an untagged item for ever (lib/demod_impl.cc:72-104); here it is passed over
                get_tags_in_range(tags, 0, at + 1, nitems_read(0) + (uint64_t)n);
                if (tags.empty()) {
                    used = n;
                    break;
                }
                used = (int)(tags[0].offset - nitems_read(0));
                continue;
            }
            lap(d_tOther);
            gr80211_sig_desc d;
            memset(&d, 0, sizeof(d));
            d.off = r.n;
            d.mcs = (int32_t)tag_value(tags, "mcs", -1.0);
            d.len = (int32_t)tag_value(tags, "len", -1.0);
            d.nsamp = (int32_t)tag_value(tags, "nsamp", -1.0);
            d.cfo = (float)tag_value(tags, "cfo", 0.0);
            d.snr = (float)tag_value(tags, "snr", 0.0);
            d.rssi = (float)tag_value(tags, "rssi", 0.0);
            d.mu_pos = (uint32_t)d_muPos;
            const tag_t *ch = tag_find(tags, "chan");
            const bool usable = !(d.nsamp < 0 || d.nsamp > (1 << 24) || !ch || ch->cvec.size() < 64);
            int running = 0;
            for (const round *q : d_flight) running += q->state.load() == R_RUN;
            const bool cut_ahead = usable && d_finish && !r.desc.empty() && d.nsamp > n - used && running < 2;
            if (!cut_ahead) d_nSeen++;  // (else the item is looked at again below)
            if (!usable) {
                // nothing a frame can be taken from: pass over this item and look for the next tagged one
                used++;
                continue;
            }
            if (cut_ahead) {
                // this frame ends in a later call: the ones before it do not wait for it (the item is looked at again),
                // nor does the call wait for them -- the next one is certain and near.  (While two rounds are running
                // they do wait, for company: a stream that comes faster than rounds return gets larger rounds.)
                d_cutAhead = true;
                const bool ok = cut();
                d_cutAhead = false;
                if (!ok) {
                    consume_each(used);
                    return 0;
                }
                continue;
            }
            r.desc.push_back(d);
            r.chan.insert(r.chan.end(), ch->cvec.begin(), ch->cvec.begin() + 64);
            d_need = d.nsamp;
            d_skip = 320;  // lib/demod_impl.cc:92
            d_state = S_COLLECT;
            lap(d_tTags);
        }
        if (d_state == S_COLLECT) {
            const int take = (int)std::min<long>(d_need, n - used);
            if (!grow(r, r.n + (size_t)take + 64)) {
                consume_each(used);
                return 0;
            }
            for (int a = 0; a < d_nrx; a++)
                memcpy(r.iq[a] + r.n, static_cast<const gr_complex *>(input_items[(size_t)a]) + used, (size_t)take * sizeof(gr_complex));
            r.n += (size_t)take;
            lap(d_tCopyIn);
            r.desc.back().n_avail += (uint32_t)take;
            d_need -= take;
            used += take;
            if (d_need == 0) d_state = S_CLEAN;
        }
        if (d_state == S_CLEAN) {
            const int take = (int)std::min<long>(d_skip, n - used);
            d_skip -= take;
            used += take;
            if (d_skip == 0) {
                d_state = S_IDLE;
                if ((int)r.desc.size() >= d_batch && !cut()) {
                    consume_each(used);
                    return 0;
                }
            }
        }
    }
    if (d_mode != M_DIRECT) consume_each(used);
    if (d_finish && d_mode != M_DIRECT) {
        // what this call completed is demodulated before it returns (the 320 items passed over after a frame carry
        // nothing: the frame is complete with its last sample)
        if (d_fill && d_state != S_COLLECT && !d_fill->desc.empty() && !cut()) return 0;
        if (!d_flight.empty() && !harvest(d_state != S_COLLECT)) return 0;
    }
    n = ninput_items[0];
    if (output_items.empty()) return 0;
    tags.clear();
    lap(d_tOther);
    if (d_outMode.load() == 0 && noutput_items > 0) {
        // decided by the first output buffer: soft bits travel from the device straight into page-locked buffers
        // (buffers that take a megabyte at a time: below that a call's share of launches and completion queries costs
        // more than copying the items by hand, 250 against 285 Msamples/s through the three-block chain at 256-KB windows)
        bool direct = (size_t)noutput_items * sizeof(float) >= DIRECT_OUT_MIN_BYTES &&
                      is_pinned(output_items[0], (size_t)noutput_items * sizeof(float));
        if (const char *e = getenv("GR80211_DEMOD_DIRECT_OUT")) direct = e[0] != '0' && is_pinned(output_items[0], (size_t)noutput_items * sizeof(float));  // experiments
        d_outMode.store(direct ? 2 : 1);
    }
    int produced = 0;
    // pieces of input on their way: a call that has nothing else to report waits for the oldest of them (below), which
    // takes microseconds, rather than for a round on the GPU
    const bool uploads_pending = d_mode == M_DIRECT && !d_up->log().empty();
    if (d_outMode.load() == 2) {
        out_stream *q[1] = {&d_out};
        void *dst[1] = {output_items[0]};
        const uint64_t wr[1] = {nitems_written(0)};
        const auto tagger = [this](int port, const tag_t &t) { add_item_tag(port, t); };
        produced = d_direct.work(q, dst, wr, 1, sizeof(float), noutput_items, used == 0, tagger, d_err);
        if (produced < 0) return 0;
        if (used == 0 && produced == 0 && n > 0 && noutput_items > 0 && !d_flight.empty() && d_direct.in_flight() == 0 && !uploads_pending) {
            // neither input taken nor output to give while rounds are on their way: wait for the oldest (see below)
            if (!harvest_front()) return 0;
            produced = d_direct.work(q, dst, wr, 1, sizeof(float), noutput_items, true, tagger, d_err);
            if (produced < 0) return 0;
        }
    } else {
        produced = d_out.pop(output_items[0], noutput_items, nitems_written(0), tags);
        if (used == 0 && produced == 0 && n > 0 && noutput_items > 0 && !d_flight.empty() && !uploads_pending) {
            // Neither input taken (every round is on its way) nor output to give: the one case in which the call waits,
            // for the oldest round, so that a caller never sees a call without progress while work is in flight (a
            // scheduler would read that as the end of the stream).
            if (!harvest_front()) return 0;
            produced = d_out.pop(output_items[0], noutput_items, nitems_written(0), tags);
        }
        for (const tag_t &t : tags) add_item_tag(0, t);
    }
    lap(d_tPop);
    if (d_mode == M_DIRECT && used == 0 && produced == 0 && !d_up->log().empty()) {
        used = settle_direct();  // nothing else moved: wait for the oldest piece on its way
        lap(d_tCopyIn);
    }
    // rounds whose soft bits have all left are free again
    while (!d_flight.empty() && d_flight.front()->state.load() == R_DRAIN && d_flight.front()->left == 0) {
        d_flight.front()->state.store(R_FREE);
        d_flight.pop_front();
    }
    return produced;
}

// Page-locked input windows: the frames' items travel to the round's device arena straight from the window, one
// asynchronous piece per call (the arena mirrors the stream from the round's first frame on, padding and all), and the
// call consumes the items whose pieces have arrived.  The state machine runs ahead of what has been consumed: it
// resumes where the last call's pieces ended.  Returns the items consumed, -1 on an error.
int demod::fill_direct(int n, gr_vector_const_void_star &input_items)
{
    const uint64_t r0 = nitems_read(0);
    upload_log &log = d_up->log();
    if (log.enqueued() < r0) log.set_enqueued(r0);
    int used = (int)std::min<uint64_t>(log.enqueued() - r0, (uint64_t)n);
    int span = used;  // window items [span, used) have been looked at and not sent yet
    // window items [span, upto) go where they belong: to the arena of the round being filled, or nowhere
    auto send_span = [&](int upto) -> bool {
        if (upto <= span) return true;
        round *r = d_fill;
        if (!r || !r->direct) {
            log.add(nullptr, 0, r0 + (uint64_t)upto);  // no frame wants them: free at once
            span = upto;
            return true;
        }
        const size_t cnt = (size_t)(upto - span);
        const uint64_t at = r0 + (uint64_t)span - r->base;
        bool pinned = true;
        const char *src[2];
        char *dst[2];
        for (int a = 0; a < d_nrx; a++) {
            src[a] = reinterpret_cast<const char *>(static_cast<const gr_complex *>(input_items[(size_t)a]) + span);
            dst[a] = r->dev[a] + at * sizeof(gr_complex);
            pinned = pinned && is_pinned(src[a], cnt * sizeof(gr_complex));
        }
        if (!d_up->send(*r->gpu, d_nrx, src, dst, cnt, sizeof(gr_complex), r0 + (uint64_t)span, pinned, d_err)) return false;
        r->n = (size_t)(at + cnt);
        span = upto;
        return true;
    };
    std::vector<tag_t> tags;
    while (used < n) {
        if (!d_fill) {
            if (!send_span(used)) return -1;
            d_fill = free_round();
            if (!d_fill) break;  // every round is on its way: the input waits
            d_fill->state.store(R_FILL);
            d_fill->n = 0;
            d_fill->direct = false;  // until its first frame
            d_fill->desc.clear();
            d_fill->chan.clear();
        }
        round &r = *d_fill;
        if (d_state == S_IDLE) {
            const uint64_t at = r0 + (uint64_t)used;
            get_tags_in_range(tags, 0, at, at + 1);
            if (tags.empty()) {
                get_tags_in_range(tags, 0, at + 1, r0 + (uint64_t)n);
                used = tags.empty() ? n : (int)(tags[0].offset - r0);
                continue;
            }
            gr80211_sig_desc d;
            memset(&d, 0, sizeof(d));
            d.mcs = (int32_t)tag_value(tags, "mcs", -1.0);
            d.len = (int32_t)tag_value(tags, "len", -1.0);
            d.nsamp = (int32_t)tag_value(tags, "nsamp", -1.0);
            d.cfo = (float)tag_value(tags, "cfo", 0.0);
            d.snr = (float)tag_value(tags, "snr", 0.0);
            d.rssi = (float)tag_value(tags, "rssi", 0.0);
            d.mu_pos = (uint32_t)d_muPos;
            const tag_t *ch = tag_find(tags, "chan");
            if (d.nsamp < 0 || d.nsamp > (1 << 24) || !ch || ch->cvec.size() < 64) {
                d_nSeen++;
                used++;
                continue;
            }
            const size_t need = (size_t)d.nsamp + 320 + 64;
            if (r.direct && (size_t)(at - r.base) + need > r.dev_cap) {
                // the arena is full: this frame opens the next round
                if (!send_span(used) || !cut()) return -1;
                continue;
            }
            if (!r.direct) {
                if (!send_span(used)) return -1;  // what came before the round's first frame is nobody's
                const size_t cap = std::max<size_t>(need, std::min<size_t>(std::max<size_t>((size_t)d_batch * 4096, (size_t)1 << 20), (size_t)1 << 25));
                for (int a = 0; a < d_nrx; a++) {
                    r.dev[a] = static_cast<char *>(r.gpu->buffer(a, (cap + 64) * sizeof(gr_complex)));
                    if (!r.dev[a]) {
                        d_err = "device memory for the frames' items exhausted";
                        return -1;
                    }
                }
                r.dev_cap = cap;
                r.base = at;
                r.direct = true;
            }
            d_nSeen++;
            d.off = at - r.base;
            r.desc.push_back(d);
            r.chan.insert(r.chan.end(), ch->cvec.begin(), ch->cvec.begin() + 64);
            d_need = d.nsamp;
            d_skip = 320;  // lib/demod_impl.cc:92
            d_state = S_COLLECT;
        }
        if (d_state == S_COLLECT) {
            const int take = (int)std::min<long>(d_need, n - used);
            r.desc.back().n_avail += (uint32_t)take;
            d_need -= take;
            used += take;
            if (d_need == 0) d_state = S_CLEAN;
        }
        if (d_state == S_CLEAN) {
            const int take = (int)std::min<long>(d_skip, n - used);
            d_skip -= take;
            used += take;
            if (d_skip == 0) {
                d_state = S_IDLE;
                if ((int)r.desc.size() >= d_batch && (!send_span(used) || !cut())) return -1;
            }
        }
    }
    if (!send_span(used)) return -1;
    if (log.enqueued() < r0 + (uint64_t)used) log.set_enqueued(r0 + (uint64_t)used);
    // what has arrived; the caller waits for the oldest piece (settle_direct) if the call would otherwise end without
    // progress, and not while there are soft bits to hand on
    const int consumed = (int)(log.settled(r0, false) - r0);
    consume_each(consumed);
    return consumed;
}

int demod::settle_direct()
{
    const uint64_t r0 = nitems_read(0);
    const int consumed = (int)(d_up->log().settled(r0, true) - r0);
    consume_each(consumed);
    return consumed;
}

// Waits for the oldest round that is still running and publishes what is ready.
bool demod::harvest_front()
{
    for (round *rp : d_flight)
        if (rp->state.load() != R_DRAIN) {
            if (rp->worker.joinable()) rp->worker.join();
            break;
        }
    return harvest(false);
}

bool demod::flush()
{
    // a frame cut off by the end of the stream is demodulated with the items it has, like the fused block does
    d_state = S_IDLE;
    if (d_mode == M_DIRECT && !d_up->log().drain()) {
        d_err = "upload of the frames' items failed";
        return false;
    }
    if (d_fill && !cut()) return false;
    return harvest(true);
}

bool demod::cut()
{
    round &r = *d_fill;
    d_fill = nullptr;
    if (r.desc.empty()) {
        r.state.store(R_FREE);
        return true;
    }
    if (!r.gpu->ok()) return false;
    // the round's stream runs behind the pieces that carried its items
    if (r.direct && !d_up->join(r.gpu->ctx())) {
        d_err = "cannot order the round behind its uploads: " + r.gpu->error();
        return false;
    }
    d_nRounds++;
    r.state.store(R_RUN);
    d_flight.push_back(&r);
    if (r.worker.joinable()) r.worker.join();
    // a round the call is going to wait for anyway runs on the caller's thread: starting one costs as much as the round
    if (d_finish && d_state != S_COLLECT && !d_cutAhead) run(r);
    else r.worker = std::thread([this, &r] { run(r); });
    return true;
}

// The round's way through the GPU, on a thread of its own: every call here may wait.
void demod::run(round &r)
{
    gpu_session &g = *r.gpu;
    gr80211_ctx *c = g.ctx();
    const int n = (int)r.desc.size();
    r.err.clear();
    r.ndp.clear();
    r.info.assign((size_t)n, gr80211_frame_info());
    r.llr = nullptr;
    auto fail = [&](const std::string &what) {
        r.err = what;
        r.state.store(R_DONE);
    };
    // direct: the items are in the arena already (fill_direct); else they travel now, from the round's page-locked memory
    char *d0 = r.direct ? r.dev[0] : static_cast<char *>(g.buffer(0, (r.n + 64) * sizeof(gr_complex)));
    char *d1 = d_nrx != 2 ? nullptr : r.direct ? r.dev[1] : static_cast<char *>(g.buffer(1, (r.n + 64) * sizeof(gr_complex)));
    void *dch = g.buffer(2, (size_t)n * 64 * sizeof(gr_complex));
    gr80211_frame_info *dinfo = static_cast<gr80211_frame_info *>(g.buffer(3, (size_t)n * sizeof(gr80211_frame_info)));
    gr_complex *hch = static_cast<gr_complex *>(g.staging(1, (size_t)n * 64 * sizeof(gr_complex)));
    gr80211_frame_info *hinfo = static_cast<gr80211_frame_info *>(g.staging(2, (size_t)n * sizeof(gr80211_frame_info)));
    if (!d0 || (d_nrx == 2 && !d1) || !dch || !dinfo || !hch || !hinfo) return fail("allocation failed");
    memcpy(hch, r.chan.data(), (size_t)n * 64 * sizeof(gr_complex));
    const float *dllr = nullptr;
    uint64_t nfl = 0;
    if ((!r.direct && (gr80211_copy_h2d_async(c, d0, r.iq[0], r.n * sizeof(gr_complex), nullptr) != GR80211_OK ||
                       (d_nrx == 2 && gr80211_copy_h2d_async(c, d1, r.iq[1], r.n * sizeof(gr_complex), nullptr) != GR80211_OK))) ||
        gr80211_copy_h2d_async(c, dch, hch, (size_t)n * 64 * sizeof(gr_complex), nullptr) != GR80211_OK ||
        gr80211_demod_tagged_batch(c, d0, d1, r.n, r.desc.data(), n, 0, dch, dinfo, &dllr, &nfl, nullptr) != GR80211_OK ||
        gr80211_copy_d2h_async(c, hinfo, dinfo, (size_t)n * sizeof(gr80211_frame_info), nullptr) != GR80211_OK)
        return fail(g.error());
    // the soft bits either come back to page-locked memory of the round now, or stay on the device and travel into the
    // scheduler's buffer when it makes room (d_outMode 2)
    float *llr = nullptr;
    r.llr_dev = nullptr;
    if (d_outMode.load() == 2) {
        r.llr_dev = dllr;
        if (!g.sync()) return fail(g.error());
    } else {
        llr = static_cast<float *>(g.staging(0, (size_t)(nfl + 1) * sizeof(float)));
        if (!llr) return fail("page-locked host memory exhausted");
        if ((nfl && gr80211_copy_d2h_async(c, llr, dllr, (size_t)nfl * sizeof(float), nullptr) != GR80211_OK) || !g.sync())
            return fail(g.error());
    }
    memcpy(r.info.data(), hinfo, (size_t)n * sizeof(gr80211_frame_info));
    r.llr = llr;
    for (int i = 0; i < n; i++) {
        const gr80211_frame_info &fi = r.info[(size_t)i];
        if (fi.status != GR80211_ST_NDP) continue;
        // sounding frame: the raw samples of the two VHT-LTF windows (lib/demod_impl.cc:238-249,265-271)
        std::vector<gr_complex> win(128);
        const uint64_t base = r.desc[(size_t)i].off + (uint64_t)fi.data_off + 8;
        for (int w = 0; w < 2; w++) {
            if (base + (uint64_t)w * 80 + 64 > r.n) continue;
            if (!r.direct) memcpy(win.data() + w * 64, r.iq[0] + base + (uint64_t)w * 80, 64 * sizeof(gr_complex));
            else if (gr80211_copy_d2h(c, win.data() + w * 64, d0 + (base + (uint64_t)w * 80) * sizeof(gr_complex), 64 * sizeof(gr_complex), nullptr) != GR80211_OK)
                return fail(g.error());
        }
        r.ndp.push_back(std::move(win));
    }
    r.state.store(R_DONE);
}

// Finished rounds, oldest first, hand their soft bits and tags to the output queue.
bool demod::harvest(bool wait)
{
    for (round *rp : d_flight) {
        round &r = *rp;
        int st = r.state.load();
        if (st == R_DRAIN) continue;
        if (st == R_RUN && !wait) break;
        if (r.worker.joinable()) r.worker.join();  // R_DONE, or waiting for it
        if (!r.err.empty()) {
            d_err = r.err;
            return false;
        }
        const int n = (int)r.desc.size();
        size_t ndp_at = 0;
        r.left = 0;
        for (int i = 0; i < n; i++) {
            const gr80211_frame_info &fi = r.info[(size_t)i];
            const bool ndp = fi.status == GR80211_ST_NDP;
            if (!(fi.status == GR80211_ST_OK || fi.status == GR80211_ST_DROP_DECODE_LIMIT || ndp)) continue;
            // lib/demod_impl.cc:221-263, lib/demod2_impl.cc:228-258: the tags sit on the frame's first soft bit
            tag_t t;
            t.offset = d_out.pushed();
            t.key = "cfo";     t.value = (double)fi.cfo_hz; d_out.tag(t);
            t.key = "snr";     t.value = (double)fi.snr; d_out.tag(t);
            if (fi.format == GR80211_F_VHT) {
                t.key = "sssnr0"; t.value = (double)fi.sssnr0; d_out.tag(t);
                if (d_nrx == 2 && fi.nSS == 2) { t.key = "sssnr1"; t.value = (double)fi.sssnr1; d_out.tag(t); }
            }
            t.key = "rssi";    t.value = (double)fi.rssi; d_out.tag(t);
            t.key = "format";  t.value = (double)fi.format; d_out.tag(t);
            t.key = "mcs";     t.value = (double)fi.mcs; d_out.tag(t);
            t.key = "len";     t.value = (double)fi.len; d_out.tag(t);
            t.key = "cr";      t.value = (double)fi.cr; d_out.tag(t);
            t.key = "ampdu";   t.value = (double)fi.ampdu; d_out.tag(t);
            t.key = "trellis"; t.value = (double)(ndp ? 0 : fi.trellis); d_out.tag(t);
            if (ndp) {
                // the raw window samples travel in a tag, 1024 items carry nothing
                t.key = "mu2x1chan"; t.value = 0.0;
                if (ndp_at < r.ndp.size()) t.cvec = r.ndp[ndp_at++];
                d_out.tag(t);
                t.cvec.clear();
                t.key = "total"; t.value = 1024.0; d_out.tag(t);
                d_out.push_zeros(1024);
                continue;
            }
            t.key = "total";   t.value = (double)fi.total; d_out.tag(t);
            if (r.llr_dev) d_out.push_dev(r.gpu.get(), r.llr_dev + fi.llr_off, (size_t)fi.total, &r.left);
            else d_out.push_ref(r.llr + fi.llr_off, (size_t)fi.total, &r.left);
        }
        r.state.store(R_DRAIN);
    }
    while (!d_flight.empty() && d_flight.front()->state.load() == R_DRAIN && d_flight.front()->left == 0) {
        d_flight.front()->state.store(R_FREE);
        d_flight.pop_front();
    }
    return true;
}

// ------------------------------------------------------------------------------------ capture_rx
capture_rx::sptr capture_rx::make(int n_rx, int device) { return sptr(new capture_rx(n_rx, device)); }

capture_rx::capture_rx(int n_rx, int device) : rx(n_rx, 1 << 30, device, 0, false)
{
    d_carryBack = DET_MARGIN + DET_HISTORY;
    if (const char *e = getenv("GR80211_CAPTURE_MIN_ROUND")) set_min_round((size_t)atoll(e));  // experiments and tests
}

void capture_rx::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    for (int &v : ninput_items_required) v = noutput_items;
}

int capture_rx::general_work(int, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    consume_each(0);
    if (!d_err.empty()) return 0;
    int n = ninput_items[0];
    if (d_nrx == 2) n = std::min(n, ninput_items[1]);
    const uint64_t r0 = nitems_read(0);
    upload_log &log = d_up.log();
    if (log.enqueued() < r0) log.set_enqueued(r0);
    if (!d_flight.empty() && poll_due(d_lastPoll) && !advance(false, false)) return 0;
    // items [lo, n) of the window are new; every one of them goes to the round being filled
    int lo = (int)std::min<uint64_t>(log.enqueued() - r0, (uint64_t)n);
    while (lo < n) {
        round &f = *d_fill;
        if (f.n == 0) f.base = r0 + (uint64_t)lo;
        // a round ends exactly at its limit, so that the rounds of a stream do not depend on the windows it came in
        const size_t room = f.n < d_maxHold ? d_maxHold - f.n : 0;
        const size_t cnt = std::min<size_t>((size_t)(n - lo), room);
        if (cnt > 0) {
            if (!reserve(f, f.n + cnt)) {
                d_err = "device memory for the held samples exhausted";
                return 0;
            }
            bool pinned = true;
            const char *src[2];
            char *dst[2];
            for (int a = 0; a < d_nrx; a++) {
                src[a] = reinterpret_cast<const char *>(static_cast<const gr_complex *>(input_items[a]) + lo);
                dst[a] = f.iq[a] + f.n * sizeof(gr_complex);
                pinned = pinned && is_pinned(src[a], cnt * sizeof(gr_complex));
            }
            if (!d_up.send(*f.gpu, d_nrx, src, dst, cnt, sizeof(gr_complex), r0 + (uint64_t)lo, pinned, d_err)) return 0;
            f.n += cnt;
            lo += (int)cnt;
        }
        if (f.n >= d_maxHold && !cut(false)) return 0;
    }
    log.set_enqueued(r0 + (uint64_t)n);
    // live reception (set_min_round): a round goes as soon as it holds enough new samples and fewer than two are on their
    // way -- looked at in every call, since it is a round's return that lets the next one go
    if (d_minNew && d_fill->n >= d_fill->n_carry + d_minNew && d_flight.size() < 2 && !cut(false)) return 0;
    consume_each((int)(log.settled(r0) - r0));
    return 0;
}

// The round's sync bytes: frame detection over its samples (antenna 1, as in the reference flowgraphs).  Sync bytes
// before d_detFrom were taken by the round before; those within DET_MARGIN of the end are left to the next round,
// which sees the samples that follow them.
bool capture_rx::before_cut(round &r, bool final)
{
    if (r.n == 0) return true;
    gr80211_ctx *c = r.gpu->ctx();
    const uint64_t end = r.base + r.n;
    const int max_desc = (int)std::min<size_t>(r.n / 80 + 16, (size_t)1 << 20);
    gr80211_frame_desc *dd = static_cast<gr80211_frame_desc *>(r.gpu->buffer(6, (size_t)max_desc * sizeof(gr80211_frame_desc)));
    int nd = 0;
    if (!dd || gr80211_detect(c, r.iq[0], r.n, dd, max_desc, &nd, nullptr) != GR80211_OK) {
        d_err = "frame detection failed: " + r.gpu->error();
        return false;
    }
    const uint64_t upto = final || end < DET_MARGIN ? end : end - DET_MARGIN;
    if (nd > 0) {
        gr80211_frame_desc *hd = static_cast<gr80211_frame_desc *>(r.gpu->staging(3, (size_t)nd * sizeof(gr80211_frame_desc)));
        if (!hd || gr80211_copy_d2h(c, hd, dd, (size_t)nd * sizeof(gr80211_frame_desc), nullptr) != GR80211_OK) {
            d_err = "frame detection failed: " + r.gpu->error();
            return false;
        }
        for (int i = 0; i < nd; i++) {
            const uint64_t pos = r.base + hd[i].sync_off;
            if (pos < d_detFrom || pos >= upto) continue;
            trigger t;
            t.pos = pos;
            t.rad = hd[i].rad;
            t.snr = hd[i].snr;
            t.rssi = hd[i].rssi;
            r.trig.push_back(t);
            r.n_new++;
            d_nSync++;
            d_nTrig++;
        }
    }
    d_detFrom = upto;
    return true;
}

bool capture_rx::flush()
{
    if (!rx::flush()) return false;
    d_detFrom = 0;  // another stream starts from nothing
    return true;
}

}  // namespace host
}  // namespace gr80211
