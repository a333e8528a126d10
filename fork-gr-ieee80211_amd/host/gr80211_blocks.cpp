// Host blocks over the C-ABI of libgr80211_rx.so (see gr80211_blocks.h).
#include "gr80211_blocks.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace gr80211 {
namespace host {

namespace {

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

void block::get_tags_in_range(std::vector<tag_t> &v, int port, uint64_t start, uint64_t end) const
{
    v.clear();
    const std::deque<tag_t> &tv = d_tags[port];
    auto it = std::lower_bound(tv.begin(), tv.end(), start, [](const tag_t &t, uint64_t o) { return t.offset < o; });
    for (; it != tv.end() && it->offset < end; ++it) v.push_back(*it);
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
            pdu_t m;
            m.blob.assign(p + at, p + at + rec);
            m.len = rec;
            message_port_pub("out", m);
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

void out_stream::push(const void *src, size_t items)
{
    if (d_head > (1u << 20) && d_head * 2 > d_buf.size()) {  // drop what has left
        d_buf.erase(d_buf.begin(), d_buf.begin() + (long)d_head);
        d_head = 0;
    }
    const uint8_t *p = static_cast<const uint8_t *>(src);
    d_buf.insert(d_buf.end(), p, p + items * d_item);
    d_pushed += items;
}

void out_stream::push_zeros(size_t items)
{
    d_buf.insert(d_buf.end(), items * d_item, 0);
    d_pushed += items;
}

int out_stream::pop(void *dst, int max_items, uint64_t written, std::vector<tag_t> &out)
{
    const size_t n = std::min<size_t>(pending(), max_items > 0 ? (size_t)max_items : 0);
    if (n == 0) return 0;
    memcpy(dst, d_buf.data() + d_head, n * d_item);
    d_head += n * d_item;
    while (!d_tags.empty() && d_tags.front().offset < d_popped + n) {
        tag_t t = d_tags.front();
        t.offset = written + (t.offset - d_popped);
        out.push_back(t);
        d_tags.pop_front();
    }
    d_popped += n;
    if (d_head == d_buf.size()) {
        d_buf.clear();
        d_head = 0;
    }
    return (int)n;
}

std::string gpu_session::error() const { return d_ctx ? gr80211_last_error(d_ctx) : "no GPU context"; }

// ------------------------------------------------------------------------------------ decode
decode::sptr decode::make(bool ifdebug, int batch_frames, int device)
{
    return sptr(new decode(ifdebug, batch_frames, device));
}

decode::decode(bool ifdebug, int batch_frames, int device)
    : block(1), d_debug(ifdebug), d_batch(batch_frames < 1 ? 1 : batch_frames), d_gpu(device), d_llr(d_gpu, sizeof(float))
{
    if (!d_gpu.ok()) d_err = "gr80211_create failed";
}

decode::~decode() {}

void decode::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    ninput_items_required[0] = noutput_items + 160;  // lib/decode_impl.cc:54-58
}

int decode::general_work(int, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    const float *in = static_cast<const float *>(input_items[0]);
    const int n = ninput_items[0];
    int used = 0;
    std::vector<tag_t> tags;
    while (used < n) {
        if (d_state == S_IDLE) {
            const uint64_t at = nitems_read(0) + (uint64_t)used;
            get_tags_in_range(tags, 0, at, at + 1);
            if (tags.empty()) {
                // The reference stays idle until an item carries tags (lib/decode_impl.cc:70-72,125-126);
                // untagged items between packets are passed over here.
                get_tags_in_range(tags, 0, at + 1, nitems_read(0) + (uint64_t)n);
                if (tags.empty()) {
                    used = n;
                    break;
                }
                used = (int)(tags[0].offset - nitems_read(0));
                continue;
            }
            t_format = (long)tag_value(tags, "format", DEFAULT_TAG);
            t_len = (long)tag_value(tags, "len", DEFAULT_TAG);
            t_nTotal = (long)tag_value(tags, "total", DEFAULT_TAG);
            t_cr = (long)tag_value(tags, "cr", DEFAULT_TAG);
            t_mcs = (long)tag_value(tags, "mcs", DEFAULT_TAG);
            t_ampdu = (long)tag_value(tags, "ampdu", DEFAULT_TAG);
            v_trellis = (long)tag_value(tags, "trellis", DEFAULT_TAG);
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
                if (d_desc.empty()) {
                    message_port_pub("out", m);
                } else {
                    // packets before it are still waiting for the GPU: it takes its turn behind them
                    d_ready.emplace_back(d_desc.size(), std::move(m));
                }
                d_nPdu++;
            } else {
                d_state = S_COLLECT;
            }
            if (d_state == S_COLLECT) {
                gr80211_dec_desc d;
                memset(&d, 0, sizeof(d));
                d.llr_off = (int64_t)d_llr.size();
                d.n_llr = (int32_t)t_nTotal;
                d.format = (int32_t)t_format;
                d.len = (int32_t)t_len;
                d.cr = (int32_t)t_cr;
                d.mcs = (int32_t)t_mcs;
                d.ampdu = (int32_t)t_ampdu;
                d.trellis = (int32_t)v_trellis;
                d_desc.push_back(d);
            }
        }
        if (d_state == S_COLLECT || d_state == S_CLEAN) {
            const long need = t_nTotal - t_nProcd;
            const int take = (int)std::min<long>(need < 0 ? 0 : need, n - used);
            if (d_state == S_COLLECT && !d_llr.append(in + used, (size_t)take)) {
                d_err = "upload of the soft bits failed: " + d_gpu.error();
                consume_each(used);
                return 0;
            }
            t_nProcd += take;
            used += take;
            if (t_nProcd >= t_nTotal) {
                const bool collected = d_state == S_COLLECT;
                d_state = S_IDLE;
                if (collected && (int)d_desc.size() >= d_batch && !run_batch()) {
                    consume_each(used);
                    return 0;
                }
            }
        }
    }
    consume_each(used);
    return 0;
}

bool decode::flush()
{
    // a packet cut off by the end of the stream is dropped, as the reference never finishes it either
    if (d_state == S_COLLECT && !d_desc.empty()) {
        d_llr.truncate((size_t)d_desc.back().llr_off);
        d_desc.pop_back();
    }
    d_state = S_IDLE;
    return run_batch();
}

bool decode::run_batch()
{
    // called between packets only: every pending descriptor has all of its soft bits
    const int n = (int)d_desc.size();
    if (n <= 0) {
        for (auto &r : d_ready) message_port_pub("out", r.second);
        d_ready.clear();
        return true;
    }
    if (!d_gpu.ok()) return false;
    const size_t nfl = d_llr.size();
    const void *dl = d_llr.data();  // uploaded window by window as the stream came in
    uint8_t *dout = static_cast<uint8_t *>(d_gpu.buffer(1, (size_t)n * OUT_STRIDE));
    int32_t *dlen = static_cast<int32_t *>(d_gpu.buffer(2, (size_t)n * 4));
    int32_t *dnp = static_cast<int32_t *>(d_gpu.buffer(3, (size_t)n * 4));
    if (!dl || !dout || !dlen || !dnp) {
        d_err = "device allocation failed";
        return false;
    }
    gr80211_ctx *c = d_gpu.ctx();
    uint8_t *out = static_cast<uint8_t *>(d_gpu.staging(0, (size_t)n * OUT_STRIDE));
    int32_t *len = static_cast<int32_t *>(d_gpu.staging(1, (size_t)n * 4));
    if (!out || !len) {
        d_err = "page-locked host memory exhausted";
        return false;
    }
    if (gr80211_decode_batch(c, static_cast<const float *>(dl), nfl, d_desc.data(), n, 0, dout, OUT_STRIDE, dlen, dnp,
                             nullptr, 0, nullptr) != GR80211_OK ||
        gr80211_copy_d2h(c, len, dlen, (size_t)n * 4, nullptr) != GR80211_OK ||
        gr80211_copy_d2h(c, out, dout, (size_t)n * OUT_STRIDE, nullptr) != GR80211_OK) {
        d_err = d_gpu.error();
        return false;
    }
    size_t r = 0;
    for (int i = 0; i < n; i++) {
        for (; r < d_ready.size() && d_ready[r].first <= (size_t)i; r++) message_port_pub("out", d_ready[r].second);
        d_nPdu += publish_slots(out + (size_t)i * OUT_STRIDE, len + i, 1, OUT_STRIDE);
    }
    for (; r < d_ready.size(); r++) message_port_pub("out", d_ready[r].second);
    d_ready.clear();
    d_llr.clear();
    d_desc.clear();
    return true;
}

// ------------------------------------------------------------------------------------ rx
rx::sptr rx::make(int n_rx, int batch_frames, int device) { return sptr(new rx(n_rx, batch_frames, device)); }

rx::rx(int n_rx, int batch_frames, int device, int n_out)
    : block(n_rx == 2 ? 3 : 2, n_out), d_nrx(n_rx == 2 ? 2 : 1), d_batch(batch_frames < 1 ? 1 : batch_frames), d_gpu(device),
      d_hold0(d_gpu, sizeof(gr_complex)), d_hold1(d_gpu, sizeof(gr_complex)), d_maxHold((size_t)1 << 22)
{
    if (!d_gpu.ok()) d_err = "gr80211_create failed";
}

rx::~rx() {}

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

// The input half of general_work: triggers and samples of this window go to the pending lists, a round of
// processing runs when enough has gathered.
int rx::take_input(gr_vector_int &ninput_items, gr_vector_const_void_star &input_items)
{
    const uint8_t *sync = static_cast<const uint8_t *>(input_items[0]);
    int n = std::min(ninput_items[0], ninput_items[1]);
    if (d_nrx == 2) n = std::min(n, ninput_items[2]);
    std::vector<tag_t> tags;
    for (int i = next_set(sync, 0, n); i < n; i = next_set(sync, i + 1, n)) {
        const uint64_t at = nitems_read(0) + (uint64_t)i;
        get_tags_in_range(tags, 0, at, at + 1);
        if (tags.empty()) continue;  // "input sync with no tag" (lib/signal_impl.cc:96-100): not a trigger
        trigger t;
        t.pos = at;
        t.rad = (float)tag_value(tags, "rad", 0.0);
        t.snr = (float)tag_value(tags, "snr", 0.0);
        t.rssi = (float)tag_value(tags, "rssi", 0.0);
        t.probed = false;
        t.status = 0;
        t.nsamp = 0;
        t.mcs = 0;
        t.len = 0;
        d_trig.push_back(t);
        d_nTrig++;
    }
    // samples are only held from the first pending trigger on
    int from = 0;
    if (d_hold0.empty()) {
        if (d_trig.empty()) {
            from = n;
        } else {
            const uint64_t first = std::max(d_trig[0].pos, nitems_read(0));
            from = (int)(first - nitems_read(0));
        }
        d_base = nitems_read(0) + (uint64_t)from;
    }
    for (int a = 0; a < d_nrx; a++) {
        const gr_complex *s = static_cast<const gr_complex *>(input_items[1 + a]);
        if (!hold(a).append(s + from, (size_t)(n - from))) {
            d_err = "upload of the samples failed: " + d_gpu.error();
            return -1;
        }
    }
    if ((int)d_trig.size() >= d_batch || d_hold0.size() >= d_maxHold) process(false);
    return n;
}

bool rx::flush() { return process(true); }

// One round: learn the span of every buffered trigger (L-SIG), replay the trigger/copy state machine of
// `signal` over them in stream order, decode the frames whose samples are complete, release what is done.
bool rx::process(bool final)
{
    const uint64_t held_end = d_base + d_hold0.size();
    // triggers at or behind the search position can never start a frame
    {
        size_t k = 0;
        while (k < d_trig.size() && d_trig[k].pos < d_cursor) k++;
        if (k) d_trig.erase(d_trig.begin(), d_trig.begin() + (long)k);
    }
    if (d_trig.empty()) {
        // nothing pending: everything held so far can go
        d_cursor = std::max(d_cursor, held_end);
        for (int a = 0; a < d_nrx; a++) hold(a).clear();
        d_base = held_end;
        return true;
    }
    if (!d_gpu.ok()) return false;
    gr80211_ctx *c = d_gpu.ctx();
    const size_t nheld = d_hold0.size();
    // the samples are on the device already: every window was uploaded when it arrived
    const void *diq0 = d_hold0.data();
    const void *diq1 = d_nrx == 2 ? d_hold1.data() : nullptr;

    // ---- 1. spans: L-SIG of every trigger that has its 224 samples (lib/signal_impl.cc:108-131)
    std::vector<gr80211_frame_desc> pd;
    std::vector<size_t> pidx;
    for (size_t j = 0; j < d_trig.size(); j++) {
        trigger &t = d_trig[j];
        if (t.probed) continue;
        const uint64_t rel = t.pos - d_base;
        if (rel + LSIG_SPAN > nheld) continue;  // `signal` waits for 224 samples too
        gr80211_frame_desc d;
        memset(&d, 0, sizeof(d));
        d.sync_off = rel;
        d.n_avail = (uint32_t)std::min<uint64_t>(nheld - rel, 0xffffffffu);
        d.rad = t.rad;
        d.snr = t.snr;
        d.rssi = t.rssi;
        d.mu_pos = (uint32_t)d_muPos;
        pd.push_back(d);
        pidx.push_back(j);
    }
    if (!pd.empty()) {
        const int np = (int)pd.size();
        gr80211_frame_info *dinfo =
            static_cast<gr80211_frame_info *>(d_gpu.buffer(2, (size_t)np * sizeof(gr80211_frame_info)));
        std::vector<gr80211_frame_info> info(np);
        if (!dinfo || gr80211_signal_batch(c, diq0, diq1, nheld, pd.data(), np, 0, dinfo, nullptr) != GR80211_OK ||
            gr80211_copy_d2h(c, info.data(), dinfo, (size_t)np * sizeof(gr80211_frame_info), nullptr) != GR80211_OK) {
            d_err = d_gpu.error();
            return false;
        }
        for (int k = 0; k < np; k++) {
            trigger &t = d_trig[pidx[k]];
            t.probed = true;
            t.status = info[k].status;
            t.nsamp = info[k].nsamp;
            t.mcs = info[k].lsig_mcs;
            t.len = info[k].lsig_len;
        }
    }

    // ---- 2. the trigger search in stream order (lib/signal_impl.cc:75-106,153-160,164-202)
    std::vector<gr80211_frame_desc> fd;
    std::vector<frame_ref> fr;
    size_t done = 0;  // triggers settled in this round
    for (size_t j = 0; j < d_trig.size(); j++) {
        const trigger &t = d_trig[j];
        if (t.pos < d_cursor) {  // inside the previous frame's span: `signal` never looks at it
            done = j + 1;
            continue;
        }
        if (!t.probed) {
            if (final) done = j + 1;  // fewer than 224 samples before the end of the stream
            break;
        }
        if (t.status == GR80211_ST_DROP_LSIG) {
            d_cursor = t.pos + LSIG_FAIL_SKIP;
            done = j + 1;
            continue;
        }
        const uint64_t span = (uint64_t)LSIG_SPAN + (uint64_t)(t.nsamp > 0 ? t.nsamp : 0);
        if (t.pos + span > held_end && !final) break;  // its samples are still arriving
        gr80211_frame_desc d;
        memset(&d, 0, sizeof(d));
        d.sync_off = t.pos - d_base;
        d.n_avail = (uint32_t)std::min<uint64_t>(span, held_end - t.pos);
        d.rad = t.rad;
        d.snr = t.snr;
        d.rssi = t.rssi;
        d.mu_pos = (uint32_t)d_muPos;
        fd.push_back(d);
        frame_ref r;
        r.pos = t.pos;
        r.mcs = t.mcs;
        r.len = t.len;
        r.nsamp = t.nsamp > 0 ? t.nsamp : 0;
        r.rad = t.rad;
        r.snr = t.snr;
        r.rssi = t.rssi;
        fr.push_back(r);
        d_cursor = t.pos + span;
        done = j + 1;
    }

    // ---- 3. the complete frames, in stream order
    if (!fd.empty() && !handle_frames(diq0, diq1, nheld, fd, fr)) return false;

    // ---- 4. release settled triggers and the samples nothing pending can need
    d_trig.erase(d_trig.begin(), d_trig.begin() + (long)done);
    uint64_t keep_from = held_end;
    if (!d_trig.empty() && !final) keep_from = std::min(keep_from, std::max(d_trig[0].pos, d_base));
    if (final) d_trig.clear();
    const size_t drop = (size_t)(keep_from - d_base);
    for (int a = 0; a < d_nrx; a++)
        if (!hold(a).release_front(drop)) {
            d_err = "release of the held samples failed: " + d_gpu.error();
            return false;
        }
    d_base = keep_from;
    return true;
}

// rx: decode the frames and publish their PDUs
bool rx::handle_frames(const void *diq0, const void *diq1, size_t nheld, std::vector<gr80211_frame_desc> &fd,
                       const std::vector<frame_ref> &)
{
    gr80211_ctx *c = d_gpu.ctx();
    const int nf = (int)fd.size();
    uint8_t *dout = static_cast<uint8_t *>(d_gpu.buffer(3, (size_t)nf * OUT_STRIDE));
    int32_t *dlen = static_cast<int32_t *>(d_gpu.buffer(4, (size_t)nf * 4));
    int32_t *dnp = static_cast<int32_t *>(d_gpu.buffer(5, (size_t)nf * 4));
    uint8_t *out = static_cast<uint8_t *>(d_gpu.staging(0, (size_t)nf * OUT_STRIDE));
    int32_t *len = static_cast<int32_t *>(d_gpu.staging(1, (size_t)nf * 4));
    if (!dout || !dlen || !dnp || !out || !len) {
        d_err = "allocation of the output buffers failed";
        return false;
    }
    if (gr80211_rx_batch(c, diq0, diq1, nheld, fd.data(), nf, 0, dout, OUT_STRIDE, dlen, dnp, nullptr, nullptr) !=
            GR80211_OK ||
        gr80211_copy_d2h(c, len, dlen, (size_t)nf * 4, nullptr) != GR80211_OK ||
        gr80211_copy_d2h(c, out, dout, (size_t)nf * OUT_STRIDE, nullptr) != GR80211_OK) {
        d_err = d_gpu.error();
        return false;
    }
    d_nFrames += (uint64_t)nf;
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
    for (int &v : ninput_items_required) v = d_out0.pending() ? 0 : noutput_items;
}

int signal::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                         gr_vector_void_star &output_items)
{
    // input is only taken while the items already produced find room downstream
    const int n = d_out0.pending() > OUT_QUEUE_LIMIT ? 0 : take_input(ninput_items, input_items);
    consume_each(n < 0 ? 0 : n);
    if ((int)output_items.size() < d_nrx) return 0;
    std::vector<tag_t> tags;
    const int produced = d_out0.pop(output_items[0], noutput_items, nitems_written(0), tags);
    for (const tag_t &t : tags) add_item_tag(0, t);
    if (d_nrx == 2) {
        tags.clear();
        d_out1.pop(output_items[1], produced, nitems_written(1), tags);  // both queues always hold the same count
    }
    return produced;
}

// signal: the frames' items and tags go to the output queues
bool signal::handle_frames(const void *diq0, const void *diq1, size_t nheld, std::vector<gr80211_frame_desc> &fd,
                           const std::vector<frame_ref> &fr)
{
    gr80211_ctx *c = d_gpu.ctx();
    const int nf = (int)fd.size();
    std::vector<uint64_t> off((size_t)nf + 1, 0);
    for (int i = 0; i < nf; i++) off[(size_t)i + 1] = off[(size_t)i] + (uint64_t)fr[(size_t)i].nsamp + 320u;
    const uint64_t total = off[(size_t)nf];
    void *do0 = d_gpu.buffer(3, (size_t)(total + 64) * sizeof(gr_complex));
    void *do1 = d_nrx == 2 ? d_gpu.buffer(4, (size_t)(total + 64) * sizeof(gr_complex)) : nullptr;
    void *dch = d_gpu.buffer(5, (size_t)nf * 64 * sizeof(gr_complex));
    gr_complex *h0 = static_cast<gr_complex *>(d_gpu.staging(0, (size_t)total * sizeof(gr_complex) + 8));
    gr_complex *h1 = d_nrx == 2 ? static_cast<gr_complex *>(d_gpu.staging(1, (size_t)total * sizeof(gr_complex) + 8)) : nullptr;
    gr_complex *hch = static_cast<gr_complex *>(d_gpu.staging(2, (size_t)nf * 64 * sizeof(gr_complex)));
    if (!do0 || (d_nrx == 2 && (!do1 || !h1)) || !dch || !h0 || !hch) {
        d_err = "allocation of the output buffers failed";
        return false;
    }
    if (gr80211_signal_emit_batch(c, diq0, diq1, nheld, fd.data(), nf, 0, off.data(), do0, do1, total, nullptr, dch,
                                  nullptr) != GR80211_OK ||
        gr80211_copy_d2h(c, h0, do0, (size_t)total * sizeof(gr_complex), nullptr) != GR80211_OK ||
        (d_nrx == 2 && gr80211_copy_d2h(c, h1, do1, (size_t)total * sizeof(gr_complex), nullptr) != GR80211_OK) ||
        gr80211_copy_d2h(c, hch, dch, (size_t)nf * 64 * sizeof(gr_complex), nullptr) != GR80211_OK) {
        d_err = d_gpu.error();
        return false;
    }
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
        d_out0.push(h0 + off[(size_t)i], cnt);
        if (d_nrx == 2) d_out1.push(h1 + off[(size_t)i], cnt);
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
    : block(n_rx == 2 ? 2 : 1, 1), d_nrx(n_rx == 2 ? 2 : 1), d_muPos(mupos ? 1 : 0), d_batch(batch_frames < 1 ? 1 : batch_frames),
      d_gpu(device), d_iq0(d_gpu, sizeof(gr_complex)), d_iq1(d_gpu, sizeof(gr_complex)), d_out(sizeof(float))
{
    if (!d_gpu.ok()) d_err = "gr80211_create failed";
}

void demod::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    // lib/demod_impl.cc:53-57; soft bits held back for lack of room need no input to leave
    for (int &v : ninput_items_required) v = d_out.pending() ? 0 : noutput_items;
}

int demod::general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                        gr_vector_void_star &output_items)
{
    int n = ninput_items[0];
    if (d_nrx == 2) n = std::min(n, ninput_items[1]);
    if (d_out.pending() > OUT_QUEUE_LIMIT) n = 0;  // input is only taken while the soft bits produced find room
    int used = 0;
    std::vector<tag_t> tags;
    while (used < n) {
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
            gr80211_sig_desc d;
            memset(&d, 0, sizeof(d));
            d.off = d_iq0.size();
            d.mcs = (int32_t)tag_value(tags, "mcs", -1.0);
            d.len = (int32_t)tag_value(tags, "len", -1.0);
            d.nsamp = (int32_t)tag_value(tags, "nsamp", -1.0);
            d.cfo = (float)tag_value(tags, "cfo", 0.0);
            d.snr = (float)tag_value(tags, "snr", 0.0);
            d.rssi = (float)tag_value(tags, "rssi", 0.0);
            d.mu_pos = (uint32_t)d_muPos;
            d_nSeen++;
            const tag_t *ch = tag_find(tags, "chan");
            if (d.nsamp < 0 || d.nsamp > (1 << 24) || !ch || ch->cvec.size() < 64) {
                // nothing a frame can be taken from: pass over this item and look for the next tagged one
                used++;
                continue;
            }
            d_desc.push_back(d);
            d_chan.insert(d_chan.end(), ch->cvec.begin(), ch->cvec.begin() + 64);
            d_need = d.nsamp;
            d_skip = 320;  // lib/demod_impl.cc:92
            d_state = S_COLLECT;
        }
        if (d_state == S_COLLECT) {
            const int take = (int)std::min<long>(d_need, n - used);
            for (int a = 0; a < d_nrx; a++) {
                const gr_complex *s = static_cast<const gr_complex *>(input_items[(size_t)a]);
                if (!iq(a).append(s + used, (size_t)take)) {
                    d_err = "upload of the samples failed: " + d_gpu.error();
                    consume_each(used);
                    return 0;
                }
            }
            d_desc.back().n_avail += (uint32_t)take;
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
                if ((int)d_desc.size() >= d_batch && !run_batch()) {
                    consume_each(used);
                    return 0;
                }
            }
        }
    }
    consume_each(used);
    if (output_items.empty()) return 0;
    tags.clear();
    const int produced = d_out.pop(output_items[0], noutput_items, nitems_written(0), tags);
    for (const tag_t &t : tags) add_item_tag(0, t);
    return produced;
}

bool demod::flush()
{
    // a frame cut off by the end of the stream is demodulated with the items it has, like the fused block does
    d_state = S_IDLE;
    return run_batch();
}

bool demod::run_batch()
{
    const int n = (int)d_desc.size();
    if (n <= 0) return true;
    if (!d_gpu.ok()) return false;
    gr80211_ctx *c = d_gpu.ctx();
    void *dch = d_gpu.buffer(0, (size_t)n * 64 * sizeof(gr_complex));
    gr80211_frame_info *dinfo = static_cast<gr80211_frame_info *>(d_gpu.buffer(1, (size_t)n * sizeof(gr80211_frame_info)));
    if (!dch || !dinfo) {
        d_err = "device allocation failed";
        return false;
    }
    std::vector<gr80211_frame_info> info((size_t)n);
    const float *dllr = nullptr;
    uint64_t nfl = 0;
    if (gr80211_copy_h2d(c, dch, d_chan.data(), (size_t)n * 64 * sizeof(gr_complex), nullptr) != GR80211_OK ||
        gr80211_demod_tagged_batch(c, d_iq0.data(), d_nrx == 2 ? d_iq1.data() : nullptr, d_iq0.size(), d_desc.data(), n, 0, dch,
                                   dinfo, &dllr, &nfl, nullptr) != GR80211_OK ||
        gr80211_copy_d2h(c, info.data(), dinfo, (size_t)n * sizeof(gr80211_frame_info), nullptr) != GR80211_OK) {
        d_err = d_gpu.error();
        return false;
    }
    float *llr = static_cast<float *>(d_gpu.staging(0, (size_t)(nfl + 1) * sizeof(float)));
    if (!llr || (nfl && gr80211_copy_d2h(c, llr, dllr, (size_t)nfl * sizeof(float), nullptr) != GR80211_OK)) {
        d_err = llr ? d_gpu.error() : std::string("page-locked host memory exhausted");
        return false;
    }
    for (int i = 0; i < n; i++) {
        const gr80211_frame_info &fi = info[(size_t)i];
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
            // sounding frame: the raw samples of the two VHT-LTF windows travel in a tag, 1024 items carry nothing
            // (lib/demod_impl.cc:238-249,265-271)
            std::vector<gr_complex> win(128);
            const uint64_t base = d_desc[(size_t)i].off + (uint64_t)fi.data_off + 8;
            bool ok = true;
            for (int w = 0; w < 2 && ok; w++)
                ok = base + (uint64_t)w * 80 + 64 <= d_iq0.size() &&
                     gr80211_copy_d2h(c, win.data() + w * 64,
                                      static_cast<const gr_complex *>(d_iq0.data()) + base + (uint64_t)w * 80,
                                      64 * sizeof(gr_complex), nullptr) == GR80211_OK;
            t.key = "mu2x1chan"; t.value = 0.0; t.cvec = win; d_out.tag(t);
            t.cvec.clear();
            t.key = "total"; t.value = 1024.0; d_out.tag(t);
            d_out.push_zeros(1024);
            continue;
        }
        t.key = "total";   t.value = (double)fi.total; d_out.tag(t);
        d_out.push(llr + fi.llr_off, (size_t)fi.total);
    }
    d_desc.clear();
    d_chan.clear();
    d_iq0.clear();
    d_iq1.clear();
    return true;
}

// ------------------------------------------------------------------------------------ capture_rx
capture_rx::sptr capture_rx::make(int n_rx, int device) { return sptr(new capture_rx(n_rx, device)); }

capture_rx::capture_rx(int n_rx, int device)
    : block(n_rx == 2 ? 2 : 1), d_nrx(n_rx == 2 ? 2 : 1), d_gpu(device), d_hold0(d_gpu, sizeof(gr_complex)),
      d_hold1(d_gpu, sizeof(gr_complex))
{
    if (!d_gpu.ok()) d_err = "gr80211_create failed";
}

capture_rx::~capture_rx() {}

void capture_rx::forecast(int noutput_items, gr_vector_int &ninput_items_required)
{
    for (int &v : ninput_items_required) v = noutput_items;
}

int capture_rx::general_work(int, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    int n = ninput_items[0];
    if (d_nrx == 2) n = std::min(n, ninput_items[1]);
    for (int a = 0; a < d_nrx; a++) {
        const gr_complex *s = static_cast<const gr_complex *>(input_items[a]);
        if (!hold(a).append(s, (size_t)n)) {
            d_err = "upload of the samples failed: " + d_gpu.error();
            consume_each(0);
            return 0;
        }
    }
    consume_each(n);
    return 0;
}

bool capture_rx::flush()
{
    const size_t n = d_hold0.size();
    if (n == 0) return true;
    if (!d_gpu.ok()) return false;
    gr80211_ctx *c = d_gpu.ctx();
    const void *diq0 = d_hold0.data();
    const void *diq1 = d_nrx == 2 ? d_hold1.data() : nullptr;
    // a frame needs at least 400 samples of preamble; sync bytes whose L-SIG check fails also take a slot
    const int max_frames = (int)std::min<size_t>(n / 80 + 16, (size_t)1 << 20);
    uint8_t *dout = static_cast<uint8_t *>(d_gpu.buffer(3, (size_t)max_frames * OUT_STRIDE));
    int32_t *dlen = static_cast<int32_t *>(d_gpu.buffer(4, (size_t)max_frames * 4));
    int32_t *dnp = static_cast<int32_t *>(d_gpu.buffer(5, (size_t)max_frames * 4));
    if (!dout || !dlen || !dnp) {
        d_err = "device allocation failed";
        return false;
    }
    int nf = 0;
    if (gr80211_set_mu_pos(c, d_muPos) != GR80211_OK ||
        gr80211_rx_capture(c, diq0, diq1, n, dout, OUT_STRIDE, dlen, dnp, max_frames, nullptr, &nf, nullptr) != GR80211_OK) {
        d_err = d_gpu.error();
        return false;
    }
    uint8_t *out = static_cast<uint8_t *>(d_gpu.staging(0, (size_t)nf * OUT_STRIDE + 1));
    int32_t *len = static_cast<int32_t *>(d_gpu.staging(1, (size_t)nf * 4 + 4));
    if (!out || !len) {
        d_err = "page-locked host memory exhausted";
        return false;
    }
    if (nf > 0 && (gr80211_copy_d2h(c, len, dlen, (size_t)nf * 4, nullptr) != GR80211_OK ||
                   gr80211_copy_d2h(c, out, dout, (size_t)nf * OUT_STRIDE, nullptr) != GR80211_OK)) {
        d_err = d_gpu.error();
        return false;
    }
    d_nFrames += (uint64_t)nf;
    d_nPdu += publish_slots(out, len, nf, OUT_STRIDE);
    for (int a = 0; a < d_nrx; a++) hold(a).clear();
    return true;
}

}  // namespace host
}  // namespace gr80211
