// Streaming C-ABI over the host blocks (include/gr80211_host.h).
#include <memory>
#include <string>

#include "../../include/gr80211_host.h"
#include "gr80211_blocks.h"

using namespace gr80211::host;

struct gr80211_block {
    std::shared_ptr<block> b;
    int kind = 0;
    int n_in = 1;
    int n_out = 0;
    gr80211_msg_fn fn = nullptr;
    void *user = nullptr;
    std::string err;
    std::vector<tag_t> out_tags;  // of the last call: the records handed out point into them
    udp_sink udp;                 // open: every message also leaves as a datagram
};

struct gr80211_udp {
    udp_sink sink;
};

namespace {

thread_local std::string g_create_err = "null block";

const char *const kTagNames[] = {"rad", "snr", "rssi", "format", "len", "total", "cr", "mcs", "ampdu", "trellis", "mu2x1chan",
                                 "cfo", "seq", "nsamp", "chan", "sssnr0", "sssnr1"};
const int kNumTags = (int)(sizeof(kTagNames) / sizeof(kTagNames[0]));

}  // namespace

extern "C" {

gr80211_block *gr80211_block_create(int kind, int batch_frames, int device)
{
    std::unique_ptr<gr80211_block> h(new gr80211_block());
    h->kind = kind;
    // no batch size: rounds sized by the traffic, so that a frame never waits for others on a quiet channel
    const int batch = batch_frames > 0 ? batch_frames : 0;
    switch (kind) {
    // decode and demod made without a batch size finish a packet in the call that completes it, as the reference does
    case GR80211_BLOCK_DECODE: h->b = decode::make(false, batch_frames > 0 ? batch_frames : 0, device); h->n_in = 1; break;
    case GR80211_BLOCK_RX: h->b = rx::make(1, batch, device); h->n_in = 2; break;
    case GR80211_BLOCK_RX2: h->b = rx::make(2, batch, device); h->n_in = 3; break;
    case GR80211_BLOCK_CAPTURE: h->b = capture_rx::make(1, device); h->n_in = 1; break;
    case GR80211_BLOCK_CAPTURE2: h->b = capture_rx::make(2, device); h->n_in = 2; break;
    case GR80211_BLOCK_SIGNAL: h->b = signal::make(1, batch, device); h->n_in = 2; h->n_out = 1; break;
    case GR80211_BLOCK_SIGNAL2: h->b = signal::make(2, batch, device); h->n_in = 3; h->n_out = 2; break;
    case GR80211_BLOCK_DEMOD: h->b = demod::make(1, 0, 2, batch_frames > 0 ? batch_frames : 0, device); h->n_in = 1; h->n_out = 1; break;
    case GR80211_BLOCK_DEMOD2: h->b = demod::make(2, 0, 2, batch_frames > 0 ? batch_frames : 0, device); h->n_in = 2; h->n_out = 1; break;
    default: g_create_err = "gr80211_block_create: unknown block kind"; return nullptr;
    }
    if (!h->b || !h->b->last_error().empty()) {
        g_create_err = h->b ? h->b->last_error() : std::string("gr80211_block_create: no block");
        return nullptr;
    }
    gr80211_block *raw = h.release();
    raw->b->set_msg_handler([raw](const std::string &, const pdu_t &m) {
        if (raw->udp.is_open()) raw->udp.send(m);
        if (raw->fn) raw->fn(raw->user, m.blob.data(), (int)m.blob.size());
    });
    return raw;
}

void gr80211_block_destroy(gr80211_block *b) { delete b; }

void *gr80211_block_alloc_input(gr80211_block *b, uint64_t bytes) { return b ? b->b->alloc_input((size_t)bytes) : nullptr; }

void gr80211_block_free_input(gr80211_block *b, void *p)
{
    if (b && p) b->b->free_input(p);
}

int gr80211_block_pin_input(gr80211_block *b, const void *p, uint64_t bytes)
{
    if (!b) return -1;
    return b->b->pin_input(p, (size_t)bytes) ? 0 : -1;
}

int gr80211_block_declare_pinned(gr80211_block *b, const void *p, uint64_t bytes)
{
    if (!b || !p || !bytes) return -1;
    b->b->declare_pinned(p, (size_t)bytes);
    return 0;
}

int gr80211_block_send_udp(gr80211_block *b, const char *host, int port)
{
    if (!b) return -1;
    if (!host) {
        b->udp.open("", 0);  // closes
        return 0;
    }
    return b->udp.open(host, port) ? 0 : -1;
}

gr80211_udp *gr80211_udp_open(const char *host, int port)
{
    if (!host) return nullptr;
    std::unique_ptr<gr80211_udp> u(new gr80211_udp());
    if (!u->sink.open(host, port)) return nullptr;
    return u.release();
}

int gr80211_udp_send(gr80211_udp *u, const uint8_t *blob, int len)
{
    if (!u || !blob || len < 0) return -1;
    pdu_t m;
    m.blob.assign(blob, blob + len);
    m.len = len;
    return u->sink.send(m) ? 0 : -1;
}

void gr80211_udp_close(gr80211_udp *u) { delete u; }

void gr80211_block_set_msg_handler(gr80211_block *b, gr80211_msg_fn fn, void *user)
{
    if (!b) return;
    b->fn = fn;
    b->user = user;
}

int gr80211_block_set_option(gr80211_block *b, int option, int value)
{
    if (!b) return -1;
    decode *d = dynamic_cast<decode *>(b->b.get());
    rx *r = dynamic_cast<rx *>(b->b.get());
    capture_rx *c = dynamic_cast<capture_rx *>(b->b.get());
    demod *dm = dynamic_cast<demod *>(b->b.get());
    if (option == GR80211_OPT_MU_POS) {
        if (value < 0 || value > 1 || d) return -1;
        if (r) r->set_mu_pos(value);
        if (c) c->set_mu_pos(value);
        if (dm) dm->set_mu_pos(value);
        return 0;
    }
    if (option == GR80211_OPT_DEBUG) {
        if (!d) return -1;
        d->set_debug(value != 0);
        return 0;
    }
    if (option == GR80211_OPT_FINISH_IN_CALL) {
        if (!d && !dm) return -1;
        if (d) d->set_finish_in_call(value != 0);
        if (dm) dm->set_finish_in_call(value != 0);
        return 0;
    }
    if (option == GR80211_OPT_MAX_HOLD) {
        if (value < 1 || dm) return -1;
        if (d) d->set_max_hold((size_t)value);
        else if (c) c->set_max_hold((size_t)value);
        else if (r) r->set_max_hold((size_t)value);
        else return -1;
        return 0;
    }
    if (option == GR80211_OPT_MIN_ROUND) {
        if (!c || value < 0) return -1;
        c->set_min_round((size_t)value);
        return 0;
    }
    if (option == GR80211_OPT_AMPDU_STRICT) {
        if (d) d->set_ampdu_strict(value != 0);
        if (r) r->set_ampdu_strict(value != 0);
        if (c) c->set_ampdu_strict(value != 0);
        return 0;
    }
    return -1;
}

int gr80211_block_n_inputs(const gr80211_block *b) { return b ? b->n_in : -1; }

int gr80211_block_forecast(gr80211_block *b, int noutput_items, int *ninput_items_required, int n_inputs)
{
    if (!b || !ninput_items_required || n_inputs != b->n_in) return -1;
    gr_vector_int req((size_t)n_inputs, 0);
    b->b->forecast(noutput_items, req);
    for (int i = 0; i < n_inputs; i++) ninput_items_required[i] = req[(size_t)i];
    return 0;
}

int gr80211_block_n_outputs(const gr80211_block *b) { return b ? b->n_out : -1; }

long long gr80211_block_pending_output(const gr80211_block *b)
{
    if (!b) return -1;
    if (const signal *sg = dynamic_cast<const signal *>(b->b.get())) return (long long)sg->pending_output();
    if (const demod *dm = dynamic_cast<const demod *>(b->b.get())) return (long long)dm->pending_output();
    return 0;
}

int gr80211_block_work(gr80211_block *b, int noutput_items, const int *ninput_items, const void *const *input_items,
                       int n_inputs, const gr80211_tag *tags, int n_tags, int *consumed)
{
    return gr80211_block_work_out(b, noutput_items, ninput_items, input_items, n_inputs, tags, n_tags, nullptr, 0, nullptr, 0,
                                  nullptr, consumed);
}

int gr80211_block_work_out(gr80211_block *b, int noutput_items, const int *ninput_items, const void *const *input_items,
                           int n_inputs, const gr80211_tag *tags, int n_tags, void *const *output_items, int n_outputs,
                           gr80211_tag *tags_out, int cap_tags_out, int *n_tags_out, int *consumed)
{
    if (!b || !ninput_items || !input_items || n_inputs != b->n_in || n_tags < 0 || (n_tags > 0 && !tags)) return -1;
    if (n_outputs != 0 && (n_outputs != b->n_out || !output_items)) return -1;
    if (n_tags_out) *n_tags_out = 0;
    // the caller hands over every tag of the offered windows with each call, so nothing is kept between calls
    b->b->clear_item_tags_in();
    for (int i = 0; i < n_tags; i++) {
        const gr80211_tag &t = tags[i];
        if (t.port >= (uint32_t)n_inputs || t.key >= (uint32_t)kNumTags) {
            b->err = "gr80211_block_work: bad tag";
            return -1;
        }
        tag_t h;
        h.offset = t.offset;
        h.key = kTagNames[t.key];
        h.value = t.value;
        if (t.cvec && t.cvec_len)
            for (uint32_t k = 0; k < t.cvec_len; k++) h.cvec.emplace_back(t.cvec[2 * k], t.cvec[2 * k + 1]);
        b->b->add_item_tag_in((int)t.port, h);
    }
    gr_vector_int nin(ninput_items, ninput_items + n_inputs);
    gr_vector_const_void_star in(input_items, input_items + n_inputs);
    gr_vector_void_star out;
    for (int i = 0; i < n_outputs; i++) out.push_back(output_items[i]);
    const int produced = b->b->general_work(noutput_items, nin, in, out);
    if (consumed) *consumed = b->b->consumed();
    b->out_tags.clear();
    for (int p = 0; p < b->b->n_outputs(); p++) {
        std::vector<tag_t> t = b->b->take_output_tags(p);
        for (tag_t &x : t) b->out_tags.push_back(std::move(x));  // only port 0 carries tags in these blocks
    }
    int nt = 0;
    bool tags_lost = false;
    for (const tag_t &x : b->out_tags) {
        if (!tags_out || nt >= cap_tags_out) {
            tags_lost = true;
            break;
        }
        int key = -1;
        for (int k = 0; k < kNumTags; k++)
            if (x.key == kTagNames[k]) key = k;
        if (key < 0) continue;
        gr80211_tag &o = tags_out[nt++];
        o.offset = x.offset;
        o.port = 0;
        o.key = (uint32_t)key;
        o.value = x.value;
        o.cvec = x.cvec.empty() ? nullptr : reinterpret_cast<const float *>(x.cvec.data());
        o.cvec_len = (uint32_t)x.cvec.size();
        o.reserved = 0;
    }
    if (n_tags_out) *n_tags_out = nt;
    b->b->commit(produced);
    if (!b->b->last_error().empty()) {
        b->err = b->b->last_error();
        return -2;
    }
    if (tags_lost) {
        b->err = "gr80211_block_work_out: " + std::to_string(b->out_tags.size()) + " output tags do not fit tags_out";
        return -3;
    }
    return produced;
}

int gr80211_block_flush(gr80211_block *b)
{
    if (!b) return -1;
    if (!b->b->flush()) {
        b->err = b->b->last_error();
        return -2;
    }
    return 0;
}

const char *gr80211_block_last_error(const gr80211_block *b) { return b ? b->err.c_str() : g_create_err.c_str(); }

}  // extern "C"
