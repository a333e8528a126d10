// Command-line scheduler for the host blocks: feeds recorded streams through general_work() in chunks the way
// the GNU Radio runtime would (bounded windows, consume_each honoured) and prints every published message.
//
//   gr80211_pump rx <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
//       triggers.txt: one "pos rad snr rssi" line per tagged sync byte; iq*.bin: interleaved float32 I/Q
//   gr80211_pump chain <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
//       the same input through three blocks, signal -> demod -> decode, joined by buffers of <chunk> items the
//       way a flowgraph joins them
//   gr80211_pump chaint <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
//       the three blocks on a thread each, as GNU Radio's thread-per-block scheduler runs them, joined by ring buffers
//       of 32 windows in memory the reading block handed out (page-locked unless GR80211_PUMP_PIN=0)
//   gr80211_pump capture <n_rx> <chunk> <iq0.bin> [iq1.bin]
//       no sync stream: frame detection runs on the GPU as well
//   gr80211_pump decode <chunk> <batch> <tags.txt> <llr.bin>
//       tags.txt: one "offset format len total cr mcs ampdu trellis" line per packet; llr.bin: float32
//
//   --udp <host>:<port> before the mode: every message also leaves as one UDP datagram holding its blob, as the
//       network.socket_pdu (UDP_CLIENT) block behind decode's "out" port sends it in the reference flowgraphs
//       (examples/cmu_v3/trxSta0.grc:301-313,1054)
//
// Output: "PDU <hex of the message blob>" per message (not with GR80211_PUMP_QUIET=1 in the environment), then
// "STAT ..." counters and the seconds spent between the first general_work call and the end of flush().
// Exit code 0 on success.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gr80211_blocks.h"

using namespace gr80211::host;

static bool read_file(const char *path, std::vector<char> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    const bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// A stream held in memory the block handed out as its input buffer (page-locked: what a GNU Radio 3.10 custom
// buffer type gives the scheduler), or in ordinary memory with GR80211_PUMP_PIN=0 in the environment (what a wrapper
// that leaves the buffers to the scheduler hands over).
struct stream_buf {
    block *owner = nullptr;
    char *p = nullptr;
    std::vector<char> plain;
    size_t bytes = 0;
    const char *data() const { return p ? p : plain.data(); }
    size_t size() const { return bytes; }
    bool load(block &b, const char *path)
    {
        std::vector<char> tmp;
        if (!read_file(path, tmp)) return false;
        bytes = tmp.size();
        const char *pin = getenv("GR80211_PUMP_PIN");
        if ((!pin || pin[0] != '0') && bytes) p = static_cast<char *>(b.alloc_input(bytes));
        if (p) {
            owner = &b;
            memcpy(p, tmp.data(), bytes);
        } else {
            plain.swap(tmp);
        }
        return true;
    }
    ~stream_buf()
    {
        if (p && owner) owner->free_input(p);
    }
};

static bool g_quiet = false;
static double g_seconds = 0.0;

static udp_sink g_udp;
static uint64_t g_udp_failed = 0;

static void on_msg(const std::string &, const pdu_t &m)
{
    if (g_udp.is_open() && !g_udp.send(m)) g_udp_failed++;
    if (g_quiet) return;
    std::string hex;
    static const char *d = "0123456789abcdef";
    for (uint8_t b : m.blob) {
        hex.push_back(d[b >> 4]);
        hex.push_back(d[b & 15]);
    }
    printf("PDU %s\n", hex.c_str());
}

// Calls general_work over [0, n_items) of the given ports in windows of at most `chunk` items.
static bool pump(block &b, const std::vector<const char *> &ports, const std::vector<size_t> &item_size, size_t n_items,
                 int chunk)
{
    size_t at = 0;
    int idle = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (at < n_items) {
        const int win = (int)std::min<size_t>((size_t)chunk, n_items - at);
        gr_vector_int nin(ports.size(), win);
        gr_vector_const_void_star in(ports.size());
        for (size_t p = 0; p < ports.size(); p++) in[p] = ports[p] + at * item_size[p];
        gr_vector_void_star out;
        b.general_work(win, nin, in, out);
        const int used = b.consumed();
        b.commit();
        if (!b.last_error().empty()) {
            fprintf(stderr, "block error: %s\n", b.last_error().c_str());
            return false;
        }
        at += (size_t)used;
        idle = used ? 0 : idle + 1;
        if (idle > 16) {
            fprintf(stderr, "no progress at item %zu\n", at);
            return false;
        }
    }
    if (!b.flush()) {
        fprintf(stderr, "flush failed: %s\n", b.last_error().c_str());
        return false;
    }
    g_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// A stream between two blocks: items and their tags (absolute offsets).
struct link {
    size_t item;
    std::vector<char> buf;   // items written by the producer; the reader is at byte `head`
    size_t head = 0;
    uint64_t base = 0;       // absolute index of the item at `head`
    std::vector<tag_t> tags; // in offset order; those before tag_head have been passed
    size_t tag_head = 0;
    explicit link(size_t it) : item(it) {}
    size_t items() const { return (buf.size() - head) / item; }
    const char *data() const { return buf.data() + head; }
    void consume(size_t n)
    {
        head += n * item;
        base += n;
        while (tag_head < tags.size() && tags[tag_head].offset < base) tag_head++;
        if (head > ((size_t)1 << 24) && head * 2 > buf.size()) {  // drop what has been read
            buf.erase(buf.begin(), buf.begin() + (long)head);
            head = 0;
            tags.erase(tags.begin(), tags.begin() + (long)tag_head);
            tag_head = 0;
        }
    }
};


// A stream between two blocks that run on threads of their own: a ring of `cap` items that one thread writes and one
// reads, followed by a mirror of its first `win` items, so that a window of up to `win` items is contiguous wherever
// it starts.  Tags travel beside it under a lock.
struct ring_link {
    size_t item, cap, win;
    char *buf = nullptr;
    block *owner = nullptr;  // the block whose alloc_input made buf (nullptr: malloc)
    std::atomic<uint64_t> w{0}, r{0};
    std::atomic<bool> closed{false};  // the writer has finished
    std::mutex m;
    std::deque<tag_t> tags;
    // mirror = false: a stream that is written once and read once from its start (a source) needs no mirror
    ring_link(size_t it, size_t windows, size_t window_items, block *pin, bool mirror = true)
        : item(it), cap(windows * window_items), win(mirror ? window_items : 0)
    {
        const size_t bytes = (cap + win) * item;
        if (pin) buf = static_cast<char *>(pin->alloc_input(bytes));
        if (buf) owner = pin;
        else buf = static_cast<char *>(malloc(bytes));
    }
    ~ring_link()
    {
        if (owner) owner->free_input(buf);
        else free(buf);
    }
    size_t readable() const { return (size_t)(w.load(std::memory_order_acquire) - r.load(std::memory_order_relaxed)); }
    size_t writable() const { return cap - (size_t)(w.load(std::memory_order_relaxed) - r.load(std::memory_order_acquire)); }
    const char *read_ptr() const { return buf + (r.load(std::memory_order_relaxed) % cap) * item; }
    char *write_ptr() { return buf + (w.load(std::memory_order_relaxed) % cap) * item; }
    // the writer has put n items at write_ptr(): keep ring and mirror alike, then publish
    void commit(size_t n)
    {
        const size_t at = (size_t)(w.load(std::memory_order_relaxed) % cap);
        // part of [at, at + n) below `win` goes to the mirror; part at or beyond `cap` (written into the mirror) goes to the front
        if (at < win) memcpy(buf + (cap + at) * item, buf + at * item, std::min(n, win - at) * item);
        if (at + n > cap) memcpy(buf, buf + cap * item, (at + n - cap) * item);
        w.store(w.load(std::memory_order_relaxed) + n, std::memory_order_release);
    }
    void consume(size_t n) { r.store(r.load(std::memory_order_relaxed) + n, std::memory_order_release); }
};

// One block of a flowgraph on its own thread: calls general_work whenever there is input or room for output.
static void run_block(block *b, std::vector<ring_link *> in, std::vector<ring_link *> out, int chunk, std::atomic<bool> *failed,
                      const char *name)
{
    bool flushed = false;
    std::vector<tag_t> tv;
    double t_work = 0, t_idle = 0, t_flush = 0;  // where the thread's time went (GR80211_BLOCK_PROFILE)
    unsigned long long n_calls = 0, n_idle = 0;
    const bool profile = getenv("GR80211_BLOCK_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        if (failed->load()) break;
        size_t avail = ~(size_t)0, room = ~(size_t)0;
        bool up_closed = true;
        for (ring_link *l : in) {
            up_closed = up_closed && l->closed.load(std::memory_order_acquire);
            avail = std::min(avail, l->readable());
        }
        for (ring_link *l : out) room = std::min(room, l->writable());
        if (in.empty()) avail = 0;
        const int nin_items = (int)std::min<size_t>(avail, (size_t)chunk);
        const int nout = out.empty() ? chunk : (int)std::min<size_t>(room, (size_t)chunk);
        if (nin_items == 0 && up_closed && !flushed) {
            const double tf = now();
            if (!b->flush()) {
                fprintf(stderr, "flush failed: %s\n", b->last_error().c_str());
                failed->store(true);
                break;
            }
            t_flush += now() - tf;
            flushed = true;
        }
        const double tw = now();
        gr_vector_int nin(in.size(), nin_items);
        gr_vector_const_void_star ip(in.size());
        gr_vector_void_star op(out.size());
        // tags change hands once: the block keeps those of items it has not consumed yet (block::commit drops the
        // others), so a call only brings the tags of the items that are new in its window
        for (size_t p = 0; p < in.size(); p++) {
            ip[p] = in[p]->read_ptr();
            const uint64_t r0 = in[p]->r.load(std::memory_order_relaxed);
            std::lock_guard<std::mutex> lk(in[p]->m);
            while (!in[p]->tags.empty() && in[p]->tags.front().offset < r0 + (uint64_t)nin_items) {
                if (in[p]->tags.front().offset >= r0) b->add_item_tag_in((int)p, std::move(in[p]->tags.front()));
                in[p]->tags.pop_front();
            }
        }
        for (size_t p = 0; p < out.size(); p++) op[p] = out[p]->write_ptr();
        const int produced = (nout > 0 || out.empty()) ? b->general_work(nout, nin, ip, op) : 0;
        const int used = (nout > 0 || out.empty()) ? b->consumed() : 0;
        if (!b->last_error().empty()) {
            fprintf(stderr, "block error: %s\n", b->last_error().c_str());
            failed->store(true);
            break;
        }
        for (size_t p = 0; p < out.size(); p++) {
            tv = b->take_output_tags((int)p);
            if (!tv.empty()) {
                std::lock_guard<std::mutex> lk(out[p]->m);
                for (tag_t &t : tv) out[p]->tags.push_back(std::move(t));
            }
            if (produced > 0) out[p]->commit((size_t)produced);
        }
        if (nout > 0 || out.empty()) b->commit(produced);
        for (ring_link *l : in) l->consume((size_t)used);
        n_calls++;
        if (used || produced) t_work += now() - tw;
        else t_idle += now() - tw;
        if (used == 0 && produced == 0) {
            n_idle++;
            if (flushed && nin_items == 0) {
                // everything upstream has been read and flushed: done once nothing is left to hand on
                if (out.empty() || room > 0) break;  // with no room the block was not asked: ask again
            }
            const double ts = now();
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            t_idle += now() - ts;
        }
    }
    if (profile)
        fprintf(stderr, "%-7s calls %llu (%llu without progress)  working %.3f s  idle %.3f s  flush %.3f s\n", name, n_calls, n_idle,
                t_work, t_idle, t_flush);
    for (ring_link *l : out) l->closed.store(true, std::memory_order_release);
}

// One general_work call of `b`: reads from `in` (all ports advance together), appends what it produces to `out`.
static bool step(block &b, std::vector<link *> in, std::vector<link *> out, int chunk, bool &progress)
{
    size_t avail = in.empty() ? 0 : in[0]->items();
    for (link *l : in) avail = std::min(avail, l->items());
    const int win = (int)std::min<size_t>(avail, (size_t)chunk);
    gr_vector_int nin(in.size(), win);
    gr_vector_const_void_star ip(in.size());
    b.clear_item_tags_in();
    for (size_t p = 0; p < in.size(); p++) {
        ip[p] = in[p]->data();
        for (size_t k = in[p]->tag_head; k < in[p]->tags.size() && in[p]->tags[k].offset < in[p]->base + (uint64_t)win; k++)
            b.add_item_tag_in((int)p, in[p]->tags[k]);
    }
    std::vector<std::vector<char>> ob(out.size());
    gr_vector_void_star op(out.size());
    for (size_t p = 0; p < out.size(); p++) {
        ob[p].resize((size_t)chunk * out[p]->item);
        op[p] = ob[p].data();
    }
    const int produced = b.general_work(chunk, nin, ip, op);
    const int used = b.consumed();
    for (size_t p = 0; p < out.size(); p++) {
        out[p]->buf.insert(out[p]->buf.end(), ob[p].begin(), ob[p].begin() + (long)((size_t)produced * out[p]->item));
        for (tag_t &t : b.take_output_tags((int)p)) out[p]->tags.push_back(std::move(t));
    }
    b.commit(produced);
    if (!b.last_error().empty()) {
        fprintf(stderr, "block error: %s\n", b.last_error().c_str());
        return false;
    }
    for (link *l : in) l->consume((size_t)used);
    progress = progress || used > 0 || produced > 0;
    return true;
}

static int run(int argc, char **argv);

int main(int argc, char **argv)
{
    if (const char *q = getenv("GR80211_PUMP_QUIET")) g_quiet = q[0] == '1';
    if (argc >= 3 && !strcmp(argv[1], "--udp")) {
        const std::string dest = argv[2];
        const size_t colon = dest.rfind(':');
        if (colon == std::string::npos || !g_udp.open(dest.substr(0, colon), atoi(dest.c_str() + colon + 1))) {
            fprintf(stderr, "cannot open a UDP socket to %s (want host:port)\n", dest.c_str());
            return 2;
        }
        argv[2] = argv[0];
        argv += 2;
        argc -= 2;
    }
    const int rc = run(argc, argv);
    if (g_udp.is_open()) printf("UDP datagrams=%llu failed=%llu\n", (unsigned long long)g_udp.datagrams_sent(), (unsigned long long)g_udp_failed);
    return rc;
}

static int run(int argc, char **argv)
{
    if (argc >= 7 && !strcmp(argv[1], "rx")) {
        const int n_rx = atoi(argv[2]), chunk = atoi(argv[3]), batch = atoi(argv[4]);
        rx::sptr b = rx::make(n_rx, batch, 0);
        if (!b->last_error().empty()) {
            fprintf(stderr, "block error: %s\n", b->last_error().c_str());
            return 1;
        }
        stream_buf iq0, iq1;
        if (!iq0.load(*b, argv[6]) || (n_rx == 2 && (argc < 8 || !iq1.load(*b, argv[7])))) {
            fprintf(stderr, "cannot read IQ file\n");
            return 2;
        }
        size_t n = iq0.size() / 8;
        if (n_rx == 2) n = std::min(n, iq1.size() / 8);
        std::vector<char> sync(n, 0);
        b->set_msg_handler(on_msg);
        FILE *f = fopen(argv[5], "r");
        if (!f) return 2;
        unsigned long long pos;
        float rad, snr, rssi;
        while (fscanf(f, "%llu %f %f %f", &pos, &rad, &snr, &rssi) == 4) {
            if (pos >= n) continue;
            sync[pos] = 1;
            tag_t t;
            t.offset = pos;
            t.key = "rad"; t.value = rad; b->add_item_tag_in(0, t);
            t.key = "snr"; t.value = snr; b->add_item_tag_in(0, t);
            t.key = "rssi"; t.value = rssi; b->add_item_tag_in(0, t);
        }
        fclose(f);
        std::vector<const char *> ports = {sync.data(), iq0.data()};
        std::vector<size_t> isz = {1, 8};
        if (n_rx == 2) {
            ports.push_back(iq1.data());
            isz.push_back(8);
        }
        const bool ok = pump(*b, ports, isz, n, chunk);
        printf("STAT triggers=%llu frames=%llu pdus=%llu\n", (unsigned long long)b->triggers_seen(),
               (unsigned long long)b->frames_decoded(), (unsigned long long)b->pdus_published());
        printf("TIME items=%zu seconds=%.6f\n", n, g_seconds);
        return ok ? 0 : 1;
    }
    if (argc >= 7 && !strcmp(argv[1], "chain")) {
        const int n_rx = atoi(argv[2]), chunk = atoi(argv[3]), batch = atoi(argv[4]);
        std::vector<char> iq0, iq1;
        if (!read_file(argv[6], iq0) || (n_rx == 2 && (argc < 8 || !read_file(argv[7], iq1)))) {
            fprintf(stderr, "cannot read IQ file\n");
            return 2;
        }
        size_t n = iq0.size() / 8;
        if (n_rx == 2) n = std::min(n, iq1.size() / 8);
        signal::sptr sg = signal::make(n_rx, batch, 0);
        demod::sptr dm = demod::make(n_rx, 0, 2, batch, 0);
        decode::sptr dc = decode::make(getenv("GR80211_PUMP_DEBUG") != nullptr, batch, 0);
        dc->set_msg_handler(on_msg);
        link l_sync(1), l_a(8), l_b(8), l_s0(8), l_s1(8), l_llr(4);
        l_sync.buf.assign(n, 0);
        l_a.buf.assign(iq0.begin(), iq0.begin() + (long)(n * 8));
        if (n_rx == 2) l_b.buf.assign(iq1.begin(), iq1.begin() + (long)(n * 8));
        FILE *f = fopen(argv[5], "r");
        if (!f) return 2;
        unsigned long long pos;
        float rad, snr, rssi;
        std::vector<tag_t> trig;
        while (fscanf(f, "%llu %f %f %f", &pos, &rad, &snr, &rssi) == 4) {
            if (pos >= n) continue;
            l_sync.buf[pos] = 1;
            tag_t t;
            t.offset = pos;
            t.key = "rad"; t.value = rad; trig.push_back(t);
            t.key = "snr"; t.value = snr; trig.push_back(t);
            t.key = "rssi"; t.value = rssi; trig.push_back(t);
        }
        fclose(f);
        std::stable_sort(trig.begin(), trig.end(), [](const tag_t &a, const tag_t &b) { return a.offset < b.offset; });
        l_sync.tags = trig;
        std::vector<link *> in_sg = {&l_sync, &l_a}, out_sg = {&l_s0}, in_dm = {&l_s0};
        if (n_rx == 2) {
            in_sg.push_back(&l_b);
            out_sg.push_back(&l_s1);
            in_dm.push_back(&l_s1);
        }
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true, flushed_sg = false, flushed_dm = false;
        for (int idle = 0; ok && idle < 4;) {
            bool progress = false;
            ok = step(*sg, in_sg, out_sg, chunk, progress) && step(*dm, in_dm, {&l_llr}, chunk, progress) &&
                 step(*dc, {&l_llr}, {}, chunk, progress);
            if (progress) {
                idle = 0;
                continue;
            }
            // nothing moves any more: end of stream for the next block that still buffers frames
            if (!flushed_sg) { ok = sg->flush(); flushed_sg = true; continue; }
            if (!flushed_dm) { ok = dm->flush(); flushed_dm = true; continue; }
            idle++;
        }
        ok = ok && dc->flush();
        g_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("STAT triggers=%llu frames=%llu pdus=%llu\n", (unsigned long long)sg->triggers_seen(),
               (unsigned long long)sg->frames_decoded(), (unsigned long long)dc->pdus_published());
        printf("TIME items=%zu seconds=%.6f\n", n, g_seconds);
        return ok ? 0 : 1;
    }
    if (argc >= 7 && !strcmp(argv[1], "chaint")) {
        const int n_rx = atoi(argv[2]), chunk = atoi(argv[3]), batch = atoi(argv[4]);
        std::vector<char> iq0, iq1;
        if (!read_file(argv[6], iq0) || (n_rx == 2 && (argc < 8 || !read_file(argv[7], iq1)))) {
            fprintf(stderr, "cannot read IQ file\n");
            return 2;
        }
        size_t n = iq0.size() / 8;
        if (n_rx == 2) n = std::min(n, iq1.size() / 8);
        signal::sptr sg = signal::make(n_rx, batch, 0);
        demod::sptr dm = demod::make(n_rx, 0, 2, batch, 0);
        decode::sptr dc = decode::make(getenv("GR80211_PUMP_DEBUG") != nullptr, batch, 0);
        dc->set_msg_handler(on_msg);
        const char *pin = getenv("GR80211_PUMP_PIN");
        const bool pinned = !pin || pin[0] != '0';
        // the sources: whole streams, written before the clock starts (one "window" as long as the stream)
        ring_link l_sync(1, 1, n + 1, nullptr, false), l_a(8, 1, n + 1, pinned ? sg.get() : nullptr, false),
            l_b(8, 1, n_rx == 2 ? n + 1 : 1, pinned ? sg.get() : nullptr, false);
        memset(l_sync.buf, 0, n + 1);
        memcpy(l_a.buf, iq0.data(), n * 8);
        if (n_rx == 2) memcpy(l_b.buf, iq1.data(), n * 8);
        FILE *f = fopen(argv[5], "r");
        if (!f) return 2;
        unsigned long long pos;
        float rad, snr, rssi;
        std::vector<tag_t> trig;
        while (fscanf(f, "%llu %f %f %f", &pos, &rad, &snr, &rssi) == 4) {
            if (pos >= n) continue;
            l_sync.buf[pos] = 1;
            tag_t t;
            t.offset = pos;
            t.key = "rad"; t.value = rad; trig.push_back(t);
            t.key = "snr"; t.value = snr; trig.push_back(t);
            t.key = "rssi"; t.value = rssi; trig.push_back(t);
        }
        fclose(f);
        std::stable_sort(trig.begin(), trig.end(), [](const tag_t &a, const tag_t &b) { return a.offset < b.offset; });
        l_sync.tags.assign(trig.begin(), trig.end());
        for (ring_link *l : {&l_sync, &l_a, &l_b}) {
            l->w.store(n);
            l->closed.store(true);
        }
        // windows a ring between two blocks holds (a scheduler's buffer size in windows)
        const size_t W = getenv("GR80211_PUMP_RING_WINDOWS") ? (size_t)std::max(2, atoi(getenv("GR80211_PUMP_RING_WINDOWS"))) : 32;
        ring_link l_s0(8, W, (size_t)chunk, pinned ? dm.get() : nullptr), l_s1(8, W, (size_t)chunk, pinned ? dm.get() : nullptr),
            l_llr(4, W, (size_t)chunk, pinned ? dc.get() : nullptr);
        std::vector<ring_link *> in_sg = {&l_sync, &l_a}, out_sg = {&l_s0}, in_dm = {&l_s0};
        if (n_rx == 2) {
            in_sg.push_back(&l_b);
            out_sg.push_back(&l_s1);
            in_dm.push_back(&l_s1);
        }
        // a block's output buffer is the next block's input buffer: tell the producer when it is page-locked
        for (ring_link *l : {&l_s0, &l_s1})
            if (l->owner) sg->declare_pinned(l->buf, (l->cap + l->win) * l->item);
        if (l_llr.owner) dm->declare_pinned(l_llr.buf, (l_llr.cap + l_llr.win) * l_llr.item);
        std::atomic<bool> failed{false};
        const auto t0 = std::chrono::steady_clock::now();
        std::thread t_sg(run_block, sg.get(), in_sg, out_sg, chunk, &failed, "signal");
        std::thread t_dm(run_block, dm.get(), in_dm, std::vector<ring_link *>{&l_llr}, chunk, &failed, "demod");
        std::thread t_dc(run_block, dc.get(), std::vector<ring_link *>{&l_llr}, std::vector<ring_link *>{}, chunk, &failed, "decode");
        t_sg.join();
        t_dm.join();
        t_dc.join();
        g_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("STAT triggers=%llu frames=%llu pdus=%llu\n", (unsigned long long)sg->triggers_seen(),
               (unsigned long long)sg->frames_decoded(), (unsigned long long)dc->pdus_published());
        printf("TIME items=%zu seconds=%.6f\n", n, g_seconds);
        return failed.load() ? 1 : 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "capture")) {
        const int n_rx = atoi(argv[2]), chunk = atoi(argv[3]);
        capture_rx::sptr b = capture_rx::make(n_rx, 0);
        stream_buf iq0, iq1;
        if (!iq0.load(*b, argv[4]) || (n_rx == 2 && (argc < 6 || !iq1.load(*b, argv[5])))) {
            fprintf(stderr, "cannot read IQ file\n");
            return 2;
        }
        size_t n = iq0.size() / 8;
        if (n_rx == 2) n = std::min(n, iq1.size() / 8);
        b->set_msg_handler(on_msg);
        std::vector<const char *> ports = {iq0.data()};
        std::vector<size_t> isz = {8};
        if (n_rx == 2) {
            ports.push_back(iq1.data());
            isz.push_back(8);
        }
        const bool ok = pump(*b, ports, isz, n, chunk);
        printf("STAT frames=%llu pdus=%llu\n", (unsigned long long)b->frames_decoded(),
               (unsigned long long)b->pdus_published());
        printf("TIME items=%zu seconds=%.6f\n", n, g_seconds);
        return ok ? 0 : 1;
    }
    if (argc >= 6 && !strcmp(argv[1], "decode")) {
        const int chunk = atoi(argv[2]), batch = atoi(argv[3]);
        decode::sptr b = decode::make(getenv("GR80211_PUMP_DEBUG") != nullptr, batch, 0);
        if (!b->last_error().empty()) {
            fprintf(stderr, "block error: %s\n", b->last_error().c_str());
            return 1;
        }
        stream_buf llr;
        if (!llr.load(*b, argv[5])) {
            fprintf(stderr, "cannot read LLR file\n");
            return 2;
        }
        const size_t n = llr.size() / 4;
        b->set_msg_handler(on_msg);
        FILE *f = fopen(argv[4], "r");
        if (!f) return 2;
        unsigned long long off;
        long v[7];
        while (fscanf(f, "%llu %ld %ld %ld %ld %ld %ld %ld", &off, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]) == 8) {
            static const char *keys[7] = {"format", "len", "total", "cr", "mcs", "ampdu", "trellis"};
            for (int k = 0; k < 7; k++) {
                tag_t t;
                t.offset = off;
                t.key = keys[k];
                t.value = (double)v[k];
                b->add_item_tag_in(0, t);
            }
        }
        fclose(f);
        const bool ok = pump(*b, {llr.data()}, {4}, n, chunk);
        printf("STAT packets=%llu pdus=%llu\n", (unsigned long long)b->packets_seen(),
               (unsigned long long)b->pdus_published());
        printf("TIME items=%zu seconds=%.6f\n", n, g_seconds);
        return ok ? 0 : 1;
    }
    fprintf(stderr, "usage: gr80211_pump [--udp host:port] <mode> ...\n"
                    "       gr80211_pump rx <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump chain <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump chaint <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump capture <n_rx> <chunk> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump decode <chunk> <batch> <tags.txt> <llr.bin>\n");
    return 2;
}
