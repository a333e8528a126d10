// Command-line scheduler for the host blocks: feeds recorded streams through general_work() in chunks the way
// the GNU Radio runtime would (bounded windows, consume_each honoured) and prints every published message.
//
//   gr80211_pump rx <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
//       triggers.txt: one "pos rad snr rssi" line per tagged sync byte; iq*.bin: interleaved float32 I/Q
//   gr80211_pump chain <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
//       the same input through three blocks, signal -> demod -> decode, joined by buffers of <chunk> items the
//       way a flowgraph joins them
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
#include <string>
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
        decode::sptr dc = decode::make(false, batch, 0);
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
        decode::sptr b = decode::make(false, batch, 0);
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
                    "       gr80211_pump capture <n_rx> <chunk> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump decode <chunk> <batch> <tags.txt> <llr.bin>\n");
    return 2;
}
