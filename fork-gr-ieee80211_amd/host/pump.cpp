// Command-line scheduler for the host blocks: feeds recorded streams through general_work() in chunks the way
// the GNU Radio runtime would (bounded windows, consume_each honoured) and prints every published message.
//
//   gr80211_pump rx <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]
//       triggers.txt: one "pos rad snr rssi" line per tagged sync byte; iq*.bin: interleaved float32 I/Q
//   gr80211_pump capture <n_rx> <chunk> <iq0.bin> [iq1.bin]
//       no sync stream: frame detection runs on the GPU as well
//   gr80211_pump decode <chunk> <batch> <tags.txt> <llr.bin>
//       tags.txt: one "offset format len total cr mcs ampdu trellis" line per packet; llr.bin: float32
//
// Output: "PDU <hex of the message blob>" per message (not with GR80211_PUMP_QUIET=1 in the environment), then
// "STAT ..." counters and the seconds spent between the first general_work call and the end of flush().
// Exit code 0 on success.
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

static bool g_quiet = false;
static double g_seconds = 0.0;

static void on_msg(const std::string &, const pdu_t &m)
{
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

int main(int argc, char **argv)
{
    if (const char *q = getenv("GR80211_PUMP_QUIET")) g_quiet = q[0] == '1';
    if (argc >= 7 && !strcmp(argv[1], "rx")) {
        const int n_rx = atoi(argv[2]), chunk = atoi(argv[3]), batch = atoi(argv[4]);
        std::vector<char> iq0, iq1;
        if (!read_file(argv[6], iq0) || (n_rx == 2 && (argc < 8 || !read_file(argv[7], iq1)))) {
            fprintf(stderr, "cannot read IQ file\n");
            return 2;
        }
        size_t n = iq0.size() / 8;
        if (n_rx == 2) n = std::min(n, iq1.size() / 8);
        std::vector<char> sync(n, 0);
        rx::sptr b = rx::make(n_rx, batch, 0);
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
    if (argc >= 5 && !strcmp(argv[1], "capture")) {
        const int n_rx = atoi(argv[2]), chunk = atoi(argv[3]);
        std::vector<char> iq0, iq1;
        if (!read_file(argv[4], iq0) || (n_rx == 2 && (argc < 6 || !read_file(argv[5], iq1)))) {
            fprintf(stderr, "cannot read IQ file\n");
            return 2;
        }
        size_t n = iq0.size() / 8;
        if (n_rx == 2) n = std::min(n, iq1.size() / 8);
        capture_rx::sptr b = capture_rx::make(n_rx, 0);
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
        std::vector<char> llr;
        if (!read_file(argv[5], llr)) {
            fprintf(stderr, "cannot read LLR file\n");
            return 2;
        }
        const size_t n = llr.size() / 4;
        decode::sptr b = decode::make(false, batch, 0);
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
    fprintf(stderr, "usage: gr80211_pump rx <n_rx> <chunk> <batch> <triggers.txt> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump capture <n_rx> <chunk> <iq0.bin> [iq1.bin]\n"
                    "       gr80211_pump decode <chunk> <batch> <tags.txt> <llr.bin>\n");
    return 2;
}
