// Host-side construction of the constant tables from the IEEE 802.11-2016 definitions.
#include <cmath>
#include <cstring>

#include "rx_common.h"

namespace gr80211 {

// L-LTF sequence on subcarriers -26..26 (IEEE 802.11-2016 equation 17-8).
static const signed char kLtf[53] = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1,
                                     -1, 1, 1, 1, 1, 0, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1,
                                     1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};

static bool pilot_bin(int b) { return b == 7 || b == 21 || b == 43 || b == 57; }

// Interleaver, first and second permutation: coded bit k is sent at position j.
static int interleave_pos(int k, int ncbpss, int nbpscs, int ncol)
{
    const int nrow = ncbpss / ncol;
    const int s = nbpscs / 2 > 1 ? nbpscs / 2 : 1;
    const int i = nrow * (k % ncol) + k / ncol;
    return s * (i / s) + (i + ncbpss - (ncol * i) / ncbpss) % s;
}

void build_tables(Tables &t)
{
    std::memset(&t, 0, sizeof(t));
    for (int k = -26; k <= 26; k++) {
        t.ltf_l[k & 63] = (float)kLtf[k + 26];
        t.ltf_nl[k & 63] = (float)kLtf[k + 26];
    }
    t.ltf_nl[(-28) & 63] = 1.0f;
    t.ltf_nl[(-27) & 63] = 1.0f;
    t.ltf_nl[27] = -1.0f;
    t.ltf_nl[28] = -1.0f;

    // Pilot polarity = scrambler output for the all-ones state, 0 -> +1, 1 -> -1.
    int st = 0x7f;
    for (int i = 0; i < 127; i++) {
        const int fb = ((st >> 6) ^ (st >> 3)) & 1;
        t.pilot_p[i] = fb ? -1.0f : 1.0f;
        st = ((st << 1) | fb) & 0x7f;
    }
    t.pilot_p[127] = 0.0f;

    const int nb_l[4] = {1, 2, 4, 6};
    for (int m = 0; m < 4; m++) {
        const int n = 48 * nb_l[m];
        for (int k = 0; k < n; k++) {
            t.deint_l[m][interleave_pos(k, n, nb_l[m], 16)] = (int16_t)k;
            t.intl_l[m][k] = (int16_t)interleave_pos(k, n, nb_l[m], 16);
        }
    }
    const int nb_n[5] = {1, 2, 4, 6, 8};
    for (int m = 0; m < 5; m++) {
        const int n = 52 * nb_n[m];
        for (int k = 0; k < n; k++) {
            const int j = interleave_pos(k, n, nb_n[m], 13);
            t.deint_nl[m][j] = (int16_t)k;
            const int r = ((j - 2 * 11 * nb_n[m]) % n + n) % n;  // third permutation, stream 2
            t.deint_nl2[m][r] = (int16_t)k;
            t.intl_nl[m][k] = (int16_t)j;
            t.intl_nl2[m][k] = (int16_t)r;
        }
    }
    for (int i = 0; i < 64; i++) {
        t.lsig_demap[i] = -1;
        t.data_idx_l[i] = -1;
        t.data_idx_nl[i] = -1;
    }
    int j = 0;
    for (int k = -26; k <= 26; k++) {
        const int b = k & 63;
        if (k == 0 || pilot_bin(b)) continue;
        t.data_idx_l[b] = (int16_t)j;
        t.lsig_demap[b] = t.deint_l[0][j];
        j++;
    }
    j = 0;
    for (int k = -28; k <= 28; k++) {
        const int b = k & 63;
        if (k == 0 || pilot_bin(b)) continue;
        t.data_idx_nl[b] = (int16_t)j++;
    }
    for (int i = 0; i < 32; i++) {
        const double a = -2.0 * M_PI * (double)i / 64.0;
        t.tw_re[i] = (float)std::cos(a);
        t.tw_im[i] = (float)std::sin(a);
    }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        t.crc[i] = c;
    }
}

uint32_t tables_checksum(const Tables &t)
{
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&t);
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < sizeof(Tables); i++) c = t.crc[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

}  // namespace gr80211
