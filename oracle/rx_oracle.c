/*
 * CPU oracle for the 802.11 OFDM RX hot path.  TEST INFRASTRUCTURE ONLY (see rx_oracle.h).
 *
 * Scalar float32 restatement of the reference algorithm, same arithmetic order:
 * C99 `float complex` arithmetic compiled by gcc yields the same operations g++ emits
 * for std::complex<float> (inline (ac-bd, ad+bc) products, __divsc3 quotients, hypotf
 * magnitudes, componentwise complex*real and complex/real).
 *
 * Constant tables are generated from the IEEE 802.11-2016 definitions (17.3.3 L-LTF,
 * 17.3.5.7/19.3.11.8 interleavers, 17.3.5.10 pilot polarity, 17.3.5.6 convolutional code)
 * instead of being transcribed; tests compare them with the reference's tables.
 *
 * Parity status: the reference C++ cannot be built in this image (every source includes
 * GNU Radio headers that are absent), so this oracle is pinned by the reference's own IQ
 * fixtures + expected MPDU, and by frames / Viterbi vectors produced by the reference's
 * Python tool box (see tests/golden/).
 */
#include "rx_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef float complex cf;

#define SHIFT 8 /* C8P_SYM_SAMP_SHIFT, lib/cloud80211phy.h:33 */

enum { QAM_BPSK = 0, QAM_QBPSK = 1, QAM_QPSK = 2, QAM_16 = 3, QAM_64 = 4, QAM_256 = 5 };
enum { CR_12 = 0, CR_23 = 1, CR_34 = 2, CR_56 = 3 };

/* ------------------------------------------------------------------ tables */
static int g_init = 0;
static float T_LTF_L[64];   /* legacy LTF sign per FFT bin (0 on unused tones) */
static float T_LTF_NL[64];  /* HT/VHT LTF sign per FFT bin */
static float T_PILOT_P[127];
static int T_LSIG_DEMAP[64]; /* FFT bin -> deinterleaved SIG position, -1 if not data */
static int T_DEINT_L[4][288];
static int T_DEINT_NL[5][416];
static int T_DEINT_NL2[5][416];
static int T_NEXT[64][2], T_OUT[64][2];
static uint32_t T_CRC[256];
static float T_TW_RE[32], T_TW_IM[32];
static int T_BITREV[64];

static const int PUNC_12[2] = {1, 1};
static const int PUNC_23[4] = {1, 1, 1, 0};
static const int PUNC_34[6] = {1, 1, 1, 0, 0, 1};
static const int PUNC_56[10] = {1, 1, 1, 0, 0, 1, 1, 0, 0, 1};

/* L-LTF, subcarriers -26..26 (IEEE 802.11-2016 eq. 17-8). */
static const signed char LTF_SEQ[53] = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1,
                                        -1, 1, 1, 1, 1, 0, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1,
                                        1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};

static int is_pilot_bin(int i) { return i == 7 || i == 21 || i == 43 || i == 57; }

/* Interleaver first+second permutation, k -> j (17.3.5.7 / 19.3.11.8.x). */
static int intlv_j(int k, int ncbpss, int nbpscs, int ncol)
{
    int nrow = ncbpss / ncol;
    int s = nbpscs / 2 > 1 ? nbpscs / 2 : 1;
    int i = nrow * (k % ncol) + k / ncol;
    return s * (i / s) + (i + ncbpss - (ncol * i) / ncbpss) % s;
}

static int nl_idx(int nbpscs) { return nbpscs == 1 ? 0 : nbpscs == 2 ? 1 : nbpscs == 4 ? 2 : nbpscs == 6 ? 3 : 4; }

void orc_init(void)
{
    if (g_init) return;
    for (int i = 0; i < 64; i++) { T_LTF_L[i] = 0.0f; T_LTF_NL[i] = 0.0f; }
    for (int k = -26; k <= 26; k++) {
        T_LTF_L[k & 63] = (float)LTF_SEQ[k + 26];
        T_LTF_NL[k & 63] = (float)LTF_SEQ[k + 26];
    }
    /* HT/VHT-LTF 20 MHz adds {1,1} at -28,-27 and {-1,-1} at 27,28 (19.3.9.4.6). */
    T_LTF_NL[(-28) & 63] = 1.0f; T_LTF_NL[(-27) & 63] = 1.0f;
    T_LTF_NL[27] = -1.0f; T_LTF_NL[28] = -1.0f;

    /* Pilot polarity p_0..126 = scrambler output for the all-ones seed, 0 -> +1, 1 -> -1. */
    {
        int st = 0x7f;
        for (int i = 0; i < 127; i++) {
            int fb = ((st >> 6) ^ (st >> 3)) & 1;
            T_PILOT_P[i] = fb ? -1.0f : 1.0f;
            st = ((st << 1) | fb) & 0x7f;
        }
    }
    /* Legacy deinterleave maps: out[map[j]] = in[j]. */
    {
        const int nb[4] = {1, 2, 4, 6};
        for (int m = 0; m < 4; m++) {
            int n = 48 * nb[m];
            for (int k = 0; k < n; k++) T_DEINT_L[m][intlv_j(k, n, nb[m], 16)] = k;
        }
        const int nbn[5] = {1, 2, 4, 6, 8};
        for (int m = 0; m < 5; m++) {
            int n = 52 * nbn[m];
            for (int k = 0; k < n; k++) {
                int j = intlv_j(k, n, nbn[m], 13);
                T_DEINT_NL[m][j] = k;
                /* third permutation, stream 2 of 2: r = (j - 2*11*nBPSCS) mod nCBPSS */
                int r = ((j - 2 * 11 * nbn[m]) % n + n) % n;
                T_DEINT_NL2[m][r] = k;
            }
        }
    }
    /* L-SIG: FFT bin -> data index (ascending frequency) -> deinterleaved position. */
    {
        int j = 0;
        for (int i = 0; i < 64; i++) T_LSIG_DEMAP[i] = -1;
        for (int k = -26; k <= 26; k++) {
            int b = k & 63;
            if (k == 0 || is_pilot_bin(b)) continue;
            T_LSIG_DEMAP[b] = T_DEINT_L[0][j++];
        }
    }
    /* Trellis: decoder state i holds the six previous inputs, newest in bit 5. */
    for (int i = 0; i < 64; i++)
        for (int b = 0; b < 2; b++) {
            int r = b;
            for (int m = 0; m < 6; m++) r |= ((i >> (5 - m)) & 1) << (m + 1);
            int a = __builtin_parity(r & 0155), bb = __builtin_parity(r & 0117);
            T_NEXT[i][b] = (i >> 1) | (b << 5);
            T_OUT[i][b] = 2 * a + bb;
        }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        T_CRC[i] = c;
    }
    for (int i = 0; i < 32; i++) {
        double a = -2.0 * M_PI * (double)i / 64.0;
        T_TW_RE[i] = (float)cos(a);
        T_TW_IM[i] = (float)sin(a);
    }
    for (int i = 0; i < 64; i++) {
        int r = 0;
        for (int b = 0; b < 6; b++) r |= ((i >> b) & 1) << (5 - b);
        T_BITREV[i] = r;
    }
    g_init = 1;
}

void orc_get_table(const char *name, int *out, int cap, int *n)
{
    orc_init();
    const int *src = NULL;
    int len = 0, tmp[832];
    if (!strcmp(name, "lsig_demap")) { src = T_LSIG_DEMAP; len = 64; }
    else if (!strncmp(name, "deint_l", 7)) { int m = name[7] - '0'; src = T_DEINT_L[m]; len = 48 * (m == 0 ? 1 : m == 1 ? 2 : m == 2 ? 4 : 6); }
    else if (!strncmp(name, "deint_nl2_", 10)) { int m = name[10] - '0'; const int nb[5] = {1, 2, 4, 6, 8}; src = T_DEINT_NL2[m]; len = 52 * nb[m]; }
    else if (!strncmp(name, "deint_nl", 8)) { int m = name[8] - '0'; const int nb[5] = {1, 2, 4, 6, 8}; src = T_DEINT_NL[m]; len = 52 * nb[m]; }
    else if (!strcmp(name, "sv_next")) { src = &T_NEXT[0][0]; len = 128; }
    else if (!strcmp(name, "sv_out")) { src = &T_OUT[0][0]; len = 128; }
    else if (!strcmp(name, "pilot_p")) { for (int i = 0; i < 127; i++) tmp[i] = (int)T_PILOT_P[i]; src = tmp; len = 127; }
    else if (!strcmp(name, "ltf_l")) { for (int i = 0; i < 64; i++) tmp[i] = (int)T_LTF_L[i]; src = tmp; len = 64; }
    else if (!strcmp(name, "ltf_nl")) { for (int i = 0; i < 64; i++) tmp[i] = (int)T_LTF_NL[i]; src = tmp; len = 64; }
    *n = len;
    for (int i = 0; i < len && i < cap; i++) out[i] = src[i];
}

/* ------------------------------------------------------------------ small helpers */
uint32_t orc_crc32(const uint8_t *p, int n)
{
    orc_init();
    uint32_t c = 0xFFFFFFFFu;
    for (int i = 0; i < n; i++) c = T_CRC[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* lib/cloud80211phy.cc:2622-2644 */
void orc_bcc_encode(const uint8_t *in, uint8_t *out, int n)
{
    int st = 0;
    for (int i = 0; i < n; i++) {
        st = ((st << 1) & 0x7e) | in[i];
        out[2 * i] = (uint8_t)__builtin_parity(st & 0155);
        out[2 * i + 1] = (uint8_t)__builtin_parity(st & 0117);
    }
}

/* lib/cloud80211phy.cc:1329-1403: CRC-8 x^8+x^2+x+1, init 0xFF, output inverted, MSB first. */
static unsigned crc8_run(const uint8_t *in, int n)
{
    unsigned c = 0xff;
    for (int i = 0; i < n; i++) {
        c <<= 1;
        if (c & 0x100) c = (c + 1) ^ 0x06;
        if (in[i]) c ^= 0x07;
    }
    return c;
}
void orc_crc8_bits(const uint8_t *in, int n, uint8_t *out8)
{
    unsigned c = 0xff - (crc8_run(in, n) & 0xff);
    for (int i = 0; i < 8; i++) out8[i] = (c >> (7 - i)) & 1;
}
static int crc8_check(const uint8_t *in, int n, const uint8_t *crc)
{
    unsigned c = crc8_run(in, n);
    for (int i = 0; i < 8; i++)
        if (crc[i]) c ^= 1u << (7 - i);
    return (c & 0xff) == 0xff;
}

/* Radix-2 DIT, float32.  The reference calls FFTW through gr::fft (lib/demod_impl.cc:541-547);
 * any exact forward DFT agrees with it to float rounding. */
static void fft64(const cf *in, cf *out)
{
    cf x[64];
    for (int i = 0; i < 64; i++) x[T_BITREV[i]] = in[i];
    for (int half = 1; half < 64; half <<= 1) {
        int step = 32 / half;
        for (int base = 0; base < 64; base += 2 * half)
            for (int k = 0; k < half; k++) {
                cf w = CMPLXF(T_TW_RE[k * step], T_TW_IM[k * step]);
                cf a = x[base + k];
                cf b = x[base + k + half];
                float br = crealf(b) * crealf(w) - cimagf(b) * cimagf(w);
                float bi = crealf(b) * cimagf(w) + cimagf(b) * crealf(w);
                x[base + k] = CMPLXF(crealf(a) + br, cimagf(a) + bi);
                x[base + k + half] = CMPLXF(crealf(a) - br, cimagf(a) - bi);
            }
    }
    memcpy(out, x, sizeof(x));
}

void orc_fft64(const float *in_ri, float *out_ri)
{
    orc_init();
    fft64((const cf *)in_ri, (cf *)out_ri);
}

/* lib/cloud80211phy.cc:2001-2088 (svSigDecoder::decode). */
void orc_sig_viterbi(const float *llrv, uint8_t *bits, int T)
{
    orc_init();
    if (T < 0 || T > 48) return;
    static __thread int his[64][49];
    int seq[49];
    float m0[64], m1[64], *pre = m0, *cur = m1;
    for (int i = 0; i < 64; i++) {
        for (int j = 0; j <= T; j++) his[i][j] = 0;
        m0[i] = -1000000000000000.0f;
        m1[i] = -1000000000000000.0f;
    }
    m0[0] = 0;
    for (int t = 0; t < T; t++) {
        float t0 = *llrv++, t1 = *llrv++;
        float tab[4] = {0, t1, t0, t1 + t0};
        for (int i = 0; i < 64; i++) {
            float a0 = pre[i] + tab[T_OUT[i][0]];
            float a1 = pre[i] + tab[T_OUT[i][1]];
            int n0 = T_NEXT[i][0], n1 = T_NEXT[i][1];
            if (a0 > cur[n0]) { cur[n0] = a0; his[n0][t + 1] = i; }
            if (a1 > cur[n1]) { cur[n1] = a1; his[n1][t + 1] = i; }
        }
        float *tmp = pre; pre = cur; cur = tmp;
        for (int i = 0; i < 64; i++) cur[i] = -1000000000000000.0f;
    }
    seq[T] = 0;
    for (int j = T; j > 0; j--) seq[j - 1] = his[seq[j]][j];
    for (int j = 0; j < T; j++) bits[j] = (seq[j + 1] == T_NEXT[seq[j]][1]) ? 1 : 0;
}

/* ------------------------------------------------------------------ SIG parsing */
/* lib/cloud80211phy.cc:650-728 */
static int check_legacy(const uint8_t *b, int *mcs, int *len, int *ndbps)
{
    if (!b[3]) return 0;
    if (b[4]) return 0;
    unsigned sum = 0;
    for (int i = 0; i < 17; i++) sum += b[i];
    if ((sum & 1) ^ b[17]) return 0;
    int rate = 0;
    for (int i = 0; i < 4; i++) rate |= b[i] << i;
    switch (rate) {
    case 11: *mcs = 0; *ndbps = 24; break;
    case 15: *mcs = 1; *ndbps = 36; break;
    case 10: *mcs = 2; *ndbps = 48; break;
    case 14: *mcs = 3; *ndbps = 72; break;
    case 9: *mcs = 4; *ndbps = 96; break;
    case 13: *mcs = 5; *ndbps = 144; break;
    case 8: *mcs = 6; *ndbps = 192; break;
    case 12: *mcs = 7; *ndbps = 216; break;
    default: *mcs = 0; *ndbps = 24; break;
    }
    *len = 0;
    for (int i = 0; i < 12; i++) *len |= b[i + 5] << i;
    if (*len > 4095 || *len < 14) return 0;
    return 1;
}

/* lib/cloud80211phy.cc:730-751 */
static int check_ht(const uint8_t *b)
{
    if (b[26] != 1) return 0;
    if (!crc8_check(b, 34, &b[34])) return 0;
    if (b[5] + b[6] + b[7] + b[28] + b[29] + b[30] + b[32] + b[33]) return 0;
    return 1;
}

/* lib/cloud80211phy.cc:753-771 */
static int check_vhta(const uint8_t *b)
{
    if (b[2] != 1 || b[23] != 1 || b[33] != 1) return 0;
    if (!crc8_check(b, 34, &b[34])) return 0;
    if (b[0] + b[1]) return 0;
    return 1;
}

static void set_rate(orc_mod_t *m, int mod, int nbpscs, int cr)
{
    m->mod = mod; m->nBPSCS = nbpscs; m->cr = cr;
}

static void set_ndbps(orc_mod_t *m)
{
    switch (m->cr) {
    case CR_12: m->nDBPS = m->nCBPS / 2; break;
    case CR_23: m->nDBPS = (m->nCBPS * 2) / 3; break;
    case CR_34: m->nDBPS = (m->nCBPS * 3) / 4; break;
    default: m->nDBPS = (m->nCBPS * 5) / 6; break;
    }
}

/* lib/cloud80211phy.cc:773-850 */
static void parse_legacy(int mcs, int len, orc_mod_t *m)
{
    static const int mod[8] = {QAM_BPSK, QAM_BPSK, QAM_QPSK, QAM_QPSK, QAM_16, QAM_16, QAM_64, QAM_64};
    static const int cr[8] = {CR_12, CR_34, CR_12, CR_34, CR_12, CR_34, CR_23, CR_34};
    static const int ndbps[8] = {24, 36, 48, 72, 96, 144, 192, 216};
    static const int nbp[8] = {1, 1, 2, 2, 4, 4, 6, 6};
    m->mcs = mcs; m->mod = mod[mcs]; m->cr = cr[mcs]; m->nDBPS = ndbps[mcs];
    m->nBPSCS = nbp[mcs]; m->nCBPS = 48 * nbp[mcs];
    m->len = len; m->nCBPSS = m->nCBPS; m->nSD = 48; m->nSS = 1; m->nLTF = 0;
    m->format = ORC_F_L; m->nSymSamp = 80;
    m->nSym = (len * 8 + 22) / m->nDBPS + (((len * 8 + 22) % m->nDBPS) != 0);
    m->ampdu = 0;
}

static int ltf_count(int nss) { return nss == 1 ? 1 : nss == 2 ? 2 : 4; }

/* lib/cloud80211phy.cc:852-998 */
static void parse_ht(const uint8_t *b, orc_mod_t *m)
{
    int mcs = 0, len = 0;
    for (int i = 0; i < 7; i++) mcs |= b[i] << i;
    for (int i = 0; i < 16; i++) len |= b[i + 8] << i;
    m->format = ORC_F_HT;
    m->nSymSamp = b[31] ? 72 : 80;
    m->ampdu = b[27] ? 1 : 0;
    m->mcs = mcs;
    switch (mcs % 8) {
    case 0: set_rate(m, QAM_BPSK, 1, CR_12); break;
    case 1: set_rate(m, QAM_QPSK, 2, CR_12); break;
    case 2: set_rate(m, QAM_QPSK, 2, CR_34); break;
    case 3: set_rate(m, QAM_16, 4, CR_12); break;
    case 4: set_rate(m, QAM_16, 4, CR_34); break;
    case 5: set_rate(m, QAM_64, 6, CR_23); break;
    case 6: set_rate(m, QAM_64, 6, CR_34); break;
    default: set_rate(m, QAM_64, 6, CR_56); break;
    }
    m->len = len;
    m->nSS = mcs / 8 + 1;
    m->nSD = 52;
    m->nCBPSS = m->nBPSCS * 52;
    m->nCBPS = m->nCBPSS * m->nSS;
    set_ndbps(m);
    m->nLTF = ltf_count(m->nSS);
    m->nSym = (len * 8 + 22) / m->nDBPS + (((len * 8 + 22) % m->nDBPS) != 0);
}

/* lib/cloud80211phy.cc:1225-1323.  Returns 0 for MCS 10..15, where the reference leaves the
 * descriptor holding values from an earlier packet (no defined result to reproduce). */
static int parse_vht_mod(int mcs, orc_mod_t *m)
{
    m->mcs = mcs;
    switch (mcs) {
    case 0: set_rate(m, QAM_BPSK, 1, CR_12); break;
    case 1: set_rate(m, QAM_QPSK, 2, CR_12); break;
    case 2: set_rate(m, QAM_QPSK, 2, CR_34); break;
    case 3: set_rate(m, QAM_16, 4, CR_12); break;
    case 4: set_rate(m, QAM_16, 4, CR_34); break;
    case 5: set_rate(m, QAM_64, 6, CR_23); break;
    case 6: set_rate(m, QAM_64, 6, CR_34); break;
    case 7: set_rate(m, QAM_64, 6, CR_56); break;
    case 8: set_rate(m, QAM_256, 8, CR_34); break;
    case 9: set_rate(m, QAM_256, 8, CR_56); break;
    default: return 0;
    }
    m->nSD = 52;
    m->nCBPSS = m->nBPSCS * 52;
    m->nCBPS = m->nCBPSS * m->nSS;
    set_ndbps(m);
    m->nLTF = ltf_count(m->nSS);
    return 1;
}

/* lib/cloud80211phy.cc:1090-1178.  Returns 1 for a single-user frame, 2 for an MU frame (group id 1..62: the
 * modulation is only known after VHT-SIG-B), 0 when the MCS is invalid (the reference then keeps fields of an
 * earlier packet). */
static int parse_vhta(const uint8_t *b, orc_mod_t *m)
{
    int gid = 0, nsts = 0, mcs = 0;
    for (int i = 0; i < 6; i++) gid |= b[i + 4] << i;
    m->format = ORC_F_VHT;
    m->nSymSamp = b[24] ? 72 : 80;
    m->ampdu = 1;
    if (!(gid == 0 || gid == 63)) {
        m->nLTF = 2;
        m->nSS = 1;
        m->nSD = 52;
        return 2;
    }
    for (int i = 0; i < 3; i++) nsts |= b[i + 10] << i;
    for (int i = 0; i < 4; i++) mcs |= b[i + 28] << i;
    m->nSS = nsts + 1;
    return parse_vht_mod(mcs, m);
}

/* lib/cloud80211phy.cc:1184-1192, MU branch of VHT-SIG-B: 16 bits of length, 4 bits of MCS. */
static int parse_vhtb_mu(const uint8_t *b, orc_mod_t *m)
{
    int l = 0, mcs = 0;
    for (int i = 0; i < 16; i++) l |= b[i] << i;
    for (int i = 0; i < 4; i++) mcs |= b[i + 16] << i;
    if (!parse_vht_mod(mcs, m)) return 0;
    m->len = l * 4;
    m->nSym = (m->len * 8 + 16 + 6) / m->nDBPS + (((m->len * 8 + 16 + 6) % m->nDBPS) != 0);
    m->nLTF = 2;
    return 1;
}

/* lib/cloud80211phy.cc:1180-1223, SU branch. */
static void parse_vhtb(const uint8_t *b, orc_mod_t *m)
{
    if ((b[17] + b[18] + b[19]) == 3) {
        int l = 0;
        for (int i = 0; i < 17; i++) l |= b[i] << i;
        m->len = l * 4;
        m->nSym = (m->len * 8 + 16 + 6) / m->nDBPS + (((m->len * 8 + 16 + 6) % m->nDBPS) != 0);
    } else {
        uint32_t pat = 0;
        for (int i = 0; i < 20; i++) pat |= (uint32_t)b[i] << i;
        if (pat == 0x422E0u) { m->len = 0; m->nSym = 0; }  /* NDP pattern 0b01000010001011100000 */
        else { m->len = -1; m->nSym = -1; }
    }
}

/* ------------------------------------------------------------------ signal */
/* lib/signal_impl.cc:108-162 + lib/cloud80211phy.cc:609-627 */
int orc_signal(const float *in_ri, int navail, float rad, orc_frame_t *f)
{
    orc_init();
    const cf *in = (const cf *)in_ri;
    memset(f, 0, sizeof(*f));
    if (navail < 224) { f->status = ORC_DROP_LSIG; return f->status; }
    cf w1[64], w2[64], ws[64], s1[64], s2[64], sg[64];
    for (int i = 0; i < 64; i++) {
        w1[i] = in[SHIFT + i] * CMPLXF(cosf((i + SHIFT) * rad), sinf((i + SHIFT) * rad));
        w2[i] = in[SHIFT + 64 + i] * CMPLXF(cosf((i + SHIFT + 64) * rad), sinf((i + SHIFT + 64) * rad));
        ws[i] = in[SHIFT + 144 + i] * CMPLXF(cosf((i + SHIFT + 144) * rad), sinf((i + SHIFT + 144) * rad));
    }
    fft64(w1, s1); fft64(w2, s2); fft64(ws, sg);

    cf h[64];
    memset(h, 0, sizeof(h));
    float llr[48];
    const int pb[4] = {7, 21, 43, 57};
    for (int p = 0; p < 4; p++) h[pb[p]] = (s1[pb[p]] + s2[pb[p]]) / (T_LTF_L[pb[p]] * 2.0f);
    cf ps = conjf(sg[7] / h[7] - sg[21] / h[21] + sg[43] / h[43] + sg[57] / h[57]);
    float pa = cabsf(ps);
    for (int i = 0; i < 64; i++)
        if (T_LSIG_DEMAP[i] > -1) {
            h[i] = (s1[i] + s2[i]) / (T_LTF_L[i] * 2.0f);
            llr[T_LSIG_DEMAP[i]] = crealf(sg[i] / h[i] * ps / pa);
        }
    memcpy(f->h_l, h, sizeof(h));
    orc_sig_viterbi(llr, f->lsig_bits, 24);
    int ndbps = 24;
    if (!check_legacy(f->lsig_bits, &f->lsig_mcs, &f->lsig_len, &ndbps)) {
        f->status = ORC_DROP_LSIG;
        return f->status;
    }
    int nsym = (f->lsig_len * 8 + 22 + ndbps - 1) / ndbps;
    f->nsamp = nsym * 80;
    f->status = ORC_OK;
    return ORC_OK;
}

/* ------------------------------------------------------------------ demod */
typedef struct {
    const cf *in0, *in1;
    int nvalid;  /* samples after L-SIG that signal copies downstream (nsamp, capped by capture) */
    float rad;
} src_t;

/* Sample k of the stream demod sees: signal's S_COPY derotation (lib/signal_impl.cc:172-173).
 * Past nsamp the reference leaves the 320-item pad uninitialised (signal_impl.cc:198); a
 * valid frame never reads it, zeros stand in. */
static cf samp(const src_t *s, int ant, int k)
{
    if (k < 0 || k >= s->nvalid) return 0.0f;
    const cf *p = ant ? s->in1 : s->in0;
    float ph = (float)(k + 224) * s->rad;
    return p[224 + k] * CMPLXF(cosf(ph), sinf(ph));
}

static void fft_at(const src_t *s, int ant, int off, cf *out)
{
    cf w[64];
    for (int i = 0; i < 64; i++) w[i] = samp(s, ant, off + SHIFT + i);
    fft64(w, out);
}

static int nl_used(int i) { return !(i == 0 || (i >= 29 && i <= 35)); }
static int l_used(int i) { return !(i == 0 || (i >= 27 && i <= 37)); }

/* lib/cloud80211phy.cc:629-648 */
static void nlsig_demod(const cf *sym1, const cf *sym2, const cf *h, float *llrht, float *llrvht)
{
    cf p1 = conjf(sym1[7] / h[7] - sym1[21] / h[21] + sym1[43] / h[43] + sym1[57] / h[57]);
    cf p2 = conjf(sym2[7] / h[7] - sym2[21] / h[21] + sym2[43] / h[43] + sym2[57] / h[57]);
    float a1 = cabsf(p1), a2 = cabsf(p2);
    for (int i = 0; i < 64; i++)
        if (T_LSIG_DEMAP[i] > -1) {
            cf m1 = sym1[i] / h[i] * p1 / a1;
            cf m2 = sym2[i] / h[i] * p2 / a2;
            llrht[T_LSIG_DEMAP[i]] = cimagf(m1);
            llrht[T_LSIG_DEMAP[i] + 48] = cimagf(m2);
            llrvht[T_LSIG_DEMAP[i]] = crealf(m1);
            llrvht[T_LSIG_DEMAP[i] + 48] = cimagf(m2);
        }
}

/* lib/cloud80211phy.cc:2090-2148 */
static void qam_to_llr(cf *q, float *o, const orc_mod_t *m)
{
    int n = m->nSD;
    if (m->mod == QAM_BPSK) {
        for (int i = 0; i < n; i++) o[i] = crealf(q[i]);
    } else if (m->mod == QAM_QPSK) {
        for (int i = 0; i < n; i++) {
            q[i] *= 1.4142135623730951f;
            o[i * 2] = crealf(q[i]);
            o[i * 2 + 1] = cimagf(q[i]);
        }
    } else if (m->mod == QAM_16) {
        for (int i = 0; i < n; i++) {
            q[i] *= 3.1622776601683795f;
            o[i * 4] = crealf(q[i]);
            o[i * 4 + 1] = 2.0f - fabsf(crealf(q[i]));
            o[i * 4 + 2] = cimagf(q[i]);
            o[i * 4 + 3] = 2.0f - fabsf(cimagf(q[i]));
        }
    } else if (m->mod == QAM_64) {
        for (int i = 0; i < n; i++) {
            q[i] *= 6.48074069840786f;
            o[i * 6] = crealf(q[i]);
            o[i * 6 + 1] = 4.0f - fabsf(crealf(q[i]));
            o[i * 6 + 2] = 2 - fabsf(4.0f - fabsf(crealf(q[i])));
            o[i * 6 + 3] = cimagf(q[i]);
            o[i * 6 + 4] = 4.0f - fabsf(cimagf(q[i]));
            o[i * 6 + 5] = 2 - fabsf(4.0f - fabsf(cimagf(q[i])));
        }
    } else if (m->mod == QAM_256) {
        for (int i = 0; i < n; i++) {
            q[i] *= 13.038404810405298f;
            o[i * 8] = crealf(q[i]);
            o[i * 8 + 1] = 8.0f - fabsf(crealf(q[i]));
            o[i * 8 + 2] = 4 - fabsf(8.0f - fabsf(crealf(q[i])));
            o[i * 8 + 3] = 2 - fabsf(4 - fabsf(8.0f - fabsf(crealf(q[i]))));
            o[i * 8 + 4] = cimagf(q[i]);
            o[i * 8 + 5] = 8.0f - fabsf(cimagf(q[i]));
            o[i * 8 + 6] = 4 - fabsf(8.0f - fabsf(cimagf(q[i])));
            o[i * 8 + 7] = 2 - fabsf(4 - fabsf(8.0f - fabsf(cimagf(q[i]))));
        }
    }
}

/* lib/cloud80211phy.cc:2150-2192 */
static void deint_l(const float *in, float *out, const orc_mod_t *m)
{
    int idx = m->nCBPS == 48 ? 0 : m->nCBPS == 96 ? 1 : m->nCBPS == 192 ? 2 : 3;
    for (int i = 0; i < m->nCBPS; i++) out[T_DEINT_L[idx][i]] = in[i];
}
/* lib/cloud80211phy.cc:2238-2338 */
static void deint_nl(const float *in, float *out, const orc_mod_t *m, int ss2)
{
    int idx = nl_idx(m->nBPSCS);
    const int *map = ss2 ? T_DEINT_NL2[idx] : T_DEINT_NL[idx];
    for (int i = 0; i < m->nCBPSS; i++) out[map[i]] = in[i];
}

typedef struct {
    cf hl[64];
    cf h[64][4];     /* d_H_NL (demod: [i][0] only) */
    cf hinv[64][4];  /* d_H_NL_INV */
    cf pl1[4], pl2[4]; /* d_pilotNlLtf, d_pilotNlLtf2 */
    float pilot[4], pilot2[4];
    int pilotP;
} chan_t;

static void pilot_shift(float *p)
{
    float t = p[0]; p[0] = p[1]; p[1] = p[2]; p[2] = p[3]; p[3] = t;
}

/* Pilot-phase reference of the single-stream paths (lib/demod_impl.cc:426-433, 520-526). */
static cf pilot_sum1(const cf *sig, const chan_t *c)
{
    float pp = T_PILOT_P[c->pilotP];
    return conjf(sig[7] * c->pilot[2] * pp + sig[21] * c->pilot[3] * pp + sig[43] * c->pilot[0] * pp +
                 sig[57] * c->pilot[1] * pp);
}

/* lib/demod_impl.cc:507-539 */
static void legacy_update(const src_t *s, int off, chan_t *c, cf *qam)
{
    cf F[64], sig[64];
    fft_at(s, 0, off, F);
    for (int i = 0; i < 64; i++)
        if (l_used(i)) sig[i] = F[i] / c->hl[i];
    cf ps = pilot_sum1(sig, c);
    c->pilotP = (c->pilotP + 1) % 127;
    float pa = cabsf(ps);
    int j = 24;
    for (int i = 0; i < 64; i++)
        if (l_used(i) && !is_pilot_bin(i)) {
            qam[j] = sig[i] * ps / pa;
            j++;
            if (j >= 48) j = 0;
        }
}

/* lib/demod_impl.cc:413-447; lib/demod2_impl.cc:476-502 (nSS == 1) */
static void nl1_update(const src_t *s, int off, chan_t *c, cf *qam)
{
    cf F[64], sig[64];
    fft_at(s, 0, off, F);
    for (int i = 0; i < 64; i++)
        if (nl_used(i)) sig[i] = F[i] / c->h[i][0];
    cf ps = pilot_sum1(sig, c);
    pilot_shift(c->pilot);
    c->pilotP = (c->pilotP + 1) % 127;
    float pa = cabsf(ps);
    int j = 26;
    for (int i = 0; i < 64; i++)
        if (nl_used(i) && !is_pilot_bin(i)) {
            qam[j] = sig[i] * ps / pa;
            j++;
            if (j >= 52) j = 0;
        }
}

/* Two-antenna zero forcing x = (H^H H)^-1 H^H y (lib/demod2_impl.cc:509-520). */
static void zf2(const cf *F1, const cf *F2, const chan_t *c, cf *s1, cf *s2)
{
    for (int i = 0; i < 64; i++)
        if (nl_used(i)) {
            cf t1 = F1[i] * conjf(c->h[i][0]) + F2[i] * conjf(c->h[i][1]);
            cf t2 = F1[i] * conjf(c->h[i][2]) + F2[i] * conjf(c->h[i][3]);
            s1[i] = t1 * c->hinv[i][0] + t2 * c->hinv[i][2];
            s2[i] = t1 * c->hinv[i][1] + t2 * c->hinv[i][3];
        }
}

/* lib/demod2_impl.cc:504-550 (HT) and 586-628 (VHT), nSS == 2 */
static void nl2_update(const src_t *s, int off, chan_t *c, int vht, cf *q0, cf *q1)
{
    cf F1[64], F2[64], s1[64], s2[64];
    fft_at(s, 0, off, F1);
    fft_at(s, 1, off, F2);
    zf2(F1, F2, c, s1, s2);
    float pp = T_PILOT_P[c->pilotP];
    const float *p2 = vht ? c->pilot : c->pilot2;
    cf ps = conjf(s1[7] * c->pilot[2] * pp * c->pl1[2] + s1[21] * c->pilot[3] * pp * c->pl1[3] +
                  s1[43] * c->pilot[0] * pp * c->pl1[0] + s1[57] * c->pilot[1] * pp * c->pl1[1] +
                  s2[7] * p2[2] * pp * c->pl2[2] + s2[21] * p2[3] * pp * c->pl2[3] +
                  s2[43] * p2[0] * pp * c->pl2[0] + s2[57] * p2[1] * pp * c->pl2[1]);
    pilot_shift(c->pilot);
    if (!vht) pilot_shift(c->pilot2);
    c->pilotP = (c->pilotP + 1) % 127;
    float pa = cabsf(ps);
    int j = 26;
    for (int i = 0; i < 64; i++)
        if (nl_used(i) && !is_pilot_bin(i)) {
            q0[j] = s1[i] * ps / pa;
            q1[j] = s2[i] * ps / pa;
            j++;
            if (j >= 52) j = 0;
        }
}

/* lib/demod_impl.cc:347-379: this station's stream out of the two VHT-LTFs of a two-user MU frame.  The second
 * LTF's sequence is LTF_NL_28_F_FLOAT_VHT22 = the HT/VHT-LTF with its pilot tones negated. */
static void mu_chan_est(const src_t *s, int off, int mu_pos, chan_t *c)
{
    cf F1[64], F2[64];
    fft_at(s, 0, off, F1);
    fft_at(s, 0, off + 80, F2);
    for (int i = 0; i < 64; i++)
        if (nl_used(i)) {
            if (mu_pos == 0) {
                c->h[i][0] = (F1[i] - F2[i]) / (T_LTF_NL[i] * 2.0f);
            } else {
                float l22 = is_pilot_bin(i) ? -T_LTF_NL[i] : T_LTF_NL[i];
                c->h[i][0] = (F1[i] / T_LTF_NL[i] + F2[i] / l22) / 2.0f;
            }
        }
}

/* lib/demod_impl.cc:380-392 (one LTF), lib/demod2_impl.cc:372-463 (two streams) */
static void nl_chan_est(const src_t *s, int off, const orc_mod_t *m, int two_ant, chan_t *c)
{
    cf F1[64], F2[64], F12[64], F22[64];
    if (m->nSS == 1 || !two_ant) {
        /* one stream, or channel sounding seen by one antenna: the first LTF only (demod_impl.cc:380-409) */
        fft_at(s, 0, off, F1);
        for (int i = 0; i < 64; i++)
            if (nl_used(i)) c->h[i][0] = F1[i] / T_LTF_NL[i];
        return;
    }
    fft_at(s, 0, off, F1);
    fft_at(s, 1, off, F2);
    fft_at(s, 0, off + 80, F12);
    fft_at(s, 1, off + 80, F22);
    for (int i = 0; i < 64; i++)
        if (nl_used(i)) {
            float l2 = T_LTF_NL[i] * 0.5f;
            c->h[i][0] = (F1[i] - F12[i]) * l2;
            c->h[i][1] = (F2[i] - F22[i]) * l2;
            c->h[i][2] = (F1[i] + F12[i]) * l2;
            c->h[i][3] = (F2[i] + F22[i]) * l2;
        }
    if (m->format == ORC_F_VHT) {
        const int pb[4] = {7, 21, 43, 57};
        for (int p = 0; p < 4; p++)
            for (int q = 0; q < 4; q++) c->h[pb[p]][q] = (c->h[pb[p] - 1][q] + c->h[pb[p] + 1][q]) / 2.0f;
    }
    for (int i = 0; i < 64; i++)
        if (nl_used(i)) {
            cf a = c->h[i][0] * conjf(c->h[i][0]) + c->h[i][1] * conjf(c->h[i][1]);
            cf b = c->h[i][0] * conjf(c->h[i][2]) + c->h[i][1] * conjf(c->h[i][3]);
            cf cc = c->h[i][2] * conjf(c->h[i][0]) + c->h[i][3] * conjf(c->h[i][1]);
            cf d = c->h[i][2] * conjf(c->h[i][2]) + c->h[i][3] * conjf(c->h[i][3]);
            cf adbc = CMPLXF(1.0f, 0.0f) / (a * d - b * cc);
            c->hinv[i][0] = adbc * d;
            c->hinv[i][1] = -adbc * b;
            c->hinv[i][2] = -adbc * cc;
            c->hinv[i][3] = adbc * a;
        }
    const int pb[4] = {43, 57, 7, 21}; /* pilot slots 0..3 */
    for (int p = 0; p < 4; p++) {
        int i = pb[p];
        cf t1 = F1[i] * conjf(c->h[i][0]) + F2[i] * conjf(c->h[i][1]);
        cf t2 = F1[i] * conjf(c->h[i][2]) + F2[i] * conjf(c->h[i][3]);
        cf e1 = t1 * c->hinv[i][0] + t2 * c->hinv[i][2];
        cf e2 = t1 * c->hinv[i][1] + t2 * c->hinv[i][3];
        if (p == 1) { c->pl1[p] = conjf(-e1); c->pl2[p] = conjf(-e2); }
        else { c->pl1[p] = conjf(e1); c->pl2[p] = conjf(e2); }
    }
}

/* lib/demod_impl.cc:449-505; lib/demod2_impl.cc:632-758 */
static void vht_sigb(const src_t *s, int off, const orc_mod_t *m, int two_ant, chan_t *c, orc_frame_t *f)
{
    cf F1[64], F2[64], s1[64], s2[64], q0[52], q1[52];
    float inted[52], coded[52];
    int two = (m->nSS == 2 && two_ant);
    if (!two) {
        fft_at(s, 0, off, F1);
        for (int i = 0; i < 64; i++)
            if (nl_used(i)) s1[i] = F1[i] / c->h[i][0];
        cf ps = conjf(s1[7] - s1[21] + s1[43] + s1[57]);
        float pa = cabsf(ps);
        int j = 26;
        for (int i = 0; i < 64; i++)
            if (nl_used(i) && !is_pilot_bin(i)) {
                q0[j] = s1[i] * ps / pa;
                inted[j] = crealf(q0[j]);
                j++;
                if (j >= 52) j = 0;
            }
    } else {
        fft_at(s, 0, off, F1);
        fft_at(s, 1, off, F2);
        zf2(F1, F2, c, s1, s2);
        cf ps = conjf(s1[7] * c->pl1[2] - s1[21] * c->pl1[3] + s1[43] * c->pl1[0] + s1[57] * c->pl1[1] +
                      s2[7] * c->pl2[2] - s2[21] * c->pl2[3] + s2[43] * c->pl2[0] + s2[57] * c->pl2[1]);
        float pa = cabsf(ps);
        int j = 26;
        for (int i = 0; i < 64; i++)
            if (nl_used(i) && !is_pilot_bin(i)) {
                q0[j] = s1[i] * ps / pa;
                q1[j] = s2[i] * ps / pa;
                inted[j] = (crealf(q0[j]) + crealf(q1[j])) / 2.0f;
                j++;
                if (j >= 52) j = 0;
            }
    }
    for (int i = 0; i < 52; i++) coded[T_DEINT_NL[0][i]] = inted[i];
    orc_sig_viterbi(coded, f->sigb_bits, 26);

    uint8_t enc[52], intl[52];
    orc_bcc_encode(f->sigb_bits, enc, 26);
    for (int j = 0; j < 52; j++) intl[j] = enc[T_DEINT_NL[0][j]];
    double n0 = 0.0, n1 = 0.0;
    for (int i = 0; i < 52; i++) {
        cf ref = intl[i] ? 1.0f : -1.0f;
        q0[i] -= ref;
        n0 += (double)(crealf(q0[i]) * crealf(q0[i]) + cimagf(q0[i]) * cimagf(q0[i]));
        if (two) {
            q1[i] -= ref;
            n1 += (double)(crealf(q1[i]) * crealf(q1[i]) + cimagf(q1[i]) * cimagf(q1[i]));
        }
    }
    f->sssnr0 = (float)(log10(52.0 / n0) * 10.0);
    if (two) f->sssnr1 = (float)(log10(52.0 / n1) * 10.0);
}

/* Test hook: the equalised constellation points of every data symbol as the demodulator hands them to
 * procSymQamToLlr (lib/demod_impl.cc:293-299, lib/demod2_impl.cc:283-296), i.e. before the constellation scaling:
 * [symbol][stream][nSD] complex, re/im interleaved, in the tone order of the reference's arrays. */
static __thread float *g_pts = NULL;
static __thread int g_pts_cap = 0, g_pts_n = 0;
void orc_capture_points(float *buf_ri, int cap_complex)
{
    g_pts = buf_ri;
    g_pts_cap = buf_ri ? cap_complex : 0;
    g_pts_n = 0;
}
int orc_captured_points(void) { return g_pts_n; }
static void keep_points(const cf *q, int n)
{
    if (!g_pts || g_pts_n + n > g_pts_cap) return;
    memcpy(g_pts + 2 * g_pts_n, q, sizeof(cf) * (size_t)n);
    g_pts_n += n;
}

int orc_demod(const float *in0_ri, const float *in1_ri, int navail, float rad, orc_frame_t *f,
              float *llr, int llr_cap)
{
    orc_init();
    if (f->status != ORC_OK) return f->status;
    int two_ant = in1_ri != NULL;
    src_t s;
    s.in0 = (const cf *)in0_ri;
    s.in1 = (const cf *)in1_ri;
    s.rad = rad;
    s.nvalid = f->nsamp;
    if (s.nvalid > navail - 224) s.nvalid = navail - 224;

    chan_t *c = (chan_t *)calloc(1, sizeof(chan_t));
    memcpy(c->hl, f->h_l, sizeof(c->hl));
    orc_mod_t *m = &f->m;
    memset(m, 0, sizeof(*m));
    int pos = 0;     /* d_nSampConsumed */
    int fmt = ORC_F_L;

    if (f->lsig_mcs == 0) {
        /* DEMOD_S_FORMAT, lib/demod_impl.cc:106-148 */
        cf F1[64], F2[64];
        float llrht[96], llrvht[96];
        fft_at(&s, 0, 0, F1);
        fft_at(&s, 0, 80, F2);
        nlsig_demod(F1, F2, c->hl, llrht, llrvht);
        orc_sig_viterbi(llrvht, f->sig_bits, 48);
        if (check_vhta(f->sig_bits)) {
            fmt = ORC_F_VHT;
            int rc = parse_vhta(f->sig_bits, m);
            if (!rc) { f->status = ORC_DROP_UNSUPPORTED; goto done; }
            f->sumu = rc == 2;
            pos = 160;
        } else {
            orc_sig_viterbi(llrht, f->sig_bits, 48);
            if (check_ht(f->sig_bits)) {
                fmt = ORC_F_HT;
                parse_ht(f->sig_bits, m);
                pos = 160;
            }
        }
    }

    int nleg = (f->lsig_len * 8 + 22 + 23) / 24;
    if (fmt == ORC_F_VHT) {
        /* DEMOD_S_VHT, lib/demod_impl.cc:150-178 / lib/demod2_impl.cc:150-177 */
        if (m->nSS > 2) { f->status = ORC_DROP_UNSUPPORTED; goto done; }
        /* two streams on one antenna: followed up to VHT-SIG-B to see whether it is a sounding NDP
         * (demod_impl.cc:393-409,452-455); two streams of data on one antenna have no defined result */
        int sounding = (m->nSS == 2 && !two_ant);
        if (f->sumu) {
            /* MU frames are the one-antenna block's business (demod_impl.cc:347-379); demod2 has no such branch */
            if (two_ant) { f->status = ORC_DROP_UNSUPPORTED; goto done; }
            mu_chan_est(&s, pos + 80, f->mu_pos, c);
        } else {
            nl_chan_est(&s, pos + 80, m, two_ant, c);
        }
        vht_sigb(&s, pos + 80 + m->nLTF * 80, m, two_ant, c, f);
        if (f->sumu) {
            if (!parse_vhtb_mu(f->sigb_bits, m)) { f->status = ORC_DROP_UNSUPPORTED; goto done; }
        } else {
            parse_vhtb(f->sigb_bits, m);
        }
        if (m->len == 0 && m->nSym == 0) {
            if (two_ant) { f->status = ORC_DROP_LEN_CHECK; goto done; } /* demod2 requires len > 0 (demod2_impl.cc:159) */
            /* NDP on one antenna: raw samples of the two LTF windows (demod_impl.cc:397-398,238-249) */
            for (int i = 0; i < 64; i++) {
                cf a = samp(&s, 0, pos + 80 + SHIFT + i), b = samp(&s, 0, pos + 80 + 80 + SHIFT + i);
                f->mu2x1chan[2 * i] = crealf(a);
                f->mu2x1chan[2 * i + 1] = cimagf(a);
                f->mu2x1chan[128 + 2 * i] = crealf(b);
                f->mu2x1chan[128 + 2 * i + 1] = cimagf(b);
            }
            f->status = ORC_NDP;
            goto done;
        }
        if (sounding) { f->status = ORC_DROP_UNSUPPORTED; goto done; }
        if (!(m->len > 0 && m->len <= 4095 && m->nSS <= 2 &&
              (nleg * 80) >= (m->nSym * m->nSymSamp + 160 + 80 + m->nLTF * 80 + 80))) {
            f->status = ORC_DROP_LEN_CHECK;
            goto done;
        }
        f->trellis = m->nSym * m->nDBPS;
        c->pilot[0] = 1.0f; c->pilot[1] = 1.0f; c->pilot[2] = 1.0f; c->pilot[3] = -1.0f;
        c->pilotP = 4;
        pos += 80 + m->nLTF * 80 + 80;
    } else if (fmt == ORC_F_HT) {
        /* DEMOD_S_HT, lib/demod_impl.cc:180-205 / lib/demod2_impl.cc:179-211 */
        if (m->nSS > 2 || (m->nSS == 2 && !two_ant)) { f->status = ORC_DROP_UNSUPPORTED; goto done; }
        nl_chan_est(&s, pos + 80, m, two_ant, c);
        if (!(m->len > 0 && m->len <= 4095 && m->nSS <= 2 &&
              (nleg * 80) >= (m->nSym * m->nSymSamp + 160 + 80 + m->nLTF * 80))) {
            f->status = ORC_DROP_LEN_CHECK;
            goto done;
        }
        f->trellis = m->len * 8 + 22;
        if (m->nSS == 1) { c->pilot[0] = 1.0f; c->pilot[1] = 1.0f; c->pilot[2] = 1.0f; c->pilot[3] = -1.0f; }
        else { c->pilot[0] = 1.0f; c->pilot[1] = 1.0f; c->pilot[2] = -1.0f; c->pilot[3] = -1.0f; }
        c->pilot2[0] = 1.0f; c->pilot2[1] = -1.0f; c->pilot2[2] = -1.0f; c->pilot2[3] = 1.0f;
        c->pilotP = 3;
        pos += 80 + m->nLTF * 80;
    } else {
        /* DEMOD_S_LEGACY, lib/demod_impl.cc:207-219 */
        parse_legacy(f->lsig_mcs, f->lsig_len, m);
        f->trellis = m->len * 8 + 22;
        c->pilot[0] = 1.0f; c->pilot[1] = 1.0f; c->pilot[2] = 1.0f; c->pilot[3] = -1.0f;
        c->pilotP = 1;
    }
    f->total = m->nSym * m->nCBPS;
    if (f->total > llr_cap) { f->status = ORC_DROP_DECODE_LIMIT; goto done; }

    /* DEMOD_S_DEMOD, lib/demod_impl.cc:279-314 / lib/demod2_impl.cc:268-321 */
    {
        cf q0[52], q1[52];
        float li0[416], li1[416], sp0[416], sp1[416];
        for (int n = 0; n < m->nSym; n++) {
            float *o = &llr[n * m->nCBPS];
            int off = pos + n * m->nSymSamp;
            if (m->format == ORC_F_L) {
                legacy_update(&s, off, c, q0);
                keep_points(q0, m->nSD);
                qam_to_llr(q0, li0, m);
                deint_l(li0, o, m);
            } else if (m->nSS == 1) {
                nl1_update(&s, off, c, q0);
                keep_points(q0, m->nSD);
                qam_to_llr(q0, li0, m);
                deint_nl(li0, o, m, 0);
            } else {
                nl2_update(&s, off, c, m->format == ORC_F_VHT, q0, q1);
                keep_points(q0, m->nSD);
                keep_points(q1, m->nSD);
                qam_to_llr(q0, li0, m);
                qam_to_llr(q1, li1, m);
                deint_nl(li0, sp0, m, 0);
                deint_nl(li1, sp1, m, 1);
                /* lib/cloud80211phy.cc:2442-2451 */
                int sb = m->nBPSCS / 2 > 1 ? m->nBPSCS / 2 : 1;
                for (int i = 0; i < m->nCBPSS / sb; i++) {
                    memcpy(&o[i * 2 * sb], &sp0[i * sb], sizeof(float) * sb);
                    memcpy(&o[(i * 2 + 1) * sb], &sp1[i * sb], sizeof(float) * sb);
                }
            }
        }
    }
    /* Convention of the batch boundary, not of the reference: a data symbol whose samples are all zero (the capture
     * ends inside the frame; the reference's `signal` would still be waiting for those samples, lib/signal_impl.cc:
     * 164-192) has a pilot sum of zero and the arithmetic above makes its soft bits 0 * 0 / 0; a tone with a zero
     * channel estimate gives infinities.  Such soft bits are handed on as 0.0, i.e. as erasures, here and in the
     * GPU demodulator alike. */
    for (int i = 0; i < f->total; i++)
        if (!isfinite(llr[i])) llr[i] = 0.0f;
done:
    free(c);
    return f->status;
}

/* ------------------------------------------------------------------ decode */
typedef struct {
    int his[64][ORC_MAX_TRELLIS + 1]; /* v_state_his, lib/decode_impl.h:73 */
    int seq[ORC_MAX_TRELLIS + 1];
    uint8_t scram[ORC_MAX_TRELLIS + 8];
    uint8_t bits[ORC_MAX_TRELLIS + 8];
} dec_ws_t;

static __thread dec_ws_t *g_ws = NULL;

/* lib/decode_impl.cc:164-302: vstb_init + vstb_update + vstb_end. */
static int viterbi_full(dec_ws_t *w, const float *llr, int len, int cr, int T)
{
    float m0[64], m1[64], *pre = m0, *cur = m1;
    for (int i = 0; i < 64; i++) {
        for (int j = 0; j <= T; j++) w->his[i][j] = 0;
        m0[i] = -1000000000000000.0f;
        m1[i] = -1000000000000000.0f;
    }
    m0[0] = 0;
    const int *punc = cr == CR_12 ? PUNC_12 : cr == CR_23 ? PUNC_23 : cr == CR_34 ? PUNC_34 : PUNC_56;
    int plen = cr == CR_12 ? 2 : cr == CR_23 ? 4 : cr == CR_34 ? 6 : 10;
    int p = 0, used = 0, t = 0;
    if (T > 0)
        while ((used + punc[p] + punc[p + 1]) <= len) {
            float t0, t1;
            if (punc[p]) t0 = llr[used++]; else t0 = 0.0f;
            if (punc[p + 1]) t1 = llr[used++]; else t1 = 0.0f;
            float tab[4] = {0.0f, t1, t0, t1 + t0};
            for (int i = 0; i < 64; i++) {
                float a0 = pre[i] + tab[T_OUT[i][0]];
                float a1 = pre[i] + tab[T_OUT[i][1]];
                int n0 = T_NEXT[i][0], n1 = T_NEXT[i][1];
                if (a0 > cur[n0]) { cur[n0] = a0; w->his[n0][t + 1] = i; }
                if (a1 > cur[n1]) { cur[n1] = a1; w->his[n1][t + 1] = i; }
            }
            float *tmp = pre; pre = cur; cur = tmp;
            for (int i = 0; i < 64; i++) cur[i] = -1000000000000000.0f;
            p += 2;
            if (p >= plen) p = 0;
            t++;
            if (t >= T) break;
        }
    if (t < T) return -1; /* the block would wait for more LLRs; a complete frame never gets here */
    w->seq[T] = 0;
    for (int j = T; j > 0; j--) w->seq[j - 1] = w->his[w->seq[j]][j];
    for (int j = 0; j < T; j++) w->scram[j] = (w->seq[j + 1] == T_NEXT[w->seq[j]][1]) ? 1 : 0;
    return used;
}

int orc_viterbi_bits(const float *llr, int nllr, int cr, int trellis, uint8_t *scram_bits)
{
    orc_init();
    if (!g_ws) g_ws = (dec_ws_t *)malloc(sizeof(dec_ws_t));
    if (trellis > ORC_MAX_TRELLIS) return -1;
    int r = viterbi_full(g_ws, llr, nllr, cr, trellis);
    if (r >= 0) memcpy(scram_bits, g_ws->scram, trellis);
    return r;
}

/* lib/decode_impl.cc:304-323 */
static void descramble(dec_ws_t *w, int T)
{
    int st = 0;
    for (int i = 0; i < 7; i++)
        if (w->scram[i]) st |= 1 << (6 - i);
    memset(w->bits, 0, 7);
    for (int i = 7; i < T; i++) {
        int fb = (!!(st & 64)) ^ (!!(st & 8));
        w->bits[i] = fb ^ (w->scram[i] & 1);
        st = ((st << 1) & 0x7e) | fb;
    }
}

static int emit_pdu(const uint8_t *bitp, int format, int len, int mcs, uint8_t *out, int cap, int *pos)
{
    uint8_t pkt[ORC_MAX_PSDU + 8];
    for (int i = 0; i < len; i++) {
        uint8_t v = 0;
        for (int j = 0; j < 8; j++) v |= bitp[i * 8 + j] << j;
        pkt[i] = v;
    }
    if (orc_crc32(pkt, len) != 558161692u) return 0; /* lib/decode_impl.cc:374,453 */
    if (*pos + len + 4 > cap) return 0;
    out[(*pos)++] = (uint8_t)format;
    out[(*pos)++] = (uint8_t)(len % 256);
    out[(*pos)++] = (uint8_t)(len / 256);
    memcpy(&out[*pos], pkt, len);
    *pos += len;
    out[(*pos)++] = (uint8_t)mcs;
    return 1;
}

/* A-MPDU delimiter check, IEEE 802.11-2016 9.7.1: CRC-8 over the first 16 bits, signature 0x4E. */
static int delimiter_ok(const uint8_t *bp)
{
    uint8_t c[8];
    orc_crc8_bits(bp, 16, c);
    for (int i = 0; i < 8; i++)
        if (c[i] != bp[16 + i]) return 0;
    for (int i = 0; i < 8; i++)
        if (bp[24 + i] != ((0x4E >> i) & 1)) return 0;
    return 1;
}

int orc_decode(const float *llr, int nllr, int format, int len, int cr, int mcs, int ampdu, int trellis,
               uint8_t *out, int out_cap, int *out_len, uint8_t *bits_out)
{
    return orc_decode_ex(llr, nllr, format, len, cr, mcs, ampdu, trellis, 0, out, out_cap, out_len, bits_out);
}

/* lib/decode_impl.cc:93-97,164-520.  strict != 0 replaces the reference's A-MPDU walk by the standard's
 * de-aggregation (IEEE 802.11-2016 10.13.3), which the reference does not implement: checked delimiters, each with
 * its own length, resynchronisation on 4-byte boundaries, for VHT and for aggregated HT PSDUs. */
int orc_decode_ex(const float *llr, int nllr, int format, int len, int cr, int mcs, int ampdu, int trellis, int strict,
                  uint8_t *out, int out_cap, int *out_len, uint8_t *bits_out)
{
    orc_init();
    *out_len = 0;
    if (len > ORC_MAX_PSDU || trellis > ORC_MAX_TRELLIS || trellis <= 0) return 0;
    if (!g_ws) g_ws = (dec_ws_t *)malloc(sizeof(dec_ws_t));
    dec_ws_t *w = g_ws;
    if (viterbi_full(w, llr, nllr, cr, trellis) < 0) return 0;
    descramble(w, trellis);
    if (bits_out) memcpy(bits_out, w->bits, trellis);
    int npdu = 0, pos = 0;
    if (strict && (format == ORC_F_VHT || (format == ORC_F_HT && ampdu))) {
        int plen = len < (trellis - 16) / 8 ? len : (trellis - 16) / 8;
        int at = 0;
        while (at + 4 <= plen) {
            const uint8_t *bp = &w->bits[16 + 8 * at];
            at += 4;
            if (!delimiter_ok(bp)) continue;
            int l = 0;
            for (int i = 0; i < 12; i++) l |= ((int)bp[4 + i]) << i;
            if (format == ORC_F_VHT) l |= (((int)bp[2]) << 12) | (((int)bp[3]) << 13);
            if (l == 0 || at + l > plen) continue;
            if (l <= ORC_MAX_PSDU) npdu += emit_pdu(&w->bits[16 + 8 * at], format, l, mcs, out, out_cap, &pos);
            at += (l + 3) & ~3;
        }
    } else if (format == ORC_F_VHT) {
        /* A-MPDU walk, lib/decode_impl.cc:329-428.  tmpLen is OR-accumulated across subframes
         * exactly as the reference does (declared outside the loop, :336). */
        int procd = 16;
        if (procd < trellis) {
            const uint8_t *bp = &w->bits[16];
            int eof, tl = 0;
            while (1) {
                procd += 32;
                if (procd > trellis) break;
                eof = bp[0];
                tl |= ((int)bp[2]) << 12;
                tl |= ((int)bp[3]) << 13;
                for (int i = 0; i < 12; i++) tl |= ((int)bp[4 + i]) << i;
                int padded = (tl / 4 + ((tl % 4) != 0)) * 4 * 8;
                procd += padded;
                if (procd > trellis) break;
                bp += 32;
                if (tl <= ORC_MAX_PSDU)
                    npdu += emit_pdu(bp, format, tl, mcs, out, out_cap, &pos);
                bp += padded;
                if (eof) break;
            }
        }
    } else if (!ampdu) {
        /* lib/decode_impl.cc:436-517 (HT A-MPDU is unimplemented in the reference, :432-435) */
        npdu += emit_pdu(&w->bits[16], format, len, mcs, out, out_cap, &pos);
    }
    *out_len = pos;
    return npdu;
}

int orc_rx_frame(const float *in0_ri, const float *in1_ri, int navail, float rad, orc_frame_t *f,
                 float *llr, int llr_cap, uint8_t *out, int out_cap, int *out_len, int *n_pdu)
{
    return orc_rx_frame_mu(in0_ri, in1_ri, navail, rad, 0, f, llr, llr_cap, out, out_cap, out_len, n_pdu);
}

int orc_rx_frame_mu(const float *in0_ri, const float *in1_ri, int navail, float rad, int mu_pos, orc_frame_t *f,
                    float *llr, int llr_cap, uint8_t *out, int out_cap, int *out_len, int *n_pdu)
{
    return orc_rx_frame_ex(in0_ri, in1_ri, navail, rad, mu_pos, 0, f, llr, llr_cap, out, out_cap, out_len, n_pdu);
}

int orc_rx_frame_ex(const float *in0_ri, const float *in1_ri, int navail, float rad, int mu_pos, int strict, orc_frame_t *f,
                    float *llr, int llr_cap, uint8_t *out, int out_cap, int *out_len, int *n_pdu)
{
    *out_len = 0;
    *n_pdu = 0;
    if (orc_signal(in0_ri, navail, rad, f) != ORC_OK) return f->status;
    f->mu_pos = mu_pos;
    if (orc_demod(in0_ri, in1_ri, navail, rad, f, llr, llr_cap) != ORC_OK) {
        if (f->status == ORC_NDP && out_cap >= 1027) {
            /* decode republishes the channel as a message of its own (decode_impl.cc:100-121) */
            out[0] = 20; /* C8P_F_VHT_CHAN */
            out[1] = (uint8_t)(1024 % 256);
            out[2] = (uint8_t)(1024 / 256);
            memcpy(out + 3, f->mu2x1chan, 1024);
            *out_len = 1027;
            *n_pdu = 1;
        }
        return f->status;
    }
    if (f->m.len > ORC_MAX_PSDU || f->trellis > ORC_MAX_TRELLIS) {
        f->status = ORC_DROP_DECODE_LIMIT;
        return f->status;
    }
    *n_pdu = orc_decode_ex(llr, f->total, f->m.format, f->m.len, f->m.cr, f->m.mcs, f->m.ampdu, f->trellis, strict,
                           out, out_cap, out_len, NULL);
    return f->status;
}
