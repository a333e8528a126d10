// HIP kernel of the synthetic transmitter for MI355X (gfx950, wave64): PSDU bytes in, baseband frames out.
//
// Produces the waveform of the reference's Python generator (tools/phy80211.py genFromMpdu 80-117 /
// genFromAmpdu 119-181 followed by genFinalSig 1077-1094): legacy preamble, L-SIG, HT-SIG or VHT-SIG-A,
// HT/VHT training, VHT-SIG-B, data field (scrambler, K=7 convolutional code, puncturing, stream parser,
// interleaver, QAM mapping, pilots, 64-point IDFT, guard interval), per-chain cyclic shifts, the tone scaling
// 1/sqrt(N_tones * N_ss) and the reference's symbol-boundary convention (the two samples either side of every
// boundary are halved, tools/phy80211header.py:910-915).
//
//   k_tx   one workgroup (4 waves) per frame.  The scrambled data bits are laid out once as a packed bit
//          string in LDS; after that every coded bit is a closed form of the symbol number: lane = FFT bin,
//          each lane knows (hoisted out of the symbol loop) which trellis step and which generator each of its
//          constellation bits comes from, reads the 7-bit encoder window from the bit string and takes its
//          parity.  Waves stride over OFDM symbols; the IDFT is a radix-2 butterfly network across the lanes.
#include "tx_kernels.h"

namespace gr80211 {

__constant__ Tables c_tx;

int upload_tx_tables(const Tables &t)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(c_tx), &t, sizeof(Tables));
}

namespace {

constexpr int TX_WAVES = 4;
constexpr int TX_BIT_WORDS = 1056;  // scrambled bits of the longest frame (4095-byte PSDU), one zero word in front

// 64-point inverse DFT across the lanes of a wave: lane l enters with X[l] and leaves with
// sum_k X[k] exp(+2 pi i k n / 64) for n = brev6(l) (no 1/64).
__device__ __forceinline__ c32 wave_idft64(c32 x, int lane)
{
    x.y = -x.y;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int half = 32 >> s;
        c32 o = mk(__shfl_xor(x.x, half), __shfl_xor(x.y, half));
        const bool up = (lane & half) != 0;
        c32 lo = up ? o : x;
        c32 hi = up ? x : o;
        c32 sum = cadd(lo, hi), dif = csub(lo, hi);
        if (s < 5) {
            const int k = (lane & (half - 1)) << s;
            dif = cmul(dif, mk(c_tx.tw_re[k], c_tx.tw_im[k]));
        }
        x = up ? dif : sum;
    }
    x.y = -x.y;
    return x;
}

__device__ __forceinline__ int bin_to_tone(int b) { return b < 32 ? b : b - 64; }
__device__ __forceinline__ int pilot_index(int b) { return b == 43 ? 0 : b == 57 ? 1 : b == 7 ? 2 : b == 21 ? 3 : -1; }

// CRC-8 of the SIG fields (IEEE 802.11-2016 19.3.9.4.4): x^8 + x^2 + x + 1, register preset to ones, the ones'
// complement sent; bits given LSB first in v.
__device__ int crc8_bits(unsigned long long v, int n)
{
    int c = 0xff;
    for (int i = 0; i < n; i++) c = ((c << 1) & 0xff) ^ ((((c >> 7) ^ (int)(v >> i)) & 1) ? 0x07 : 0);
    c = 0xff - c;
    // transmitted c7 first: bit i of the result is the i-th transmitted bit
    int r = 0;
    for (int i = 0; i < 8; i++) r |= ((c >> (7 - i)) & 1) << i;
    return r;
}

// Rate-1/2 encoder output for step t of the bit string `bits` (bit t of the string = word t/32 + 1, bit t%32;
// word 0 is zero = the encoder's initial state).  o = 0: generator 133, o = 1: generator 171.
__device__ __forceinline__ int coded_bit(const uint32_t *bits, int t, int o)
{
    const int w = (t >> 5) + 1;
    const unsigned long long v = ((unsigned long long)bits[w] << 32) | bits[w - 1];
    const unsigned win = (unsigned)(v >> ((t & 31) + 26)) & 0x7f;  // bit 6 = newest
    return __popc(win & (o ? 0171u : 0133u)) & 1;
}

// Axis level of nb Gray-coded bits (first transmitted bit in bit 0 of v): -(2^nb - 1) .. 2^nb - 1.
__device__ __forceinline__ float gray_level(int v, int nb)
{
    int acc = 0, g = 0;
    for (int i = 0; i < nb; i++) {
        acc ^= (v >> i) & 1;
        g = (g << 1) | acc;
    }
    return (float)(2 * g - ((1 << nb) - 1));
}

}  // namespace

__host__ __device__ bool tx_mod(int fmt, int mcs, int nss, int len, TxMod &m)
{
    const int l_nb[8] = {1, 1, 2, 2, 4, 4, 6, 6}, l_cr[8] = {CR_12, CR_34, CR_12, CR_34, CR_12, CR_34, CR_23, CR_34};
    const int n_nb[10] = {1, 2, 2, 4, 4, 6, 6, 6, 8, 8};
    const int n_cr[10] = {CR_12, CR_12, CR_34, CR_12, CR_34, CR_23, CR_34, CR_56, CR_34, CR_56};
    const int num[4] = {1, 2, 3, 5}, den[4] = {2, 3, 4, 6};
    m.fmt = fmt;
    m.mcs = mcs;
    // length 0 = VHT null data packet: training and SIG fields only (tools/phy80211.py:160-175)
    if (len < 0 || len > GR80211_MAX_PSDU || (len == 0 && fmt != GR80211_F_VHT)) return false;
    if (fmt == GR80211_F_L) {
        if (mcs < 0 || mcs > 7) return false;
        m.nss = 1;
        m.nbpscs = l_nb[mcs];
        m.cr = l_cr[mcs];
        m.nsd = 48;
    } else if (fmt == GR80211_F_HT) {
        if (mcs < 0 || mcs > 15) return false;
        m.nss = mcs / 8 + 1;
        m.nbpscs = n_nb[mcs % 8];
        m.cr = n_cr[mcs % 8];
        m.nsd = 52;
    } else if (fmt == GR80211_F_VHT) {
        if (mcs < 0 || mcs > 8 || nss < 1 || nss > 2) return false;
        m.nss = nss;
        m.nbpscs = n_nb[mcs];
        m.cr = n_cr[mcs];
        m.nsd = 52;
    } else {
        return false;
    }
    m.ncbpss = m.nbpscs * m.nsd;
    m.ncbps = m.ncbpss * m.nss;
    m.ndbps = m.ncbps * num[m.cr] / den[m.cr];
    m.nltf = fmt == GR80211_F_L ? 0 : m.nss == 1 ? 1 : 2;
    m.nsym = len == 0 ? 0 : (8 * len + 16 + 6 + m.ndbps - 1) / m.ndbps;
    m.psdu = len;
    m.npad_eof = 0;
    if (fmt == GR80211_F_VHT && len > 0) {
        m.psdu = (m.nsym * m.ndbps - 16 - 6) / 8;
        m.npad_eof = (m.psdu - len) / 4;
    }
    if (fmt == GR80211_F_L) {
        m.lsig_len = len;
    } else {
        const int tx = 20 + 8 + 4 + 4 * m.nltf + (fmt == GR80211_F_VHT ? 4 : 0) + 4 * m.nsym;
        m.lsig_len = (tx - 20) / 4 * 3 - 3;
        if (m.lsig_len > 4095) return false;
    }
    m.npre = fmt == GR80211_F_L ? 400 : 400 + 160 + 80 + 80 * m.nltf + (fmt == GR80211_F_VHT ? 80 : 0);
    m.nsamp = m.npre + 80 * m.nsym;
    return true;
}

namespace {

// One 64-sample symbol body from the lanes' IDFT outputs -> `len` output samples of stream `out`:
// sample j of the segment is body[(j + lead) & 63] (lead = 48: 16-sample guard interval; 32: the double-length
// training fields).  `sh` = cyclic shift in samples (the chain's CSD), `gain` includes 1/64 and the tone scaling.
__device__ __forceinline__ void emit(c32 x, int lane, c32 *body, c32 *out, int seg_off, int len, int lead, int sh, float gain,
                                     float rad, bool first_seg, bool last_seg)
{
    const int n = brev6(lane);
    body[(n - sh) & 63] = cmulf(x, gain);
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < len; j += 64) {
        c32 v = body[(j + lead) & 63];
        if ((j == 0 && !first_seg) || (j == len - 1 && !last_seg)) v = cmulf(v, 0.5f);
        if (rad != 0.0f) {
            // the angle grows with the sample index: reduce it in double before the float sine / cosine
            double a = (double)(seg_off + j) * (double)rad;
            a -= 6.283185307179586 * rint(a * 0.15915494309189535);
            float sn, cs;
            sincosf((float)a, &sn, &cs);
            v = cmul(v, mk(cs, sn));
        }
        out[seg_off + j] = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// The scrambled data field of one PSDU as a packed bit string: bits[w + 1] = bits 32w .. 32w+31 (bits[0] = 0 is the
// encoder's initial state).  Service field (16 zeros; VHT: the SIG-B CRC in its second byte), PSDU bytes, VHT EOF
// padding delimiters, zero padding; xor the scrambler sequence; six tail bits forced to zero at tail_at.
__device__ __forceinline__ void build_bits(uint32_t *bits, const uint8_t *src, int apep, int npad_eof, int nbits, int tail_at,
                                           int sigb_crc, int scrambler, int tid)
{
    // scrambler x^7 + x^4 + 1: its 127-bit output period for this seed, twice over in 192 bits
    unsigned long long e0 = 0, e1 = 0, e2 = 0;
    {
        unsigned st = (unsigned)scrambler;
        unsigned long long lo = 0, hi = 0;
        for (int i = 0; i < 127; i++) {
            const unsigned fb = ((st >> 6) ^ (st >> 3)) & 1u;
            if (i < 64) lo |= (unsigned long long)fb << i;
            else hi |= (unsigned long long)fb << (i - 64);
            st = ((st << 1) & 0x7eu) | fb;
        }
        e0 = lo;
        e1 = hi | (lo << 63);
        e2 = (lo >> 1) | (hi << 63);
    }
    const int nwords = (nbits + 31) / 32;
    const unsigned eof_delim = 0x4E000001u | ((unsigned)crc8_bits(1ull, 16) << 16);
    for (int w = tid; w < nwords; w += 64 * TX_WAVES) {
        unsigned data = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 4 * w - 2 + k;  // PSDU byte that occupies bits 8k .. 8k+7 of this word
            unsigned by = 0;
            if (j == -1) by = (unsigned)sigb_crc;  // VHT: second service byte carries the SIG-B CRC
            else if (j >= 0 && j < apep) by = src[j];
            else if (j >= apep && j < apep + 4 * npad_eof) by = (eof_delim >> (8 * ((j - apep) & 3))) & 0xff;
            data |= by << (8 * k);
        }
        const int p = (32 * w) % 127;
        const unsigned long long a = p < 64 ? e0 : e1, b = p < 64 ? e1 : e2;
        const int sh = p & 63;
        const unsigned seq = (unsigned)(sh ? (a >> sh) | (b << (64 - sh)) : a);
        unsigned v = data ^ seq;
        // tail bits are zero after scrambling; VHT scrambles nbits - 6 bits and appends the tail
        const int lo = tail_at - 32 * w;
        if (lo < 32 && lo > -6) {
            const unsigned long long msk = 0x3full << (lo + 6);
            v &= ~(unsigned)(msk >> 6);
        }
        if (32 * w + 32 > nbits) v &= (nbits & 31) ? ((1u << (nbits & 31)) - 1u) : 0xffffffffu;
        bits[w + 1] = v;
    }
    if (tid == 0) {
        bits[0] = 0;
        bits[nwords + 1] = 0;
    }
}

}  // namespace

__global__ __launch_bounds__(64 * TX_WAVES) void k_tx(const uint8_t *__restrict__ psdu, unsigned long long psdu_bytes,
                                                      const gr80211_tx_desc *__restrict__ descs, int n,
                                                      c32 *__restrict__ iq0, c32 *__restrict__ iq1,
                                                      unsigned long long arena_len, int32_t *__restrict__ n_samples)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ uint32_t s_bits[TX_BIT_WORDS];
    __shared__ uint32_t s_sig[4];  // [0] L-SIG, [1..2] HT-SIG / VHT-SIG-A, [3] VHT-SIG-B
    __shared__ c32 s_body[TX_WAVES][64];
    const gr80211_tx_desc d = descs[f];
    TxMod m;
    bool ok = tx_mod(d.format, d.mcs, d.nss, d.psdu_len, m);
    ok = ok && (unsigned long long)d.psdu_off + (unsigned long long)d.psdu_len <= psdu_bytes;
    ok = ok && d.out_off + (unsigned long long)(ok ? m.nsamp : 0) <= arena_len && (m.nss == 1 || iq1 != nullptr);
    ok = ok && d.scrambler >= 1 && d.scrambler <= 127;
    if (!ok) {
        if (tid == 0 && n_samples) n_samples[f] = -1;
        return;
    }
    if (tid == 0 && n_samples) n_samples[f] = m.nsamp;
    const uint8_t *src = psdu + d.psdu_off;
    const bool vht = m.fmt == GR80211_F_VHT, ht = m.fmt == GR80211_F_HT, leg = m.fmt == GR80211_F_L;

    // ---- SIG field bits (LSB = first transmitted bit) -------------------------------------------------
    // VHT-SIG-B: length in 4-byte units (17 bits) + 3 reserved ones; its CRC-8 rides in the service field.
    // (a null data packet carries the fixed pattern of IEEE 802.11-2016 table 21-15 instead)
    const unsigned sigb = !vht ? 0u : d.psdu_len == 0 ? 0x422E0u : (((unsigned)(d.psdu_len + 3) / 4) & 0x1ffff) | (7u << 17);
    const int sigb_crc = vht ? crc8_bits(sigb, 20) : 0;
    if (tid == 0) {
        const int rate_bits[8] = {0xB, 0xF, 0xA, 0xE, 0x9, 0xD, 0x8, 0xC};  // R1..R4, R1 first
        unsigned ls = (unsigned)rate_bits[leg ? m.mcs : 0] | ((unsigned)m.lsig_len << 5);
        ls |= (unsigned)(__popc(ls) & 1) << 17;
        s_sig[0] = ls;
        unsigned long long sg = 0;
        if (ht) {
            // mcs(7) cbw(1) length(16) | smoothing=1 not-sounding=1 reserved=1 aggregation=0 stbc=00 fec=0 sgi=0 ness=00
            sg = (unsigned long long)m.mcs | ((unsigned long long)d.psdu_len << 8) | (7ull << 24);
            sg |= (unsigned long long)crc8_bits(sg, 34) << 34;
        } else if (vht) {
            // A1: bw(2)=0 rsvd=1 stbc=0 gid(6)=0 nsts-1(3) paid(9)=0 txop_ps=0 rsvd=1
            // A2: sgi=0 sgi_disamb=0 coding=0 ldpc_extra=0 mcs(4) beamformed=0 rsvd=1
            sg = (1ull << 2) | ((unsigned long long)(m.nss - 1) << 10) | (1ull << 23);
            sg |= ((unsigned long long)m.mcs << 28) | (1ull << 33);
            sg |= (unsigned long long)crc8_bits(sg, 34) << 34;
        }
        s_sig[1] = (uint32_t)sg;
        s_sig[2] = (uint32_t)(sg >> 32);
        s_sig[3] = sigb;
    }

    // ---- scrambled data bits as a packed string
    const int nbits = m.nsym * m.ndbps;
    build_bits(s_bits, src, d.psdu_len, m.npad_eof, nbits, vht ? nbits - 6 : 16 + 8 * d.psdu_len, sigb_crc, d.scrambler, tid);
    __syncthreads();

    // ---- per-lane constants ---------------------------------------------------------------------------
    const int kt = bin_to_tone(lane);
    const int pil = pilot_index(lane);
    const int d_l = c_tx.data_idx_l[lane], d_nl = c_tx.data_idx_nl[lane];
    const float ltf_l = c_tx.ltf_l[lane], ltf_nl = c_tx.ltf_nl[lane];
    // L-STF / HT-STF tones: +-4, 8, .., 24 with sqrt(1/2)(1+j) times the sign pattern
    float stf = 0.0f;
    if (kt != 0 && (kt & 3) == 0 && kt >= -24 && kt <= 24) {
        const int sgn[13] = {1, -1, 1, -1, -1, 1, 0, -1, -1, 1, 1, 1, 1};
        stf = 0.70710678118654752f * (float)sgn[(kt + 24) / 4];
    }
    const float base_pil[4] = {1.0f, 1.0f, 1.0f, -1.0f};
    const int didx = m.nbpscs == 1 ? 0 : m.nbpscs == 2 ? 1 : m.nbpscs == 4 ? 2 : m.nbpscs == 6 ? 3 : 4;
    // data bits of this lane's tone: trellis step within the symbol and generator of each, per stream
    const int P = m.cr == CR_12 ? 1 : m.cr == CR_23 ? 2 : m.cr == CR_34 ? 3 : 5;
    const int K = P + 1;
    const int spar = m.nbpscs / 2 > 1 ? m.nbpscs / 2 : 1;
    const int dd = leg ? d_l : d_nl;
    uint16_t tpos[2][8];
    unsigned gsel[2] = {0, 0};
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            tpos[s][b] = 0;
            if (dd >= 0 && b < m.nbpscs && s < m.nss) {
                const int j = dd * m.nbpscs + b;
                const int kk = leg ? c_tx.deint_l[didx][j] : s == 0 ? c_tx.deint_nl[didx][j] : c_tx.deint_nl2[didx][j];
                const int c = m.nss == 1 ? kk : ((kk / spar) * 2 + s) * spar + (kk % spar);
                const int q = c / K, r = c % K;
                tpos[s][b] = (uint16_t)(P * q + (r < 2 ? 0 : r - 1));
                gsel[s] |= (unsigned)(r & 1) << b;
            }
        }
    const float qnorm = m.nbpscs == 1 ? 1.0f : m.nbpscs == 2 ? 0.70710678118654752f : m.nbpscs == 4 ? 0.31622776601683794f
                        : m.nbpscs == 6 ? 0.15430334996209191f : 0.076696498884737041f;

    c32 *body = s_body[wave];
    const int nseg = 3 + (leg ? 0 : 3 + m.nltf + (vht ? 1 : 0)) + m.nsym;
    const float amp = d.scale / 64.0f;
    const float g12 = amp / sqrtf(12.0f * (float)m.nss), g52 = amp / sqrtf(52.0f * (float)m.nss),
                g56 = amp / sqrtf(56.0f * (float)m.nss);
    const int first_data = nseg - m.nsym;

    enum { K_STF, K_LLTF, K_SIG, K_NLSTF, K_NLLTF, K_SIGB, K_DATA };
    for (int q = wave; q < nseg; q += TX_WAVES) {
        // segment geometry and kind
        int off, len = 80, lead = 48, kind;
        if (q == 0) { off = 0; len = 160; lead = 32; kind = K_STF; }
        else if (q == 1) { off = 160; len = 160; lead = 32; kind = K_LLTF; }
        else {
            off = 320 + 80 * (q - 2);
            if (q >= first_data) kind = K_DATA;
            else if (q <= 4) kind = K_SIG;
            else if (q == 5) kind = K_NLSTF;
            else if (q < 6 + m.nltf) kind = K_NLLTF;
            else kind = K_SIGB;
        }
        const int ns = q - first_data;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s >= m.nss) break;
            c32 X = mk(0.0f, 0.0f);
            float gain = g52;
            int sh = s == 0 ? 0 : 4;  // legacy part: -200 ns on the second chain = 4 samples
            if (kind == K_STF || kind == K_NLSTF) {
                X = mk(stf, stf);
                gain = g12;
            } else if (kind == K_LLTF) {
                X = mk(ltf_l, 0.0f);
            } else if (kind == K_SIG) {
                // BPSK / QBPSK SIG symbol: 24 bits -> 48 coded bits, legacy interleaver and tone plan
                float v = 0.0f;
                if (d_l >= 0) {
                    const int kk = c_tx.deint_l[0][d_l];
                    const int t = (kk >> 1) + (q == 4 ? 24 : 0);
                    const unsigned long long str =
                        q == 2 ? (unsigned long long)s_sig[0] : ((unsigned long long)s_sig[2] << 32) | s_sig[1];
                    const unsigned win = (unsigned)((str << 6) >> t) & 0x7f;  // encoder starts from the zero state
                    v = (__popc(win & ((kk & 1) ? 0171u : 0133u)) & 1) ? 1.0f : -1.0f;
                } else if (pil >= 0) {
                    v = base_pil[pil];  // polarity p_0 .. p_2 = +1
                }
                const bool qb = (ht && q >= 3) || (vht && q == 4);
                X = (qb && pil < 0) ? mk(0.0f, v) : mk(v, 0.0f);
            } else if (kind == K_NLLTF) {
                const int il = q - 6;
                float v = ltf_nl * ((s == 0 && il == 1) ? -1.0f : 1.0f);       // P_HTLTF rows (1 -1), (1 1)
                if (vht && pil >= 0) v = ltf_nl * (il == 0 ? 1.0f : -1.0f);    // VHT pilots: R_HTLTF row 1 on every stream
                X = mk(v, 0.0f);
                gain = g56;
            } else if (kind == K_SIGB) {
                // VHT-SIG-B: 26 bits -> 52 coded bits, 13-column interleaver, same on every stream
                float v = 0.0f;
                if (d_nl >= 0) {
                    const int kk = c_tx.deint_nl[0][d_nl];
                    const unsigned win = (unsigned)(((unsigned long long)s_sig[3] << 6) >> (kk >> 1)) & 0x7f;
                    v = (__popc(win & ((kk & 1) ? 0171u : 0133u)) & 1) ? 1.0f : -1.0f;
                } else if (pil >= 0) {
                    v = base_pil[pil];
                }
                X = mk(v, 0.0f);
                gain = g56;
            } else {
                gain = leg ? g52 : g56;
                if (dd >= 0) {
                    const int t0 = ns * m.ndbps;
                    int chip = 0;
#pragma unroll
                    for (int b = 0; b < 8; b++)
                        if (b < m.nbpscs) chip |= coded_bit(s_bits, t0 + tpos[s][b], (gsel[s] >> b) & 1) << b;
                    if (m.nbpscs == 1) {
                        X = mk(chip ? 1.0f : -1.0f, 0.0f);
                    } else {
                        const int nb = m.nbpscs >> 1;
                        X = mk(gray_level(chip & ((1 << nb) - 1), nb) * qnorm, gray_level(chip >> nb, nb) * qnorm);
                    }
                } else if (pil >= 0) {
                    float pv;
                    if (leg) {
                        pv = base_pil[pil];
                    } else if (ht && m.nss == 2) {
                        const int r = (pil + ns) & 3;  // (1 1 -1 -1) and (1 -1 -1 1)
                        pv = s == 0 ? (r < 2 ? 1.0f : -1.0f) : ((r == 0 || r == 3) ? 1.0f : -1.0f);
                    } else {
                        pv = base_pil[(pil + ns) & 3];
                    }
                    const int pidx = leg ? 1 : ht ? 3 : 4;
                    X = mk(pv * c_tx.pilot_p[(pidx + ns) % 127], 0.0f);
                }
            }
            if (kind >= K_NLSTF && !(kind == K_DATA && leg)) sh = s == 0 ? 0 : 8;  // HT/VHT part: -400 ns
            const c32 x = wave_idft64(X, lane);
            emit(x, lane, body, (s == 0 ? iq0 : iq1) + d.out_off, off, len, lead, sh, gain, d.cfo_rad, q == 0, q == nseg - 1);
        }
    }
}

void launch_tx(const uint8_t *psdu, uint64_t psdu_bytes, const gr80211_tx_desc *desc, int n, void *iq0, void *iq1,
               uint64_t arena_len, int32_t *n_samples, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_tx, dim3(n), dim3(64 * TX_WAVES), 0, s, psdu, (unsigned long long)psdu_bytes, desc, n, (c32 *)iq0,
                       (c32 *)iq1, (unsigned long long)arena_len, n_samples);
}

// =========================================================================================
// k_tx_mu: two-user VHT MU-MIMO frames (tools/phy80211.py genAmpduMu 183-224, 596-662, 815-838): one space-time
// stream per user, each with its own A-MPDU, MCS and VHT-SIG-B; from the VHT-STF on, every tone of the two streams
// is mixed onto the two transmit chains by the steering matrix Q of that tone (the streams' cyclic shifts are
// applied before Q, as phase ramps).  The legacy part and VHT-SIG-A go out unsteered like a two-chain frame.
// =========================================================================================
__global__ __launch_bounds__(64 * TX_WAVES) void k_tx_mu(const uint8_t *__restrict__ psdu, unsigned long long psdu_bytes,
                                                         const gr80211_tx_mu_desc *__restrict__ descs, int n,
                                                         const c32 *__restrict__ qmat, int n_q, c32 *__restrict__ iq0,
                                                         c32 *__restrict__ iq1, unsigned long long arena_len,
                                                         int32_t *__restrict__ n_samples)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ uint32_t s_bits[2][TX_BIT_WORDS];
    __shared__ uint32_t s_sig[5];  // [0] L-SIG, [1..2] VHT-SIG-A, [3..4] VHT-SIG-B of user 0 / 1
    __shared__ c32 s_body[TX_WAVES][64];
    const gr80211_tx_mu_desc d = descs[f];
    TxMod mu[2];
    bool ok = d.group_id >= 1 && d.group_id <= 62 && d.q_index >= 0 && d.q_index < n_q && d.scrambler >= 1 && d.scrambler <= 127;
    int nsym = 0;
    for (int u = 0; u < 2; u++) {
        ok = ok && d.psdu_len[u] >= 4 && (d.psdu_len[u] & 3) == 0 && tx_mod(GR80211_F_VHT, d.mcs[u], 1, d.psdu_len[u], mu[u]);
        ok = ok && (unsigned long long)d.psdu_off[u] + (unsigned long long)d.psdu_len[u] <= psdu_bytes;
        if (ok && mu[u].nsym > nsym) nsym = mu[u].nsym;
    }
    // frame timing of a two-stream VHT frame with the longer user's symbol count
    const int txus = 20 + 8 + 4 + 4 * 2 + 4 + 4 * nsym;
    const int lsig_len = (txus - 20) / 4 * 3 - 3;
    const int nsamp = 400 + 160 + 80 + 160 + 80 + 80 * nsym;
    ok = ok && lsig_len <= 4095 && d.out_off + (unsigned long long)nsamp <= arena_len;
    // both users are padded to the common symbol count: their PSDUs must still fit what the receiver accepts
    for (int u = 0; u < 2; u++) ok = ok && (nsym * mu[u].ndbps - 22) / 8 <= GR80211_MAX_PSDU + 3;
    if (!ok) {
        if (tid == 0 && n_samples) n_samples[f] = -1;
        return;
    }
    if (tid == 0 && n_samples) n_samples[f] = nsamp;

    // ---- SIG bits.  MU VHT-SIG-B (20 MHz): length / 4 in 16 bits, MCS in 4 bits; its CRC-8 goes to the service field
    unsigned sigb[2];
    int sigb_crc[2];
    for (int u = 0; u < 2; u++) {
        sigb[u] = ((unsigned)(d.psdu_len[u] / 4) & 0xffffu) | ((unsigned)d.mcs[u] << 16);
        sigb_crc[u] = crc8_bits(sigb[u], 20);
    }
    if (tid == 0) {
        unsigned ls = 0xBu | ((unsigned)lsig_len << 5);
        ls |= (unsigned)(__popc(ls) & 1) << 17;
        s_sig[0] = ls;
        // A1: bw = 0, reserved 1, stbc 0, group id, one space-time stream for users 0 and 1, none for 2 and 3,
        //     txop_ps 0, reserved 1.  A2: sgi 0 0, coding of user 0 = BCC, ldpc extra 0, coding of user 1 = BCC,
        //     two absent users reserved 1 1, reserved 1, reserved 1, reserved 1
        unsigned long long sg = (1ull << 2) | ((unsigned long long)d.group_id << 4) | (1ull << 10) | (1ull << 13) | (1ull << 23);
        sg |= 0x1full << 29;
        sg |= (unsigned long long)crc8_bits(sg, 34) << 34;
        s_sig[1] = (uint32_t)sg;
        s_sig[2] = (uint32_t)(sg >> 32);
        s_sig[3] = sigb[0];
        s_sig[4] = sigb[1];
    }
    for (int u = 0; u < 2; u++) {
        const int nbits = nsym * mu[u].ndbps;
        const int psdu_on_air = (nbits - 22) / 8;
        build_bits(s_bits[u], psdu + d.psdu_off[u], d.psdu_len[u], (psdu_on_air - d.psdu_len[u]) / 4, nbits, nbits - 6,
                   sigb_crc[u], d.scrambler, tid);
    }
    __syncthreads();

    // ---- per-lane constants
    const int kt = bin_to_tone(lane);
    const int pil = pilot_index(lane);
    const int d_l = c_tx.data_idx_l[lane], d_nl = c_tx.data_idx_nl[lane];
    const float ltf_l = c_tx.ltf_l[lane], ltf_nl = c_tx.ltf_nl[lane];
    float stf = 0.0f;
    if (kt != 0 && (kt & 3) == 0 && kt >= -24 && kt <= 24) {
        const int sgn[13] = {1, -1, 1, -1, -1, 1, 0, -1, -1, 1, 1, 1, 1};
        stf = 0.70710678118654752f * (float)sgn[(kt + 24) / 4];
    }
    const float base_pil[4] = {1.0f, 1.0f, 1.0f, -1.0f};
    uint16_t tpos[2][8];
    unsigned gsel[2] = {0, 0};
    float qnorm[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int nb = mu[u].nbpscs;
        const int didx = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : nb == 6 ? 3 : 4;
        const int P = mu[u].cr == CR_12 ? 1 : mu[u].cr == CR_23 ? 2 : mu[u].cr == CR_34 ? 3 : 5;
        const int K = P + 1;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            tpos[u][b] = 0;
            if (d_nl >= 0 && b < nb) {
                const int c = c_tx.deint_nl[didx][d_nl * nb + b];
                const int q = c / K, r = c % K;
                tpos[u][b] = (uint16_t)(P * q + (r < 2 ? 0 : r - 1));
                gsel[u] |= (unsigned)(r & 1) << b;
            }
        }
        qnorm[u] = nb == 1 ? 1.0f : nb == 2 ? 0.70710678118654752f : nb == 4 ? 0.31622776601683794f
                   : nb == 6 ? 0.15430334996209191f : 0.076696498884737041f;
    }
    // steering matrix of this lane's tone: chain r gets q[r][0] * stream 0 + q[r][1] * stream 1
    const c32 *qm = qmat + ((size_t)d.q_index * 64 + lane) * 4;
    const c32 q00 = qm[0], q01 = qm[1], q10 = qm[2], q11 = qm[3];
    // -400 ns on the second stream = phase +2 pi 8 k / 64 on tone k
    const float csc[8] = {1.0f, 0.70710678118654752f, 0.0f, -0.70710678118654752f, -1.0f, -0.70710678118654752f, 0.0f,
                          0.70710678118654752f};
    const c32 cs1 = mk(csc[kt & 7], csc[(kt + 6) & 7]);  // cos, sin of pi k / 4

    c32 *body = s_body[wave];
    const int nseg = 9 + nsym;
    const float amp = d.scale / 64.0f;
    const float g12 = amp / sqrtf(24.0f), g52 = amp / sqrtf(104.0f), g56 = amp / sqrtf(112.0f);
    for (int q = wave; q < nseg; q += TX_WAVES) {
        int off, len = 80, lead = 48;
        if (q == 0) { off = 0; len = 160; lead = 32; }
        else if (q == 1) { off = 160; len = 160; lead = 32; }
        else off = 320 + 80 * (q - 2);
        if (q <= 4) {
            // unsteered legacy part: the same tones on both chains, -200 ns (4 samples) on the second
            c32 X = mk(0.0f, 0.0f);
            float gain = g52;
            if (q == 0) { X = mk(stf, stf); gain = g12; }
            else if (q == 1) X = mk(ltf_l, 0.0f);
            else {
                float v = 0.0f;
                if (d_l >= 0) {
                    const int kk = c_tx.deint_l[0][d_l];
                    const int t = (kk >> 1) + (q == 4 ? 24 : 0);
                    const unsigned long long str =
                        q == 2 ? (unsigned long long)s_sig[0] : ((unsigned long long)s_sig[2] << 32) | s_sig[1];
                    const unsigned win = (unsigned)((str << 6) >> t) & 0x7f;
                    v = (__popc(win & ((kk & 1) ? 0171u : 0133u)) & 1) ? 1.0f : -1.0f;
                } else if (pil >= 0) {
                    v = base_pil[pil];
                }
                X = (q == 4 && pil < 0) ? mk(0.0f, v) : mk(v, 0.0f);
            }
            const c32 x = wave_idft64(X, lane);
            emit(x, lane, body, iq0 + d.out_off, off, len, lead, 0, gain, d.cfo_rad, q == 0, false);
            emit(x, lane, body, iq1 + d.out_off, off, len, lead, 4, gain, d.cfo_rad, q == 0, false);
            continue;
        }
        // steered part: the two streams' tone values
        c32 X0 = mk(0.0f, 0.0f), X1 = mk(0.0f, 0.0f);
        float gain = g56;
        if (q == 5) {
            X0 = X1 = mk(stf, stf);
            gain = g12;
        } else if (q <= 7) {
            const int il = q - 6;
            float v0 = ltf_nl * (il == 1 ? -1.0f : 1.0f), v1 = ltf_nl;  // P rows (1 -1), (1 1)
            if (pil >= 0) v0 = v1 = ltf_nl * (il == 0 ? 1.0f : -1.0f);  // pilots: R row on both streams
            X0 = mk(v0, 0.0f);
            X1 = mk(v1, 0.0f);
        } else if (q == 8) {
            float v[2] = {0.0f, 0.0f};
            if (d_nl >= 0) {
                const int kk = c_tx.deint_nl[0][d_nl];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const unsigned win = (unsigned)(((unsigned long long)s_sig[3 + u] << 6) >> (kk >> 1)) & 0x7f;
                    v[u] = (__popc(win & ((kk & 1) ? 0171u : 0133u)) & 1) ? 1.0f : -1.0f;
                }
            } else if (pil >= 0) {
                v[0] = v[1] = base_pil[pil];
            }
            X0 = mk(v[0], 0.0f);
            X1 = mk(v[1], 0.0f);
        } else {
            const int ns = q - 9;
            c32 X[2] = {mk(0.0f, 0.0f), mk(0.0f, 0.0f)};
            if (d_nl >= 0) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int nb = mu[u].nbpscs, t0 = ns * mu[u].ndbps;
                    int chip = 0;
#pragma unroll
                    for (int b = 0; b < 8; b++)
                        if (b < nb) chip |= coded_bit(s_bits[u], t0 + tpos[u][b], (gsel[u] >> b) & 1) << b;
                    if (nb == 1) X[u] = mk(chip ? 1.0f : -1.0f, 0.0f);
                    else X[u] = mk(gray_level(chip & ((1 << (nb >> 1)) - 1), nb >> 1) * qnorm[u],
                                   gray_level(chip >> (nb >> 1), nb >> 1) * qnorm[u]);
                }
            } else if (pil >= 0) {
                const float pv = base_pil[(pil + ns) & 3] * c_tx.pilot_p[(4 + ns) % 127];
                X[0] = X[1] = mk(pv, 0.0f);
            }
            X0 = X[0];
            X1 = X[1];
        }
        X1 = cmul(X1, cs1);
        const c32 Y0 = cadd(cmul(q00, X0), cmul(q01, X1)), Y1 = cadd(cmul(q10, X0), cmul(q11, X1));
        const c32 y0 = wave_idft64(Y0, lane);
        emit(y0, lane, body, iq0 + d.out_off, off, len, lead, 0, gain, d.cfo_rad, false, q == nseg - 1);
        const c32 y1 = wave_idft64(Y1, lane);
        emit(y1, lane, body, iq1 + d.out_off, off, len, lead, 0, gain, d.cfo_rad, false, q == nseg - 1);
    }
}

void launch_tx_mu(const uint8_t *psdu, uint64_t psdu_bytes, const gr80211_tx_mu_desc *desc, int n, const void *qmat, int n_q,
                  void *iq0, void *iq1, uint64_t arena_len, int32_t *n_samples, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_tx_mu, dim3(n), dim3(64 * TX_WAVES), 0, s, psdu, (unsigned long long)psdu_bytes, desc, n,
                       (const c32 *)qmat, n_q, (c32 *)iq0, (c32 *)iq1, (unsigned long long)arena_len, n_samples);
}

}  // namespace gr80211
