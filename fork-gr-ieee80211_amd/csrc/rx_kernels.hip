// HIP kernels of the 802.11 OFDM RX hot path for MI355X (gfx950, wave64).
//
//   k_preamble   one wavefront per frame: CFO-derotated L-LTF/L-SIG FFTs, LS channel estimate,
//                L-SIG / HT-SIG / VHT-SIG-A / VHT-SIG-B soft decoding (lane = trellis state),
//                HT/VHT channel estimate incl. the 2x2 Gram inverse.
//   k_sched_*    bucket frames by code rate and trellis length into groups of 64 (decoder lanes).
//   k_demod<1|2> one workgroup per frame, its waves stride over the OFDM symbols: FFT-64 by lane shuffles, one-tap
//                or 2x2 ZF equalisation, pilot phase tracking, QAM soft demap, deinterleave / stream deparse
//                folded into the place a soft bit takes in the wave's LDS stage.
//   k_decode     one wavefront per group (four groups per workgroup), lane = frame: 64-state add-compare-select
//                fully in registers, depuncturing folded into the per-rate unrolled step sequence, LLRs staged
//                (transposed) through LDS, survivor bits streamed to HBM; then the same wave walks them back:
//                full-frame traceback, descrambling, (A-)MPDU assembly, FCS.
//   k_viterbi / k_traceback   the two halves of k_decode as kernels of their own (k_traceback also follows the
//                four-waves-per-group forward pass of vit_split.hip).
//
// Built with -ffp-contract=off: the decoder's path metrics must be the plain float32 sums the
// reference forms (lib/decode_impl.cc:231-258), and the demodulator follows the reference's
// operation order so that soft bits agree to rounding.
#include <utility>

#include "acs_common.h"
#include "split_forward.h"
#include "rx_kernels.h"
#include "assemble.h"

namespace gr80211 {

__constant__ Tables c_tab;

int upload_tables(const Tables &t)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &t, sizeof(Tables));
}

// =========================================================================================
// wave-level building blocks
// =========================================================================================

// 64-point forward DFT across the lanes of one wavefront (radix-2 decimation in frequency).
// Lane l enters with x[l] and leaves with X[brev6(l)].  tw: the lane's twiddle of each of the first five stages,
// negated (fft_twiddles), fetched once per kernel instead of once per transform.
__device__ __forceinline__ void fft_twiddles(c32 (&tw)[5], int lane)
{
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int k = (lane & ((32 >> s) - 1)) << s;
        tw[s] = mk(-c_tab.tw_re[k], -c_tab.tw_im[k]);
    }
}
// A stage pairs lane l with lane l ^ half; the lower lane keeps lo + hi, the upper one (lo - hi) * w.  Written
// without telling the two apart before the end: the sum does not care about the order, and the upper lane's
// x - o is -(lo - hi) exactly, whose sign the negated twiddle takes back (every product changes sign exactly, so
// the result is the same bit for bit).  Two selects per stage instead of six.
__device__ __forceinline__ c32 wave_fft64_tw(c32 x, int lane, const c32 (&tw)[5])
{
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int half = 32 >> s;
        const c32 o = mk(__shfl_xor(x.x, half), __shfl_xor(x.y, half));
        const bool up = (lane & half) != 0;
        const c32 sum = cadd(x, o), d = csub(x, o);
        const c32 dif = s < 5 ? cmul(d, tw[s]) : cneg(d);
        x = mk(up ? dif.x : sum.x, up ? dif.y : sum.y);
    }
    return x;
}

constexpr int brev6c(int v)
{
    int r = 0;
    for (int b = 0; b < 6; b++) r |= ((v >> b) & 1) << (5 - b);
    return r;
}
constexpr int LN7 = brev6c(7), LN21 = brev6c(21), LN43 = brev6c(43), LN57 = brev6c(57);

// Value of lane L in every lane (v_readlane: the result is wave-uniform and costs no LDS crossbar trip).
template <int L>
__device__ __forceinline__ c32 lane_c(c32 v)
{
    return mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), L)),
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), L)));
}

__device__ __forceinline__ bool is_pilot_bin(int b) { return b == 7 || b == 21 || b == 43 || b == 57; }

// Sample k of the stream `demod` sees: the samples after L-SIG, CFO-derotated by
// exp(j*(k+224)*rad) (lib/signal_impl.cc:172-173).  Only nvalid samples are forwarded.
__device__ __forceinline__ c32 load_samp(const c32 *in, int k, int nvalid, float rad)
{
    if (k >= nvalid) return mk(0.0f, 0.0f);
    c32 v = in[224 + k];
    if (rad != 0.0f) {
        float sn, cs;
        sincosf((float)(k + 224) * rad, &sn, &cs);
        v = cmul(v, mk(cs, sn));
    }
    return v;
}

// Rate-1/2 64-state soft Viterbi over <= 48 steps with one trellis state per lane
// (svSigDecoder::decode, lib/cloud80211phy.cc:2001-2088).  llr lives in LDS; the decoded bits come back as one word.  The decision mask of step t
// (one bit per state, a wave-wide ballot) is parked in lane t of a register pair, and the walk back from state 0 runs
// on the scalar unit: no memory, no single-lane vector code.
__device__ __forceinline__ float pick3(float t1, float t0, float t01, uint32_t k1, uint32_t k2, uint32_t k3)
{
    // one of the three values, or 0.0f: the masks are all ones for at most one of them (no branches, no selects
    // the compiler could turn into branches)
    return __uint_as_float((__float_as_uint(t1) & k1) | (__float_as_uint(t0) & k2) | (__float_as_uint(t01) & k3));
}

// One add-compare-select step of the state in this lane; the decision goes into bit `bit` of hist.
__device__ __forceinline__ void sig_acs(float &m, uint32_t &hist, float t0, float t1, int p0, int p1, uint32_t bit,
                                        uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b1, uint32_t b2, uint32_t b3)
{
    const float t01 = t1 + t0;
    const float m0 = __shfl(m, p0), m1 = __shfl(m, p1);
    const float c0 = m0 + pick3(t1, t0, t01, a1, a2, a3), c1 = m1 + pick3(t1, t0, t01, b1, b2, b3);
    const bool g = c1 > c0;  // ties keep the even predecessor, as the reference's strict '>' does
    m = g ? c1 : c0;
    hist |= g ? bit : 0u;
}

__device__ __forceinline__ unsigned long long wave_sig_viterbi(const float *llr, int T, int lane)
{
    const int j = lane & 31, b = lane >> 5, p0 = 2 * j, p1 = 2 * j + 1;
    const int o0 = sv_out(p0, b), o1 = sv_out(p1, b);
    // branch metric of output pair o: 0, t1, t0, t1 + t0 for o = 0..3
    const uint32_t a1 = o0 == 1 ? ~0u : 0u, a2 = o0 == 2 ? ~0u : 0u, a3 = o0 == 3 ? ~0u : 0u;
    const uint32_t b1 = o1 == 1 ? ~0u : 0u, b2 = o1 == 2 ? ~0u : 0u, b3 = o1 == 3 ? ~0u : 0u;
    float m = lane == 0 ? 0.0f : NEG_INIT;
    // every lane keeps the decisions of its own state, one bit per step; the walk back from state 0 then runs on
    // the scalar unit, fetching the word of the state it stands on with a lane read: no memory, no one-lane code
    uint32_t hlo = 0, hhi = 0;
    const int Tlo = T < 32 ? T : 32;
    __syncthreads();
    for (int t = 0; t < Tlo; t++) sig_acs(m, hlo, llr[2 * t], llr[2 * t + 1], p0, p1, 1u << t, a1, a2, a3, b1, b2, b3);
    for (int t = 32; t < T; t++) sig_acs(m, hhi, llr[2 * t], llr[2 * t + 1], p0, p1, 1u << (t - 32), a1, a2, a3, b1, b2, b3);
    unsigned long long decoded = 0;
    int st = 0;
    for (int t = T - 1; t >= 32; t--) {
        decoded |= (unsigned long long)(st >> 5) << t;
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)hhi, st);
        st = ((st & 31) << 1) | (int)((w >> (t - 32)) & 1u);
    }
    for (int t = Tlo - 1; t >= 0; t--) {
        decoded |= (unsigned long long)(st >> 5) << t;
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)hlo, st);
        st = ((st & 31) << 1) | (int)((w >> t) & 1u);
    }
    return decoded;  // bit t = decoded bit t, the same value in every lane (scalar registers)
}

// ---- SIG field checks and parsers (run by lane 0 on bits in LDS) -------------------------
struct ModS {
    int format, ampdu, nSym, nSymSamp, nSD, nSS, nLTF, mcs, len, mod, cr, nBPSCS, nDBPS, nCBPS, nCBPSS;
};

// The decoded bits of a SIG field, bit i of the word = bit i of the field.  The word is wave-uniform, so what the
// checks and parsers below do with it runs on the scalar unit.
struct SigBits {
    unsigned long long w;
    __device__ __forceinline__ unsigned operator[](int i) const { return (unsigned)((w >> i) & 1ull); }
};

__device__ __forceinline__ unsigned crc8_run(SigBits in, int n)
{
    unsigned c = 0xff;
    for (int i = 0; i < n; i++) c = crc8_step(c, in[i]);
    return c;
}
__device__ __forceinline__ bool crc8_check(SigBits in, int n)  // the CRC's eight bits follow the n it covers
{
    unsigned c = crc8_run(in, n);
    for (int i = 0; i < 8; i++)
        if (in[n + i]) c ^= 1u << (7 - i);
    return (c & 0xff) == 0xff;
}

// lib/cloud80211phy.cc:650-728
__device__ __forceinline__ bool check_legacy(SigBits b, int *mcs, int *len, int *ndbps)
{
    if (!b[3]) return false;
    if (b[4]) return false;
    unsigned sum = 0;
    for (int i = 0; i < 17; i++) sum += b[i];
    if ((sum & 1u) ^ b[17]) return false;
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
    int l = 0;
    for (int i = 0; i < 12; i++) l |= b[i + 5] << i;
    *len = l;
    return !(l > 4095 || l < 14);
}
// lib/cloud80211phy.cc:730-751
__device__ __forceinline__ bool check_ht(SigBits b)
{
    if (b[26] != 1) return false;
    if (!crc8_check(b, 34)) return false;
    if (b[5] + b[6] + b[7] + b[28] + b[29] + b[30] + b[32] + b[33]) return false;
    return true;
}
// lib/cloud80211phy.cc:753-771
__device__ __forceinline__ bool check_vhta(SigBits b)
{
    if (b[2] != 1 || b[23] != 1 || b[33] != 1) return false;
    if (!crc8_check(b, 34)) return false;
    if (b[0] + b[1]) return false;
    return true;
}
__device__ void set_ndbps(ModS *m)
{
    switch (m->cr) {
    case CR_12: m->nDBPS = m->nCBPS / 2; break;
    case CR_23: m->nDBPS = (m->nCBPS * 2) / 3; break;
    case CR_34: m->nDBPS = (m->nCBPS * 3) / 4; break;
    default: m->nDBPS = (m->nCBPS * 5) / 6; break;
    }
}
__device__ int ltf_count(int nss) { return nss == 1 ? 1 : nss == 2 ? 2 : 4; }

// lib/cloud80211phy.cc:773-850
__device__ void parse_legacy(int mcs, int len, ModS *m)
{
    const int mod[8] = {QAM_BPSK, QAM_BPSK, QAM_QPSK, QAM_QPSK, QAM_16, QAM_16, QAM_64, QAM_64};
    const int cr[8] = {CR_12, CR_34, CR_12, CR_34, CR_12, CR_34, CR_23, CR_34};
    const int ndbps[8] = {24, 36, 48, 72, 96, 144, 192, 216};
    const int nbp[8] = {1, 1, 2, 2, 4, 4, 6, 6};
    m->mcs = mcs; m->mod = mod[mcs]; m->cr = cr[mcs]; m->nDBPS = ndbps[mcs];
    m->nBPSCS = nbp[mcs]; m->nCBPS = 48 * nbp[mcs];
    m->len = len; m->nCBPSS = m->nCBPS; m->nSD = 48; m->nSS = 1; m->nLTF = 0;
    m->format = GR80211_F_L; m->nSymSamp = 80;
    m->nSym = (len * 8 + 22) / m->nDBPS + (((len * 8 + 22) % m->nDBPS) != 0);
    m->ampdu = 0;
}
// HT MCS%8 and VHT MCS 0..7 share one modulation / code-rate ladder.
__device__ void set_rate_ladder(int r, ModS *m)
{
    const int mod[10] = {QAM_BPSK, QAM_QPSK, QAM_QPSK, QAM_16, QAM_16, QAM_64, QAM_64, QAM_64, QAM_256, QAM_256};
    const int nbp[10] = {1, 2, 2, 4, 4, 6, 6, 6, 8, 8};
    const int cr[10] = {CR_12, CR_12, CR_34, CR_12, CR_34, CR_23, CR_34, CR_56, CR_34, CR_56};
    m->mod = mod[r]; m->nBPSCS = nbp[r]; m->cr = cr[r];
}
// lib/cloud80211phy.cc:852-998
__device__ __forceinline__ void parse_ht(SigBits b, ModS *m)
{
    int mcs = 0, len = 0;
    for (int i = 0; i < 7; i++) mcs |= b[i] << i;
    for (int i = 0; i < 16; i++) len |= b[i + 8] << i;
    m->format = GR80211_F_HT;
    m->nSymSamp = b[31] ? 72 : 80;
    m->ampdu = b[27] ? 1 : 0;
    m->mcs = mcs;
    set_rate_ladder(mcs % 8, m);
    m->len = len;
    m->nSS = mcs / 8 + 1;
    m->nSD = 52;
    m->nCBPSS = m->nBPSCS * 52;
    m->nCBPS = m->nCBPSS * m->nSS;
    set_ndbps(m);
    m->nLTF = ltf_count(m->nSS);
    m->nSym = (len * 8 + 22) / m->nDBPS + (((len * 8 + 22) % m->nDBPS) != 0);
}
// Rate-dependent fields of a VHT MCS (modParserVht, lib/cloud80211phy.cc:1225-1323); false for MCS 10..15, for
// which the reference keeps the fields of an earlier packet.
__device__ bool set_vht_mcs(int mcs, ModS *m)
{
    m->mcs = mcs;
    if (mcs > 9) return false;
    set_rate_ladder(mcs, m);
    m->nSD = 52;
    m->nCBPSS = m->nBPSCS * 52;
    m->nCBPS = m->nCBPSS * m->nSS;
    set_ndbps(m);
    m->nLTF = ltf_count(m->nSS);
    return true;
}
// lib/cloud80211phy.cc:1090-1178.  Returns 1 for a single-user frame, 2 for an MU frame (group id 1..62: one
// stream for this station, two VHT-LTFs, modulation known after VHT-SIG-B), 0 for an invalid MCS.
__device__ __forceinline__ int parse_vhta(SigBits b, ModS *m)
{
    int gid = 0, nsts = 0, mcs = 0;
    for (int i = 0; i < 6; i++) gid |= b[i + 4] << i;
    m->format = GR80211_F_VHT;
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
    return set_vht_mcs(mcs, m) ? 1 : 0;
}
// lib/cloud80211phy.cc:1184-1192: VHT-SIG-B of an MU frame, 16 bits of length and 4 bits of MCS.
__device__ __forceinline__ bool parse_vhtb_mu(SigBits b, ModS *m)
{
    int l = 0, mcs = 0;
    for (int i = 0; i < 16; i++) l |= b[i] << i;
    for (int i = 0; i < 4; i++) mcs |= b[i + 16] << i;
    if (!set_vht_mcs(mcs, m)) return false;
    m->len = l * 4;
    m->nSym = (m->len * 8 + 16 + 6) / m->nDBPS + (((m->len * 8 + 16 + 6) % m->nDBPS) != 0);
    m->nLTF = 2;
    return true;
}
// lib/cloud80211phy.cc:1180-1223, single-user branch
__device__ __forceinline__ void parse_vhtb(SigBits b, ModS *m)
{
    if ((b[17] + b[18] + b[19]) == 3) {
        int l = 0;
        for (int i = 0; i < 17; i++) l |= b[i] << i;
        m->len = l * 4;
        m->nSym = (m->len * 8 + 16 + 6) / m->nDBPS + (((m->len * 8 + 16 + 6) % m->nDBPS) != 0);
    } else {
        unsigned pat = 0;
        for (int i = 0; i < 20; i++) pat |= (unsigned)b[i] << i;
        if (pat == 0x422E0u) { m->len = 0; m->nSym = 0; }
        else { m->len = -1; m->nSym = -1; }
    }
}

// Two-antenna zero forcing for one tone: x = (H^H H)^-1 H^H y (lib/demod2_impl.cc:515-518).
__device__ __forceinline__ void zf2(c32 F1, c32 F2, const c32 (&h)[4], const c32 (&hi)[4], c32 &s1, c32 &s2)
{
    const c32 t1 = cadd(cmul(F1, cconj(h[0])), cmul(F2, cconj(h[1])));
    const c32 t2 = cadd(cmul(F1, cconj(h[2])), cmul(F2, cconj(h[3])));
    s1 = cadd(cmul(t1, hi[0]), cmul(t2, hi[2]));
    s2 = cadd(cmul(t1, hi[1]), cmul(t2, hi[3]));
}

// =========================================================================================
// k_preamble
// =========================================================================================
constexpr int CH2 = 520;  // per-frame 2x2 channel block: H[64][4], Hinv[64][4], pilot refs [2][4]

// The part of the arena a descriptor may touch: its claim cut to the arena (a descriptor that starts outside the
// arena gets nothing).  Every kernel that reads samples derives its bounds from here.
__device__ __forceinline__ int frame_window(const gr80211_frame_desc &d, uint64_t arena_len, uint64_t *start)
{
    uint64_t so = d.sync_off;
    uint64_t navail64 = d.n_avail;
    if (so >= arena_len) { so = 0; navail64 = 0; }
    if (so + navail64 > arena_len) navail64 = arena_len - so;
    *start = so;
    return (int)(navail64 > 0x7fffffffull ? 0x7fffffffull : navail64);
}

// A descriptor made from `signal`'s tags (gr80211_demod_tagged_batch) points at the first item `demod` sees: the
// 224 samples between the trigger byte and that item (L-LTF, L-SIG) are not in the arena.
constexpr uint32_t DESC_AFTER_SIGNAL = 1u;

// Base pointers such that base[224 + k] is item k of the stream `demod` sees, for either kind of descriptor, and the
// number of samples the frame has counted from the (real or virtual) trigger byte.  The pointers may lie before
// the arena; nothing below base[224] is read through them unless the preamble is there.
__device__ __forceinline__ int frame_origin(const gr80211_frame_desc &d, uint64_t arena_len, const c32 *iq0,
                                            const c32 *iq1, const c32 **in0, const c32 **in1)
{
    uint64_t so;
    const int navail = frame_window(d, arena_len, &so);
    const int gone = (d.reserved & DESC_AFTER_SIGNAL) ? 224 : 0;
    *in0 = iq0 + so - gone;
    *in1 = iq1 ? iq1 + so - gone : nullptr;
    return navail > 0x7fffffff - gone ? 0x7fffffff : navail + gone;
}

__global__ __launch_bounds__(64) void k_preamble(const c32 *__restrict__ iq0, const c32 *__restrict__ iq1,
                                                 uint64_t arena_len, const gr80211_frame_desc *__restrict__ descs,
                                                 int n, gr80211_frame_info *__restrict__ info,
                                                 c32 *__restrict__ chan1, c32 *__restrict__ chan2,
                                                 const gr80211_sig_desc *__restrict__ sig,
                                                 const c32 *__restrict__ chan_in, c32 *__restrict__ chanl_out)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    const int lane = threadIdx.x;
    const int bin = brev6(lane);

    __shared__ float s_llr[2][96];
    __shared__ ModS s_m;
    __shared__ int s_ctl[8];  // 0 status, 1 lsig_mcs, 2 lsig_len, 3 nsamp, 4 fmt, 5 trellis, 6 data_off, 7 MU frame
    __shared__ float s_snr[2];
    __shared__ double s_red[64];

    const gr80211_frame_desc d = descs[f];
    const c32 *in0, *in1;
    const int navail = frame_origin(d, arena_len, iq0, iq1, &in0, &in1);
    const float rad = d.rad;
    const bool two_ant = in1 != nullptr;
    const bool after_signal = (d.reserved & DESC_AFTER_SIGNAL) != 0;

    if (lane < 8) s_ctl[lane] = 0;
    if (lane == 0) {
        ModS z = {};
        s_m = z;
        s_snr[0] = 0.0f;
        s_snr[1] = 0.0f;
    }
    __syncthreads();

    c32 tw[5];
    fft_twiddles(tw, lane);
    const float ltfl = c_tab.ltf_l[bin];
    const float ltfn = c_tab.ltf_nl[bin];
    const bool lused = ltfl != 0.0f;
    const bool nused = ltfn != 0.0f;
    const int demap = c_tab.lsig_demap[bin];
    c32 hl = mk(0.0f, 0.0f);
    int nvalid = 0;
    int pos = 0;

    if (navail < 224 || (after_signal && (!sig || !chan_in))) {
        if (lane == 0) s_ctl[0] = GR80211_ST_DROP_LSIG;
        __syncthreads();
        goto finish;
    }

    if (after_signal) {
        // what `signal` found travels in the tags (lib/demod_impl.cc:75-100): L-SIG rate and length, the sample
        // count and the L-LTF channel; fields a correct `signal` cannot have sent are refused
        const gr80211_sig_desc sd = sig[f];
        if (lused) hl = chan_in[(size_t)f * 64 + bin];
        if (lane == 0) {
            const int ndbps_l[8] = {24, 36, 48, 72, 96, 144, 192, 216};
            if (sd.mcs < 0 || sd.mcs > 7 || sd.len < 1 || sd.len > 4095 ||
                sd.nsamp != ((sd.len * 8 + 22 + ndbps_l[sd.mcs & 7] - 1) / ndbps_l[sd.mcs & 7]) * 80) {
                s_ctl[0] = GR80211_ST_DROP_LSIG;
            } else {
                s_ctl[1] = sd.mcs;
                s_ctl[2] = sd.len;
                s_ctl[3] = sd.nsamp;
            }
        }
        __syncthreads();
        if (s_ctl[0] != GR80211_ST_OK) goto finish;
    } else
    // ---- signal: L-LTF x2 + L-SIG (lib/signal_impl.cc:108-162, lib/cloud80211phy.cc:609-627)
    {
        c32 x1 = in0[SHIFT + lane], x2 = in0[SHIFT + 64 + lane], xs = in0[SHIFT + 144 + lane];
        if (rad != 0.0f) {
            float sn, cs;
            sincosf((lane + SHIFT) * rad, &sn, &cs);
            x1 = cmul(x1, mk(cs, sn));
            sincosf((lane + SHIFT + 64) * rad, &sn, &cs);
            x2 = cmul(x2, mk(cs, sn));
            sincosf((lane + SHIFT + 144) * rad, &sn, &cs);
            xs = cmul(xs, mk(cs, sn));
        }
        const c32 S1 = wave_fft64_tw(x1, lane, tw), S2 = wave_fft64_tw(x2, lane, tw), SG = wave_fft64_tw(xs, lane, tw);
        c32 e = mk(0.0f, 0.0f);
        if (lused) {
            hl = cdivf(cadd(S1, S2), ltfl * 2.0f);
            e = cdiv(SG, hl);
        }
        const c32 e7 = lane_c<LN7>(e), e21 = lane_c<LN21>(e), e43 = lane_c<LN43>(e), e57 = lane_c<LN57>(e);
        const c32 ps = cconj(cadd(cadd(csub(e7, e21), e43), e57));
        const float pa = cabs32(ps);
        if (demap >= 0) s_llr[0][demap] = cdivf(cmul(e, ps), pa).x;
        const SigBits lsig = {wave_sig_viterbi(s_llr[0], 24, lane)};
        if (lane == 0) {
            int mcs = 0, len = 0, ndbps = 24;
            if (!check_legacy(lsig, &mcs, &len, &ndbps)) {
                s_ctl[0] = GR80211_ST_DROP_LSIG;
            } else {
                s_ctl[1] = mcs;
                s_ctl[2] = len;
                s_ctl[3] = ((len * 8 + 22 + ndbps - 1) / ndbps) * 80;
            }
        }
        __syncthreads();
        if (s_ctl[0] != GR80211_ST_OK) goto finish;
    }
    if (chanl_out) chanl_out[(size_t)f * 64 + bin] = hl;  // the "chan" tag of `signal` (lib/signal_impl.cc:143)
    nvalid = s_ctl[3];
    if (nvalid > navail - 224) nvalid = navail - 224;

    // ---- format detection (lib/demod_impl.cc:106-148, lib/cloud80211phy.cc:629-648)
    if (s_ctl[1] == 0) {
        const c32 F1 = wave_fft64_tw(load_samp(in0, SHIFT + lane, nvalid, rad), lane, tw);
        const c32 F2 = wave_fft64_tw(load_samp(in0, 80 + SHIFT + lane, nvalid, rad), lane, tw);
        c32 e1 = mk(0.0f, 0.0f), e2 = mk(0.0f, 0.0f);
        if (lused) {
            e1 = cdiv(F1, hl);
            e2 = cdiv(F2, hl);
        }
        const c32 p1 = cconj(cadd(cadd(csub(lane_c<LN7>(e1), lane_c<LN21>(e1)), lane_c<LN43>(e1)), lane_c<LN57>(e1)));
        const c32 p2 = cconj(cadd(cadd(csub(lane_c<LN7>(e2), lane_c<LN21>(e2)), lane_c<LN43>(e2)), lane_c<LN57>(e2)));
        const float a1 = cabs32(p1), a2 = cabs32(p2);
        if (demap >= 0) {
            const c32 m1 = cdivf(cmul(e1, p1), a1), m2 = cdivf(cmul(e2, p2), a2);
            s_llr[0][demap] = m1.x;       // VHT-SIG-A: BPSK then QBPSK
            s_llr[0][demap + 48] = m2.y;
            s_llr[1][demap] = m1.y;       // HT-SIG: QBPSK twice
            s_llr[1][demap + 48] = m2.y;
        }
        const SigBits siga = {wave_sig_viterbi(s_llr[0], 48, lane)};
        if (lane == 0) {
            if (check_vhta(siga)) {
                s_ctl[4] = GR80211_F_VHT;
                const int kind = parse_vhta(siga, &s_m);
                s_ctl[7] = kind == 2;
                // MU frames are the one-antenna block's business (lib/demod_impl.cc:347-379); demod2 has no such branch
                if (kind == 0 || (kind == 2 && two_ant)) s_ctl[0] = GR80211_ST_DROP_UNSUPPORTED;
            } else {
                s_ctl[4] = -1;
            }
        }
        __syncthreads();
        if (s_ctl[0] != GR80211_ST_OK) goto finish;
        if (s_ctl[4] == -1) {
            const SigBits htsig = {wave_sig_viterbi(s_llr[1], 48, lane)};
            if (lane == 0) {
                if (check_ht(htsig)) {
                    s_ctl[4] = GR80211_F_HT;
                    parse_ht(htsig, &s_m);
                } else {
                    s_ctl[4] = GR80211_F_L;
                }
            }
            __syncthreads();
        }
        if (s_ctl[4] != GR80211_F_L) pos = 160;
    }

    if (s_ctl[4] == GR80211_F_L) {
        // lib/demod_impl.cc:207-219
        if (lane == 0) {
            parse_legacy(s_ctl[1], s_ctl[2], &s_m);
            s_ctl[5] = s_m.len * 8 + 22;
            s_ctl[6] = 0;
        }
        chan1[(size_t)f * 64 + bin] = hl;
        __syncthreads();
        goto finish;
    }

    // ---- HT / VHT: channel estimate (+ VHT-SIG-B) -----------------------------------------
    {
        const int fmt = s_ctl[4];
        const int nltf = s_m.nLTF;
        // A two-stream VHT frame on one antenna is only followed as far as VHT-SIG-B, with the channel of the first
        // VHT-LTF, to see whether it is a sounding NDP (lib/demod_impl.cc:393-409,452-455).
        const bool sounding = s_m.nSS == 2 && !two_ant && fmt == GR80211_F_VHT;
        if (s_m.nSS > 2 || (s_m.nSS == 2 && !two_ant && !sounding)) {
            if (lane == 0) s_ctl[0] = GR80211_ST_DROP_UNSUPPORTED;
            __syncthreads();
            goto finish;
        }
        const int nss = sounding ? 1 : s_m.nSS;  // streams the equaliser separates
        const int off = pos + 80;
        c32 h[4], hi[4];
        c32 hn = mk(0.0f, 0.0f);
        c32 pl1 = mk(0.0f, 0.0f), pl2 = mk(0.0f, 0.0f);  // LTF-derived pilot references (pilot lanes)
        const bool mu = s_ctl[7] != 0;
        if (mu) {
            // this station's stream out of the two VHT-LTFs of a two-user MU frame (lib/demod_impl.cc:347-379);
            // the second LTF's sequence has its pilot tones negated (LTF_NL_28_F_FLOAT_VHT22)
            const c32 F1 = wave_fft64_tw(load_samp(in0, off + SHIFT + lane, nvalid, rad), lane, tw);
            const c32 F2 = wave_fft64_tw(load_samp(in0, off + 80 + SHIFT + lane, nvalid, rad), lane, tw);
            if (nused) {
                if (d.mu_pos == 0) hn = cdivf(csub(F1, F2), ltfn * 2.0f);
                else hn = cdivf(cadd(cdivf(F1, ltfn), cdivf(F2, is_pilot_bin(bin) ? -ltfn : ltfn)), 2.0f);
            }
            chan1[(size_t)f * 64 + bin] = hn;
        } else if (nss == 1) {
            // lib/demod_impl.cc:380-392
            const c32 F1 = wave_fft64_tw(load_samp(in0, off + SHIFT + lane, nvalid, rad), lane, tw);
            if (nused) hn = cdivf(F1, ltfn);
            chan1[(size_t)f * 64 + bin] = hn;
        } else {
            // lib/demod2_impl.cc:372-463
            const c32 F1 = wave_fft64_tw(load_samp(in0, off + SHIFT + lane, nvalid, rad), lane, tw);
            const c32 F2 = wave_fft64_tw(load_samp(in1, off + SHIFT + lane, nvalid, rad), lane, tw);
            const c32 F12 = wave_fft64_tw(load_samp(in0, off + 80 + SHIFT + lane, nvalid, rad), lane, tw);
            const c32 F22 = wave_fft64_tw(load_samp(in1, off + 80 + SHIFT + lane, nvalid, rad), lane, tw);
            const float l2 = ltfn * 0.5f;
            h[0] = cmulf(csub(F1, F12), l2);
            h[1] = cmulf(csub(F2, F22), l2);
            h[2] = cmulf(cadd(F1, F12), l2);
            h[3] = cmulf(cadd(F2, F22), l2);
            if (fmt == GR80211_F_VHT) {
                const int lm = brev6((bin + 63) & 63), lp = brev6((bin + 1) & 63);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const c32 a = shfl_c(h[q], lm), b = shfl_c(h[q], lp);
                    if (is_pilot_bin(bin)) h[q] = cdivf(cadd(a, b), 2.0f);
                }
            }
            if (nused) {
                const c32 a = cadd(cmul(h[0], cconj(h[0])), cmul(h[1], cconj(h[1])));
                const c32 b = cadd(cmul(h[0], cconj(h[2])), cmul(h[1], cconj(h[3])));
                const c32 c = cadd(cmul(h[2], cconj(h[0])), cmul(h[3], cconj(h[1])));
                const c32 dd = cadd(cmul(h[2], cconj(h[2])), cmul(h[3], cconj(h[3])));
                const c32 adbc = cdiv(mk(1.0f, 0.0f), csub(cmul(a, dd), cmul(b, c)));
                hi[0] = cmul(adbc, dd);
                hi[1] = cmul(cneg(adbc), b);
                hi[2] = cmul(cneg(adbc), c);
                hi[3] = cmul(adbc, a);
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) { h[q] = mk(0.0f, 0.0f); hi[q] = mk(0.0f, 0.0f); }
            }
            c32 *c2 = chan2 + (size_t)f * CH2;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                c2[bin * 4 + q] = h[q];
                c2[256 + bin * 4 + q] = hi[q];
            }
            if (is_pilot_bin(bin)) {
                c32 e1, e2;
                zf2(F1, F2, h, hi, e1, e2);
                if (bin == 57) { e1 = cneg(e1); e2 = cneg(e2); }
                pl1 = cconj(e1);
                pl2 = cconj(e2);
                const int slot = bin == 43 ? 0 : bin == 57 ? 1 : bin == 7 ? 2 : 3;
                c2[512 + slot] = pl1;
                c2[516 + slot] = pl2;
            }
        }

        const int nleg = (s_ctl[2] * 8 + 22 + 23) / 24;
        if (fmt == GR80211_F_VHT) {
            // VHT-SIG-B (lib/demod_impl.cc:449-505, lib/demod2_impl.cc:632-758)
            const int ob = off + nltf * 80;
            c32 q0 = mk(0.0f, 0.0f), q1 = mk(0.0f, 0.0f);
            const int jn = c_tab.data_idx_nl[bin];
            float inted = 0.0f;
            if (nss == 1) {
                const c32 F1 = wave_fft64_tw(load_samp(in0, ob + SHIFT + lane, nvalid, rad), lane, tw);
                c32 e = mk(0.0f, 0.0f);
                if (nused) e = cdiv(F1, hn);
                const c32 ps = cconj(cadd(cadd(csub(lane_c<LN7>(e), lane_c<LN21>(e)), lane_c<LN43>(e)), lane_c<LN57>(e)));
                const float pa = cabs32(ps);
                q0 = cdivf(cmul(e, ps), pa);
                inted = q0.x;
            } else {
                const c32 F1 = wave_fft64_tw(load_samp(in0, ob + SHIFT + lane, nvalid, rad), lane, tw);
                const c32 F2 = wave_fft64_tw(load_samp(in1, ob + SHIFT + lane, nvalid, rad), lane, tw);
                c32 s1, s2;
                zf2(F1, F2, h, hi, s1, s2);
                const c32 u1 = cmul(s1, pl1), u2 = cmul(s2, pl2);  // meaningful on pilot lanes
                c32 acc = lane_c<LN7>(u1);
                acc = csub(acc, lane_c<LN21>(u1));
                acc = cadd(acc, lane_c<LN43>(u1));
                acc = cadd(acc, lane_c<LN57>(u1));
                acc = cadd(acc, lane_c<LN7>(u2));
                acc = csub(acc, lane_c<LN21>(u2));
                acc = cadd(acc, lane_c<LN43>(u2));
                acc = cadd(acc, lane_c<LN57>(u2));
                const c32 ps = cconj(acc);
                const float pa = cabs32(ps);
                q0 = cdivf(cmul(s1, ps), pa);
                q1 = cdivf(cmul(s2, ps), pa);
                inted = (q0.x + q1.x) / 2.0f;
            }
            if (jn >= 0) s_llr[0][c_tab.deint_nl[0][jn]] = inted;
            const SigBits sigb = {wave_sig_viterbi(s_llr[0], 26, lane)};
            // per-stream SNR from the re-encoded SIG-B (float64 accumulation as in the reference)
            double n0 = 0.0, n1 = 0.0;
            if (jn >= 0) {
                const int k = c_tab.deint_nl[0][jn];  // coded bit carried by this tone
                const int i = k >> 1;
                // the encoder's seven-bit window ending at input bit i (bits before the first read as 0), oldest bit
                // lowest: the generators 0155 / 0117, which count the newest bit lowest, turn into 0133 / 0171
                const unsigned win = (unsigned)((sigb.w << 6) >> i) & 0x7Fu;
                const int cb = (k & 1) ? parity7(win & 0171) : parity7(win & 0133);
                const float ref = cb ? 1.0f : -1.0f;
                const c32 r0 = mk(q0.x - ref, q0.y - 0.0f);
                n0 = (double)(r0.x * r0.x + r0.y * r0.y);
                if (nss == 2) {
                    const c32 r1 = mk(q1.x - ref, q1.y - 0.0f);
                    n1 = (double)(r1.x * r1.x + r1.y * r1.y);
                }
            }
            // ordered sum over data tones 0..51 (the reference's accumulation order)
            s_red[jn >= 0 ? jn : 52 + (lane % 12)] = n0;
            __syncthreads();
            if (lane == 0) {
                double acc = 0.0;
                for (int i = 0; i < 52; i++) acc += s_red[i];
                s_snr[0] = (float)(log10(52.0 / acc) * 10.0);
            }
            __syncthreads();
            if (nss == 2) {
                s_red[jn >= 0 ? jn : 52 + (lane % 12)] = n1;
                __syncthreads();
                if (lane == 0) {
                    double acc = 0.0;
                    for (int i = 0; i < 52; i++) acc += s_red[i];
                    s_snr[1] = (float)(log10(52.0 / acc) * 10.0);
                }
                __syncthreads();
            }
            if (lane == 0) {
                bool known = true;
                if (mu) known = parse_vhtb_mu(sigb, &s_m);
                else parse_vhtb(sigb, &s_m);
                const ModS m = s_m;
                if (!known) {
                    s_ctl[0] = GR80211_ST_DROP_UNSUPPORTED;
                } else if (m.len == 0 && m.nSym == 0) {
                    // NDP: one antenna reports the raw VHT-LTF samples (lib/demod_impl.cc:238-249,265-271);
                    // the two-antenna block requires len > 0 and drops it (lib/demod2_impl.cc:159)
                    s_ctl[0] = two_ant ? GR80211_ST_DROP_LEN_CHECK : GR80211_ST_NDP;
                    s_ctl[6] = pos + 80;
                } else if (sounding) {
                    s_ctl[0] = GR80211_ST_DROP_UNSUPPORTED;  // two streams of data on one antenna
                } else if (!(m.len > 0 && m.len <= 4095 && m.nSS <= 2 &&
                             (nleg * 80) >= (m.nSym * m.nSymSamp + 160 + 80 + m.nLTF * 80 + 80))) {
                    s_ctl[0] = GR80211_ST_DROP_LEN_CHECK;
                } else {
                    s_ctl[5] = m.nSym * m.nDBPS;
                    s_ctl[6] = pos + 80 + m.nLTF * 80 + 80;
                }
            }
        } else {
            if (lane == 0) {
                const ModS m = s_m;
                if (!(m.len > 0 && m.len <= 4095 && m.nSS <= 2 &&
                      (nleg * 80) >= (m.nSym * m.nSymSamp + 160 + 80 + m.nLTF * 80))) {
                    s_ctl[0] = GR80211_ST_DROP_LEN_CHECK;
                } else {
                    s_ctl[5] = m.len * 8 + 22;
                    s_ctl[6] = pos + 80 + m.nLTF * 80;
                }
            }
        }
        __syncthreads();
    }

finish:
    if (lane == 0) {
        gr80211_frame_info fi;
        const ModS m = s_m;
        fi.status = s_ctl[0];
        fi.lsig_mcs = s_ctl[1];
        fi.lsig_len = s_ctl[2];
        fi.nsamp = s_ctl[3];
        fi.format = m.format; fi.mcs = m.mcs; fi.len = m.len; fi.cr = m.cr; fi.ampdu = m.ampdu;
        fi.trellis = s_ctl[5];
        fi.total = s_ctl[0] == GR80211_ST_OK ? m.nSym * m.nCBPS : 0;
        fi.nSym = m.nSym; fi.nSS = m.nSS; fi.nLTF = m.nLTF; fi.nCBPS = m.nCBPS; fi.nDBPS = m.nDBPS;
        fi.nBPSCS = m.nBPSCS; fi.mod = m.mod;
        fi.sssnr0 = s_snr[0]; fi.sssnr1 = s_snr[1];
        // rad * 20e6 / 2pi (lib/signal_impl.cc:136); `demod` hands the tag on as it came (lib/demod_impl.cc:82,225)
        fi.cfo_hz = after_signal && sig ? sig[f].cfo : rad * 3183098.8618379068f;
        fi.snr = d.snr; fi.rssi = d.rssi;
        fi.slot = -1;
        fi.llr_off = 0;
        fi.data_off = s_ctl[6];
        fi.nSymSamp = m.nSymSamp;
        info[f] = fi;
    }
}

// =========================================================================================
// k_desc_to_info: decode-only entry, tags -> frame records
// =========================================================================================
__global__ void k_desc_to_info(const gr80211_dec_desc *__restrict__ desc, int n, unsigned long long llr_floats,
                               gr80211_frame_info *__restrict__ info)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gr80211_dec_desc d = desc[i];
    gr80211_frame_info fi = {};
    fi.status = GR80211_ST_OK;
    fi.format = d.format; fi.len = d.len; fi.cr = d.cr; fi.mcs = d.mcs; fi.ampdu = d.ampdu;
    fi.trellis = d.trellis; fi.total = d.n_llr; fi.llr_off = d.llr_off; fi.slot = -1;
    if (d.cr < 0 || d.cr > 3 || d.n_llr < 0 || d.llr_off < 0 ||
        (unsigned long long)d.llr_off + (unsigned long long)d.n_llr > llr_floats) {  // soft bits outside the arena
        fi.status = GR80211_ST_DROP_UNSUPPORTED;
    } else if (d.trellis > 0 && d.trellis <= GR80211_MAX_TRELLIS) {
        // soft bits the trellis consumes; with fewer the reference block would wait for more input
        // and never finish the packet (lib/decode_impl.cc:210)
        const int P = d.cr == CR_12 ? 1 : d.cr == CR_23 ? 2 : d.cr == CR_34 ? 3 : 5;
        const int q = d.trellis / P, rem = d.trellis % P;
        const int need = q * (P + 1) + (rem ? rem + 1 : 0);
        if (d.n_llr < need) fi.status = GR80211_ST_DROP_DECODE_LIMIT;
    }
    info[i] = fi;
}

// =========================================================================================
// scheduler: decode admission, counting sort by (code rate, trellis/64) into groups of 64 lanes,
// LLR arena offsets.  Four small kernels: histogram -> scan -> assign -> group offsets.
// =========================================================================================
__device__ __forceinline__ uint32_t round64(uint32_t v) { return (v + 63u) & ~63u; }

constexpr int SCHED_BINS = 4 * SCHED_NB;
constexpr int SCHED_BLOCK = 256;

// ws layout (uint32): [0, SCHED_BINS) histogram -> bin cursors; then crstart[5]; then n_valid.
__device__ __forceinline__ bool llr_eligible(int st) { return st == GR80211_ST_OK || st == GR80211_ST_DROP_DECODE_LIMIT; }

__global__ __launch_bounds__(SCHED_BLOCK) void k_sched_hist(gr80211_frame_info *__restrict__ info, int n,
                                                            uint32_t *__restrict__ ws,
                                                            unsigned long long *__restrict__ blk_llr)
{
    __shared__ unsigned long long red[SCHED_BLOCK];
    __shared__ uint32_t lh[SCHED_BINS];  // block-private histogram: batches of equal frames hit one bin
    for (int k = threadIdx.x; k < SCHED_BINS; k += SCHED_BLOCK) lh[k] = 0;
    __syncthreads();
    const int i = blockIdx.x * SCHED_BLOCK + threadIdx.x;
    unsigned long long mine = 0;
    if (i < n) {
        int st = info[i].status;
        const int tr = info[i].trellis, len = info[i].len, cr = info[i].cr;
        // decode admission (lib/decode_impl.cc:93)
        if (st == GR80211_ST_OK && (len > GR80211_MAX_PSDU || tr > GR80211_MAX_TRELLIS || tr <= 0 || cr < 0 || cr > 3)) {
            st = GR80211_ST_DROP_DECODE_LIMIT;
            info[i].status = st;
        }
        if (st == GR80211_ST_OK) atomicAdd(&lh[cr * SCHED_NB + tr / 64], 1u);
        if (llr_eligible(st)) mine = (unsigned long long)info[i].total;
    }
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int k = threadIdx.x; k < SCHED_BINS; k += SCHED_BLOCK)
        if (lh[k]) atomicAdd(&ws[k], lh[k]);
    for (int o = SCHED_BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) blk_llr[blockIdx.x] = red[0];
}

// Single workgroup: exclusive scans of the bin histogram (code-rate starts aligned to 64 lanes) and of
// the per-block LLR sums.
__global__ __launch_bounds__(1024) void k_sched_scan(uint32_t *__restrict__ ws, unsigned long long *__restrict__ blk_llr,
                                                     int nblk, int assign_llr, int slot_cap, int group_cap,
                                                     int32_t *__restrict__ slot_frame, GroupInfo *__restrict__ groups,
                                                     SchedTotals *__restrict__ totals, unsigned long long llr_cap)
{
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t binpart[1024];
    __shared__ uint32_t crs[5];
    __shared__ int s_over;
    const int tid = threadIdx.x;
    // --- LLR block sums
    const int per = (nblk + 1023) / 1024;
    const int b0 = tid * per < nblk ? tid * per : nblk;
    const int b1 = b0 + per < nblk ? b0 + per : nblk;
    unsigned long long sum = 0;
    for (int b = b0; b < b1; b++) sum += blk_llr[b];
    part[tid] = sum;
    // --- bins: thread t owns bins [3t, 3t+3) of the 2052 (684 threads)
    uint32_t bsum = 0;
    for (int k = 0; k < 3; k++) {
        const int b = tid * 3 + k;
        if (b < SCHED_BINS) bsum += ws[b];
    }
    binpart[tid] = bsum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 1024; k++) { const unsigned long long v = part[k]; part[k] = run; run += v; }
        totals->llr_floats = assign_llr ? run : 0;
        s_over = assign_llr && run > llr_cap;  // more soft bits than the caller has room for
        // SCHED_NB = 513 bins per rate = 171 threads per rate
        uint32_t p = 0, cnt = 0;
        for (int cr = 0; cr < 4; cr++) {
            p = round64(p);
            crs[cr] = p;
            for (int k = cr * 171; k < (cr + 1) * 171; k++) { const uint32_t v = binpart[k]; binpart[k] = p; p += v; cnt += v; }
        }
        crs[4] = round64(p);
        totals->n_valid = (int)cnt;
    }
    __syncthreads();
    {
        unsigned long long run = part[tid];
        for (int b = b0; b < b1; b++) { const unsigned long long v = blk_llr[b]; blk_llr[b] = run; run += v; }
        uint32_t p = binpart[tid];
        for (int k = 0; k < 3; k++) {
            const int b = tid * 3 + k;
            if (b < SCHED_BINS) { const uint32_t v = ws[b]; ws[b] = p; p += v; }
        }
        if (tid < 5) ws[SCHED_BINS + tid] = crs[tid];
    }
    const int n_slots = (int)crs[4];
    const int n_groups = n_slots / 64;
    const bool refuse = n_slots > slot_cap || n_groups > group_cap || s_over;
    if (tid == 0) totals->n_groups = refuse ? -1 : n_groups;
    if (refuse) return;
    for (int sl = tid; sl < n_slots; sl += 1024) slot_frame[sl] = -1;
    for (int g = tid; g < n_groups; g += 1024) {
        const uint32_t s0 = (uint32_t)g * 64u;
        GroupInfo gi;
        gi.cr = s0 >= crs[3] ? 3 : s0 >= crs[2] ? 2 : s0 >= crs[1] ? 1 : 0;
        gi.steps = 0; gi.nchunks = 0; gi.words = 0; gi.dec_off = 0; gi.bits_off = 0;
        groups[g] = gi;
    }
}

__global__ __launch_bounds__(SCHED_BLOCK) void k_sched_assign(gr80211_frame_info *__restrict__ info, int n, int assign_llr,
                                                              uint32_t *__restrict__ ws,
                                                              const unsigned long long *__restrict__ blk_llr,
                                                              int32_t *__restrict__ slot_frame,
                                                              GroupInfo *__restrict__ groups,
                                                              const SchedTotals *__restrict__ totals)
{
    __shared__ unsigned long long sc[SCHED_BLOCK];
    __shared__ uint32_t lh[SCHED_BINS];
    __shared__ uint32_t lbase[SCHED_BINS];
    if (totals->n_groups < 0) return;
    const int i = blockIdx.x * SCHED_BLOCK + threadIdx.x;
    int st = GR80211_ST_DROP_LSIG;
    unsigned long long mine = 0;
    if (i < n) {
        st = info[i].status;
        if (llr_eligible(st)) mine = (unsigned long long)info[i].total;
    }
    // block-level exclusive scan keeps the LLR runs in frame order
    sc[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < SCHED_BLOCK; o <<= 1) {
        unsigned long long v = 0;
        if ((int)threadIdx.x >= o) v = sc[threadIdx.x - o];
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    // slots: rank inside the block through a block-private histogram, one global atomic per used bin
    for (int k = threadIdx.x; k < SCHED_BINS; k += SCHED_BLOCK) lh[k] = 0;
    __syncthreads();
    int bin = -1;
    uint32_t rank = 0;
    if (i < n && st == GR80211_ST_OK) {
        bin = info[i].cr * SCHED_NB + info[i].trellis / 64;
        rank = atomicAdd(&lh[bin], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SCHED_BINS; k += SCHED_BLOCK) {
        const uint32_t c = lh[k];
        if (c) lbase[k] = atomicAdd(&ws[k], c);
    }
    __syncthreads();
    if (i >= n) return;
    if (assign_llr && llr_eligible(st)) info[i].llr_off = (int64_t)(blk_llr[blockIdx.x] + sc[threadIdx.x] - mine);
    if (bin >= 0) {
        const uint32_t slot = lbase[bin] + rank;
        info[i].slot = (int32_t)slot;
        slot_frame[slot] = i;
        atomicMax(&groups[slot / 64].steps, info[i].trellis);
    } else {
        info[i].slot = -1;
    }
}

__global__ __launch_bounds__(1024) void k_sched_groups(GroupInfo *__restrict__ groups, SchedTotals *__restrict__ totals,
                                                       unsigned long long dec_cap, unsigned long long bits_cap)
{
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long part2[1024];
    const int n_groups = totals->n_groups;
    const int tid = threadIdx.x;
    if (n_groups < 0) return;
    const int gper = (n_groups + 1023) / 1024;
    const int g0 = tid * gper < n_groups ? tid * gper : n_groups;
    const int g1 = g0 + gper < n_groups ? g0 + gper : n_groups;
    unsigned long long dsum = 0, bsum = 0;
    for (int g = g0; g < g1; g++) {
        const int cs = chunk_steps(groups[g].cr);
        const int nch = (groups[g].steps + cs - 1) / cs;
        const int steps = nch * cs;
        groups[g].nchunks = nch;
        groups[g].steps = steps;
        groups[g].words = (steps + 31) / 32;
        dsum += (unsigned long long)steps * 128ull;
        bsum += (unsigned long long)((steps + 31) / 32) * 64ull;
    }
    part[tid] = dsum;
    part2[tid] = bsum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long r1 = 0, r2 = 0;
        for (int k = 0; k < 1024; k++) {
            const unsigned long long v1 = part[k], v2 = part2[k];
            part[k] = r1; part2[k] = r2;
            r1 += v1; r2 += v2;
        }
        totals->dec_words = r1;
        totals->bits_words = r2;
        if (r1 > dec_cap || r2 > bits_cap) totals->n_groups = -1;  // no room for the survivor or bit words: not run
    }
    __syncthreads();
    unsigned long long r1 = part[tid], r2 = part2[tid];
    for (int g = g0; g < g1; g++) {
        groups[g].dec_off = r1;
        groups[g].bits_off = r2;
        r1 += (unsigned long long)groups[g].steps * 128ull;
        r2 += (unsigned long long)groups[g].words * 64ull;
    }
}

// The whole scheduler in one workgroup, for batches of at most 4096 frames: what the four kernels above do, with the
// histogram, the scans and the group records in LDS; a thread takes a run of consecutive frames.  A small batch is a
// matter of latency, and five launches of tiny kernels cost a single frame a sixth of its way through the decoder
// (0.060 -> 0.010 ms; 2000 frames 0.065 -> 0.031, 4000 frames 0.065 -> 0.052; from 8000 frames on one workgroup is the
// slower way, 0.096 ms).  Same admission rule, same layout (frames of a bin in whatever order their threads' atomics
// land, as above), same totals and refusals.
constexpr int SCHED_ONE_MAX = 4096;
constexpr int SCHED_ONE_GROUPS = 96;  // 4096 frames in groups of 64, every code rate's run rounded up: at most 67
__global__ __launch_bounds__(1024) void k_sched_one(gr80211_frame_info *__restrict__ info, int n, int assign_llr,
                                                    int32_t *__restrict__ slot_frame, int slot_cap,
                                                    GroupInfo *__restrict__ groups, int group_cap,
                                                    SchedTotals *__restrict__ totals, unsigned long long llr_cap,
                                                    unsigned long long dec_cap, unsigned long long bits_cap)
{
    __shared__ uint32_t bins[SCHED_BINS];
    __shared__ unsigned long long sc[1024];
    __shared__ uint32_t part[1024];
    __shared__ uint32_t crs[5];
    __shared__ int gsteps[SCHED_ONE_GROUPS];
    __shared__ int s_refuse;
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;  // frames per thread: thread t has frames [t * per, t * per + per)
    const int f0 = tid * per < n ? tid * per : n, f1 = f0 + per < n ? f0 + per : n;
    for (int k = tid; k < SCHED_BINS; k += 1024) bins[k] = 0;
    for (int k = tid; k < SCHED_ONE_GROUPS; k += 1024) gsteps[k] = 0;
    __syncthreads();
    // decode admission (lib/decode_impl.cc:93), histogram, rank inside the bin (parked in the frame's slot field)
    unsigned long long mine = 0;
    for (int i = f0; i < f1; i++) {
        int st = info[i].status;
        const int tr = info[i].trellis, len = info[i].len, cr = info[i].cr;
        if (st == GR80211_ST_OK && (len > GR80211_MAX_PSDU || tr > GR80211_MAX_TRELLIS || tr <= 0 || cr < 0 || cr > 3)) {
            st = GR80211_ST_DROP_DECODE_LIMIT;
            info[i].status = st;
        }
        info[i].slot = st == GR80211_ST_OK ? (int32_t)atomicAdd(&bins[cr * SCHED_NB + tr / 64], 1u) : -1;
        if (llr_eligible(st)) mine += (unsigned long long)info[i].total;
    }
    sc[tid] = mine;
    __syncthreads();
    // soft-bit runs in frame order: inclusive scan over the threads' sums
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned long long v = tid >= o ? sc[tid - o] : 0;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    // bins: thread t owns bins [3t, 3t + 3) (171 threads per code rate); inclusive scan over the threads' sums
    uint32_t b3[3], bsum = 0;
    for (int k = 0; k < 3; k++) {
        const int b = tid * 3 + k;
        b3[k] = b < SCHED_BINS ? bins[b] : 0;
        bsum += b3[k];
    }
    part[tid] = bsum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t p = 0, cnt = 0;
        for (int cr = 0; cr < 4; cr++) {
            p = round64(p);
            crs[cr] = p;
            const uint32_t t = part[171 * (cr + 1) - 1] - (cr ? part[171 * cr - 1] : 0u);
            p += t;
            cnt += t;
        }
        crs[4] = round64(p);
        const unsigned long long llr_total = sc[1023];
        totals->n_valid = (int)cnt;
        totals->llr_floats = assign_llr ? llr_total : 0;
        const int n_slots = (int)crs[4];
        s_refuse = n_slots > slot_cap || n_slots / 64 > group_cap || n_slots / 64 > SCHED_ONE_GROUPS ||
                   (assign_llr && llr_total > llr_cap);
    }
    __syncthreads();
    if (tid < 684) {
        const int rate = tid / 171;
        uint32_t p = crs[rate] + part[tid] - bsum - (rate ? part[171 * rate - 1] : 0u);
        for (int k = 0; k < 3; k++) {
            const int b = tid * 3 + k;
            if (b < SCHED_BINS) {
                bins[b] = p;
                p += b3[k];
            }
        }
    }
    const int n_slots = (int)crs[4];
    const int n_groups = n_slots / 64;
    if (s_refuse) {
        if (tid == 0) totals->n_groups = -1;
        return;
    }
    for (int sl = tid; sl < n_slots; sl += 1024) slot_frame[sl] = -1;
    __syncthreads();  // the bins' starts and the cleared slots are in place
    unsigned long long run = sc[tid] - mine;
    for (int i = f0; i < f1; i++) {
        const int st = info[i].status;
        if (llr_eligible(st)) {
            if (assign_llr) info[i].llr_off = (int64_t)run;
            run += (unsigned long long)info[i].total;
        }
        if (st == GR80211_ST_OK) {
            const int tr = info[i].trellis;
            const uint32_t slot = bins[info[i].cr * SCHED_NB + tr / 64] + (uint32_t)info[i].slot;
            info[i].slot = (int32_t)slot;
            slot_frame[slot] = i;
            atomicMax(&gsteps[slot / 64], tr);
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long r1 = 0, r2 = 0;
        for (int g = 0; g < n_groups; g++) {
            const uint32_t s0 = (uint32_t)g * 64u;
            GroupInfo gi;
            gi.cr = s0 >= crs[3] ? 3 : s0 >= crs[2] ? 2 : s0 >= crs[1] ? 1 : 0;
            const int cs = chunk_steps(gi.cr);
            gi.nchunks = (gsteps[g] + cs - 1) / cs;
            gi.steps = gi.nchunks * cs;
            gi.words = (gi.steps + 31) / 32;
            gi.dec_off = r1;
            gi.bits_off = r2;
            groups[g] = gi;
            r1 += (unsigned long long)gi.steps * 128ull;
            r2 += (unsigned long long)gi.words * 64ull;
        }
        totals->dec_words = r1;
        totals->bits_words = r2;
        totals->n_groups = (r1 > dec_cap || r2 > bits_cap) ? -1 : n_groups;  // no room for the survivor or bit words: not run
    }
}

__global__ void k_zero_u32(uint32_t *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// =========================================================================================
// k_demod: one workgroup (4 waves) per frame, each wave strides over OFDM symbols
// =========================================================================================
// Max-log soft demap of one axis after constellation scaling (lib/cloud80211phy.cc:2090-2148):
// o0 = v, o1 = E - |v|, o2 = E/2 - |o1|, o3 = 2 - |o2| with E = 2, 4, 8 for 2, 3, 4 bits per axis.
__device__ __forceinline__ void axis_llr(float v, int nb, float &o0, float &o1, float &o2, float &o3)
{
    const float edge = nb == 2 ? 2.0f : nb == 3 ? 4.0f : 8.0f;
    o0 = v;
    o1 = edge - fabsf(v);
    o2 = edge * 0.5f - fabsf(o1);
    o3 = 2.0f - fabsf(o2);
}

__device__ __forceinline__ float qam_scale(int mod)
{
    return mod == QAM_QPSK ? 1.4142135623730951f : mod == QAM_16 ? 3.1622776601683795f
         : mod == QAM_64 ? 6.48074069840786f : mod == QAM_256 ? 13.038404810405298f : 1.0f;
}

constexpr int DEMOD_WAVES = 4;
constexpr int DEMOD_STAGE = 832;  // soft bits of one OFDM symbol: 2 streams x 52 tones x 8 bits

// Places of one tone's soft bits in the symbol's LDS stage after deinterleaving (and, for two
// streams, stream de-parsing, lib/cloud80211phy.cc:2442-2451): kI[i] for bit i of the I axis,
// kQ[i] for bit i of the Q axis.  They do not change from symbol to symbol.
__device__ __forceinline__ void tone_slots(int (&kI)[4], int (&kQ)[4], const int16_t *map, int jd, int nb, int nax,
                                           int ss, int sblk, bool two)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int vi = 0, vq = 0;
        if (jd >= 0 && i < nax) {
            vi = map[jd * nb + i];
            if (two) vi = (vi / sblk) * 2 * sblk + ss * sblk + (vi % sblk);
            if (nb >= 2) {
                vq = map[jd * nb + nax + i];
                if (two) vq = (vq / sblk) * 2 * sblk + ss * sblk + (vq % sblk);
            }
        }
        kI[i] = vi;
        kQ[i] = vq;
    }
}

// A symbol whose samples are all zero (the capture ends inside the frame, digital silence) has a pilot sum of zero,
// and the reference's arithmetic then makes every soft bit of it 0 * 0 / 0; a tone whose channel estimate is zero
// gives infinities.  Such soft bits leave as 0.0, i.e. as erasures: the decoder's path metrics stay finite.
__device__ __forceinline__ void erase_unless(bool ok, float &a, float &b, float &c, float &d)
{
    a = ok ? a : 0.0f;
    b = ok ? b : 0.0f;
    c = ok ? c : 0.0f;
    d = ok ? d : 0.0f;
}

// Soft bits of one tone -> the symbol's LDS stage.
__device__ __forceinline__ void put_llrs(float *stage, c32 q, int mod, float scale, int nb, int nax,
                                         const int (&kI)[4], const int (&kQ)[4])
{
    if (mod != QAM_BPSK) q = cmulf(q, scale);
    float li0, li1, li2, li3, lq0, lq1, lq2, lq3;
    axis_llr(q.x, nax, li0, li1, li2, li3);
    axis_llr(q.y, nax, lq0, lq1, lq2, lq3);
    // every soft bit of an axis is a finite function of the axis value, so one test per axis covers them
    erase_unless(__builtin_isfinite(q.x), li0, li1, li2, li3);
    erase_unless(__builtin_isfinite(q.y), lq0, lq1, lq2, lq3);
    stage[kI[0]] = li0;
    if (nb >= 2) {
        stage[kQ[0]] = lq0;
        if (nax >= 2) {
            stage[kI[1]] = li1;
            stage[kQ[1]] = lq1;
            if (nax >= 3) {
                stage[kI[2]] = li2;
                stage[kQ[2]] = lq2;
                if (nax >= 4) {
                    stage[kI[3]] = li3;
                    stage[kQ[3]] = lq3;
                }
            }
        }
    }
}

// Experiment (GR80211_AXIS_LLR=1): instead of its soft bits a tone leaves its scaled axis values, one float per
// constellation axis; every soft bit of an axis is a fold of that one number, which the decoder's tile loader
// applies (axis_fold).  A symbol then takes nSS * nSD * (1 or 2) floats, tone by tone in the order of the data
// subcarriers, stream after stream, at a stride of its own in the frame's soft-bit run.
__device__ __forceinline__ void put_axes(float *stage, c32 q, int mod, float scale, int nb, int slot)
{
    if (mod != QAM_BPSK) q = cmulf(q, scale);
    stage[slot] = q.x;
    if (nb >= 2) stage[slot + 1] = q.y;
}

// The wave's staged symbol -> the frame's LLR run, 16 bytes per lane (runs are 16-byte aligned: every frame's
// soft-bit count is a multiple of 4).  Always P store instructions, whatever the symbol's size: lanes past the
// end repeat the last four soft bits (the same bytes to the same place).  With a fixed number of stores behind
// it, the wait for the next symbol's prefetched samples can leave them in flight.
template <int P>
__device__ __forceinline__ void flush_stage(const float *stage, float *dst, int ncbps, int lane)
{
    __builtin_amdgcn_wave_barrier();
    float4 v[P];
    int at[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int i = lane * 4 + p * 256;
        at[p] = i < ncbps ? i : ncbps - 4;
        v[p] = *reinterpret_cast<const float4 *>(stage + at[p]);
    }
#pragma unroll
    for (int p = 0; p < P; p++) *reinterpret_cast<float4 *>(dst + at[p]) = v[p];
    __builtin_amdgcn_wave_barrier();
}

// Sample k of the stream `demod` sees, as it lies in memory; src points at that stream's sample 0 (or, for a frame
// without any, at some sample of the arena).  The load is issued whatever k is, so that every symbol costs the
// same number of memory instructions; samples past nvalid read as zero (lib/signal_impl.cc:164-202 forwards
// only what arrived).
__device__ __forceinline__ c32 fetch_samp(const c32 *src, int k, int nvalid)
{
    const c32 v = src[k < nvalid ? k : 0];
    return k < nvalid ? v : mk(0.0f, 0.0f);
}

// exp(j*(k+224)*rad): the CFO derotation `signal` applies to sample k of that stream (lib/signal_impl.cc:172-173).
__device__ __forceinline__ c32 derot(int k, float rad)
{
    if (rad == 0.0f) return mk(1.0f, 0.0f);
    float sn, cs;
    sincosf((float)(k + 224) * rad, &sn, &cs);
    return mk(cs, sn);
}


// One workgroup per frame; the instantiation for the frame's number of spatial streams does the work, the other
// one leaves at once (two kernels keep the one-stream path's registers, and with them its occupancy, free of the
// 2x2 channel matrices).  At most 96 registers per lane (the attribute counts in pairs on this target): five waves
// of the two-stream form per SIMD instead of four (103 registers without it; four values spilled), which this
// latency-bound kernel turns into 1.29 -> 1.10 ms at 50 000 two-stream frames; the one-stream form needs 67 anyway.
template <int NSS, bool AXIS>
__global__ __launch_bounds__(64 * DEMOD_WAVES) __attribute__((amdgpu_num_vgpr(48)))
void k_demod(const c32 *__restrict__ iq0, const c32 *__restrict__ iq1,
                                                            uint64_t arena_len,
                                                            const gr80211_frame_desc *__restrict__ descs, int n,
                                                            const gr80211_frame_info *__restrict__ info,
                                                            const c32 *__restrict__ chan1, const c32 *__restrict__ chan2,
                                                            float *__restrict__ llr, const SchedTotals *__restrict__ totals)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    if (totals && totals->n_groups < 0) return;  // the scheduler refused the batch: its soft bits have no place
    const int st = info[f].status;
    if (!(st == GR80211_ST_OK || st == GR80211_ST_DROP_DECODE_LIMIT)) return;
    if ((info[f].nSS == 1) != (NSS == 1)) return;
    // Beside the decoder waves of another batch in flight this kernel's short, memory-bound waves go first: the
    // decoder's waves alternate between priorities 2 and 0 and would otherwise keep them from issuing most of the time.
    __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) float s_stage[DEMOD_WAVES][DEMOD_STAGE / 2 * NSS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // symbol counters stay scalar
    const int bin = brev6(lane);
    float *stage = s_stage[wave];
    const gr80211_frame_info fi = info[f];
    const gr80211_frame_desc d = descs[f];
    const c32 *b0, *b1;
    const int navail = frame_origin(d, arena_len, iq0, iq1, &b0, &b1);
    const float rad = d.rad;
    int nvalid = fi.nsamp;
    if (nvalid > navail - 224) nvalid = navail - 224;
    // a frame without data samples still needs an address that can be read: the arena's first sample
    const c32 *in0 = nvalid > 0 ? b0 + 224 : iq0;

    const int fmt = fi.format, nb = fi.nBPSCS, ncbps = fi.nCBPS, mod = fi.mod;
    const bool legacy = fmt == GR80211_F_L;
    const bool used = legacy ? (c_tab.ltf_l[bin] != 0.0f) : (c_tab.ltf_nl[bin] != 0.0f);
    const int jd = legacy ? c_tab.data_idx_l[bin] : c_tab.data_idx_nl[bin];
    const int pslot = bin == 43 ? 0 : bin == 57 ? 1 : bin == 7 ? 2 : 3;
    const int didx = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : nb == 6 ? 3 : 4;
    const float scale = qam_scale(mod);
    const int nax = nb == 1 ? 1 : nb / 2;  // bits per axis (BPSK: one bit, I axis only)
    float *out = llr + fi.llr_off;
    const int pp0 = legacy ? 1 : fmt == GR80211_F_HT ? 3 : 4;
    const int nsym = fi.nSym, symsamp = fi.nSymSamp, data_off = fi.data_off;

    // what does not change from symbol to symbol sits in registers before the loop starts: the twiddles, and
    // the 127 pilot polarities spread over the lanes (lane l holds entries l and 64 + l)
    c32 tw[5];
    fft_twiddles(tw, lane);
    const float pol0 = c_tab.pilot_p[lane], pol1 = c_tab.pilot_p[lane < 63 ? 64 + lane : 64];
    asm volatile("; loop invariants resident" ::"v"(tw[0].x), "v"(tw[0].y), "v"(tw[1].x), "v"(tw[1].y), "v"(tw[2].x),
                 "v"(tw[2].y), "v"(tw[3].x), "v"(tw[3].y), "v"(tw[4].x), "v"(tw[4].y), "v"(pol0), "v"(pol1));

    if (NSS == 1) {
        const c32 h = chan1[(size_t)f * 64 + bin];
        const int16_t *map = legacy ? c_tab.deint_l[didx] : c_tab.deint_nl[didx];
        int kI[4], kQ[4];
        tone_slots(kI, kQ, map, jd, nb, nax, 0, 1, false);
        asm volatile("; loop invariants resident" ::"v"(h.x), "v"(h.y), "v"(kI[0]), "v"(kI[1]), "v"(kI[2]), "v"(kI[3]),
                     "v"(kQ[0]), "v"(kQ[1]), "v"(kQ[2]), "v"(kQ[3]));
        int kn = data_off + wave * symsamp + SHIFT + lane;
        c32 xn = fetch_samp(in0, kn, nvalid);
        // the first symbol's samples are waited for here, so that inside the loop the only wait on memory is the
        // one for the samples fetched a whole symbol earlier
        asm volatile("; first symbol resident" ::"v"(xn.x), "v"(xn.y));
        for (int sym = wave; sym < nsym; sym += DEMOD_WAVES) {
            const c32 x = cmul(xn, derot(kn, rad));
            // the next symbol's samples travel while this one is worked on (past the last symbol the load repeats
            // an address that is valid anyway)
            kn += DEMOD_WAVES * symsamp;
            xn = fetch_samp(in0, kn, sym + DEMOD_WAVES < nsym ? nvalid : 0);
            const c32 F = wave_fft64_tw(x, lane, tw);
            c32 e = mk(0.0f, 0.0f);
            if (used) e = cdiv(F, h);
            // pilot term of this lane's tone: sig * pilot * polarity (lib/demod_impl.cc:426-430);
            // base pattern {1,1,1,-1}, rotated one place per symbol for HT/VHT
            const int pi = (pp0 + sym) % 127;
            const float pp = __int_as_float(pi < 64 ? __builtin_amdgcn_readlane(__float_as_int(pol0), pi)
                                                    : __builtin_amdgcn_readlane(__float_as_int(pol1), pi - 64));
            const int pr = legacy ? pslot : ((pslot + sym) & 3);
            const float pv = pr == 3 ? -1.0f : 1.0f;
            const c32 u = cmulf(cmulf(e, pv), pp);
            const c32 ps = cconj(cadd(cadd(cadd(lane_c<LN7>(u), lane_c<LN21>(u)), lane_c<LN43>(u)), lane_c<LN57>(u)));
            const float pa = cabs32(ps);
            if constexpr (AXIS) {
                const int na = nb == 1 ? 1 : 2, nax_sym = (legacy ? 48 : 52) * na;
                if (jd >= 0) put_axes(stage, cdivf(cmul(e, ps), pa), mod, scale, nb, jd * na);
                flush_stage<1>(stage, out + (size_t)sym * nax_sym, nax_sym, lane);
                continue;
            }
            if (jd >= 0) put_llrs(stage, cdivf(cmul(e, ps), pa), mod, scale, nb, nax, kI, kQ);
            flush_stage<2>(stage, out + (size_t)sym * ncbps, ncbps, lane);
        }
    } else {
        // two spatial streams on two antennas (lib/demod2_impl.cc:504-550, 586-628)
        const c32 *in1 = nvalid > 0 ? b1 + 224 : iq1;
        const bool pil = is_pilot_bin(bin);
        const c32 *c2 = chan2 + (size_t)f * CH2;
        c32 h[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { h[q] = c2[bin * 4 + q]; hi[q] = c2[256 + bin * 4 + q]; }
        const c32 pl1 = pil ? c2[512 + pslot] : mk(0.0f, 0.0f);
        const c32 pl2 = pil ? c2[516 + pslot] : mk(0.0f, 0.0f);
        const bool vht = fmt == GR80211_F_VHT;
        const int16_t *map1 = c_tab.deint_nl[didx];
        const int16_t *map2 = c_tab.deint_nl2[didx];
        const int sblk = nb / 2 > 1 ? nb / 2 : 1;  // stream parser block
        int kI1[4], kQ1[4], kI2[4], kQ2[4];
        tone_slots(kI1, kQ1, map1, jd, nb, nax, 0, sblk, true);
        tone_slots(kI2, kQ2, map2, jd, nb, nax, 1, sblk, true);
        asm volatile("; loop invariants resident" ::"v"(h[0].x), "v"(h[1].x), "v"(h[2].x), "v"(h[3].x), "v"(hi[0].x),
                     "v"(hi[1].x), "v"(hi[2].x), "v"(hi[3].x), "v"(pl1.x), "v"(pl2.x), "v"(kI1[3]), "v"(kQ1[3]),
                     "v"(kI2[3]), "v"(kQ2[3]));
        int kn = data_off + wave * symsamp + SHIFT + lane;
        c32 xn1 = fetch_samp(in0, kn, nvalid), xn2 = fetch_samp(in1, kn, nvalid);
        asm volatile("; first symbol resident" ::"v"(xn1.x), "v"(xn1.y), "v"(xn2.x), "v"(xn2.y));
        for (int sym = wave; sym < nsym; sym += DEMOD_WAVES) {
            const c32 rot = derot(kn, rad);
            const c32 x1 = cmul(xn1, rot), x2 = cmul(xn2, rot);
            kn += DEMOD_WAVES * symsamp;
            const int nv = sym + DEMOD_WAVES < nsym ? nvalid : 0;
            xn1 = fetch_samp(in0, kn, nv);
            xn2 = fetch_samp(in1, kn, nv);
            const c32 F1 = wave_fft64_tw(x1, lane, tw);
            const c32 F2 = wave_fft64_tw(x2, lane, tw);
            c32 s1 = mk(0.0f, 0.0f), s2 = mk(0.0f, 0.0f);
            if (used) zf2(F1, F2, h, hi, s1, s2);
            const int pi = (pp0 + sym) % 127;
            const float pp = __int_as_float(pi < 64 ? __builtin_amdgcn_readlane(__float_as_int(pol0), pi)
                                                    : __builtin_amdgcn_readlane(__float_as_int(pol1), pi - 64));
            const int pr = (pslot + sym) & 3;
            // PILOT_VHT {1,1,1,-1} on both streams; PILOT_HT_2_1 {1,1,-1,-1}, PILOT_HT_2_2 {1,-1,-1,1}
            const float pv1 = vht ? (pr == 3 ? -1.0f : 1.0f) : (pr >= 2 ? -1.0f : 1.0f);
            const float pv2 = vht ? (pr == 3 ? -1.0f : 1.0f) : ((pr == 1 || pr == 2) ? -1.0f : 1.0f);
            const c32 u1 = cmul(cmulf(cmulf(s1, pv1), pp), pl1);
            const c32 u2 = cmul(cmulf(cmulf(s2, pv2), pp), pl2);
            c32 acc = lane_c<LN7>(u1);
            acc = cadd(acc, lane_c<LN21>(u1));
            acc = cadd(acc, lane_c<LN43>(u1));
            acc = cadd(acc, lane_c<LN57>(u1));
            acc = cadd(acc, lane_c<LN7>(u2));
            acc = cadd(acc, lane_c<LN21>(u2));
            acc = cadd(acc, lane_c<LN43>(u2));
            acc = cadd(acc, lane_c<LN57>(u2));
            const c32 ps = cconj(acc);
            const float pa = cabs32(ps);
            if constexpr (AXIS) {
                const int na = nb == 1 ? 1 : 2, nax_sym = 2 * 52 * na;
                if (jd >= 0) {
                    put_axes(stage, cdivf(cmul(s1, ps), pa), mod, scale, nb, jd * na);
                    put_axes(stage, cdivf(cmul(s2, ps), pa), mod, scale, nb, (52 + jd) * na);
                }
                flush_stage<1>(stage, out + (size_t)sym * nax_sym, nax_sym, lane);
                continue;
            }
            if (jd >= 0) {
                put_llrs(stage, cdivf(cmul(s1, ps), pa), mod, scale, nb, nax, kI1, kQ1);
                put_llrs(stage, cdivf(cmul(s2, ps), pa), mod, scale, nb, nax, kI2, kQ2);
            }
            flush_stage<4>(stage, out + (size_t)sym * ncbps, ncbps, lane);
        }
    }
}

// =========================================================================================
// k_viterbi: lane = frame, 64 states in registers
// =========================================================================================
// Four state updates (states N, N+8, N+16, N+24) written phase by phase (adds, subtracts, maxima, decision shifts) so that no
// instruction sits next to the one it depends on: on gfx950 a dependent VALU pair costs ~9.5 cycles
// against ~4.3 for independent issue from one wave.
//
// Register-only form of the reference's "a1 > a0 ? a1 : a0" plus survivor bit (lib/decode_impl.cc:248-258),
// three VALU instructions and no scalar mask: the metric is max(a0, a1); the decision is the sign of
// a0 - a1, which is set exactly when a1 > a0 (two distinct floats never subtract to zero with
// subnormals enabled, and a tie gives +0, i.e. the even predecessor as the strict '>' does);
// v_alignbit shifts it into a decision accumulator.  Each of the four states feeds its own
// accumulator, which removes the 32-long shift chain of a single decision word.
template <int KIND, int N>
__device__ __forceinline__ void acs_quad(const float (&m)[64], float (&o)[64], float t0, float t1, float t01,
                                         uint32_t (&dw)[4])
{
    float a0[4], a1[4], df[4];
#define GR_ACS_IN(K)                                                              \
    {                                                                             \
        constexpr int n = N + 8 * K, j = n & 31, b = n >> 5, p0 = 2 * j, p1 = 2 * j + 1; \
        a0[K] = add_tab<KIND, sv_out(p0, b)>(m[p0], t0, t1, t01);                 \
        a1[K] = add_tab<KIND, sv_out(p1, b)>(m[p1], t0, t1, t01);                 \
    }
    GR_ACS_IN(0) GR_ACS_IN(1) GR_ACS_IN(2) GR_ACS_IN(3)
#undef GR_ACS_IN
#pragma unroll
    for (int k = 0; k < 4; k++) df[k] = a0[k] - a1[k];
#pragma unroll
    for (int k = 0; k < 4; k++) o[N + 8 * k] = metric_max(a0[k], a1[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) dw[k] = __builtin_amdgcn_alignbit(dw[k], __float_as_uint(df[k]), 31);
}

template <int KIND, int BASE, int... N>
__device__ __forceinline__ void acs_range4(const float (&m)[64], float (&o)[64], float t0, float t1, float t01,
                                           uint32_t (&dw)[4], std::integer_sequence<int, N...>)
{
    (acs_quad<KIND, BASE + N>(m, o, t0, t1, t01, dw), ...);
}

// One trellis step for 64 frames.  Survivor bits of states 0..31 go to w0, of states 32..63 to w1;
// within a word state n sits at bit dec_bit(n).
template <int KIND>
__device__ __forceinline__ void acs_step(const float (&m)[64], float (&o)[64], float t0, float t1, uint32_t &w0,
                                         uint32_t &w1)
{
    const float t01 = t1 + t0;
    uint32_t qa[4] = {0, 0, 0, 0}, qb[4] = {0, 0, 0, 0};
    acs_range4<KIND, 0>(m, o, t0, t1, t01, qa, std::make_integer_sequence<int, 8>{});
    acs_range4<KIND, 32>(m, o, t0, t1, t01, qb, std::make_integer_sequence<int, 8>{});
    // accumulator k holds states 8k .. 8k+7 of its half (first in the top bit of its byte)
    w0 = qa[0] | (qa[1] << 8) | (qa[2] << 16) | (qa[3] << 24);
    w1 = qb[0] | (qb[1] << 8) | (qb[2] << 16) | (qb[3] << 24);
}

// Bit position of state n's survivor bit in the 64-bit word {w1:w0} of its step.
__device__ __forceinline__ int dec_bit(int n) { return n ^ 7; }

// Step S (0 .. 2P-1) of a pair of puncture periods whose soft bits start at c.  Even steps go from
// A to B, odd steps back, so a pair of periods always starts and ends in A.
template <int CR, int S>
__device__ __forceinline__ void trellis_step(float (&A)[64], float (&B)[64], const float *c, uint32_t &w0, uint32_t &w1)
{
    constexpr int P = Rate<CR>::P, ps = S % P, kind = step_kind(CR, ps);
    constexpr int col = (S / P) * Rate<CR>::COLS + step_col(ps);
    const float t0 = kind == 2 ? 0.0f : c[col];
    const float t1 = kind == 0 ? c[col + 1] : kind == 2 ? c[col] : 0.0f;
    if constexpr (S & 1) acs_step<kind>(B, A, t0, t1, w0, w1);
    else acs_step<kind>(A, B, t0, t1, w0, w1);
}

// Steps 2U and 2U+1; their four survivor words leave as one 16-byte store per lane
// (survivor layout: [step pair][lane][w0(even) w1(even) w0(odd) w1(odd)]).
template <int CR, int U>
__device__ __forceinline__ void step_pair(float (&A)[64], float (&B)[64], const float *c, uint32_t *dp)
{
    uint4 v;
    trellis_step<CR, 2 * U>(A, B, c, v.x, v.y);
    trellis_step<CR, 2 * U + 1>(A, B, c, v.z, v.w);
    *reinterpret_cast<uint4 *>(dp + U * 256) = v;
}

template <int CR, int... U>
__device__ __forceinline__ void two_periods(float (&A)[64], float (&B)[64], const float *c, uint32_t *dp,
                                            std::integer_sequence<int, U...>)
{
    (step_pair<CR, U>(A, B, c, dp), ...);
}


// 64 rows (frames) x 64 columns of soft bits, transposed through LDS.  Each row is one LDS-DMA load
// (global_load_lds_dword: lane l of the wave writes tile[row][l]), so all 64 are in flight together and cost no
// VGPRs while the 128 path-metric registers are live.  Columns past a frame's last soft bit are clamped to it:
// they only feed trellis steps after that frame's end.
typedef __attribute__((address_space(3))) float lds_float;

__device__ __forceinline__ void load_tile(const float *__restrict__ llr, unsigned long long mybase, int myn, int c0,
                                          lds_float *tile)
{
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const unsigned lo = __builtin_amdgcn_readlane((int)(mybase & 0xffffffffull), r);
        const unsigned hi = __builtin_amdgcn_readlane((int)(mybase >> 32), r);
        const int nr = __builtin_amdgcn_readlane(myn, r);
        const unsigned long long base = ((unsigned long long)hi << 32) | lo;
        int idx = c0 < nr ? c0 : nr - 1;
        idx = idx < 0 ? 0 : idx;
        const float *src = llr + base + (unsigned long long)idx;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(tile + r * TILE_STRIDE), 4, 0, 0);
    }
}

// The same for a chunk that lies inside every frame of the group (all chunks but the last few): no clamping, so a
// row's address is a wave-uniform base plus the lane's column, which costs a third of the instructions per row.
// Lanes without a frame read soft bits of some other frame at that column; nothing uses them.
__device__ __forceinline__ void load_tile_inside(const float *__restrict__ llr, unsigned long long mybase, int col0, int lane,
                                                 lds_float *tile)
{
    // every lane's row address in bytes; the row's own comes out of it with two v_readlane
    const unsigned long long mine = (unsigned long long)(llr + mybase);
    const unsigned long long step = (unsigned long long)col0 * sizeof(float);
#pragma unroll
    for (int r = 0; r < 64; r++) {
        const unsigned lo = __builtin_amdgcn_readlane((int)(mine & 0xffffffffull), r);
        const unsigned hi = __builtin_amdgcn_readlane((int)(mine >> 32), r);
        const float *rowp = (const float *)((((unsigned long long)hi << 32) | lo) + step);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp + lane),
                                         (__attribute__((address_space(3))) void *)(tile + r * TILE_STRIDE), 4, 0, 0);
    }
}

// The tile whose first column is col0: without clamping where every frame of the group (the shortest has `least`
// soft bits) reaches past the tile's 64 columns.
__device__ __forceinline__ void fetch_tile(const float *__restrict__ llr, unsigned long long mybase, int myn, int least,
                                           int col0, int lane, lds_float *tile)
{
    if (col0 + 64 <= least) load_tile_inside(llr, mybase, col0, lane, tile);
    else load_tile(llr, mybase, myn, col0 + lane, tile);
}


// ---- experiment: soft bits folded out of axis values by the tile loader (k_demod's put_axes) ----------------------
// The modulation every frame of a group shares (wave-uniform).
struct AxisMod {
    int nb, ncbps, nss, legacy, didx, nsd, na, nax, sym_floats;
};

// Column c of a frame's soft-bit stream (decoder order) -> where its axis value lies in the frame's run, and which
// fold of it the soft bit is: the inverse of the stream de-parser (lib/cloud80211phy.cc:2442-2451), the interleaver
// (coded bit k is sent at position j = tone * nBPSCS + bit) and the I-bits-then-Q-bits order of a tone.
__device__ __forceinline__ void axis_map(const AxisMod &m, int c, int &ofs, int &level)
{
    const int n = c / m.ncbps, p = c - n * m.ncbps;
    int iss = 0, k = p;
    if (m.nss == 2) {
        const int sblk = m.nb / 2 > 1 ? m.nb / 2 : 1;
        const int blk = p / sblk;
        iss = blk & 1;
        k = (blk >> 1) * sblk + (p - blk * sblk);
    }
    const int j = m.legacy ? c_tab.intl_l[m.didx][k] : iss ? c_tab.intl_nl2[m.didx][k] : c_tab.intl_nl[m.didx][k];
    const int tone = j / m.nb, b = j - tone * m.nb;
    const int ax = b / m.nax;
    level = b - ax * m.nax;
    ofs = n * m.sym_floats + (iss * m.nsd + tone) * m.na + ax;
}

// Soft bit `level` of an axis whose scaled value is v: axis_llr's operations, one erasure test per axis as in put_llrs.
__device__ __forceinline__ float axis_fold(float v, int level, int nax)
{
    const float edge = nax == 2 ? 2.0f : nax == 3 ? 4.0f : 8.0f;
    const float o1 = edge - fabsf(v);
    const float o2 = edge * 0.5f - fabsf(o1);
    const float o3 = 2.0f - fabsf(o2);
    const float r = level == 0 ? v : level == 1 ? o1 : level == 2 ? o2 : o3;
    return __builtin_isfinite(v) ? r : 0.0f;
}

// The tile whose first column is col0, from axis values.  Inside every frame: one row load per frame as in
// load_tile_inside, the lane's column offset being where the column's axis value lies; then every lane folds its
// column, row by row, in LDS.  At the frames' ends (columns clamped per frame) values are loaded and folded one row at
// a time.
__device__ __forceinline__ void fetch_tile_axis(const float *__restrict__ llr, unsigned long long mybase, int myn, int least,
                                                int col0, int lane, lds_float *tile, const AxisMod &m)
{
    float *t = (float *)tile;
    if (col0 + 64 <= least) {
        int ofs, level;
        axis_map(m, col0 + lane, ofs, level);
        const unsigned long long mine = (unsigned long long)(llr + mybase);
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const unsigned lo = __builtin_amdgcn_readlane((int)(mine & 0xffffffffull), r);
            const unsigned hi = __builtin_amdgcn_readlane((int)(mine >> 32), r);
            const float *rowp = (const float *)(((unsigned long long)hi << 32) | lo);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp + ofs),
                                             (__attribute__((address_space(3))) void *)(tile + r * TILE_STRIDE), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (m.nax > 1) {
#pragma unroll 8
            for (int r = 0; r < 64; r++) t[r * TILE_STRIDE + lane] = axis_fold(t[r * TILE_STRIDE + lane], level, m.nax);
        }
    } else {
        for (int r = 0; r < 64; r++) {
            const unsigned lo = __builtin_amdgcn_readlane((int)(mybase & 0xffffffffull), r);
            const unsigned hi = __builtin_amdgcn_readlane((int)(mybase >> 32), r);
            const int nr = __builtin_amdgcn_readlane(myn, r);
            const unsigned long long base = ((unsigned long long)hi << 32) | lo;
            int c = col0 + lane;
            c = c < nr ? c : nr - 1;
            c = c < 0 ? 0 : c;
            int ofs, level;
            axis_map(m, c, ofs, level);
            const float v = llr[base + (unsigned long long)ofs];
            t[r * TILE_STRIDE + lane] = m.nax > 1 ? axis_fold(v, level, m.nax) : (__builtin_isfinite(v) ? v : 0.0f);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Forward pass of one group.  TWO_TILES: the wave has room for two tiles in LDS (a batch of at most one group per
// SIMD) and fetches the next chunk's soft bits while it works on this one; a wave that shares its SIMD leaves the
// fetch latency to its partner instead.
// Chunks [ch0, ch1) of a group's forward pass: A holds the path metrics before chunk ch0 on entry and after chunk
// ch1 - 1 on return (a chunk is an even number of steps, so the metrics end where they started).
template <int CR, bool TWO_TILES, bool AXIS = false>
__device__ __forceinline__ void viterbi_chunks(float (&A)[64], const float *__restrict__ llr, unsigned long long mybase, int myn,
                                               int ch0, int ch1, uint32_t *__restrict__ dec, float *tile, int lane,
                                               const AxisMod *am = nullptr)
{
    constexpr int P = Rate<CR>::P, COLS = Rate<CR>::COLS, NPER = Rate<CR>::NPER;
    constexpr int CH = COLS * NPER;    // LLR columns per tile
    constexpr int STEPS = P * NPER;    // trellis steps per tile
    static_assert(STEPS / 2 < 63, "the wait below counts one survivor store per step pair");
    float B[64];
#ifndef GR80211_PRIO_SHIFT
#define GR80211_PRIO_SHIFT 13
#endif
    // Two waves that share a SIMD are served oldest first: the older one runs at the speed of a wave that has
    // the SIMD to itself and the younger one is left to finish alone, at a lone wave's poor issue rate.  The
    // waves therefore trade priority in time slices of 2^GR80211_PRIO_SHIFT ticks of the 100-MHz clock (both
    // read the same clock; the wave slot number tells them apart), so that both stay paired until the end.
    const unsigned slot_parity = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) & 1u;  // HW_ID.wave_id
    // soft bits the shortest frame of the group has: chunks below that mark need no clamping (lanes without a
    // frame do not count)
    int least = myn > 0 ? myn : 0x7fffffff;
    for (int d = 32; d > 0; d >>= 1) least = min(least, __shfl_xor(least, d));
    least = __builtin_amdgcn_readfirstlane(least);
    lds_float *lt = (lds_float *)tile;  // the wave's tile(s), as the LDS addresses the DMA loads take
    if (TWO_TILES && ch0 < ch1) fetch_tile(llr, mybase, myn, least, ch0 * CH, lane, lt + (ch0 & 1) * (64 * TILE_STRIDE));
    for (int ch = ch0; ch < ch1; ch++) {
        float *cur = tile;
        if (TWO_TILES) {
            cur = tile + (ch & 1) * (64 * TILE_STRIDE);
            // this chunk's soft bits were requested a chunk ago; the only younger memory operations are the
            // survivor stores of the chunk in between (none before the first chunk), which may stay in flight
            if (ch == ch0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEPS / 2) : "memory");
            // the other tile's last readers were the steps of the previous chunk: their values have arrived
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ch + 1 < ch1) fetch_tile(llr, mybase, myn, least, (ch + 1) * CH, lane, lt + ((ch + 1) & 1) * (64 * TILE_STRIDE));
        } else {
            if ((((unsigned)(__builtin_amdgcn_s_memrealtime() >> GR80211_PRIO_SHIFT)) ^ slot_parity) & 1u)
                __builtin_amdgcn_s_setprio(2);
            else
                __builtin_amdgcn_s_setprio(0);
            // the tile belongs to this wave alone: its reads of the previous chunk have returned (their values were
            // used), so the loads below may overwrite it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (AXIS) {
                fetch_tile_axis(llr, mybase, myn, least, ch * CH, lane, lt, *am);
            } else {
                fetch_tile(llr, mybase, myn, least, ch * CH, lane, lt);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float *row = cur + lane * TILE_STRIDE;
        uint32_t *dp = dec + (size_t)ch * STEPS * 128 + lane * 4;
#pragma unroll 1
        for (int pp = 0; pp < NPER; pp += 2)
            two_periods<CR>(A, B, row + pp * COLS, dp + (size_t)pp * P * 128, std::make_integer_sequence<int, P>{});
    }
}

// Forward pass of one group from its first chunk to its last.
template <int CR, bool TWO_TILES, bool AXIS = false>
__device__ __forceinline__ void viterbi_group(const float *__restrict__ llr, unsigned long long mybase, int myn,
                                              int nchunks, uint32_t *__restrict__ dec, float *tile, int lane,
                                              const AxisMod *am = nullptr)
{
    float A[64];
#pragma unroll
    for (int i = 0; i < 64; i++) A[i] = i == 0 ? 0.0f : NEG_INIT;
    viterbi_chunks<CR, TWO_TILES, AXIS>(A, llr, mybase, myn, 0, nchunks, dec, tile, lane, am);
}

__global__ __launch_bounds__(64) void k_viterbi(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                                const int32_t *__restrict__ slot_frame,
                                                const GroupInfo *__restrict__ groups, int n_groups,
                                                uint32_t *__restrict__ dec, const SchedTotals *__restrict__ totals)
{
    const int g = blockIdx.x;
    if (g >= groups_to_run(n_groups, totals)) return;
    const int lane = threadIdx.x;
    __shared__ float tile[64 * TILE_STRIDE];
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    // a lane without a frame reads element 0 of the arena (always present) and ignores it
    const int myn = fr >= 0 ? info[fr].total : 0;
    const unsigned long long mybase = myn > 0 ? (unsigned long long)info[fr].llr_off : 0ull;
    uint32_t *d = dec + gi.dec_off;
    switch (gi.cr) {
    case CR_12: viterbi_group<CR_12, false>(llr, mybase, myn, gi.nchunks, d, tile, lane); break;
    case CR_23: viterbi_group<CR_23, false>(llr, mybase, myn, gi.nchunks, d, tile, lane); break;
    case CR_34: viterbi_group<CR_34, false>(llr, mybase, myn, gi.nchunks, d, tile, lane); break;
    default: viterbi_group<CR_56, false>(llr, mybase, myn, gi.nchunks, d, tile, lane); break;
    }
}

// =========================================================================================
// k_traceback: traceback + descramble + packet assembly + FCS, lane = frame
// =========================================================================================
// One group's traceback and packet assembly; s_crc is the CRC-32 byte table in LDS.
__device__ __forceinline__ void traceback_group(const uint32_t *__restrict__ dec, const gr80211_frame_info *__restrict__ info,
                                                const GroupInfo &gi, const int fr, const int lane, const uint32_t *s_crc,
                                                uint32_t *__restrict__ bitsw, uint8_t *__restrict__ out, int out_stride,
                                                int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu,
                                                uint8_t *__restrict__ bits_out, int bits_stride, int strict)
{
    const int T = fr >= 0 ? info[fr].trellis : 0;
    uint32_t *bw = bitsw + gi.bits_off + lane;

    // ---- traceback from state 0 at step T (lib/decode_impl.cc:282-302).  The survivor words do not
    // depend on the path, so they are fetched into a ring of four register sets (16 steps each):
    // three sets of 1-KiB coalesced loads stay in flight while the dependent chain walks the fourth.
    // The walk keeps the path in a shift register S: the state after step t is S & 63, its newest bit
    // (bit 5) is decoded bit t, and one step back is S = (S << 1) | survivor bit.  Decoded bits thus
    // line up in S in time order and leave 16 at a time; per step the chain is xor, 64-bit shift,
    // and-or.  Steps at or above a lane's own trellis length produce bits nobody reads; S is reset
    // when the lane's last step is reached, which can only happen in the topmost blocks.
    {
        constexpr int TB = 16;
        uint32_t S = 0;
        uint4 wa[TB / 2], wb[TB / 2], wc[TB / 2], wd[TB / 2];
        const int steps = gi.steps;          // even: whole pairs of puncture periods
        const int nb32 = (steps + 31) >> 5;  // 32-step blocks = decoded words per lane
        const int umax = (steps - 1) >> 1;
        int tmin = fr >= 0 ? T : 0x7fffffff;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int o = __shfl_xor(tmin, m);
            tmin = o < tmin ? o : tmin;
        }
        tmin = __builtin_amdgcn_readfirstlane(tmin);
        const uint4 *d4 = reinterpret_cast<const uint4 *>(dec + gi.dec_off) + lane;
        // half-block h (from the top): steps 16*(2*nb32-1-h)+15 .. 16*(2*nb32-1-h)
#define GR_TB_FETCH(W, H)                                    \
    {                                                        \
        const int uhi = (16 * (2 * nb32 - 1 - (H)) + 15) >> 1; \
        _Pragma("unroll") for (int k = 0; k < TB / 2; k++) { \
            int u = uhi - k;                                 \
            u = u > umax ? umax : u;                         \
            u = u < 0 ? 0 : u;                               \
            W[k] = d4[(size_t)u * 64];                       \
        }                                                    \
    }
#define GR_TB_WALK(W, THI, CHECK)                                                                   \
    {                                                                                               \
        _Pragma("unroll") for (int k = 0; k < TB; k++) {                                            \
            if (CHECK) S = ((THI) - k == T - 1) ? 0u : S;                                           \
            const uint32_t w0 = (k & 1) ? W[k >> 1].x : W[k >> 1].z;                                \
            const uint32_t w1 = (k & 1) ? W[k >> 1].y : W[k >> 1].w;                                \
            const unsigned long long w = ((unsigned long long)w1 << 32) | w0;                       \
            const uint32_t sh = (uint32_t)(w >> ((S ^ 7u) & 63u));                                  \
            S = (S << 1) | (sh & 1u);                                                               \
        }                                                                                           \
    }
        // one 32-step block = two half-blocks; S >> 6 after a half holds its 16 decoded bits
#define GR_TB_BLOCK(WH, WL, B32, HNEXT)                                                             \
    if ((B32) >= 0) {                                                                               \
        const int thi = 32 * (B32) + 31;                                                            \
        uint32_t hi16, lo16;                                                                        \
        if (thi >= tmin - 1) {                                                                      \
            GR_TB_WALK(WH, thi, true)                                                               \
            hi16 = (S >> 6) & 0xffffu;                                                              \
            GR_TB_FETCH(WH, HNEXT)                                                                  \
            GR_TB_WALK(WL, thi - 16, true)                                                          \
            lo16 = (S >> 6) & 0xffffu;                                                              \
            GR_TB_FETCH(WL, (HNEXT) + 1)                                                            \
        } else {                                                                                    \
            GR_TB_WALK(WH, thi, false)                                                              \
            hi16 = (S >> 6) & 0xffffu;                                                              \
            GR_TB_FETCH(WH, HNEXT)                                                                  \
            GR_TB_WALK(WL, thi - 16, false)                                                         \
            lo16 = (S >> 6) & 0xffffu;                                                              \
            GR_TB_FETCH(WL, (HNEXT) + 1)                                                            \
        }                                                                                           \
        bw[(size_t)(B32) * 64] = lo16 | (hi16 << 16);                                               \
    }
        GR_TB_FETCH(wa, 0)
        GR_TB_FETCH(wb, 1)
        GR_TB_FETCH(wc, 2)
        GR_TB_FETCH(wd, 3)
        for (int b32 = nb32 - 1, h = 4; b32 >= 0; b32 -= 2, h += 4) {
            GR_TB_BLOCK(wa, wb, b32, h)
            GR_TB_BLOCK(wc, wd, b32 - 1, h + 2)
        }
#undef GR_TB_BLOCK
#undef GR_TB_FETCH
#undef GR_TB_WALK
    }
    if (fr < 0) return;
    assemble_frame(bw, 64, info[fr], fr, gi.words, s_crc, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict);
}

__global__ __launch_bounds__(64) void k_traceback(const uint32_t *__restrict__ dec, const gr80211_frame_info *__restrict__ info,
                                                  const int32_t *__restrict__ slot_frame,
                                                  const GroupInfo *__restrict__ groups, int n_groups,
                                                  uint32_t *__restrict__ bitsw, uint8_t *__restrict__ out, int out_stride,
                                                  int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu,
                                                  uint8_t *__restrict__ bits_out, int bits_stride, int strict,
                                                  const SchedTotals *__restrict__ totals)
{
    const int g = blockIdx.x;
    if (g >= groups_to_run(n_groups, totals)) return;
    const int lane = threadIdx.x;
    __shared__ uint32_t s_crc[256];
    for (int i = lane; i < 256; i += 64) s_crc[i] = c_tab.crc[i];
    __syncthreads();
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    traceback_group(dec, info, gi, fr, lane, s_crc, bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict);
}

// =========================================================================================
// k_decode: Viterbi forward pass and traceback of one group in one wave.  A wave that finishes its
// trellis early (it had a SIMD to itself, or a short group) walks its survivors back while other
// waves still run the add-compare-select recursion, so the memory-bound walk hides behind the
// VALU-bound recursion instead of following it as a second kernel.
// =========================================================================================
#ifdef GR80211_TIMELINE
// Debug build only: per group {start, end of forward pass, end, hardware id} in 100-MHz ticks.
__device__ unsigned long long g_timeline[4 * 8192];
int read_timeline(unsigned long long *dst, int n_groups)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 4 * (size_t)n_groups);
}
#endif

// WAVES groups per workgroup, one per wave; the waves share nothing and never meet at a barrier (their trellises
// differ in length).  Four: the hardware spreads the waves of a workgroup over the four SIMDs of a CU, which a batch
// of at most one group per SIMD needs (one-wave workgroups get stacked two to a SIMD beside idle SIMDs).  One: for
// larger batches, where single waves spread more evenly over the CUs than whole workgroups of four (measured at
// 1563 groups: 12.0 against 12.5 ms).
constexpr int DEC_WAVES = 4;

template <bool TWO_TILES, int WAVES, bool AXIS = false>
__device__ __forceinline__ void decode_wave(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                            const int32_t *__restrict__ slot_frame,
                                            const GroupInfo *__restrict__ groups, int n_groups,
                                            uint32_t *__restrict__ dec, uint32_t *__restrict__ bitsw,
                                            uint8_t *__restrict__ out, int out_stride, int32_t *__restrict__ out_len,
                                            int32_t *__restrict__ n_pdu, uint8_t *__restrict__ bits_out, int bits_stride,
                                            int strict, const SchedTotals *__restrict__ totals)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g0 = blockIdx.x * WAVES + wv;
    n_groups = groups_to_run(n_groups, totals);
    if (g0 >= n_groups) return;
    // the scheduler numbers the groups of a code rate by rising trellis length; workgroups start in index order, so
    // the numbering is walked backwards and the longest trellises start first (a mixed batch ends 6 % sooner)
    const int g = n_groups - 1 - g0;
    const int lane = threadIdx.x & 63;
#ifdef GR80211_TIMELINE
    if (lane == 0 && g < 8192) {
        g_timeline[4 * g] = wall_clock64();
        g_timeline[4 * g + 3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    }
#endif
    __shared__ float tiles[WAVES][(TWO_TILES ? 2 : 1) * 64 * TILE_STRIDE];
    float *tile = tiles[wv];
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    const int myn = fr >= 0 ? info[fr].total : 0;
    const unsigned long long mybase = myn > 0 ? (unsigned long long)info[fr].llr_off : 0ull;
    uint32_t *d = dec + gi.dec_off;
    AxisMod am = {};
    if constexpr (AXIS) {
        // every frame of the group must share one modulation: a group that does not is left undecoded
        const int key = fr >= 0 ? (info[fr].nBPSCS | (info[fr].nCBPS << 4) | (info[fr].nSS << 16) | ((info[fr].format == GR80211_F_L) << 20)) : -1;
        int first = key;
        for (int dd = 32; dd > 0; dd >>= 1) { const int o = __shfl_xor(first, dd); first = first > o ? first : o; }
        first = __builtin_amdgcn_readfirstlane(first);
        if (__builtin_amdgcn_ballot_w64(key >= 0 && key != first) != 0) return;
        am.nb = first & 15;
        am.ncbps = (first >> 4) & 0xfff;
        am.nss = (first >> 16) & 15;
        am.legacy = (first >> 20) & 1;
        am.didx = am.nb == 1 ? 0 : am.nb == 2 ? 1 : am.nb == 4 ? 2 : am.nb == 6 ? 3 : 4;
        am.nsd = am.legacy ? 48 : 52;
        am.na = am.nb == 1 ? 1 : 2;
        am.nax = am.nb == 1 ? 1 : am.nb / 2;
        am.sym_floats = am.nss * am.nsd * am.na;
    }
    switch (gi.cr) {
    case CR_12: viterbi_group<CR_12, TWO_TILES, AXIS>(llr, mybase, myn, gi.nchunks, d, tile, lane, &am); break;
    case CR_23: viterbi_group<CR_23, TWO_TILES, AXIS>(llr, mybase, myn, gi.nchunks, d, tile, lane, &am); break;
    case CR_34: viterbi_group<CR_34, TWO_TILES, AXIS>(llr, mybase, myn, gi.nchunks, d, tile, lane, &am); break;
    default: viterbi_group<CR_56, TWO_TILES, AXIS>(llr, mybase, myn, gi.nchunks, d, tile, lane, &am); break;
    }
    // every lane reads back only survivor words it stored itself; drain the stores first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(3);  // the walk back is a short dependent chain between long memory waits
#ifdef GR80211_TIMELINE
    if (lane == 0 && g < 8192) g_timeline[4 * g + 1] = wall_clock64();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t *s_crc = reinterpret_cast<uint32_t *>(tile);  // the wave's LLR tile is dead from here on
    for (int i = lane; i < 256; i += 64) s_crc[i] = c_tab.crc[i];
    __builtin_amdgcn_wave_barrier();
    traceback_group(dec, info, gi, fr, lane, s_crc, bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict);
#ifdef GR80211_TIMELINE
    if (lane == 0 && g < 8192) g_timeline[4 * g + 2] = wall_clock64();
#endif
}

template <bool TWO_TILES, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_decode(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                               const int32_t *__restrict__ slot_frame,
                                               const GroupInfo *__restrict__ groups, int n_groups,
                                               uint32_t *__restrict__ dec, uint32_t *__restrict__ bitsw,
                                               uint8_t *__restrict__ out, int out_stride, int32_t *__restrict__ out_len,
                                               int32_t *__restrict__ n_pdu, uint8_t *__restrict__ bits_out, int bits_stride,
                                               int strict, const SchedTotals *__restrict__ totals)
{
    decode_wave<TWO_TILES, WAVES>(llr, info, slot_frame, groups, n_groups, dec, bitsw, out, out_stride, out_len, n_pdu,
                                  bits_out, bits_stride, strict, totals);
}

// k_decode<false, 1> held to 216 registers per lane (the attribute counts in pairs on this target): two such waves
// leave 80 of a SIMD's 512 registers, which is room for a wave of k_preamble or k_demod of the next batch.
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(108)))
void k_decode_lean(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                   const int32_t *__restrict__ slot_frame, const GroupInfo *__restrict__ groups, int n_groups,
                   uint32_t *__restrict__ dec, uint32_t *__restrict__ bitsw, uint8_t *__restrict__ out, int out_stride,
                   int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu, uint8_t *__restrict__ bits_out,
                   int bits_stride, int strict, const SchedTotals *__restrict__ totals)
{
    decode_wave<false, 1>(llr, info, slot_frame, groups, n_groups, dec, bitsw, out, out_stride, out_len, n_pdu, bits_out,
                          bits_stride, strict, totals);
}

// The same with the tile loader that folds soft bits out of axis values (experiment, GR80211_AXIS_LLR=1).
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(108)))
void k_decode_lean_axis(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                        const int32_t *__restrict__ slot_frame, const GroupInfo *__restrict__ groups, int n_groups,
                        uint32_t *__restrict__ dec, uint32_t *__restrict__ bitsw, uint8_t *__restrict__ out, int out_stride,
                        int32_t *__restrict__ out_len, int32_t *__restrict__ n_pdu, uint8_t *__restrict__ bits_out,
                        int bits_stride, int strict, const SchedTotals *__restrict__ totals)
{
    decode_wave<false, 1, true>(llr, info, slot_frame, groups, n_groups, dec, bitsw, out, out_stride, out_len, n_pdu, bits_out,
                                bits_stride, strict, totals);
}

// =========================================================================================
// k_decode_split2: forward pass by two waves per group (split_forward.h), then the walk back by the wave that
// stored the group's survivor words (wave 0 stores every record, so it reads back only what it stored itself) while
// the other wave's SIMD goes to the next workgroup.  For batches of more groups than the four-wave form is good for
// and at most one per SIMD: a lone one-trellis wave leaves half of its SIMD's issue slots unused.
// =========================================================================================
__global__ __launch_bounds__(128) void k_decode_split2(const float *__restrict__ llr, const gr80211_frame_info *__restrict__ info,
                                                       const int32_t *__restrict__ slot_frame,
                                                       const GroupInfo *__restrict__ groups, int n_groups,
                                                       uint32_t *__restrict__ dec, uint32_t *__restrict__ bitsw,
                                                       uint8_t *__restrict__ out, int out_stride, int32_t *__restrict__ out_len,
                                                       int32_t *__restrict__ n_pdu, uint8_t *__restrict__ bits_out, int bits_stride,
                                                       int strict, const SchedTotals *__restrict__ totals)
{
    n_groups = groups_to_run(n_groups, totals);
    if ((int)blockIdx.x >= n_groups) return;  // whole workgroup
    const int g = n_groups - 1 - (int)blockIdx.x;  // longest trellises first, as k_decode
    const int lane = threadIdx.x & 63;
    const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float tile[64 * TILE_STRIDE];
    __shared__ float xch[2 * XCH2_BUF];
    const GroupInfo gi = groups[g];
    const int fr = slot_frame[g * 64 + lane];
    const int myn = fr >= 0 ? info[fr].total : 0;
    const unsigned long long mybase = myn > 0 ? (unsigned long long)info[fr].llr_off : 0ull;
    uint32_t *d = dec + gi.dec_off;
    switch (gi.cr) {
    case CR_12: split2_dispatch<CR_12>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_23: split2_dispatch<CR_23>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    case CR_34: split2_dispatch<CR_34>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    default: split2_dispatch<CR_56>(wi, llr, mybase, myn, gi.nchunks, d, tile, xch, lane); break;
    }
    // the other wave's last look at the soft-bit tile came before the last barrier of the forward pass
    if (wi != 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t *s_crc = reinterpret_cast<uint32_t *>(tile);
    for (int i = lane; i < 256; i += 64) s_crc[i] = c_tab.crc[i];
    __builtin_amdgcn_wave_barrier();
    traceback_group(dec, info, gi, fr, lane, s_crc, bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict);
}

// =========================================================================================
// k_ndp_emit: the channel message of a sounding NDP (lib/decode_impl.cc:100-121): type byte 20, length 1024,
// then the 64 + 64 time samples of the two VHT-LTF FFT windows as `signal` forwards them, i.e. CFO-derotated
// (lib/demod_impl.cc:397-398: copies of sig1[8..72) and sig1[88..152) of the samples after VHT-STF).
// One wave per frame, lane = sample of each window.
// =========================================================================================
__global__ __launch_bounds__(256) void k_ndp_emit(const c32 *__restrict__ iq0, uint64_t arena_len,
                                                  const gr80211_frame_desc *__restrict__ descs,
                                                  int n, const gr80211_frame_info *__restrict__ info,
                                                  uint8_t *__restrict__ out, int out_stride, int32_t *__restrict__ out_len,
                                                  int32_t *__restrict__ n_pdu, const SchedTotals *__restrict__ totals)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n) return;
    if (totals && totals->n_groups < 0) return;  // a refused batch publishes nothing, channel messages included
    const gr80211_frame_info fi = info[f];
    if (fi.status != GR80211_ST_NDP) return;
    constexpr int NBYTES = 128 * 8;
    if (out_stride < NBYTES + 3) return;  // no room: nothing is published, as for any PDU that does not fit
    const int lane = threadIdx.x & 63;
    const gr80211_frame_desc d = descs[f];
    const c32 *in0, *unused;
    const int navail = frame_origin(d, arena_len, iq0, nullptr, &in0, &unused);
    int nvalid = fi.nsamp;
    if (nvalid > navail - 224) nvalid = navail - 224;
    uint8_t *o = out + (size_t)f * out_stride;
    for (int w = 0; w < 2; w++) {
        const c32 v = load_samp(in0, fi.data_off + w * 80 + SHIFT + lane, nvalid, d.rad);
        // byte-wise: the slot is only byte aligned (3-byte header)
        const uint32_t re = __float_as_uint(v.x), im = __float_as_uint(v.y);
        uint8_t *p = o + 3 + (w * 64 + lane) * 8;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            p[b] = (uint8_t)(re >> (8 * b));
            p[4 + b] = (uint8_t)(im >> (8 * b));
        }
    }
    if (lane == 0) {
        o[0] = 20;  // C8P_F_VHT_CHAN
        o[1] = (uint8_t)(NBYTES % 256);
        o[2] = (uint8_t)(NBYTES / 256);
        out_len[f] = NBYTES + 3;
        n_pdu[f] = 1;
    }
}

// Tags of `signal` -> descriptors of the kind the kernels above read.
__global__ void k_sig_to_desc(const gr80211_sig_desc *__restrict__ sig, int n, gr80211_frame_desc *__restrict__ desc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gr80211_sig_desc sd = sig[i];
    gr80211_frame_desc d;
    d.sync_off = sd.off;
    d.n_avail = sd.n_avail;
    d.rad = 0.0f;  // `signal` has taken the CFO out
    d.snr = sd.snr;
    d.rssi = sd.rssi;
    d.mu_pos = sd.mu_pos;
    d.reserved = DESC_AFTER_SIGNAL;
    desc[i] = d;
}

// The items `signal` writes for one frame (lib/signal_impl.cc:164-202): nsamp samples after L-SIG, CFO taken out,
// then the 320 items of padding (zeros here, left unwritten by the reference).  One workgroup per frame.
__global__ __launch_bounds__(256) void k_signal_copy(const c32 *__restrict__ iq0, const c32 *__restrict__ iq1,
                                                     uint64_t arena_len, const gr80211_frame_desc *__restrict__ descs,
                                                     int n, const gr80211_frame_info *__restrict__ info,
                                                     const unsigned long long *__restrict__ out_off,
                                                     c32 *__restrict__ out0, c32 *__restrict__ out1,
                                                     unsigned long long out_len)
{
    const int f = blockIdx.x;
    if (f >= n) return;
    const gr80211_frame_info fi = info[f];
    if (fi.status == GR80211_ST_DROP_LSIG) return;
    const gr80211_frame_desc d = descs[f];
    const c32 *in0, *in1;
    const int navail = frame_origin(d, arena_len, iq0, iq1, &in0, &in1);
    int nvalid = fi.nsamp;
    if (nvalid > navail - 224) nvalid = navail - 224;
    const unsigned long long o = out_off[f];
    const int total = fi.nsamp + 320;
    if (o > out_len || (unsigned long long)total > out_len - o) return;  // no room: the caller sized the arena wrongly
    for (int k = threadIdx.x; k < total; k += blockDim.x) {
        out0[o + k] = k < fi.nsamp ? load_samp(in0, k, nvalid, d.rad) : mk(0.0f, 0.0f);
        if (out1) out1[o + k] = k < fi.nsamp && in1 ? load_samp(in1, k, nvalid, d.rad) : mk(0.0f, 0.0f);
    }
}

// What k_preamble left for the frames path[0..n) names, moved to places 0..n-1 of a second set of buffers: the
// capture path runs the preamble over every sync byte to learn the frames' spans, then goes on with the frames
// `signal` would deal with, whose records and channels it already has.  One wave per frame.
__global__ __launch_bounds__(64) void k_pre_gather(const int32_t *__restrict__ path, int n,
                                                   const gr80211_frame_info *__restrict__ info_in,
                                                   const c32 *__restrict__ chan1_in, const c32 *__restrict__ chan2_in,
                                                   gr80211_frame_info *__restrict__ info_out, c32 *__restrict__ chan1_out,
                                                   c32 *__restrict__ chan2_out)
{
    const int m = blockIdx.x;
    if (m >= n) return;
    const int i = path[m], lane = threadIdx.x;
    if (lane == 0) info_out[m] = info_in[i];
    chan1_out[(size_t)m * 64 + lane] = chan1_in[(size_t)i * 64 + lane];
    if (chan2_in)
        for (int k = lane; k < CH2; k += 64) chan2_out[(size_t)m * CH2 + k] = chan2_in[(size_t)i * CH2 + k];
}

__global__ void k_clear_outputs(int32_t *a, int32_t *b, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; b[i] = 0; }
}

// =========================================================================================
// launchers
// =========================================================================================
void launch_preamble(const void *iq0, const void *iq1, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                     gr80211_frame_info *info, void *chan1, void *chan2, const gr80211_sig_desc *sig, const void *chan_in,
                     void *chanl_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_preamble, dim3(n), dim3(64), 0, s, (const c32 *)iq0, (const c32 *)iq1, arena_len, desc, n,
                       info, (c32 *)chan1, (c32 *)chan2, sig, (const c32 *)chan_in, (c32 *)chanl_out);
}
void launch_sig_to_desc(const gr80211_sig_desc *sig, int n, gr80211_frame_desc *desc, hipStream_t s)
{
    hipLaunchKernelGGL(k_sig_to_desc, dim3((n + 255) / 256), dim3(256), 0, s, sig, n, desc);
}
void launch_signal_copy(const void *iq0, const void *iq1, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                        const gr80211_frame_info *info, const uint64_t *out_off, void *out0, void *out1, uint64_t out_len,
                        hipStream_t s)
{
    hipLaunchKernelGGL(k_signal_copy, dim3(n), dim3(256), 0, s, (const c32 *)iq0, (const c32 *)iq1, arena_len, desc, n,
                       info, (const unsigned long long *)out_off, (c32 *)out0, (c32 *)out1, (unsigned long long)out_len);
}
void launch_desc_to_info(const gr80211_dec_desc *desc, int n, uint64_t llr_floats, gr80211_frame_info *info, hipStream_t s)
{
    hipLaunchKernelGGL(k_desc_to_info, dim3((n + 255) / 256), dim3(256), 0, s, desc, n, (unsigned long long)llr_floats, info);
}
void launch_schedule(gr80211_frame_info *info, int n, int assign_llr, int32_t *slot_frame, int slot_cap,
                     GroupInfo *groups, int group_cap, SchedTotals *totals, uint32_t *ws, unsigned long long *blk_llr,
                     SchedCaps caps, hipStream_t s)
{
    static const bool one = [] {  // GR80211_SCHED_ONE=0: small batches take the four-kernel form too (A/B measurements)
        const char *e = getenv("GR80211_SCHED_ONE");
        return !e || e[0] != '0';
    }();
    if (n <= SCHED_ONE_MAX && one) {
        hipLaunchKernelGGL(k_sched_one, dim3(1), dim3(1024), 0, s, info, n, assign_llr, slot_frame, slot_cap, groups, group_cap,
                           totals, (unsigned long long)caps.llr_floats, (unsigned long long)caps.dec_words,
                           (unsigned long long)caps.bits_words);
        return;
    }
    const int nblk = (n + SCHED_BLOCK - 1) / SCHED_BLOCK;
    const int nws = SCHED_BINS + 8;
    hipLaunchKernelGGL(k_zero_u32, dim3((nws + 255) / 256), dim3(256), 0, s, ws, nws);
    hipLaunchKernelGGL(k_sched_hist, dim3(nblk), dim3(SCHED_BLOCK), 0, s, info, n, ws, blk_llr);
    hipLaunchKernelGGL(k_sched_scan, dim3(1), dim3(1024), 0, s, ws, blk_llr, nblk, assign_llr, slot_cap, group_cap,
                       slot_frame, groups, totals, (unsigned long long)caps.llr_floats);
    hipLaunchKernelGGL(k_sched_assign, dim3(nblk), dim3(SCHED_BLOCK), 0, s, info, n, assign_llr, ws, blk_llr,
                       slot_frame, groups, totals);
    hipLaunchKernelGGL(k_sched_groups, dim3(1), dim3(1024), 0, s, groups, totals, (unsigned long long)caps.dec_words,
                       (unsigned long long)caps.bits_words);
}
void launch_demod(const void *iq0, const void *iq1, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                  const gr80211_frame_info *info, const void *chan1, const void *chan2, float *llr,
                  const SchedTotals *totals, hipStream_t s, bool axis)
{
    if (axis) {
        hipLaunchKernelGGL((k_demod<1, true>), dim3(n), dim3(256), 0, s, (const c32 *)iq0, (const c32 *)iq1, arena_len, desc, n,
                           info, (const c32 *)chan1, (const c32 *)chan2, llr, totals);
        if (iq1)
            hipLaunchKernelGGL((k_demod<2, true>), dim3(n), dim3(256), 0, s, (const c32 *)iq0, (const c32 *)iq1, arena_len,
                               desc, n, info, (const c32 *)chan1, (const c32 *)chan2, llr, totals);
        return;
    }
    hipLaunchKernelGGL((k_demod<1, false>), dim3(n), dim3(256), 0, s, (const c32 *)iq0, (const c32 *)iq1, arena_len, desc, n,
                       info, (const c32 *)chan1, (const c32 *)chan2, llr, totals);
    // two-stream frames only exist where a second antenna does
    if (iq1)
        hipLaunchKernelGGL((k_demod<2, false>), dim3(n), dim3(256), 0, s, (const c32 *)iq0, (const c32 *)iq1, arena_len, desc,
                           n, info, (const c32 *)chan1, (const c32 *)chan2, llr, totals);
}
void launch_viterbi(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                    const GroupInfo *groups, int n_groups, uint32_t *dec, const SchedTotals *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_viterbi, dim3(n_groups), dim3(64), 0, s, llr, info, slot_frame, groups, n_groups, dec, totals);
}
void launch_traceback(const uint32_t *dec, const gr80211_frame_info *info, const int32_t *slot_frame,
                      const GroupInfo *groups, int n_groups, uint32_t *bitsw, uint8_t *out, int out_stride,
                      int32_t *out_len, int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict,
                      const SchedTotals *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_traceback, dim3(n_groups), dim3(64), 0, s, dec, info, slot_frame, groups, n_groups, bitsw,
                       out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict, totals);
}
void launch_decode(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame,
                   const GroupInfo *groups, int n_groups, uint32_t *dec, uint32_t *bitsw, uint8_t *out, int out_stride,
                   int32_t *out_len, int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict,
                   const SchedTotals *totals, bool large_form, hipStream_t s, bool axis)
{
    if (axis) {
        hipLaunchKernelGGL(k_decode_lean_axis, dim3(n_groups), dim3(64), 0, s, llr, info, slot_frame, groups, n_groups, dec,
                           bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict, totals);
        return;
    }
    // A batch of at most one group per SIMD runs fastest with exactly one wave on every SIMD (a lone wave finishes
    // its trellis in three quarters of the time a paired one needs): workgroups of four waves with two LDS tiles each,
    // 130 KB per workgroup, so a CU takes one workgroup and 256 workgroups cover the 256 CUs.  Larger batches run one
    // group per workgroup with one tile (16 KB), two waves per SIMD.
    if (n_groups <= 1024 && !large_form)
        hipLaunchKernelGGL((k_decode<true, DEC_WAVES>), dim3((n_groups + DEC_WAVES - 1) / DEC_WAVES), dim3(64 * DEC_WAVES), 0, s,
                           llr, info, slot_frame, groups, n_groups, dec, bitsw, out, out_stride, out_len, n_pdu, bits_out,
                           bits_stride, strict, totals);
    else
        hipLaunchKernelGGL(k_decode_lean, dim3(n_groups), dim3(64), 0, s, llr, info, slot_frame, groups, n_groups, dec,
                           bitsw, out, out_stride, out_len, n_pdu, bits_out, bits_stride, strict, totals);
}
void launch_decode_split2(const float *llr, const gr80211_frame_info *info, const int32_t *slot_frame, const GroupInfo *groups,
                          int n_groups, uint32_t *dec, uint32_t *bitsw, uint8_t *out, int out_stride, int32_t *out_len,
                          int32_t *n_pdu, uint8_t *bits_out, int bits_stride, int strict, const SchedTotals *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_decode_split2, dim3(n_groups), dim3(128), 0, s, llr, info, slot_frame, groups, n_groups, dec, bitsw, out,
                       out_stride, out_len, n_pdu, bits_out, bits_stride, strict, totals);
}
void launch_ndp_emit(const void *iq0, uint64_t arena_len, const gr80211_frame_desc *desc, int n,
                     const gr80211_frame_info *info, uint8_t *out, int out_stride, int32_t *out_len, int32_t *n_pdu,
                     const SchedTotals *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_ndp_emit, dim3((n + 3) / 4), dim3(256), 0, s, (const c32 *)iq0, arena_len, desc, n, info, out, out_stride,
                       out_len, n_pdu, totals);
}
void launch_pre_gather(const int32_t *path, int n, const gr80211_frame_info *info_in, const void *chan1_in,
                       const void *chan2_in, gr80211_frame_info *info_out, void *chan1_out, void *chan2_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_pre_gather, dim3(n), dim3(64), 0, s, path, n, info_in, (const c32 *)chan1_in, (const c32 *)chan2_in,
                       info_out, (c32 *)chan1_out, (c32 *)chan2_out);
}
void launch_clear_outputs(int32_t *out_len, int32_t *n_pdu, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_clear_outputs, dim3((n + 255) / 256), dim3(256), 0, s, out_len, n_pdu, n);
}

}  // namespace gr80211
