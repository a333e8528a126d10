// Shared definitions of the MI355X RX hot path: constant tables and device helpers.
// gfx950 only; wavefront = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gr80211_rx.h"

namespace gr80211 {

enum { QAM_BPSK = 0, QAM_QBPSK = 1, QAM_QPSK = 2, QAM_16 = 3, QAM_64 = 4, QAM_256 = 5 };
enum { CR_12 = 0, CR_23 = 1, CR_34 = 2, CR_56 = 3 };

constexpr int SHIFT = 8;            // FFT window starts 8 samples into the 16-sample GI (lib/cloud80211phy.h:33)
constexpr float NEG_INIT = -1000000000000000.0f;  // start metric of unreachable states (lib/decode_impl.cc:174)

// Constant tables, built on the host from the IEEE 802.11-2016 definitions and uploaded once.
struct Tables {
    float ltf_l[64];        // L-LTF sign per FFT bin (0 = unused tone)          17.3.3
    float ltf_nl[64];       // HT/VHT-LTF sign per FFT bin                       19.3.9.4.6
    float pilot_p[128];     // pilot polarity p_0..126                            17.3.5.10
    int16_t lsig_demap[64]; // FFT bin -> deinterleaved SIG bit position, -1 = no data
    int16_t data_idx_l[64]; // FFT bin -> data subcarrier index 0..47, -1
    int16_t data_idx_nl[64];// FFT bin -> data subcarrier index 0..51, -1
    int16_t deint_l[4][288];   // out[map[j]] = in[j]                             17.3.5.7
    int16_t deint_nl[5][416];  //                                                 19.3.11.8.x
    int16_t deint_nl2[5][416]; // second spatial stream (rotation 11)
    int16_t intl_l[4][288];    // the interleavers themselves: coded bit k is sent at position map[k]
    int16_t intl_nl[5][416];
    int16_t intl_nl2[5][416];
    float tw_re[32], tw_im[32];// exp(-2*pi*i*k/64)
    uint32_t crc[256];         // reflected CRC-32 0xEDB88320
};

void build_tables(Tables &t);
uint32_t tables_checksum(const Tables &t);

// ----------------------------------------------------------------------------------------
// Convolutional code (17.3.5.6): generators 133/171 octal.  Decoder state i holds the six
// previous input bits with the newest in bit 5; next(i,b) = (i>>1)|(b<<5).
// sv_out returns 2*A+B for the branch leaving state i with input bit b.
__host__ __device__ constexpr int parity7(int v)
{
    int p = 0;
    for (int k = 0; k < 7; k++) p ^= (v >> k) & 1;
    return p;
}
__host__ __device__ constexpr int sv_out(int i, int b)
{
    int r = b;
    for (int m = 0; m < 6; m++) r |= ((i >> (5 - m)) & 1) << (m + 1);
    return 2 * parity7(r & 0155) + parity7(r & 0117);
}

#ifdef __HIPCC__
struct c32 {
    float x, y;
};
__device__ __forceinline__ c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cneg(c32 a) { return mk(-a.x, -a.y); }
__device__ __forceinline__ c32 cconj(c32 a) { return mk(a.x, -a.y); }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c32 cmulf(c32 a, float s) { return mk(a.x * s, a.y * s); }
__device__ __forceinline__ c32 cdivf(c32 a, float s) { return mk(a.x / s, a.y / s); }
// Complex quotient by Smith's method, the operation order of the C runtime's float complex
// division that the reference's std::complex<float> operator/ resolves to.
__device__ __forceinline__ c32 cdiv(c32 p, c32 q)
{
    // if |c| < |d|: r = c/d, den = c*r + d, x = (a*r + b)/den, y = (b*r - a)/den
    // else        : r = d/c, den = d*r + c, x = (b*r + a)/den, y = (b - a*r)/den
    // written with selects so that a wave never executes both branches (3 divisions, not 6)
    const float a = p.x, b = p.y, c = q.x, d = q.y;
    const bool sw = fabsf(c) < fabsf(d);
    const float num = sw ? c : d, den0 = sw ? d : c;
    const float ratio = num / den0;
    const float denom = (num * ratio) + den0;
    const float xm = sw ? a : b, xa = sw ? b : a;
    const float x = ((xm * ratio) + xa) / denom;
    const float yr = sw ? (b * ratio) : (a * ratio);
    const float y = (sw ? (yr - a) : (b - yr)) / denom;
    return mk(x, y);
}
// |z| as the C library's hypotf computes it: double-precision sum of squares, one rounding.
__device__ __forceinline__ float cabs32(c32 a)
{
    double x = (double)a.x, y = (double)a.y;
    return (float)sqrt(x * x + y * y);
}
__device__ __forceinline__ c32 shfl_c(c32 v, int src) { return mk(__shfl(v.x, src), __shfl(v.y, src)); }
__device__ __forceinline__ int brev6(int v) { return (int)(__brev((unsigned)v) >> 26); }
#endif

}  // namespace gr80211
