// Frame detection on MI355X: the work of presiso -> trigger -> sync of the reference's receive flowgraph
// (examples/presiso.grc:68-137,218-229; lib/trigger_impl.cc:59-117; lib/sync_impl.cc:61-196) over a whole
// capture at once.  gfx950, wave64.
//
// The three blocks are sequential state machines over the sample stream; here they become
//   k_det_ac      one pass over the samples (the HBM-bound part): the normalised 16-lag autocorrelation of
//                 presiso, thresholded at trigger's 0.3 into one bit per sample;
//   k_det_edges   rising / falling edges of that bit stream = trigger's plateaus ("runs"), in order;
//   k_det_trigger trigger's plateau counter / 80-sample countdown and sync's "111 items consumed per
//                 trigger" rule, replayed per chain of runs that are close enough to influence each other;
//   k_det_sync    per trigger (one wave each): the conjugate average sync would hold at that moment, the 64-lag
//                 L-LTF autocorrelation over 111 positions, its 80 % shoulder midpoint, CFO, SNR, RSSI;
//   k_det_emit    the sync bytes that were produced, in stream order, as frame descriptors.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detect_kernels.h"

namespace gr80211 {

namespace {

constexpr float PLATEAU_AC = 0.3f;   // lib/trigger_impl.cc:78
constexpr int PLATEAU_LEN = 20;      // lib/trigger_impl.cc:87
constexpr int COUNT_DOWN = 80;       // lib/trigger_impl.cc:91
constexpr int SYNC_BUF = 240;        // SYNC_MAX_BUF_LEN, lib/sync_impl.h:30
constexpr int SYNC_RES = SYNC_BUF - 128 - 1;  // SYNC_MAX_RES_LEN = 111, lib/sync_impl.h:31
// A run can only be influenced by what happened before it if the previous run ended less than this many samples
// before its own 21st sample: a countdown started in the previous run fires at most 79 samples after that run's
// last sample and sync skips the 111 items that follow a trigger byte, so from 20 + 79 + 111 = 210 samples on
// neither the countdown, nor the skipped items, nor the conjugate value sync holds can reach into this run.
constexpr long long INDEPENDENT_GAP = 256;

struct c32d {
    float x, y;
};

__device__ __forceinline__ c32d ld(const c32d *x, long long i, long long n)
{
    c32d z;
    z.x = 0.0f;
    z.y = 0.0f;
    return (i < 0 || i >= n) ? z : x[i];  // before the capture: the zero history GNU Radio blocks start with
}

// |z| as the C library's hypotf computes it (what std::abs(std::complex<float>) resolves to).
__device__ __forceinline__ float cabs_ref(float x, float y)
{
    const double dx = (double)x, dy = (double)y;
    return (float)sqrt(dx * dx + dy * dy);
}

}  // namespace

// -----------------------------------------------------------------------------------------
// k_det_ac: bit n of `mask` = ( |sum_{k<48} x[n-k-16] conj(x[n-k])| / sum_{k<64} |x[n-k]|^2 > 0.3 ).
// One wave walks AC_ROWS rows of 256 consecutive samples, four per lane (coalesced 16-byte loads, the next row in
// flight while this one is worked on).  A lane sums its four lag products and powers in order, the lane totals
// are scanned across the wave (six DPP adds per quantity: a quarter of what one sample per lane would need, and
// the DPP adds are what the kernel is priced by), and every sample's prefix sum of the row goes to a two-row ring
// in LDS.  The window sums are differences of prefix sums: over the last 48 samples P[i] - P[i-48] (12 lanes back),
// over the last 64 P[i] - P[i-64] (16 lanes back), reaching into the previous row through its total.  No sum is
// longer than 256 terms, so nothing drifts the way the reference's running sums do.  Reads 8 B per sample from
// HBM once (x[n-16] is read again, from cache), writes 1 bit per sample.
// -----------------------------------------------------------------------------------------
constexpr int AC_SPL = 4;             // samples per lane
constexpr int AC_ROW = 64 * AC_SPL;   // samples per row
constexpr int AC_ROWS = 16;           // rows per wave
constexpr int AC_WAVES = 4;           // waves per workgroup

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float scan_step(float v)
{
    // lanes the control does not reach add 0
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float wave_scan(float v)
{
    v = scan_step<0x111, 0xf>(v);  // row_shr:1
    v = scan_step<0x112, 0xf>(v);  // row_shr:2
    v = scan_step<0x114, 0xf>(v);  // row_shr:4
    v = scan_step<0x118, 0xf>(v);  // row_shr:8
    v = scan_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = scan_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

struct row_sums {
    float sr, si, sp;  // inclusive scans of the lag product (re, im) and of the power
    float tr, ti, tp;  // row totals
};

// scans of the 64-sample row whose samples are xc, given the previous row's samples xq (for x[n-16] of the first
// 16 lanes); used where single plateaus are looked at again (k_det_sync)
__device__ __forceinline__ row_sums scan_row(c32d xc, c32d xq, int lane)
{
    const int src = (lane + 48) & 63;  // lane - 16, wrapping into the previous row
    const float ax = __shfl(xc.x, src), ay = __shfl(xc.y, src);
    const float bx = __shfl(xq.x, src), by = __shfl(xq.y, src);
    c32d a;  // x[n-16]
    a.x = lane >= 16 ? ax : bx;
    a.y = lane >= 16 ? ay : by;
    // multiply_conjugate_cc: a * conj(x)
    const float mr = a.x * xc.x + a.y * xc.y;
    const float mi = a.y * xc.x - a.x * xc.y;
    const float pw = xc.x * xc.x + xc.y * xc.y;
    row_sums r;
    r.sr = wave_scan(mr);
    r.si = wave_scan(mi);
    r.sp = wave_scan(pw);
    r.tr = __shfl(r.sr, 63);
    r.ti = __shfl(r.si, 63);
    r.tp = __shfl(r.sp, 63);
    return r;
}

struct quad {
    c32d v[AC_SPL];
};

// Samples [i, i + 4), i a multiple of 4.  INSIDE: the whole quad lies in the capture and the capture starts on a
// 16-byte boundary, so it travels as two 16-byte loads that nothing guards (a wave in the middle of the capture
// then issues the same memory instructions for every row, and the wait for a prefetched row can leave younger
// loads in flight).  Otherwise sample by sample; outside the capture samples read as zero.
template <bool INSIDE>
__device__ __forceinline__ quad ld4(const c32d *x, long long i, long long n)
{
    quad q;
    if (INSIDE) {
        const float4 a = *reinterpret_cast<const float4 *>(x + i);
        const float4 b = *reinterpret_cast<const float4 *>(x + i + 2);
        q.v[0].x = a.x; q.v[0].y = a.y; q.v[1].x = a.z; q.v[1].y = a.w;
        q.v[2].x = b.x; q.v[2].y = b.y; q.v[3].x = b.z; q.v[3].y = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < AC_SPL; j++) q.v[j] = ld(x, i + j, n);
    }
    return q;
}

__device__ __forceinline__ float lane63(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// prefix sums of one row into ring[par]; the row totals come back
__device__ __forceinline__ void put_row(float (*ring)[3][AC_ROW], const quad &xc, const quad &xa, int par, int lane,
                                        float &tr, float &ti, float &tp)
{
    float mr[AC_SPL], mi[AC_SPL], pw[AC_SPL];
#pragma unroll
    for (int j = 0; j < AC_SPL; j++) {
        const c32d a = xa.v[j], c = xc.v[j];
        mr[j] = a.x * c.x + a.y * c.y;  // multiply_conjugate_cc: x[n-16] * conj(x[n])
        mi[j] = a.y * c.x - a.x * c.y;
        pw[j] = c.x * c.x + c.y * c.y;
    }
#pragma unroll
    for (int j = 1; j < AC_SPL; j++) {
        mr[j] += mr[j - 1];
        mi[j] += mi[j - 1];
        pw[j] += pw[j - 1];
    }
    const float sr = wave_scan(mr[AC_SPL - 1]), si = wave_scan(mi[AC_SPL - 1]), sp = wave_scan(pw[AC_SPL - 1]);
    const float er = sr - mr[AC_SPL - 1], ei = si - mi[AC_SPL - 1], ep = sp - pw[AC_SPL - 1];  // lanes before this one
    float4 vr, vi, vp;
    vr.x = er + mr[0]; vr.y = er + mr[1]; vr.z = er + mr[2]; vr.w = er + mr[3];
    vi.x = ei + mi[0]; vi.y = ei + mi[1]; vi.z = ei + mi[2]; vi.w = ei + mi[3];
    vp.x = ep + pw[0]; vp.y = ep + pw[1]; vp.z = ep + pw[2]; vp.w = ep + pw[3];
    *reinterpret_cast<float4 *>(&ring[par][0][lane * AC_SPL]) = vr;
    *reinterpret_cast<float4 *>(&ring[par][1][lane * AC_SPL]) = vi;
    *reinterpret_cast<float4 *>(&ring[par][2][lane * AC_SPL]) = vp;
    tr = lane63(vr.w);
    ti = lane63(vi.w);
    tp = lane63(vp.w);
}

// The rows of one wave; their mask words are left in words[] (8 per row).
template <bool INSIDE>
__device__ __forceinline__ void ac_rows(const c32d *__restrict__ x, long long n, long long base, int lane,
                                        float (*ring)[3][AC_ROW], uint32_t *words)
{
    // the row before this wave's first one seeds the window sums
    float ptr_, pti_, ptp_;
    {
        const quad xp = ld4<INSIDE>(x, base - AC_ROW + lane * AC_SPL, n);
        const quad xq = ld4<INSIDE>(x, base - AC_ROW + lane * AC_SPL - 16, n);
        put_row(ring, xp, xq, 1, lane, ptr_, pti_, ptp_);
    }
    quad xn = ld4<INSIDE>(x, base + lane * AC_SPL, n);
    quad an = ld4<INSIDE>(x, base + lane * AC_SPL - 16, n);
    for (int r = 0; r < AC_ROWS; r++) {
        const long long n0 = base + (long long)r * AC_ROW;
        if (!INSIDE && n0 >= n) break;
        const int par = r & 1;
        const quad xc = xn, xa = an;
        // the next row travels while this one is worked on (INSIDE: the row after the wave's last one exists too)
        xn = ld4<INSIDE>(x, n0 + AC_ROW + lane * AC_SPL, n);
        an = ld4<INSIDE>(x, n0 + AC_ROW + lane * AC_SPL - 16, n);
        float tr, ti, tp;
        __builtin_amdgcn_wave_barrier();
        put_row(ring, xc, xa, par, lane, tr, ti, tp);
        __builtin_amdgcn_wave_barrier();
        // prefix sums 48 and 64 samples back: 12 and 16 lanes back, in the previous row for the first lanes
        const bool in48 = lane >= 12, in64 = lane >= 16;
        const float4 cr = *reinterpret_cast<const float4 *>(&ring[par][0][lane * AC_SPL]);
        const float4 ci = *reinterpret_cast<const float4 *>(&ring[par][1][lane * AC_SPL]);
        const float4 cp = *reinterpret_cast<const float4 *>(&ring[par][2][lane * AC_SPL]);
        const float4 br = *reinterpret_cast<const float4 *>(&ring[in48 ? par : par ^ 1][0][((lane + 52) & 63) * AC_SPL]);
        const float4 bi = *reinterpret_cast<const float4 *>(&ring[in48 ? par : par ^ 1][1][((lane + 52) & 63) * AC_SPL]);
        const float4 bp = *reinterpret_cast<const float4 *>(&ring[in64 ? par : par ^ 1][2][((lane + 48) & 63) * AC_SPL]);
        const float cra[4] = {cr.x, cr.y, cr.z, cr.w}, cia[4] = {ci.x, ci.y, ci.z, ci.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
        const float bra[4] = {br.x, br.y, br.z, br.w}, bia[4] = {bi.x, bi.y, bi.z, bi.w}, bpa[4] = {bp.x, bp.y, bp.z, bp.w};
        uint32_t nib = 0;
#pragma unroll
        for (int j = 0; j < AC_SPL; j++) {
            const float wr = in48 ? cra[j] - bra[j] : cra[j] + (ptr_ - bra[j]);
            const float wi = in48 ? cia[j] - bia[j] : cia[j] + (pti_ - bia[j]);
            const float wp = in64 ? cpa[j] - bpa[j] : cpa[j] + (ptp_ - bpa[j]);
            // complex_to_mag, divide, "> 0.3" with the square root and the division taken out: |w|^2 > (0.3 wp)^2.
            // wp is a sum of squares (never negative); where it is zero so is the lag sum and the bit stays clear,
            // as 0/0 compares false.
            const float lim = PLATEAU_AC * wp;
            if ((INSIDE || n0 + lane * AC_SPL + j < n) && wr * wr + wi * wi > lim * lim) nib |= 1u << j;
        }
        // eight lanes make one mask word: OR the nibbles together inside each group of eight lanes
        uint32_t w = nib << (4 * (lane & 7));
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x104, 0xf, 0xf, false);  // row_shl:4 (lane l takes l + 4)
        if ((lane & 7) == 0) words[r * 8 + (lane >> 3)] = w;
        ptr_ = tr;
        pti_ = ti;
        ptp_ = tp;
    }
}

__global__ __launch_bounds__(64 * AC_WAVES) void k_det_ac(const c32d *__restrict__ x, long long n, uint32_t *__restrict__ mask,
                                                         int wide)
{
    // per wave: two rows x (re, im, power) x 256 prefix sums, and the wave's mask words until it is done
    __shared__ __attribute__((aligned(16))) float s_ring[AC_WAVES][2][3][AC_ROW];
    __shared__ uint32_t s_words[AC_WAVES][AC_ROWS * 8];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long)blockIdx.x * AC_WAVES + wv;
    const long long base = wave * (long long)(AC_ROWS * AC_ROW);
    if (base >= n) return;
    const long long nwords = (n + 31) >> 5;
    uint32_t *words = s_words[wv];
    for (int k = lane; k < AC_ROWS * 8; k += 64) words[k] = 0;
    __builtin_amdgcn_wave_barrier();
    // a wave whose rows, the row before them and the row after them lie inside the capture needs no guards
    const bool inside = wide && base >= AC_ROW + 16 && base + (long long)(AC_ROWS + 1) * AC_ROW <= n;
    if (inside) ac_rows<true>(x, n, base, lane, s_ring[wv], words);
    else ac_rows<false>(x, n, base, lane, s_ring[wv], words);
    __builtin_amdgcn_wave_barrier();
    const long long w0 = base >> 5;
    for (int k = lane; k < AC_ROWS * 8; k += 64)
        if (w0 + k < nwords) mask[w0 + k] = words[k];
}

// -----------------------------------------------------------------------------------------
// Edges of the plateau bit stream.  Pass 0 counts the rising edges of each block of words, pass 1 writes the
// rising and falling edge positions at the offsets an exclusive scan of the counts gave.
// A run r is [rise[r], fall[r]); a run still open at the end of the capture falls at n.
// -----------------------------------------------------------------------------------------
constexpr int EDGE_WORDS = 1024;  // mask words per workgroup
constexpr int EDGE_THREADS = 256;

__device__ __forceinline__ void edge_words(const uint32_t *mask, long long nwords, long long w, uint32_t &rise, uint32_t &fall)
{
    const uint32_t cur = w < nwords ? mask[w] : 0u;
    const uint32_t prev_top = (w > 0 && w - 1 < nwords) ? (mask[w - 1] >> 31) : 0u;
    const uint32_t before = (cur << 1) | prev_top;  // bit k = sample 32w+k-1
    rise = cur & ~before;
    fall = ~cur & before;
}

__global__ __launch_bounds__(EDGE_THREADS) void k_det_edge_count(const uint32_t *__restrict__ mask, long long nwords,
                                                                 uint32_t *__restrict__ blk_rise, uint32_t *__restrict__ blk_fall)
{
    __shared__ uint32_t s_r[EDGE_THREADS / 64], s_f[EDGE_THREADS / 64];
    const long long w0 = (long long)blockIdx.x * EDGE_WORDS;
    uint32_t nr = 0, nf = 0;
    for (int i = threadIdx.x; i < EDGE_WORDS; i += EDGE_THREADS) {
        uint32_t r, f;
        // one word past the end is looked at too: a run open at the end of the capture falls there
        edge_words(mask, nwords, w0 + i, r, f);
        if (w0 + i > nwords) r = f = 0;
        nr += __popc(r);
        nf += __popc(f);
    }
    for (int d = 32; d > 0; d >>= 1) {
        nr += __shfl_down(nr, d);
        nf += __shfl_down(nf, d);
    }
    if ((threadIdx.x & 63) == 0) {
        s_r[threadIdx.x >> 6] = nr;
        s_f[threadIdx.x >> 6] = nf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (int w = 0; w < EDGE_THREADS / 64; w++) {
            a += s_r[w];
            b += s_f[w];
        }
        blk_rise[blockIdx.x] = a;
        blk_fall[blockIdx.x] = b;
    }
}

// Exclusive scan of up to a few hundred thousand block counts by one workgroup; totals[slot] = sum.
__global__ __launch_bounds__(1024) void k_det_scan(uint32_t *__restrict__ a, uint32_t *__restrict__ b, int nblk,
                                                   uint32_t *__restrict__ totals, int slot)
{
    __shared__ uint32_t s_a[1024], s_b[1024];
    __shared__ uint32_t s_ca, s_cb;
    if (threadIdx.x == 0) s_ca = s_cb = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t va = i < nblk ? a[i] : 0u, vb = (b && i < nblk) ? b[i] : 0u;
        s_a[threadIdx.x] = va;
        s_b[threadIdx.x] = vb;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t ta = threadIdx.x >= d ? s_a[threadIdx.x - d] : 0u;
            const uint32_t tb = threadIdx.x >= d ? s_b[threadIdx.x - d] : 0u;
            __syncthreads();
            s_a[threadIdx.x] += ta;
            s_b[threadIdx.x] += tb;
            __syncthreads();
        }
        if (i < nblk) {
            a[i] = s_ca + s_a[threadIdx.x] - va;
            if (b) b[i] = s_cb + s_b[threadIdx.x] - vb;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            s_ca += s_a[1023];
            s_cb += s_b[1023];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[slot] = s_ca;
        totals[slot + 1] = s_cb;
    }
}

__global__ __launch_bounds__(EDGE_THREADS) void k_det_edge_emit(const uint32_t *__restrict__ mask, long long nwords, long long n,
                                                                const uint32_t *__restrict__ blk_rise,
                                                                const uint32_t *__restrict__ blk_fall, long long cap,
                                                                long long *__restrict__ rise, long long *__restrict__ fall)
{
    // one wave per workgroup walks the block's words in order so that positions come out sorted
    const int lane = threadIdx.x;
    if (lane >= 64) return;
    const long long w0 = (long long)blockIdx.x * EDGE_WORDS;
    long long pr = blk_rise[blockIdx.x], pf = blk_fall[blockIdx.x];
    for (int base = 0; base < EDGE_WORDS; base += 64) {
        const long long w = w0 + base + lane;
        uint32_t r, f;
        edge_words(mask, nwords, w, r, f);
        if (w > nwords) r = f = 0;
        int cr = __popc(r), cf = __popc(f);
        int xr = cr, xf = cf;
        for (int d = 1; d < 64; d <<= 1) {
            const int ur = __shfl_up(xr, d), uf = __shfl_up(xf, d);
            if (lane >= d) {
                xr += ur;
                xf += uf;
            }
        }
        long long or_ = pr + xr - cr, of = pf + xf - cf;
        while (r) {
            const int k = __ffs(r) - 1;
            r &= r - 1;
            if (or_ < cap) rise[or_] = w * 32 + k;
            or_++;
        }
        while (f) {
            const int k = __ffs(f) - 1;
            f &= f - 1;
            long long p = w * 32 + k;
            if (p > n) p = n;
            if (of < cap) fall[of] = p;
            of++;
        }
        pr += __shfl(xr, 63);
        pf += __shfl(xf, 63);
    }
}

// -----------------------------------------------------------------------------------------
// k_det_trigger: one thread per run; the thread of a run that nothing earlier can influence walks its chain.
//   trigger (lib/trigger_impl.cc:78-107): in a run that starts at a, the countdown starts at the first sample
//   i >= a+20 at which no countdown is running (a countdown started at s ends, and fires the trigger byte, at
//   s+79; the check at that very sample still sees it running) -- once per run -- and fires even if the run
//   ends earlier.
//   sync (lib/sync_impl.cc:76-93,94-131): a trigger byte it scans makes it consume 111 items from there, emitted
//   or not; trigger bytes inside those items are never seen; fewer than 240 items left = never processed.
// Per run: seen_at[r] = trigger position sync acts on (-1: none), scan_lo[r] = first item sync scanned after
// its previous trigger, prev_seen[r] = run of that previous trigger (-1: none).
// -----------------------------------------------------------------------------------------
__global__ void k_det_trigger(const long long *__restrict__ rise, const long long *__restrict__ fall, long long nruns,
                              long long n, long long *__restrict__ seen_at, long long *__restrict__ scan_lo,
                              long long *__restrict__ prev_seen)
{
    const long long r0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r0 >= nruns) return;
    const bool head = r0 == 0 || rise[r0] + PLATEAU_LEN > fall[r0 - 1] + INDEPENDENT_GAP;
    if (!head) return;
    long long fire_prev = -(1ll << 60), last_seen = -(1ll << 60), last_seen_run = -1;
    for (long long r = r0; r < nruns; r++) {
        const long long a = rise[r], e = fall[r];
        if (r > r0 && a + PLATEAU_LEN > fall[r - 1] + INDEPENDENT_GAP) break;  // next chain
        long long t = -1;
        if (e - a > PLATEAU_LEN) {
            long long s = a + PLATEAU_LEN;
            if (fire_prev + 1 > s) s = fire_prev + 1;
            if (s <= e - 1) {
                t = s + COUNT_DOWN - 1;
                fire_prev = t;
            }
        }
        long long seen = -1;
        if (t >= 0 && t < n && t >= last_seen + SYNC_RES && n - t >= SYNC_BUF) seen = t;
        seen_at[r] = seen;
        scan_lo[r] = last_seen < 0 ? 0 : last_seen + SYNC_RES;
        prev_seen[r] = last_seen_run;
        if (seen >= 0) {
            last_seen = seen;
            last_seen_run = r;
        }
    }
}

// -----------------------------------------------------------------------------------------
// k_det_sync: one wave per run whose trigger byte sync acts on.
//   1. The conjugate average sync holds when it meets the byte (lib/sync_impl.cc:85-88): the 48-sample lag sum at
//      the last "new maximum" flag (trigger bit 1, lib/trigger_impl.cc:81-86) it scanned -- the first position of
//      the maximum of the autocorrelation over the latest plateau before the byte.  The autocorrelation is
//      recomputed with k_det_ac's row scans on the same 64-sample grid, so it is the value the plateau bits came
//      from.  Flags inside the 111 items sync skipped after an earlier byte were never seen; then the value found
//      for that earlier byte is still in place.
//   2. The 64-lag L-LTF autocorrelation over 111 positions (lib/sync_impl.cc:139-171): the 240 samples go to
//      LDS, lanes compute |x|^2 and the lag products in parallel, then the three running sums are stepped in the
//      reference's order (a short dependent chain of float adds; every lane steps it and keeps the sums of its
//      own positions), and lanes finish their positions in parallel.
//   3. Maximum (first), 80 % shoulders by ballot, midpoint; CFO (lib/sync_impl.cc:173-191) with the 128 rotations
//      spread over the lanes and the 64 products summed in order.
// -----------------------------------------------------------------------------------------
namespace {

// first position of the maximum of ac over [a, hi) if it is >= lo (else -1) and the lag sum there
__device__ long long plateau_peak(const c32d *x, long long n, long long a, long long hi, long long lo, int lane, float &vr,
                                  float &vi)
{
    const long long row0 = a >> 6, row1 = (hi - 1) >> 6;
    c32d xq = ld(x, (row0 - 2) * 64 + lane, n);
    c32d xp = ld(x, (row0 - 1) * 64 + lane, n);
    row_sums prev = scan_row(xp, xq, lane);
    float best = 0.0f, br = 0.0f, bi = 0.0f;
    long long at = -1;
    for (long long row = row0; row <= row1; row++) {
        const long long n0 = row * 64;
        const c32d xc = ld(x, n0 + lane, n);
        const row_sums cur = scan_row(xc, xp, lane);
        const int s48 = (lane + 16) & 63;
        const float cr48 = __shfl(cur.sr, s48), ci48 = __shfl(cur.si, s48);
        const float pr48 = __shfl(prev.sr, s48), pi48 = __shfl(prev.si, s48);
        const float wr = lane >= 48 ? cur.sr - cr48 : cur.sr + (prev.tr - pr48);
        const float wi = lane >= 48 ? cur.si - ci48 : cur.si + (prev.ti - pi48);
        const float wp = cur.sp + (prev.tp - prev.sp);
        float ac = sqrtf(wr * wr + wi * wi) / wp;
        const long long i = n0 + lane;
        if (i < a || i >= hi || !(ac > best)) ac = -1.0f;  // only a value above the running maximum sets the flag
        // first lane holding the row's maximum
        float m = ac;
        for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (m > best) {
            const unsigned long long who = __ballot(ac == m);
            const int l = __ffsll((long long)who) - 1;
            best = m;
            at = n0 + l;
            br = __shfl(wr, l);
            bi = __shfl(wi, l);
        }
        xp = xc;
        prev = cur;
    }
    vr = br;
    vi = bi;
    return at >= lo ? at : -1;
}

}  // namespace

__global__ __launch_bounds__(64) void k_det_sync(const c32d *__restrict__ x, long long n, const long long *__restrict__ rise,
                                                 const long long *__restrict__ fall, long long nruns,
                                                 const long long *__restrict__ seen_at, const long long *__restrict__ scan_lo,
                                                 const long long *__restrict__ prev_seen, long long *__restrict__ out_pos,
                                                 float *__restrict__ out_rad, float *__restrict__ out_snr,
                                                 float *__restrict__ out_rssi, uint32_t *__restrict__ emitted)
{
    const long long r = blockIdx.x;
    if (r >= nruns) return;
    const int lane = threadIdx.x;
    if (lane == 0) emitted[r] = 0;
    const long long T = seen_at[r];
    if (T < 0) return;
    __shared__ c32d s_x[SYNC_BUF];
    __shared__ c32d s_pr[176];   // sig[i] * conj(sig[i+64])
    __shared__ float s_mm[SYNC_BUF];  // |sig[i]|^2 as the reference forms it: abs * abs
    __shared__ c32d s_c[64];

    // ---- 2. L-LTF autocorrelation
    for (int j = lane; j < SYNC_BUF; j += 64) s_x[j] = x[T + j];  // T + 240 <= n by construction
    __syncthreads();
    for (int j = lane; j < SYNC_BUF; j += 64) {
        const c32d v = s_x[j];
        const float m = cabs_ref(v.x, v.y);
        s_mm[j] = m * m;
        if (j < 176) {
            const c32d w = s_x[j + 64];
            c32d p;
            p.x = v.x * w.x + v.y * w.y;  // a * conj(b): (ac+bd, bc-ad)
            p.y = v.y * w.x - v.x * w.y;
            s_pr[j] = p;
        }
    }
    __syncthreads();
    float msr = 0.0f, msi = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (int i = 0; i < 64; i++) {
        const c32d p = s_pr[i];
        msr += p.x;
        msi += p.y;
        s1 += s_mm[i];
        s2 += s_mm[i + 64];
    }
    // lane l keeps the sums at positions l and l + 64
    float k_r[2] = {0.0f, 0.0f}, k_i[2] = {0.0f, 0.0f}, k_1[2] = {1.0f, 1.0f}, k_2[2] = {1.0f, 1.0f};
#pragma unroll 1
    for (int i = 0; i < SYNC_RES; i++) {
        const bool mine = (i & 63) == lane;
        const int h = i >> 6;
        if (mine) {
            if (h == 0) { k_r[0] = msr; k_i[0] = msi; k_1[0] = s1; k_2[0] = s2; }
            else { k_r[1] = msr; k_i[1] = msi; k_1[1] = s1; k_2[1] = s2; }
        }
        const c32d po = s_pr[i], pn = s_pr[i + 64];
        const float m0 = s_mm[i], m1 = s_mm[i + 64], m2 = s_mm[i + 128];
        msr -= po.x;
        msi -= po.y;
        s1 -= m0;
        s2 -= m1;
        msr += pn.x;
        msi += pn.y;
        s1 += m1;
        s2 += m2;
    }
    float acv[2];
    acv[0] = cabs_ref(k_r[0], k_i[0]) / sqrtf(k_1[0]) / sqrtf(k_2[0]);
    acv[1] = lane + 64 < SYNC_RES ? cabs_ref(k_r[1], k_i[1]) / sqrtf(k_1[1]) / sqrtf(k_2[1]) : -1.0f;
    // std::max_element: the first maximum
    float best = fmaxf(acv[0], acv[1]);
    for (int d = 32; d > 0; d >>= 1) best = fmaxf(best, __shfl_xor(best, d));
    const unsigned long long e0 = __ballot(acv[0] == best), e1 = __ballot(acv[1] == best);
    const int mx = e0 ? __ffsll((long long)e0) - 1 : 64 + __ffsll((long long)e1) - 1;
    if (!((double)best > 0.5)) return;  // lib/sync_impl.cc:99 (uniform)
    const float lim = (float)((double)best * 0.8);
    const unsigned long long b0 = __ballot(acv[0] < lim), b1 = __ballot(lane + 64 < SYNC_RES && acv[1] < lim);
    // l = last position <= mx below the limit (mx if none), rr = first position >= mx below it (mx if none)
    int l = mx, rr = mx;
    {
        const unsigned long long lo0 = mx >= 63 ? b0 : (b0 & ((2ull << mx) - 1ull));
        const unsigned long long lo1 = mx >= 64 ? (b1 & ((2ull << (mx - 64)) - 1ull)) : 0ull;
        if (lo1) l = 64 + 63 - __clzll((long long)lo1);
        else if (lo0) l = 63 - __clzll((long long)lo0);
        const unsigned long long hi0 = mx >= 64 ? 0ull : (b0 & ~((1ull << mx) - 1ull));
        const unsigned long long hi1 = mx >= 64 ? (b1 & ~((1ull << (mx - 64)) - 1ull)) : b1;
        if (hi0) rr = __ffsll((long long)hi0) - 1;
        else if (hi1) rr = 64 + __ffsll((long long)hi1) - 1;
    }
    const int m = (l + rr) / 2;
    const float best_pwr = mx < 64 ? __shfl(k_1[0], mx) : __shfl(k_1[1], mx - 64);

    // ---- 1. conjugate average
    float cvr = 0.0f, cvi = 0.0f;
    {
        long long run = r;
        for (int guard = 0; guard < 64 && run >= 0; guard++) {
            const long long t = seen_at[run], lo = scan_lo[run];
            // the latest run that starts before the trigger byte holds the last flag before it; if that flag lies
            // in skipped items so does every earlier one
            long long q = run;
            while (q + 1 < nruns && rise[q + 1] < t) q++;
            const long long hi = fall[q] < t ? fall[q] : t;
            if (plateau_peak(x, n, rise[q], hi, lo, lane, cvr, cvi) >= 0) break;
            cvr = cvi = 0.0f;
            run = prev_seen[run];
        }
    }

    // ---- 3. CFO: coarse from the 16-lag average, fine from the two LTF halves
    const float step_stf = atan2f(cvi, cvr) / 16.0f;
    {
        const c32d a = s_x[m + lane], b = s_x[m + lane + 64];
        float ca, sa, cb, sb;
        sincosf(lane * step_stf, &sa, &ca);
        sincosf((lane + 64) * step_stf, &sb, &cb);
        const float ar = a.x * ca - a.y * sa, ai = a.x * sa + a.y * ca;
        const float br = b.x * cb - b.y * sb, bi = b.x * sb + b.y * cb;
        c32d p;
        p.x = ar * br + ai * bi;
        p.y = ai * br - ar * bi;
        s_c[lane] = p;
    }
    __syncthreads();
    float csr = 0.0f, csi = 0.0f;
    for (int i = 0; i < 64; i++) {
        csr += s_c[i].x;
        csi += s_c[i].y;
    }
    const float step_ltf = atan2f(csi / 64.0f, csr / 64.0f) / 64.0f;
    if (lane == 0) {
        const double mxd = (double)best;
        out_pos[r] = T + m;
        out_rad[r] = step_stf + step_ltf;
        out_snr[r] = (float)(10.0 * log10(mxd / (1.0 - mxd)));
        out_rssi[r] = best_pwr / 64.0f;
        emitted[r] = 1;
    }
}

// -----------------------------------------------------------------------------------------
// k_det_emit: the emitted sync bytes, in stream order, as frame descriptors.
// -----------------------------------------------------------------------------------------
__global__ void k_det_flag_count(const uint32_t *__restrict__ emitted, long long nruns, uint32_t *__restrict__ blk)
{
    __shared__ uint32_t s[4];
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = r < nruns ? emitted[r] : 0u;
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void k_det_emit(const uint32_t *__restrict__ emitted, long long nruns, const uint32_t *__restrict__ blk,
                           const long long *__restrict__ pos, const float *__restrict__ rad, const float *__restrict__ snr,
                           const float *__restrict__ rssi, long long n, gr80211_frame_desc *__restrict__ desc, int max_desc,
                           uint32_t mu_pos)
{
    __shared__ uint32_t s[256];
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    const uint32_t v = r < nruns ? emitted[r] : 0u;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = threadIdx.x >= (unsigned)d ? s[threadIdx.x - d] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    if (!v) return;
    const long long o = (long long)blk[blockIdx.x] + s[threadIdx.x] - 1;
    if (o >= max_desc) return;
    gr80211_frame_desc d;
    d.sync_off = (uint64_t)pos[r];
    const long long left = n - pos[r];
    d.n_avail = left > 0xffffffffll ? 0xffffffffu : (uint32_t)left;
    d.rad = rad[r];
    d.snr = snr[r];
    d.rssi = rssi[r];
    d.mu_pos = mu_pos;
    d.reserved = 0;
    desc[o] = d;
}

// -----------------------------------------------------------------------------------------
// The trigger search of `signal` over the sync bytes of a capture (lib/signal_impl.cc:75-106,153-202): a sync
// byte starts a frame; while the frame's 224 + nsamp samples pass (80 after a failed L-SIG check) later sync
// bytes are not looked at.  Sequential in the reference; here next[i] = the first sync byte `signal` looks at
// after it dealt with byte i, and the bytes it deals with are the path 0, next[0], next[next[0]], ... found by
// pointer doubling.  Byte index nd stands for "end of capture".
// -----------------------------------------------------------------------------------------
__global__ void k_span_next(const gr80211_frame_desc *__restrict__ desc, const gr80211_frame_info *__restrict__ info, int nd,
                            long long n, int32_t *__restrict__ next)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nd) return;
    if (i == nd) {
        next[i] = nd;
        return;
    }
    const long long pos = (long long)desc[i].sync_off;
    if (n - pos < 224) {  // `signal` waits for 224 samples that never come
        next[i] = nd;
        return;
    }
    const gr80211_frame_info fi = info[i];
    const long long lim = pos + (fi.status == GR80211_ST_DROP_LSIG ? 80 : 224 + (fi.nsamp > 0 ? fi.nsamp : 0));
    int lo = i + 1, hi = nd;  // first j in (i, nd] with sync_off >= lim
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)desc[mid].sync_off >= lim) hi = mid;
        else lo = mid + 1;
    }
    next[i] = lo;
}

// path[m + half] = jump[path[m]] for m < half; then jump = jump o jump
__global__ void k_span_extend(const int32_t *__restrict__ jump, int32_t *__restrict__ path, int half, int cap)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < half && m + half < cap) path[m + half] = jump[path[m]];
}
__global__ void k_span_square(const int32_t *__restrict__ jump, int32_t *__restrict__ jump2, int nd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nd) jump2[i] = jump[jump[i]];
}
__global__ void k_span_init(int32_t *__restrict__ path, uint32_t *__restrict__ count)
{
    path[0] = 0;
    *count = 0;
}

// The bytes on the path become the descriptors of the RX batch; a frame sees its own samples only.
__global__ void k_span_gather(const int32_t *__restrict__ path, int cap, int nd, const gr80211_frame_desc *__restrict__ desc,
                              const gr80211_frame_info *__restrict__ info, long long n,
                              gr80211_frame_desc *__restrict__ out, uint32_t *__restrict__ count)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= cap) return;
    const int i = path[m];
    if (i >= nd) return;
    gr80211_frame_desc d = desc[i];
    const gr80211_frame_info fi = info[i];
    const long long left = n - (long long)d.sync_off;
    long long span = 224 + (long long)(fi.nsamp > 0 ? fi.nsamp : 0);
    if (span > left) span = left;
    d.n_avail = (uint32_t)span;
    out[m] = d;
    atomicAdd(count, 1u);
}

// -----------------------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------------------
long long det_mask_words(long long n) { return (n + 31) / 32 + 1; }
int det_edge_blocks(long long n) { return (int)((det_mask_words(n) + EDGE_WORDS) / EDGE_WORDS); }

void launch_det_ac(const void *iq, long long n, uint32_t *mask, hipStream_t s)
{
    const long long waves = (n + AC_ROWS * AC_ROW - 1) / (AC_ROWS * AC_ROW);
    hipLaunchKernelGGL(k_det_ac, dim3((unsigned)((waves + AC_WAVES - 1) / AC_WAVES)), dim3(64 * AC_WAVES), 0, s, (const c32d *)iq, n, mask,
                       ((uintptr_t)iq & 15) == 0 ? 1 : 0);
}

void launch_det_edge_count(const uint32_t *mask, long long n, uint32_t *blk_rise, uint32_t *blk_fall, uint32_t *totals,
                           hipStream_t s)
{
    const long long nwords = (n + 31) / 32;
    const int nblk = det_edge_blocks(n);
    hipLaunchKernelGGL(k_det_edge_count, dim3(nblk), dim3(EDGE_THREADS), 0, s, mask, nwords, blk_rise, blk_fall);
    hipLaunchKernelGGL(k_det_scan, dim3(1), dim3(1024), 0, s, blk_rise, blk_fall, nblk, totals, 0);
}

void launch_det_edge_emit(const uint32_t *mask, long long n, const uint32_t *blk_rise, const uint32_t *blk_fall,
                          long long cap, long long *rise, long long *fall, hipStream_t s)
{
    const long long nwords = (n + 31) / 32;
    hipLaunchKernelGGL(k_det_edge_emit, dim3(det_edge_blocks(n)), dim3(64), 0, s, mask, nwords, n, blk_rise, blk_fall,
                       cap, rise, fall);
}

void launch_det_trigger_sync(const void *iq, long long n, const long long *rise, const long long *fall, long long nruns,
                             long long *seen_at, long long *scan_lo, long long *prev_seen, long long *pos, float *rad,
                             float *snr, float *rssi, uint32_t *emitted, uint32_t *blk, uint32_t *totals, hipStream_t s)
{
    if (nruns <= 0) return;
    const int nb = (int)((nruns + 255) / 256);
    // a run that is not the head of a chain gets its entries from the head's walk; clear first for runs that
    // no walk reaches (none by construction, but the arrays are reused between calls)
    hipLaunchKernelGGL(k_det_trigger, dim3(nb), dim3(256), 0, s, rise, fall, nruns, n, seen_at, scan_lo, prev_seen);
    hipLaunchKernelGGL(k_det_sync, dim3((unsigned)nruns), dim3(64), 0, s, (const c32d *)iq, n, rise, fall,
                       nruns, seen_at, scan_lo, prev_seen, pos, rad, snr, rssi, emitted);
    hipLaunchKernelGGL(k_det_flag_count, dim3(nb), dim3(256), 0, s, emitted, nruns, blk);
    hipLaunchKernelGGL(k_det_scan, dim3(1), dim3(1024), 0, s, blk, (uint32_t *)nullptr, nb, totals, 2);
}

void launch_det_emit(const uint32_t *emitted, long long nruns, const uint32_t *blk, const long long *pos, const float *rad,
                     const float *snr, const float *rssi, long long n, gr80211_frame_desc *desc, int max_desc, uint32_t mu_pos,
                     hipStream_t s)
{
    if (nruns <= 0) return;
    const int nb = (int)((nruns + 255) / 256);
    hipLaunchKernelGGL(k_det_emit, dim3(nb), dim3(256), 0, s, emitted, nruns, blk, pos, rad, snr, rssi, n, desc, max_desc, mu_pos);
}

// path (capacity nd + 1 entries) and two jump buffers of nd + 1 entries; *count = bytes on the path.
void launch_span_filter(const gr80211_frame_desc *desc, const gr80211_frame_info *info, int nd, long long n,
                        int32_t *jump_a, int32_t *jump_b, int32_t *path, gr80211_frame_desc *out, uint32_t *count,
                        hipStream_t s)
{
    const int nb = (nd + 1 + 255) / 256;
    hipLaunchKernelGGL(k_span_next, dim3(nb), dim3(256), 0, s, desc, info, nd, n, jump_a);
    hipLaunchKernelGGL(k_span_init, dim3(1), dim3(1), 0, s, path, count);
    int32_t *ja = jump_a, *jb = jump_b;
    const int cap = nd + 1;
    for (int half = 1; half < cap; half <<= 1) {
        hipLaunchKernelGGL(k_span_extend, dim3((half + 255) / 256), dim3(256), 0, s, ja, path, half, cap);
        hipLaunchKernelGGL(k_span_square, dim3(nb), dim3(256), 0, s, ja, jb, nd);
        int32_t *t = ja;
        ja = jb;
        jb = t;
    }
    hipLaunchKernelGGL(k_span_gather, dim3((cap + 255) / 256), dim3(256), 0, s, path, cap, nd, desc, info, n, out, count);
}

}  // namespace gr80211
