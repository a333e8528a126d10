// C-ABI entry points of libgr80211_rx.so (see include/gr80211_rx.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>
#include <string>

#include "detect_kernels.h"
#include "rx_kernels.h"
#include "tx_kernels.h"

using namespace gr80211;

// the record layouts are part of the C-ABI: bindings in other languages restate them
static_assert(sizeof(gr80211_frame_desc) == 32 && sizeof(gr80211_frame_info) == 112 && sizeof(gr80211_dec_desc) == 40 &&
                  sizeof(gr80211_sig_desc) == 48 && sizeof(gr80211_tx_desc) == 40 && sizeof(gr80211_tx_mu_desc) == 56,
              "C-ABI record sizes");

namespace {

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    bool fixed = false;  // a piece of the fixed scratch array: never grown, never freed
};

// ---- work memory of gr80211_decode_batch_noalloc: one array per GPU, part of the code object, so the entry point
// never allocates.  In host code the array's name is only a handle for hipGetSymbolAddress; g_scratch[device] is
// the address of that device's copy and is what kernels, copies and memsets take.
constexpr int NA_MAX_FRAMES = 1024;
constexpr size_t NA_SCRATCH_BYTES = (size_t)48 << 20;
static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];
void *g_scratch[64] = {};

}  // namespace

struct PoolLeaseBase {
    // from here on the call's kernels touch the pooled buffer of that kind (0 soft bits, 1 survivor words and decoded
    // bits): wait, on the device, for its last user
    virtual void engage(int kind) = 0;
    virtual ~PoolLeaseBase() = default;
};

struct gr80211_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // workspaces (grown on demand, never shrunk)
    Buf desc, ddesc, info, chan1, chan2, llr, slot_frame, groups, dec, bitsw, totals_d, sched_ws, blk_llr;
    // frame detection
    Buf det_mask, det_blk, det_runs, det_rec, det_totals_d, cap_desc, cap_ws;
    Buf tx_desc, sig_desc, chanl, out_off;
    Buf info_alt, chan1_alt, chan2_alt;  // the capture path's second set of frame records and channels
    Buf pack_ws;                         // gr80211_pack_pdus: per-frame offsets and per-block sums
    // a batch between gr80211_rx_submit and gr80211_rx_finish
    struct {
        bool active;
        int n;
        const void *iq0, *iq1;
        uint64_t arena_len;
        const gr80211_frame_desc *desc;
        hipStream_t stream;
    } pending = {};
    uint32_t *det_totals_h = nullptr;  // pinned
    SchedTotals *totals_h = nullptr;  // pinned
    gr80211_dec_desc *na_desc_h = nullptr;  // pinned: descriptors of gr80211_decode_batch_noalloc on their way to the GPU
    Buf sel_index;                          // gr80211_rx_resume: which of the frames of the last gr80211_signal_batch go on
    int signal_n = 0;                       // frames of the last gr80211_signal_batch whose records and channels are still held
    bool signal_two = false;                // ... and whether it ran on two antennas
    uint32_t *pull_flag_h = nullptr;        // pinned: number of the last gr80211_pull_h2d copy that has read all of its source
    unsigned *pull_ctr_d = nullptr;         // device: workgroups of the running pull copy that have finished
    long long pull_seq = 0;
    hipStream_t pull_stream = nullptr;      // the one stream this context's pull copies go to (they share a counter and a flag)
    bool pull_stream_set = false;
    void *pull_flag_view = nullptr;         // the flag word as the GPU addresses it
    // Descriptors the caller keeps in ordinary host memory travel through page-locked memory of the context: the
    // runtime's copy from ordinary memory waits for everything its hardware queue holds, other streams' decoders included.
    struct HostStage {
        void *p = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;  // end of the last copy out of p
        bool used = false;
    } hstage[4];
    unsigned hstage_next = 0;
    uint64_t pull_gen = 0;                  // state of the list of page-locked allocations the range below was taken from
    uintptr_t pull_lo = 0, pull_hi = 0;     // the page-locked range the last pull copy read from ...
    intptr_t pull_to_dev = 0;               // ... and what turns its host addresses into the GPU's
    static constexpr int N_MARK = 256;
    hipEvent_t mark_ev[N_MARK] = {};        // gr80211_marker_*: created on first use
    long long mark_next = 0;
    struct PoolLeaseBase *lease = nullptr;  // the work set a batch call in progress holds
    int pool_sets = 0;     // > 0: soft bits, survivor words and decoded bits come from the device's pool of work sets
    int ampdu_strict = 0;  // de-aggregate A-MPDUs by the standard's rules instead of the reference's
    uint32_t mu_pos = 0;  // MU group position written into the descriptors frame detection makes
    bool profiling = false;
    bool split_decode = false;  // GR80211_SPLIT_DECODE=1: forward pass and traceback as two kernels
    int group_waves = -1;       // GR80211_GROUP_WAVES=1|2|4: wavefronts per decoder group (-1: by batch size); 10: one, in the
                                // form batches of more than 1024 groups take (k_decode_lean) whatever the size; 64: one
                                // wavefront per frame (lane = trellis state)
    bool axis_llr = false;      // GR80211_AXIS_LLR=1 (experiment): the demodulator leaves one value per constellation axis, the
                                // one-wave decoder folds the soft bits out of it; batches of one modulation per group only
    int lane_state_max = 18000;
    int lane_state_default = 18000;  // what gr80211_set_lane_state_max(-1) goes back to (GR80211_LANE_STATE_MAX)  // batches of at most this many frames take the one-wavefront-per-frame form
                                // (GR80211_LANE_STATE_MAX; 0 switches the form off)
    hipEvent_t ev[7] = {};
    float ms[6] = {};
};

namespace {

Tables g_tables;
std::once_flag g_tables_once;
bool g_uploaded[64] = {};
std::mutex g_upload_mutex;

const Tables &tables()
{
    std::call_once(g_tables_once, [] { build_tables(g_tables); });
    return g_tables;
}

int fail(gr80211_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(c, expr)                                                    \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) return fail((c), GR80211_ERR_HIP, #expr, e__); \
    } while (0)

int ensure(gr80211_ctx *c, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return GR80211_OK;
    if (b.fixed) return fail(c, GR80211_ERR_NOMEM, "the batch needs more work memory than the fixed scratch array holds");
    // grow with headroom so steady-state batches of similar size never reallocate (an eighth; a thirty-second for
    // buffers of a gigabyte and more, where an eighth is gigabytes)
    size_t want = bytes + (bytes >= ((size_t)1 << 30) ? bytes / 32 : bytes / 8) + 4096;
    if (b.p) {
        hipError_t e = hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
        if (e != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipFree", e);
    }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, GR80211_ERR_NOMEM, "hipMalloc workspace", e);
    }
    b.cap = want;
    return GR80211_OK;
}

void release(Buf &b)
{
    if (b.p && !b.fixed) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

hipStream_t pick(gr80211_ctx *c, void *stream) { return stream ? (hipStream_t)stream : c->stream; }

// ---- work buffers shared by the contexts of a device -------------------------------------------------------------
// The three large work buffers of a batch are live for part of it only: the soft bits from the demodulator to the end
// of the decoder, the survivor words and decoded bits for the length of the decoder.  Contexts that rotate over batches
// (one stream each, so that kernels of neighbouring batches overlap) therefore need fewer of them than there are
// contexts, and fewer survivor buffers than soft-bit buffers: decoders overlap two at a time (the next one fills the
// wave slots the running one leaves free), while the demodulator of the batch after next should not have to wait for
// the oldest decoder to end.  A context that has joined the pool takes a soft-bit buffer and a survivor buffer for the
// length of one batch call and hands them back with an event recorded behind the batch's last kernel; the next taker's
// stream waits for that event on the device (the host never blocks), and only from the first kernel on that touches
// the buffer: the demodulator for the soft bits, the decoder for the survivor words.
enum { POOL_SOFT = 0, POOL_SURV = 1, POOL_KINDS = 2 };
struct PoolItem {
    Buf a, b;                  // POOL_SOFT: soft bits, -; POOL_SURV: survivor words, decoded bits
    hipEvent_t ev = nullptr;   // end of the last batch that used it
    bool pending = false;      // ev has been recorded and may not have completed
    bool busy = false;         // held by a batch call that is still queuing its kernels (another host thread)
    uint64_t released = 0;     // order of hand-backs: the one handed back longest ago is taken first
};
struct DevicePool {
    std::vector<PoolItem *> items[POOL_KINDS];
    int users = 0;             // contexts of this device that have joined
    int max_items[POOL_KINDS] = {0, 0};
    size_t cap[POOL_KINDS][2] = {{0, 0}, {0, 0}};  // the largest buffers of each kind so far (bytes)
    uint64_t clock = 0;
};
std::mutex g_pool_mutex;
DevicePool g_pool[64];

// A batch call's hold on pooled buffers: they stand in for the context's own for the length of the call.
struct PoolLease : PoolLeaseBase {
    gr80211_ctx *c;
    PoolItem *it[POOL_KINDS] = {nullptr, nullptr};
    hipStream_t s;
    bool must_wait[POOL_KINDS] = {false, false};  // the buffer's last batch may still be running
    Buf own_llr, own_dec, own_bitsw;
    PoolLease(gr80211_ctx *cc, hipStream_t stream);
    ~PoolLease() override;
    void engage(int kind) override;
    PoolLease(const PoolLease &) = delete;
    PoolLease &operator=(const PoolLease &) = delete;

private:
    PoolItem *take(DevicePool &dp, int kind, bool &wait);
    static void level(Buf &b, size_t bytes);
};

// Under g_pool_mutex.  nullptr: every buffer of the kind is held by a call in progress on another thread (or no event
// could be made): the context's own buffer serves.
PoolItem *PoolLease::take(DevicePool &dp, int kind, bool &wait)
{
    PoolItem *done = nullptr, *oldest = nullptr;
    for (PoolItem *w : dp.items[kind]) {
        if (w->busy) continue;
        if (w->pending && hipEventQuery(w->ev) == hipSuccess) w->pending = false;
        if (!w->pending) {
            if (!done || w->released < done->released) done = w;
        } else if (!oldest || w->released < oldest->released) {
            oldest = w;
        }
    }
    (void)hipGetLastError();  // hipErrorNotReady from the queries above is not an error
    wait = false;
    if (done) return done;
    if ((int)dp.items[kind].size() < dp.max_items[kind]) {
        PoolItem *w = new PoolItem();
        if (hipEventCreateWithFlags(&w->ev, hipEventDisableTiming) != hipSuccess) {
            delete w;
            return nullptr;
        }
        dp.items[kind].push_back(w);
        return w;
    }
    // its last batch is still running: this batch's kernels that touch the buffer queue up behind it on the device
    // (engage), the ones before them do not
    wait = oldest != nullptr;
    return oldest;
}

PoolLease::PoolLease(gr80211_ctx *cc, hipStream_t stream) : c(cc), s(stream)
{
    if (c->pool_sets <= 0) return;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    DevicePool &dp = g_pool[c->device];
    own_llr = c->llr, own_dec = c->dec, own_bitsw = c->bitsw;
    for (int k = 0; k < POOL_KINDS; k++) {
        it[k] = take(dp, k, must_wait[k]);
        if (!it[k]) continue;
        it[k]->busy = true;
        // Every buffer of a kind is as large as the largest of them: a batch queued without the wait must find the
        // room the batches before it found, whichever buffer it happens to get.  (Freeing memory waits for the
        // device, so a buffer still in use by an earlier batch is safe to replace; sizes settle within the first
        // batches.)
        level(it[k]->a, dp.cap[k][0]);
        level(it[k]->b, dp.cap[k][1]);
    }
    if (it[POOL_SOFT]) c->llr = it[POOL_SOFT]->a;
    if (it[POOL_SURV]) c->dec = it[POOL_SURV]->a, c->bitsw = it[POOL_SURV]->b;
    if (it[POOL_SOFT] || it[POOL_SURV]) c->lease = this;
}

void PoolLease::engage(int kind)
{
    if (!it[kind] || !must_wait[kind]) return;
    must_wait[kind] = false;
    if (hipStreamWaitEvent(s, it[kind]->ev, 0) != hipSuccess) (void)hipEventSynchronize(it[kind]->ev);
}

PoolLease::~PoolLease()
{
    if (!it[POOL_SOFT] && !it[POOL_SURV]) return;
    c->lease = nullptr;
    // the call may have grown the buffers: the pool keeps what the context holds now
    if (it[POOL_SOFT]) it[POOL_SOFT]->a = c->llr, c->llr = own_llr;
    if (it[POOL_SURV]) it[POOL_SURV]->a = c->dec, it[POOL_SURV]->b = c->bitsw, c->dec = own_dec, c->bitsw = own_bitsw;
    bool recorded[POOL_KINDS] = {false, false};
    bool all = true;
    for (int k = 0; k < POOL_KINDS; k++)
        if (it[k]) all = (recorded[k] = hipEventRecord(it[k]->ev, s) == hipSuccess) && all;
    if (!all) (void)hipStreamSynchronize(s);  // without the event a buffer is only safe once the stream has run dry
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    DevicePool &dp = g_pool[c->device];
    for (int k = 0; k < POOL_KINDS; k++) {
        if (!it[k]) continue;
        it[k]->pending = recorded[k];
        it[k]->busy = false;
        it[k]->released = ++dp.clock;
        dp.cap[k][0] = std::max(dp.cap[k][0], it[k]->a.cap);
        dp.cap[k][1] = std::max(dp.cap[k][1], it[k]->b.cap);
    }
}

// Makes b at least `bytes` large, without headroom of its own (the size comes from a sibling that has it already).
void PoolLease::level(Buf &b, size_t bytes)
{
    if (b.cap >= bytes) return;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    if (hipMalloc(&b.p, bytes) == hipSuccess) b.cap = bytes;
    else b.p = nullptr, (void)hipGetLastError();  // the batch's own ensure() reports the shortage
}

// A context leaves the pool (or is destroyed): with the last user gone the buffers are freed.
void pool_leave(gr80211_ctx *c)
{
    if (c->pool_sets <= 0) return;
    c->pool_sets = 0;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    DevicePool &dp = g_pool[c->device];
    if (--dp.users > 0) return;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < POOL_KINDS; k++) {
        for (PoolItem *w : dp.items[k]) {
            release(w->a);
            release(w->b);
            if (w->ev) (void)hipEventDestroy(w->ev);
            delete w;
        }
        dp.items[k].clear();
        dp.max_items[k] = 0;
        dp.cap[k][0] = dp.cap[k][1] = 0;
    }
    dp.users = 0;
}

// Between gr80211_rx_submit and gr80211_rx_finish the context's workspaces belong to the submitted batch: every other
// batch entry point checks this before it stages a descriptor, grows a buffer or launches a kernel.
int refuse_if_pending(gr80211_ctx *c, const char *who)
{
    if (!c->pending.active) return GR80211_OK;
    c->err = who;
    c->err += ": a batch submitted with gr80211_rx_submit is waiting for gr80211_rx_finish";
    return GR80211_ERR_ARG;
}

void mark(gr80211_ctx *c, int i, hipStream_t s)
{
    if (c->profiling) (void)hipEventRecord(c->ev[i], s);
}

int finish_timing(gr80211_ctx *c, hipStream_t s)
{
    if (!c->profiling) return GR80211_OK;
    HIP_TRY(c, hipStreamSynchronize(s));
    for (int i = 0; i < 5; i++) {
        float v = 0.0f;
        if (hipEventElapsedTime(&v, c->ev[i], c->ev[i + 1]) != hipSuccess) v = 0.0f;
        c->ms[i] = v;
    }
    float v = 0.0f;
    if (hipEventElapsedTime(&v, c->ev[0], c->ev[5]) != hipSuccess) v = 0.0f;
    c->ms[5] = v;
    return GR80211_OK;
}

// What the scheduler kernels will report for a batch, worked out on the host from what the host knows of the frames
// (their decoder tags), so that work buffers can be sized -- or a batch refused -- without waiting for the device:
// the frames the decoder admits (k_desc_to_info / k_sched_hist), counted per code rate and 64-step length bin and
// cut into groups of 64 in that order.  The group count and the soft-bit count are exact; a group runs as many steps
// as its longest frame, which is at most the longest frame of the last bin it takes frames from, so the survivor and
// bit words are upper bounds.  get(i, &admitted, &cr, &trellis, &soft_bits): soft_bits counts towards llr_floats
// whether or not the frame is admitted (the demodulator writes them for every frame it keeps).
template <class Get>
SchedTotals host_schedule(int n, Get get)
{
    static thread_local uint32_t cnt[4][SCHED_NB];
    static thread_local int32_t longest[4][SCHED_NB];
    memset(cnt, 0, sizeof(cnt));
    memset(longest, 0, sizeof(longest));
    SchedTotals t = {};
    for (int i = 0; i < n; i++) {
        bool admitted = false;
        int cr = 0, trellis = 0;
        long long soft = 0;
        get(i, &admitted, &cr, &trellis, &soft);
        t.llr_floats += (uint64_t)(soft > 0 ? soft : 0);
        if (!admitted) continue;
        t.n_valid++;
        cnt[cr][trellis / 64]++;
        if (trellis > longest[cr][trellis / 64]) longest[cr][trellis / 64] = trellis;
    }
    for (int cr = 0; cr < 4; cr++) {
        const int cs = chunk_steps(cr);
        uint32_t in_group = 0;
        int top = 0;
        auto close = [&] {
            const int steps = (top + cs - 1) / cs * cs;
            t.dec_words += (uint64_t)steps * 128u;
            t.bits_words += (uint64_t)((steps + 31) / 32) * 64u;
            t.n_groups++;
            in_group = 0;
            top = 0;
        };
        for (int b = 0; b < SCHED_NB; b++) {
            uint32_t left = cnt[cr][b];
            while (left > 0) {
                const uint32_t take = left < 64 - in_group ? left : 64 - in_group;
                in_group += take;
                left -= take;
                top = longest[cr][b];
                if (in_group == 64) close();
            }
        }
        if (in_group) close();
    }
    return t;
}

// Decoder admission of a frame given by its tags (k_desc_to_info, k_sched_hist; lib/decode_impl.cc:93,210).
bool decoder_admits(int cr, int trellis, int len, long long n_llr)
{
    if (cr < 0 || cr > 3 || trellis <= 0 || trellis > GR80211_MAX_TRELLIS || len > GR80211_MAX_PSDU) return false;
    const int P = cr == CR_12 ? 1 : cr == CR_23 ? 2 : cr == CR_34 ? 3 : 5;
    const int q = trellis / P, rem = trellis % P;
    return n_llr >= (long long)q * (P + 1) + (rem ? rem + 1 : 0);
}

// Shared tail, first half: the scheduler's kernels and the copy of its totals to the host, all asynchronous.
// max_groups >= 0: the decoder will be launched for that many groups (the host worked the count out itself); the
// scheduler is told, so that a batch for which the device arrives at more is marked as refused instead of being
// decoded in part.
int launch_sched_stage(gr80211_ctx *c, hipStream_t s, int n, bool from_iq, SchedCaps caps = SCHED_NO_CAPS, int max_groups = -1)
{
    if (c->pending.active) return fail(c, GR80211_ERR_ARG, "a batch submitted with gr80211_rx_submit is waiting for gr80211_rx_finish");
    const int slot_cap = n + 4 * 64 + 64;
    const int group_room = slot_cap / 64 + 1;  // what the group array holds
    const int group_cap = max_groups >= 0 && max_groups < group_room ? max_groups : group_room;
    int rc;
    if ((rc = ensure(c, c->slot_frame, (size_t)slot_cap * sizeof(int32_t)))) return rc;
    if ((rc = ensure(c, c->groups, (size_t)group_room * sizeof(GroupInfo)))) return rc;
    if ((rc = ensure(c, c->totals_d, sizeof(SchedTotals)))) return rc;
    if ((rc = ensure(c, c->sched_ws, (size_t)SCHED_WS_WORDS * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->blk_llr, (size_t)(n / 256 + 2) * sizeof(unsigned long long)))) return rc;
    gr80211_frame_info *info = (gr80211_frame_info *)c->info.p;

    launch_schedule(info, n, from_iq ? 1 : 0, (int32_t *)c->slot_frame.p, slot_cap, (GroupInfo *)c->groups.p,
                    group_cap, (SchedTotals *)c->totals_d.p, (uint32_t *)c->sched_ws.p,
                    (unsigned long long *)c->blk_llr.p, caps, s);
    mark(c, 2, s);
    HIP_TRY(c, hipMemcpyAsync(c->totals_h, c->totals_d.p, sizeof(SchedTotals), hipMemcpyDeviceToHost, s));
    return GR80211_OK;
}

// Shared tail, second half: waits for the scheduler's totals (they size the work buffers) unless the caller knows
// them already (`known`: group and soft-bit counts exact, survivor and bit words upper bounds), then (demod) ->
// viterbi -> traceback.
int finish_decode_stages(gr80211_ctx *c, hipStream_t s, int n, bool from_iq, const void *d_iq0, const void *d_iq1,
                         uint64_t arena_len, const gr80211_frame_desc *d_desc, const float *d_llr_in, uint8_t *d_out,
                         int out_stride, int32_t *d_out_len, int32_t *d_n_pdu, uint8_t *d_bits, int bits_stride, bool decode,
                         const float **llr_used, uint64_t *llr_floats, const SchedTotals *known = nullptr)
{
    int rc;
    gr80211_frame_info *info = (gr80211_frame_info *)c->info.p;
    if (!known) HIP_TRY(c, hipStreamSynchronize(s));
    const SchedTotals tot = known ? *known : *c->totals_h;
    if (tot.n_groups < 0) return fail(c, GR80211_ERR_ARG, "scheduler capacity exceeded");

    // a batch launched without waiting for the scheduler hands its kernels the scheduler's record on the device
    const SchedTotals *dt = known ? (const SchedTotals *)c->totals_d.p : nullptr;
    const float *llr = d_llr_in;
    if (c->lease) c->lease->engage(0);  // the demodulator writes the (possibly shared) soft-bit buffer
    if (from_iq) {
        if ((rc = ensure(c, c->llr, (size_t)(tot.llr_floats + 64) * sizeof(float)))) return rc;
        launch_demod(d_iq0, d_iq1, arena_len, d_desc, n, info, c->chan1.p, c->chan2.p, (float *)c->llr.p, dt, s,
                     c->axis_llr && decode);
        llr = (const float *)c->llr.p;
    }
    if (llr_used) *llr_used = llr;
    if (llr_floats) *llr_floats = tot.llr_floats;
    mark(c, 3, s);
    if (decode) {
        launch_clear_outputs(d_out_len, d_n_pdu, n, s);
        if (from_iq && !d_iq1) launch_ndp_emit(d_iq0, arena_len, d_desc, n, info, d_out, out_stride, d_out_len, d_n_pdu, dt, s);
        if (tot.n_groups > 0) {
            if ((rc = ensure(c, c->dec, (size_t)(tot.dec_words + 128) * sizeof(uint32_t)))) return rc;
            if ((rc = ensure(c, c->bitsw, (size_t)(tot.bits_words + 64) * sizeof(uint32_t)))) return rc;
            if (c->lease) c->lease->engage(1);  // the decoder writes the (possibly shared) survivor words
            // a batch of few groups leaves SIMDs idle for a whole trellis: share each group among four waves
            int gw = c->group_waves;
            // four waves per group while their quarter-waves fit two to a SIMD, two up to one group per SIMD, one beyond
            // (tools/width_sweep.sh: 500 groups 5.6 / 6.5 / 7.7 ms by four / two / one waves, 625 groups 7.8 / 6.8 / 7.9,
            // 1024 groups 11.4 / 8.0 / 8.9, 1125 groups 13.0 / 11.8 / 10.6)
            if (gw == 10) gw = 1;  // one wave per group in the form large batches take
            // few frames: one wave per frame, one state per lane (the frame count is exact when it comes from the
            // scheduler, an upper bound when the batch was launched without waiting for it)
            const bool axis = c->axis_llr && from_iq;  // soft bits handed in from outside are soft bits
            if (axis) gw = 1;                           // only the one-wave form has the folding loader
            if (gw != 1 && gw != 2 && gw != 4 && gw != 64)
                gw = tot.n_valid <= c->lane_state_max ? 64 : tot.n_groups <= 512 ? 4 : tot.n_groups <= 1024 ? 2 : 1;
            if (gw == 64) {
                launch_decode_lane_state(llr, info, (const int32_t *)c->slot_frame.p, (const GroupInfo *)c->groups.p, tot.n_groups,
                                         (uint32_t *)c->dec.p, (uint32_t *)c->bitsw.p, d_out, out_stride, d_out_len, d_n_pdu, d_bits,
                                         bits_stride, c->ampdu_strict, dt, s);
                mark(c, 4, s);
            } else if (gw == 2 && !c->split_decode) {
                launch_decode_split2(llr, info, (const int32_t *)c->slot_frame.p, (const GroupInfo *)c->groups.p, tot.n_groups,
                                     (uint32_t *)c->dec.p, (uint32_t *)c->bitsw.p, d_out, out_stride, d_out_len, d_n_pdu, d_bits,
                                     bits_stride, c->ampdu_strict, dt, s);
                mark(c, 4, s);
            } else if (gw > 1) {
                if (gw == 2)
                    launch_viterbi_split2(llr, info, (const int32_t *)c->slot_frame.p, (const GroupInfo *)c->groups.p,
                                          tot.n_groups, (uint32_t *)c->dec.p, dt, s);
                else
                    launch_viterbi_split(llr, info, (const int32_t *)c->slot_frame.p, (const GroupInfo *)c->groups.p,
                                         tot.n_groups, (uint32_t *)c->dec.p, dt, s);
                mark(c, 4, s);
                launch_traceback((const uint32_t *)c->dec.p, info, (const int32_t *)c->slot_frame.p,
                                 (const GroupInfo *)c->groups.p, tot.n_groups, (uint32_t *)c->bitsw.p, d_out,
                                 out_stride, d_out_len, d_n_pdu, d_bits, bits_stride, c->ampdu_strict, dt, s);
            } else if (c->split_decode) {
                launch_viterbi(llr, info, (const int32_t *)c->slot_frame.p, (const GroupInfo *)c->groups.p,
                               tot.n_groups, (uint32_t *)c->dec.p, dt, s);
                mark(c, 4, s);
                launch_traceback((const uint32_t *)c->dec.p, info, (const int32_t *)c->slot_frame.p,
                                 (const GroupInfo *)c->groups.p, tot.n_groups, (uint32_t *)c->bitsw.p, d_out,
                                 out_stride, d_out_len, d_n_pdu, d_bits, bits_stride, c->ampdu_strict, dt, s);
            } else {
                // one kernel: the timing slot of the forward pass then covers the traceback as well
                launch_decode(llr, info, (const int32_t *)c->slot_frame.p, (const GroupInfo *)c->groups.p,
                              tot.n_groups, (uint32_t *)c->dec.p, (uint32_t *)c->bitsw.p, d_out, out_stride,
                              d_out_len, d_n_pdu, d_bits, bits_stride, c->ampdu_strict, dt, c->group_waves == 10, s, axis);
                mark(c, 4, s);
            }
        } else {
            mark(c, 4, s);
        }
    } else {
        mark(c, 4, s);
    }
    mark(c, 5, s);
    HIP_TRY(c, hipGetLastError());
    return GR80211_OK;
}

// Shared tail: schedule -> (demod) -> viterbi -> traceback.
int run_decode_stages(gr80211_ctx *c, hipStream_t s, int n, bool from_iq, const void *d_iq0, const void *d_iq1,
                      uint64_t arena_len, const gr80211_frame_desc *d_desc, const float *d_llr_in, uint8_t *d_out, int out_stride,
                      int32_t *d_out_len, int32_t *d_n_pdu, uint8_t *d_bits, int bits_stride, bool decode,
                      const float **llr_used, uint64_t *llr_floats, const SchedTotals *known = nullptr)
{
    int rc;
    SchedCaps caps = SCHED_NO_CAPS;
    if (known) {
        // the work buffers reach the size the host worked out before the scheduler runs, and the scheduler is told
        // what there is: should the device arrive at more (the host was told wrong), the batch is not run
        if (from_iq && (rc = ensure(c, c->llr, (size_t)(known->llr_floats + 64) * sizeof(float)))) return rc;
        if (decode && known->n_groups > 0) {
            if ((rc = ensure(c, c->dec, (size_t)(known->dec_words + 128) * sizeof(uint32_t)))) return rc;
            if ((rc = ensure(c, c->bitsw, (size_t)(known->bits_words + 64) * sizeof(uint32_t)))) return rc;
        }
        if (from_iq) caps.llr_floats = c->llr.cap / sizeof(float) - 64;
        caps.dec_words = decode ? (c->dec.cap >= 128 * sizeof(uint32_t) ? c->dec.cap / sizeof(uint32_t) - 128 : 0) : ~0ull;
        caps.bits_words = decode ? (c->bitsw.cap >= 64 * sizeof(uint32_t) ? c->bitsw.cap / sizeof(uint32_t) - 64 : 0) : ~0ull;
    }
    if ((rc = launch_sched_stage(c, s, n, from_iq, caps, known ? known->n_groups : -1))) return rc;
    return finish_decode_stages(c, s, n, from_iq, d_iq0, d_iq1, arena_len, d_desc, d_llr_in, d_out, out_stride, d_out_len,
                                d_n_pdu, d_bits, bits_stride, decode, llr_used, llr_floats, known);
}

// Host memory of any kind to the device, asynchronously: the bytes are copied to page-locked memory of the context
// first (the caller's array is free on return), one of four areas in turn, each waited for before it is used again.
int upload_small(gr80211_ctx *c, hipStream_t s, void *d_dst, const void *h_src, size_t bytes)
{
    gr80211_ctx::HostStage &h = c->hstage[c->hstage_next++ & 3];
    if (!h.ev && hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipEventCreate");
    if (h.used) HIP_TRY(c, hipEventSynchronize(h.ev));
    if (bytes > h.cap) {
        if (h.p) (void)hipHostFree(h.p);
        h.p = nullptr;
        h.cap = 0;
        const size_t cap = bytes + bytes / 2 + 4096;
        if (hipHostMalloc(&h.p, cap) != hipSuccess) {
            h.p = nullptr;
            return fail(c, GR80211_ERR_NOMEM, "page-locked memory for the descriptors");
        }
        h.cap = cap;
    }
    memcpy(h.p, h_src, bytes);
    HIP_TRY(c, hipMemcpyAsync(d_dst, h.p, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipEventRecord(h.ev, s));
    h.used = true;
    return GR80211_OK;
}

int stage_desc(gr80211_ctx *c, hipStream_t s, const void *desc, size_t bytes, unsigned flags, Buf &dst, const void **out)
{
    if (flags & GR80211_FLAG_DESC_ON_DEVICE) {
        *out = desc;
        return GR80211_OK;
    }
    int rc = ensure(c, dst, bytes);
    if (rc) return rc;
    if ((rc = upload_small(c, s, dst.p, desc, bytes))) return rc;
    *out = dst.p;
    return GR80211_OK;
}

int front_end(gr80211_ctx *c, hipStream_t s, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
              const gr80211_frame_desc *desc, int n, unsigned flags, const gr80211_frame_desc **d_desc,
              const gr80211_sig_desc *d_sig = nullptr, const void *d_chan_in = nullptr, void *d_chanl_out = nullptr)
{
    int rc;
    const void *dd = nullptr;
    if (c->pending.active) return fail(c, GR80211_ERR_ARG, "a batch submitted with gr80211_rx_submit is waiting for gr80211_rx_finish");
    c->signal_n = 0;  // the frame records and channels of an earlier gr80211_signal_batch are about to be overwritten
    if ((rc = stage_desc(c, s, desc, (size_t)n * sizeof(gr80211_frame_desc), flags, c->desc, &dd))) return rc;
    if ((rc = ensure(c, c->info, (size_t)n * sizeof(gr80211_frame_info)))) return rc;
    if ((rc = ensure(c, c->chan1, (size_t)n * 64 * 8))) return rc;
    if (d_iq1 && (rc = ensure(c, c->chan2, (size_t)n * 520 * 8))) return rc;
    mark(c, 0, s);
    launch_preamble(d_iq0, d_iq1, arena_len, (const gr80211_frame_desc *)dd, n, (gr80211_frame_info *)c->info.p,
                    c->chan1.p, c->chan2.p, d_sig, d_chan_in, d_chanl_out, s);
    mark(c, 1, s);
    *d_desc = (const gr80211_frame_desc *)dd;
    return GR80211_OK;
}

// Detection up to the per-run records (left in the context's workspaces).  *n_emit = sync bytes emitted.
int detect_core(gr80211_ctx *c, hipStream_t s, const void *d_iq, long long n, long long *nruns_out, int *n_emit)
{
    *nruns_out = 0;
    *n_emit = 0;
    int rc;
    const int eb = det_edge_blocks(n);
    if ((rc = ensure(c, c->det_mask, (size_t)det_mask_words(n) * 4))) return rc;
    if ((rc = ensure(c, c->det_blk, (size_t)eb * 2 * 4 + 64))) return rc;
    if ((rc = ensure(c, c->det_totals_d, 4 * 4))) return rc;
    uint32_t *mask = (uint32_t *)c->det_mask.p;
    uint32_t *blk_r = (uint32_t *)c->det_blk.p, *blk_f = blk_r + eb;
    uint32_t *tot_d = (uint32_t *)c->det_totals_d.p;
    launch_det_ac(d_iq, n, mask, s);
    launch_det_edge_count(mask, n, blk_r, blk_f, tot_d, s);
    HIP_TRY(c, hipMemcpyAsync(c->det_totals_h, tot_d, 2 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const long long nruns = c->det_totals_h[0];
    if (c->det_totals_h[1] != c->det_totals_h[0]) return fail(c, GR80211_ERR_HIP, "frame detection: edge lists disagree");
    if (nruns == 0) return GR80211_OK;
    // per run: rise, fall, seen_at, scan_lo, prev_seen, pos (8 B each); rad, snr, rssi, emitted (4 B each)
    if ((rc = ensure(c, c->det_runs, (size_t)nruns * 6 * 8))) return rc;
    if ((rc = ensure(c, c->det_rec, (size_t)nruns * 4 * 4 + (size_t)(nruns / 256 + 2) * 4))) return rc;
    long long *rise = (long long *)c->det_runs.p, *fall = rise + nruns, *seen_at = fall + nruns,
              *scan_lo = seen_at + nruns, *prev_seen = scan_lo + nruns, *pos = prev_seen + nruns;
    float *rad = (float *)c->det_rec.p, *snr = rad + nruns, *rssi = snr + nruns;
    uint32_t *emitted = (uint32_t *)(rssi + nruns), *blk2 = emitted + nruns;
    launch_det_edge_emit(mask, n, blk_r, blk_f, nruns, rise, fall, s);
    launch_det_trigger_sync(d_iq, n, rise, fall, nruns, seen_at, scan_lo, prev_seen, pos, rad, snr, rssi, emitted, blk2,
                            tot_d, s);
    HIP_TRY(c, hipMemcpyAsync(c->det_totals_h + 2, tot_d + 2, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    *nruns_out = nruns;
    *n_emit = (int)c->det_totals_h[2];
    return GR80211_OK;
}

// The emitted sync bytes as descriptors (the first max_desc of them), after detect_core.
void detect_emit(gr80211_ctx *c, hipStream_t s, long long nruns, long long n, gr80211_frame_desc *d_desc, int max_desc)
{
    long long *rise = (long long *)c->det_runs.p, *pos = rise + 5 * nruns;
    float *rad = (float *)c->det_rec.p, *snr = rad + nruns, *rssi = snr + nruns;
    uint32_t *emitted = (uint32_t *)(rssi + nruns), *blk2 = emitted + nruns;
    launch_det_emit(emitted, nruns, blk2, pos, rad, snr, rssi, n, d_desc, max_desc, c->mu_pos, s);
}

}  // namespace

namespace {
// Page-locked allocations this library has handed out, each with the address the GPU reads it by: a pull copy finds
// its source here instead of asking the runtime on every call.
struct HostRange {
    uintptr_t lo, hi;
    intptr_t to_dev;  // device view minus host address
};
std::mutex g_host_mutex;
std::vector<HostRange> g_host_ranges;
std::atomic<uint64_t> g_host_gen{1};  // bumped whenever the list changes: a context's remembered range is only good for one value

bool host_range_of(const void *p, size_t bytes, HostRange *out)
{
    std::lock_guard<std::mutex> lk(g_host_mutex);
    const uintptr_t a = (uintptr_t)p;
    for (const HostRange &r : g_host_ranges)
        if (a >= r.lo && a + bytes <= r.hi) {
            *out = r;
            return true;
        }
    return false;
}
}  // namespace

namespace {
// Host-to-device copy as a kernel: the GPU reads page-locked host memory over the link itself.  Costs the host one
// kernel launch whatever the size (the runtime's own asynchronous copy spends host time in proportion to the bytes).
// The workgroup that finishes last writes the copy's number to a page-locked word, which is how the host learns,
// without a call into the runtime, that the source may be overwritten.
typedef unsigned int pull_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int pull_u2 __attribute__((ext_vector_type(2)));
template <class V>
__global__ __launch_bounds__(256) void k_pull(V *__restrict__ dst, const V *__restrict__ src, size_t n_wide,
                                              unsigned char *__restrict__ dst_tail, const unsigned char *__restrict__ src_tail,
                                              int n_tail, unsigned *__restrict__ finished, uint32_t *__restrict__ flag, uint32_t number)
{
    // a few short waves that wait on the link: beside decoder waves they should issue whenever they can
    __builtin_amdgcn_s_setprio(3);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_wide; i += stride)
        dst[i] = __builtin_nontemporal_load(src + i);
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
    if (!flag) return;  // a copy nobody waits for by number
    // The number tells the host that the source has been read, nothing about the destination (work that follows on
    // the stream sees that as usual): every load of this workgroup has returned once its threads are here (each
    // store above needed its load's data), so plain relaxed operations do -- a release here would write back the
    // whole L2 of an XCD for every workgroup, with a decoder's survivor words in it.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add(finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1) {
            __hip_atomic_store(finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next copy of this stream
            __hip_atomic_store(flag, number, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Launches the copy kernel with the widest element both addresses are aligned to: 16 bytes, 8 (a stream of complex
// samples cut at an odd item), 4 or 1.
void launch_link_copy(void *dst, const void *src, size_t bytes, unsigned *finished, uint32_t *flag, uint32_t number, hipStream_t on)
{
    const uintptr_t both = (uintptr_t)src | (uintptr_t)dst;
    const int width = !(both & 15) ? 16 : !(both & 7) ? 8 : !(both & 3) ? 4 : 1;
    const size_t n_wide = bytes / width;
    const int n_tail = (int)(bytes - n_wide * width);
    size_t blocks = (n_wide + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    unsigned char *dt = (unsigned char *)dst + n_wide * width;
    const unsigned char *st = (const unsigned char *)src + n_wide * width;
    const dim3 grid((unsigned)blocks), wg(256);
    if (width == 16)
        hipLaunchKernelGGL(k_pull<pull_u4>, grid, wg, 0, on, (pull_u4 *)dst, (const pull_u4 *)src, n_wide, dt, st, n_tail, finished, flag, number);
    else if (width == 8)
        hipLaunchKernelGGL(k_pull<pull_u2>, grid, wg, 0, on, (pull_u2 *)dst, (const pull_u2 *)src, n_wide, dt, st, n_tail, finished, flag, number);
    else if (width == 4)
        hipLaunchKernelGGL(k_pull<unsigned>, grid, wg, 0, on, (unsigned *)dst, (const unsigned *)src, n_wide, dt, st, n_tail, finished, flag, number);
    else
        hipLaunchKernelGGL(k_pull<unsigned char>, grid, wg, 0, on, (unsigned char *)dst, (const unsigned char *)src, n_wide, dt, st, n_tail,
                           finished, flag, number);
}
}  // namespace


namespace {
// ---- gr80211_pack_pdus: the PDU slots of a batch as one run of bytes, each frame's blob rounded up to 16 bytes.
// Three small kernels: padded sizes scanned inside blocks of 1024 frames, the block sums scanned by one workgroup,
// then one wave per frame copies its blob to its place.
constexpr int PACK_BLOCK = 1024;
__device__ __forceinline__ unsigned pack_pad16(int len) { return len > 0 ? ((unsigned)len + 15u) & ~15u : 0u; }

__global__ __launch_bounds__(PACK_BLOCK) void k_pack_sizes(const int32_t *__restrict__ out_len, int n, int stride,
                                                            unsigned long long *__restrict__ off, unsigned long long *__restrict__ blk)
{
    __shared__ unsigned wsum[PACK_BLOCK / 64];
    const int i = blockIdx.x * PACK_BLOCK + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int len = i < n ? out_len[i] : 0;
    if (len > stride) len = stride;  // a slot never holds more than its stride
    const unsigned mine = pack_pad16(len);
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned before = 0;
    for (int k = 0; k < w; k++) before += wsum[k];
    if (i < n) off[i] = (unsigned long long)(before + incl - mine);
    if (threadIdx.x == PACK_BLOCK - 1) blk[blockIdx.x] = (unsigned long long)(before + incl);
}

__global__ __launch_bounds__(1024) void k_pack_scan_blocks(unsigned long long *__restrict__ blk, int nblk,
                                                            unsigned long long *__restrict__ total)
{
    // nblk <= 1024 (at most 2^20 frames per call): one element per thread
    __shared__ unsigned long long wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long mine = t < nblk ? blk[t] : 0ull;
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (int k = 0; k < w; k++) before += wsum[k];
    if (t < nblk) blk[t] = before + incl - mine;
    if (t == 1023) *total = before + incl;
}

__global__ __launch_bounds__(256) void k_pack_copy(const uint8_t *__restrict__ out, int stride, const int32_t *__restrict__ out_len,
                                                   int n, const unsigned long long *__restrict__ off,
                                                   const unsigned long long *__restrict__ blk, uint8_t *__restrict__ packed,
                                                   unsigned long long cap, unsigned long long *__restrict__ off_out)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= n) return;
    int len = out_len[f];
    if (len > stride) len = stride;
    const unsigned padded = pack_pad16(len);
    const unsigned long long at = blk[f / PACK_BLOCK] + off[f];
    if (off_out && lane == 0) off_out[f] = at;
    if (padded == 0 || at + padded > cap) return;  // no room: the caller sees total > cap
    const pull_u4 *src = reinterpret_cast<const pull_u4 *>(out + (size_t)f * stride);
    pull_u4 *dst = reinterpret_cast<pull_u4 *>(packed + at);
    const int nq = (int)(padded / 16);
    for (int q = lane; q < nq; q += 64) {
        pull_u4 v = src[q];
        // bytes between the blob's end and the 16-byte boundary leave as zeros, whatever the slot held there
        const int left = len - q * 16;
        if (left < 16) {
            unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int lb = left - 4 * k;
                wv[k] = lb >= 4 ? wv[k] : lb <= 0 ? 0u : wv[k] & ((1u << (8 * lb)) - 1u);
            }
            v.x = wv[0], v.y = wv[1], v.z = wv[2], v.w = wv[3];
        }
        dst[q] = v;
    }
}
}  // namespace

extern "C" {

const char *gr80211_version(void) { return "gr80211_rx 0.1 (gfx950)"; }

uint32_t gr80211_table_checksum(void) { return tables_checksum(tables()); }

int gr80211_create(gr80211_ctx **out, int device)
{
    if (!out) return GR80211_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GR80211_ERR_HIP;
    if (device < 0 || device >= ndev || device >= 64) return GR80211_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return GR80211_ERR_HIP;
    gr80211_ctx *c = new gr80211_ctx();
    c->device = device;
    if (const char *e = getenv("GR80211_SPLIT_DECODE")) c->split_decode = e[0] == '1';
    if (const char *e = getenv("GR80211_GROUP_WAVES")) c->group_waves = atoi(e);
    if (const char *e = getenv("GR80211_LANE_STATE_MAX")) c->lane_state_max = c->lane_state_default = atoi(e);
    if (const char *e = getenv("GR80211_AXIS_LLR")) c->axis_llr = e[0] == '1';
    if (const char *e = getenv("GR80211_WORK_POOL")) {
        const int sets = atoi(e);
        if (sets > 0) (void)gr80211_use_work_pool(c, sets);
    }
    {
        std::lock_guard<std::mutex> lk(g_upload_mutex);
        if (!g_uploaded[device]) {
            // this device's copy of the scratch array: the address is taken once, with the device current
            void *sp = nullptr;
            if (upload_tables(tables()) != 0 || upload_tx_tables(tables()) != 0 ||
                hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) {
                delete c;
                return GR80211_ERR_HIP;
            }
            g_scratch[device] = sp;
            g_uploaded[device] = true;
        }
    }
    if (hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return GR80211_ERR_HIP;
    }
    if (hipHostMalloc((void **)&c->totals_h, sizeof(SchedTotals)) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return GR80211_ERR_NOMEM;
    }
    if (hipHostMalloc((void **)&c->det_totals_h, 4 * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void **)&c->na_desc_h, NA_MAX_FRAMES * sizeof(gr80211_dec_desc)) != hipSuccess) {
        gr80211_destroy(c);
        return GR80211_ERR_NOMEM;
    }
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            gr80211_destroy(c);
            return GR80211_ERR_HIP;
        }
    *out = c;
    return GR80211_OK;
}

void gr80211_destroy(gr80211_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    pool_leave(c);
    for (Buf *b : {&c->desc, &c->ddesc, &c->info, &c->chan1, &c->chan2, &c->llr, &c->slot_frame, &c->groups, &c->dec,
                   &c->bitsw, &c->totals_d, &c->sched_ws, &c->blk_llr, &c->det_mask, &c->det_blk, &c->det_runs,
                   &c->det_rec, &c->det_totals_d, &c->cap_desc, &c->cap_ws, &c->tx_desc, &c->sig_desc, &c->chanl, &c->out_off,
                   &c->info_alt, &c->chan1_alt, &c->chan2_alt, &c->sel_index, &c->pack_ws})
        release(*b);
    for (auto &e : c->mark_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->totals_h) (void)hipHostFree(c->totals_h);
    if (c->det_totals_h) (void)hipHostFree(c->det_totals_h);
    if (c->na_desc_h) (void)hipHostFree(c->na_desc_h);
    if (c->pull_flag_h) (void)hipHostFree(c->pull_flag_h);
    for (auto &h : c->hstage) {
        if (h.p) (void)hipHostFree(h.p);
        if (h.ev) (void)hipEventDestroy(h.ev);
    }
    if (c->pull_ctr_d) (void)hipFree(c->pull_ctr_d);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *gr80211_last_error(gr80211_ctx *c) { return c ? c->err.c_str() : "null context"; }

int gr80211_sync(gr80211_ctx *c, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(pick(c, stream)));
    return GR80211_OK;
}

void *gr80211_get_stream(gr80211_ctx *c) { return c ? (void *)c->stream : nullptr; }

int gr80211_rx_batch(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                     const gr80211_frame_desc *desc, int n, unsigned flags, uint8_t *d_out, int out_stride,
                     int32_t *d_out_len, int32_t *d_n_pdu, gr80211_frame_info *d_info, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_rx_batch")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_iq0 || !desc || n < 0 || !d_out || !d_out_len || !d_n_pdu || out_stride < 8)
        return fail(c, GR80211_ERR_ARG, "gr80211_rx_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_rx_batch: at most 1048576 frames per call");
    if (n == 0) return GR80211_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    PoolLease lease(c, s);
    const gr80211_frame_desc *dd = nullptr;
    SchedTotals room = {};
    const bool no_wait = (flags & GR80211_FLAG_NO_WAIT) != 0;
    if (no_wait) {
        // what the buffers hold is what the batch may use; groups are cut per code rate, so four may stay part full
        if (c->llr.cap < 65 * sizeof(float) || c->dec.cap < 129 * sizeof(uint32_t) || c->bitsw.cap < 65 * sizeof(uint32_t))
            return fail(c, GR80211_ERR_ARG, "gr80211_rx_batch: GR80211_FLAG_NO_WAIT before a batch without it has sized the work buffers");
        room.n_groups = n / 64 + 4;
        room.n_valid = n;
        room.llr_floats = c->llr.cap / sizeof(float) - 64;
        room.dec_words = c->dec.cap / sizeof(uint32_t) - 128;
        room.bits_words = c->bitsw.cap / sizeof(uint32_t) - 64;
    }
    int rc = front_end(c, s, d_iq0, d_iq1, arena_len, desc, n, flags, &dd);
    if (rc) return rc;
    rc = run_decode_stages(c, s, n, true, d_iq0, d_iq1, arena_len, dd, nullptr, d_out, out_stride, d_out_len, d_n_pdu, nullptr, 0,
                           true, nullptr, nullptr, no_wait ? &room : nullptr);
    if (rc) return rc;
    if (d_info)
        HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, s));
    return finish_timing(c, s);
}

int gr80211_rx_submit(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                      const gr80211_frame_desc *desc, int n, unsigned flags, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (c->pending.active) return fail(c, GR80211_ERR_ARG, "gr80211_rx_submit: the batch submitted before has not been finished");
    if (!d_iq0 || !desc || n < 0) return fail(c, GR80211_ERR_ARG, "gr80211_rx_submit: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_rx_submit: at most 1048576 frames per call");
    c->pending = {false, n, d_iq0, d_iq1, arena_len, nullptr, nullptr};
    if (n > 0) {
        HIP_TRY(c, hipSetDevice(c->device));
        hipStream_t s = pick(c, stream);
        c->pending.stream = s;
        int rc = front_end(c, s, d_iq0, d_iq1, arena_len, desc, n, flags, &c->pending.desc);
        if (!rc) rc = launch_sched_stage(c, s, n, true);
        if (rc) return rc;
    }
    c->pending.active = true;  // from here to gr80211_rx_finish every other batch call is refused
    return GR80211_OK;
}

int gr80211_rx_finish(gr80211_ctx *c, uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu,
                      gr80211_frame_info *d_info)
{
    if (!c) return GR80211_ERR_ARG;
    if (!c->pending.active) return fail(c, GR80211_ERR_ARG, "gr80211_rx_finish: no batch has been submitted");
    const auto p = c->pending;
    // a call that is refused for its arguments leaves the submitted batch in place: the caller may finish it again
    if (p.n > 0 && (!d_out || !d_out_len || !d_n_pdu || out_stride < 8))
        return fail(c, GR80211_ERR_ARG, "gr80211_rx_finish: bad argument");
    if (p.n > 0) HIP_TRY(c, hipSetDevice(c->device));
    c->pending.active = false;  // from here on the batch is spent, whatever the launches below return
    if (p.n == 0) return GR80211_OK;
    PoolLease lease(c, p.stream);
    int rc = finish_decode_stages(c, p.stream, p.n, true, p.iq0, p.iq1, p.arena_len, p.desc, nullptr, d_out, out_stride, d_out_len,
                                  d_n_pdu, nullptr, 0, true, nullptr, nullptr);
    if (rc) return rc;
    if (d_info)
        HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)p.n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, p.stream));
    return finish_timing(c, p.stream);
}

int gr80211_detect(gr80211_ctx *c, const void *d_iq, uint64_t n_samples, gr80211_frame_desc *d_desc, int max_desc,
                   int *n_desc, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_detect")) return rc_busy;
    if (!n_desc) return fail(c, GR80211_ERR_ARG, "gr80211_detect: bad argument");
    *n_desc = 0;
    if (n_samples == 0) return GR80211_OK;
    if (!d_iq || !d_desc || max_desc < 0 || n_samples > (1ull << 40))
        return fail(c, GR80211_ERR_ARG, "gr80211_detect: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    long long nruns = 0;
    int rc = detect_core(c, s, d_iq, (long long)n_samples, &nruns, n_desc);
    if (rc) return rc;
    if (*n_desc == 0) return GR80211_OK;
    detect_emit(c, s, nruns, (long long)n_samples, d_desc, max_desc);
    HIP_TRY(c, hipGetLastError());
    if (*n_desc > max_desc) return fail(c, GR80211_ERR_ARG, "gr80211_detect: more sync points than max_desc");
    return GR80211_OK;
}

int gr80211_rx_capture(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t n_samples, uint8_t *d_out,
                       int out_stride, int32_t *d_out_len, int32_t *d_n_pdu, int max_frames,
                       gr80211_frame_desc *d_desc_out, int *n_frames, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_rx_capture")) return rc_busy;
    if (!n_frames) return fail(c, GR80211_ERR_ARG, "gr80211_rx_capture: bad argument");
    *n_frames = 0;
    if (n_samples == 0) return GR80211_OK;
    if (!d_iq0 || !d_out || !d_out_len || !d_n_pdu || out_stride < 8 || max_frames < 0 || n_samples > (1ull << 40))
        return fail(c, GR80211_ERR_ARG, "gr80211_rx_capture: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    const long long n = (long long)n_samples;
    long long nruns = 0;
    int nd = 0;
    int rc = detect_core(c, s, d_iq0, n, &nruns, &nd);
    if (rc) return rc;
    if (nd == 0) return GR80211_OK;
    if (nd > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_rx_capture: at most 1048576 sync bytes per call");
    // sync bytes -> L-SIG of each -> the ones `signal` deals with
    if ((rc = ensure(c, c->cap_desc, (size_t)nd * 2 * sizeof(gr80211_frame_desc)))) return rc;
    if ((rc = ensure(c, c->cap_ws, (size_t)(nd + 1) * 3 * sizeof(int32_t) + 64))) return rc;
    gr80211_frame_desc *d_all = (gr80211_frame_desc *)c->cap_desc.p, *d_sel = d_all + nd;
    int32_t *jump_a = (int32_t *)c->cap_ws.p, *jump_b = jump_a + nd + 1, *path = jump_b + nd + 1;
    uint32_t *tot_d = (uint32_t *)c->det_totals_d.p;
    detect_emit(c, s, nruns, n, d_all, nd);
    const gr80211_frame_desc *dd = nullptr;
    if ((rc = front_end(c, s, d_iq0, d_iq1, n_samples, d_all, nd, GR80211_FLAG_DESC_ON_DEVICE, &dd))) return rc;
    launch_span_filter(d_all, (const gr80211_frame_info *)c->info.p, nd, n, jump_a, jump_b, path, d_sel, tot_d + 3, s);
    HIP_TRY(c, hipMemcpyAsync(c->det_totals_h + 3, tot_d + 3, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const int nf = (int)c->det_totals_h[3];
    *n_frames = nf;
    if (nf > max_frames) return fail(c, GR80211_ERR_ARG, "gr80211_rx_capture: more frames than max_frames");
    if (nf == 0) return GR80211_OK;
    if (d_desc_out)
        HIP_TRY(c, hipMemcpyAsync(d_desc_out, d_sel, (size_t)nf * sizeof(gr80211_frame_desc), hipMemcpyDeviceToDevice, s));
    // the preamble has run over these frames already (a frame's span does not change what it finds): their records
    // and channels move to the front of a second set of buffers, which then takes the first set's place
    if ((rc = ensure(c, c->info_alt, (size_t)nf * sizeof(gr80211_frame_info)))) return rc;
    if ((rc = ensure(c, c->chan1_alt, (size_t)nf * 64 * 8))) return rc;
    if (d_iq1 && (rc = ensure(c, c->chan2_alt, (size_t)nf * 520 * 8))) return rc;
    launch_pre_gather(path, nf, (const gr80211_frame_info *)c->info.p, c->chan1.p, d_iq1 ? c->chan2.p : nullptr,
                      (gr80211_frame_info *)c->info_alt.p, c->chan1_alt.p, c->chan2_alt.p, s);
    std::swap(c->info, c->info_alt);
    std::swap(c->chan1, c->chan1_alt);
    if (d_iq1) std::swap(c->chan2, c->chan2_alt);
    mark(c, 1, s);
    PoolLease lease(c, s);
    rc = run_decode_stages(c, s, nf, true, d_iq0, d_iq1, n_samples, d_sel, nullptr, d_out, out_stride, d_out_len, d_n_pdu, nullptr, 0,
                           true, nullptr, nullptr);
    if (rc) return rc;
    return finish_timing(c, s);
}

int gr80211_signal_batch(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                         const gr80211_frame_desc *desc, int n, unsigned flags, gr80211_frame_info *d_info, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_signal_batch")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_iq0 || !desc || n < 0 || !d_info) return fail(c, GR80211_ERR_ARG, "gr80211_signal_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_signal_batch: at most 1048576 frames per call");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    const gr80211_frame_desc *dd = nullptr;
    int rc = front_end(c, s, d_iq0, d_iq1, arena_len, desc, n, flags, &dd);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipGetLastError());
    c->signal_n = n;
    c->signal_two = d_iq1 != nullptr;
    return GR80211_OK;
}

int gr80211_signal_emit_batch(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                              const gr80211_frame_desc *desc, int n, unsigned flags, const uint64_t *out_off, void *d_out0,
                              void *d_out1, uint64_t out_len, gr80211_frame_info *d_info, void *d_chan, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_signal_emit_batch")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_iq0 || !desc || n < 0 || !out_off || !d_out0 || (d_iq1 && !d_out1) || arena_len == 0)
        return fail(c, GR80211_ERR_ARG, "gr80211_signal_emit_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_signal_emit_batch: at most 1048576 frames per call");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    int rc;
    const void *d_off = nullptr;
    if ((rc = stage_desc(c, s, out_off, (size_t)n * sizeof(uint64_t), flags, c->out_off, &d_off))) return rc;
    if (!d_chan) {
        if ((rc = ensure(c, c->chanl, (size_t)n * 64 * 8))) return rc;
        d_chan = c->chanl.p;
    }
    const gr80211_frame_desc *dd = nullptr;
    if ((rc = front_end(c, s, d_iq0, d_iq1, arena_len, desc, n, flags, &dd, nullptr, nullptr, d_chan))) return rc;
    launch_signal_copy(d_iq0, d_iq1, arena_len, dd, n, (const gr80211_frame_info *)c->info.p, (const uint64_t *)d_off, d_out0,
                       d_iq1 ? d_out1 : nullptr, out_len, s);
    if (d_info)
        HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipGetLastError());
    return GR80211_OK;
}

int gr80211_demod_tagged_batch(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                               const gr80211_sig_desc *desc, int n, unsigned flags, const void *d_chan,
                               gr80211_frame_info *d_info, const float **d_llr, uint64_t *llr_floats, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_demod_tagged_batch")) return rc_busy;
    if (!d_iq0 || !desc || n < 0 || !d_llr || !d_chan || arena_len == 0)
        return fail(c, GR80211_ERR_ARG, "gr80211_demod_tagged_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_demod_tagged_batch: at most 1048576 frames per call");
    *d_llr = nullptr;
    if (llr_floats) *llr_floats = 0;
    if (n == 0) return GR80211_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    int rc;
    const void *d_sig = nullptr;
    if ((rc = stage_desc(c, s, desc, (size_t)n * sizeof(gr80211_sig_desc), flags, c->sig_desc, &d_sig))) return rc;
    // the tags become descriptors of the kind the kernels read, marked as starting after L-SIG
    if ((rc = ensure(c, c->desc, (size_t)n * sizeof(gr80211_frame_desc)))) return rc;
    launch_sig_to_desc((const gr80211_sig_desc *)d_sig, n, (gr80211_frame_desc *)c->desc.p, s);
    const gr80211_frame_desc *dd = nullptr;
    if ((rc = front_end(c, s, d_iq0, d_iq1, arena_len, (const gr80211_frame_desc *)c->desc.p, n, GR80211_FLAG_DESC_ON_DEVICE, &dd,
                        (const gr80211_sig_desc *)d_sig, d_chan)))
        return rc;
    rc = run_decode_stages(c, s, n, true, d_iq0, d_iq1, arena_len, dd, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, false,
                           d_llr, llr_floats);
    if (rc) return rc;
    if (d_info)
        HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, s));
    return finish_timing(c, s);
}

int gr80211_demod_batch(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len,
                        const gr80211_frame_desc *desc, int n, unsigned flags, gr80211_frame_info *d_info,
                        const float **d_llr, uint64_t *llr_floats, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_demod_batch")) return rc_busy;
    if (!d_iq0 || !desc || n < 0 || !d_llr) return fail(c, GR80211_ERR_ARG, "gr80211_demod_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_demod_batch: at most 1048576 frames per call");
    *d_llr = nullptr;
    if (llr_floats) *llr_floats = 0;
    if (n == 0) return GR80211_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    const gr80211_frame_desc *dd = nullptr;
    int rc = front_end(c, s, d_iq0, d_iq1, arena_len, desc, n, flags, &dd);
    if (rc) return rc;
    rc = run_decode_stages(c, s, n, true, d_iq0, d_iq1, arena_len, dd, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, false,
                           d_llr, llr_floats);
    if (rc) return rc;
    if (d_info)
        HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, s));
    return finish_timing(c, s);
}

int gr80211_decode_batch(gr80211_ctx *c, const float *d_llr, uint64_t llr_floats, const gr80211_dec_desc *desc, int n,
                         unsigned flags, uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu,
                         uint8_t *d_bits, int bits_stride, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_decode_batch")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_llr || !desc || n < 0 || !d_out || !d_out_len || !d_n_pdu || out_stride < 8)
        return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch: bad argument");
    if (d_bits && (bits_stride < 4 || (bits_stride & 3)))
        return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch: bits_stride must be a multiple of 4");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch: at most 1048576 frames per call");
    if (n == 0) return GR80211_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    // descriptors must be validated against the LLR arena on the host when they are host-resident
    if (!(flags & GR80211_FLAG_DESC_ON_DEVICE)) {
        for (int i = 0; i < n; i++) {
            if (desc[i].llr_off < 0 || desc[i].n_llr < 0 ||
                (uint64_t)desc[i].llr_off + (uint64_t)desc[i].n_llr > llr_floats)
                return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch: descriptor outside the LLR arena");
        }
    }
    int rc;
    const void *dd = nullptr;
    // Descriptors on the host say what the scheduler kernels will arrive at: the work buffers are sized from them and
    // nothing is waited for (descriptors on the device: the scheduler's totals are waited for, as before).
    SchedTotals known = {};
    const bool on_host = !(flags & GR80211_FLAG_DESC_ON_DEVICE);
    if (on_host) {
        known = host_schedule(n, [&](int i, bool *admitted, int *cr, int *trellis, long long *soft) {
            const gr80211_dec_desc &d = desc[i];
            *admitted = decoder_admits(d.cr, d.trellis, d.len, d.n_llr);
            *cr = d.cr;
            *trellis = d.trellis;
            *soft = 0;
        });
    }
    if ((rc = stage_desc(c, s, desc, (size_t)n * sizeof(gr80211_dec_desc), flags, c->ddesc, &dd))) return rc;
    c->signal_n = 0;
    if ((rc = ensure(c, c->info, (size_t)n * sizeof(gr80211_frame_info)))) return rc;
    mark(c, 0, s);
    launch_desc_to_info((const gr80211_dec_desc *)dd, n, llr_floats, (gr80211_frame_info *)c->info.p, s);
    mark(c, 1, s);
    PoolLease lease(c, s);
    rc = run_decode_stages(c, s, n, false, nullptr, nullptr, 0, nullptr, d_llr, d_out, out_stride, d_out_len, d_n_pdu,
                           d_bits, bits_stride, true, nullptr, nullptr, on_host ? &known : nullptr);
    if (rc) return rc;
    return finish_timing(c, s);
}

int gr80211_decode_batch_noalloc(gr80211_ctx *c, const float *d_llr, uint64_t llr_floats, const gr80211_dec_desc *h_desc, int n,
                                 uint8_t *d_out, int out_stride, int32_t *d_out_len, int32_t *d_n_pdu, uint8_t *d_bits,
                                 int bits_stride, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_decode_batch_noalloc")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_llr || !h_desc || n < 0 || !d_out || !d_out_len || !d_n_pdu || out_stride < 8)
        return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch_noalloc: bad argument");
    if (d_bits && (bits_stride < 4 || (bits_stride & 3)))
        return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch_noalloc: bits_stride must be a multiple of 4");
    if (n > NA_MAX_FRAMES) return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch_noalloc: at most 1024 frames per call");
    for (int i = 0; i < n; i++)
        if (h_desc[i].llr_off < 0 || h_desc[i].n_llr < 0 || (uint64_t)h_desc[i].llr_off + (uint64_t)h_desc[i].n_llr > llr_floats)
            return fail(c, GR80211_ERR_ARG, "gr80211_decode_batch_noalloc: descriptor outside the LLR arena");

    // The scratch array in pieces (256-byte aligned), sized for NA_MAX_FRAMES frames; survivor words take what is left.
    struct Carver {
        size_t at, size;
        char *base;
        Buf take(size_t bytes)
        {
            bytes = (bytes + 255) & ~(size_t)255;
            Buf b;
            b.fixed = true;
            if (at + bytes <= size) {
                b.p = base ? base + at : nullptr;
                b.cap = bytes;
                at += bytes;
            }
            return b;
        }
    };
    const int slot_cap = NA_MAX_FRAMES + 4 * 64 + 64, group_cap = slot_cap / 64 + 1;
    const size_t piece[7] = {NA_MAX_FRAMES * sizeof(gr80211_dec_desc), NA_MAX_FRAMES * sizeof(gr80211_frame_info),
                             (size_t)slot_cap * sizeof(int32_t), (size_t)group_cap * sizeof(GroupInfo), sizeof(SchedTotals),
                             (size_t)SCHED_WS_WORDS * sizeof(uint32_t),
                             (size_t)(NA_MAX_FRAMES / 256 + 2) * sizeof(unsigned long long)};
    size_t fixed_part = 0;
    for (size_t b : piece) fixed_part += (b + 255) & ~(size_t)255;
    const size_t bits_bytes = (size_t)1 << 20;
    const size_t dec_bytes = NA_SCRATCH_BYTES - fixed_part - bits_bytes;

    // what the scheduler kernels will ask for, worked out here so that a batch beyond the array is refused before
    // anything is launched
    const SchedTotals need = host_schedule(n, [&](int i, bool *admitted, int *cr, int *trellis, long long *soft) {
        const gr80211_dec_desc &d = h_desc[i];
        *admitted = decoder_admits(d.cr, d.trellis, d.len, d.n_llr);
        *cr = d.cr;
        *trellis = d.trellis;
        *soft = 0;
    });
    if ((need.dec_words + 128) * sizeof(uint32_t) > dec_bytes || (need.bits_words + 64) * sizeof(uint32_t) > bits_bytes) {
        char msg[200];
        snprintf(msg, sizeof(msg),
                 "gr80211_decode_batch_noalloc: the batch needs %llu bytes of survivor words, the scratch array holds %llu",
                 (unsigned long long)((need.dec_words + 128) * sizeof(uint32_t)), (unsigned long long)dec_bytes);
        return fail(c, GR80211_ERR_ARG, msg);
    }

    HIP_TRY(c, hipSetDevice(c->device));
    char *base = static_cast<char *>(g_scratch[c->device]);
    if (!base) return fail(c, GR80211_ERR_HIP, "gr80211_decode_batch_noalloc: no scratch array on this device");
    hipStream_t s = pick(c, stream);
    Carver cv = {0, NA_SCRATCH_BYTES, base};
    // the context's own work buffers step aside for the duration of the call
    struct Swap {
        gr80211_ctx *c;
        Buf ddesc, info, slot_frame, groups, totals_d, sched_ws, blk_llr, dec, bitsw;
        explicit Swap(gr80211_ctx *cc)
            : c(cc), ddesc(cc->ddesc), info(cc->info), slot_frame(cc->slot_frame), groups(cc->groups), totals_d(cc->totals_d),
              sched_ws(cc->sched_ws), blk_llr(cc->blk_llr), dec(cc->dec), bitsw(cc->bitsw) {}
        ~Swap()
        {
            c->ddesc = ddesc; c->info = info; c->slot_frame = slot_frame; c->groups = groups; c->totals_d = totals_d;
            c->sched_ws = sched_ws; c->blk_llr = blk_llr; c->dec = dec; c->bitsw = bitsw;
        }
    } keep(c);
    c->ddesc = cv.take(piece[0]);
    c->info = cv.take(piece[1]);
    c->slot_frame = cv.take(piece[2]);
    c->groups = cv.take(piece[3]);
    c->totals_d = cv.take(piece[4]);
    c->sched_ws = cv.take(piece[5]);
    c->blk_llr = cv.take(piece[6]);
    c->bitsw = cv.take(bits_bytes);
    c->dec = cv.take(dec_bytes);

    // the descriptors travel through page-locked memory the context has held since it was created; the copy has
    // left that memory when the call returns (the stream is synchronised below, for the scheduler's totals)
    memcpy(c->na_desc_h, h_desc, (size_t)n * sizeof(gr80211_dec_desc));
    HIP_TRY(c, hipMemcpyAsync(c->ddesc.p, c->na_desc_h, (size_t)n * sizeof(gr80211_dec_desc), hipMemcpyHostToDevice, s));
    mark(c, 0, s);
    launch_desc_to_info((const gr80211_dec_desc *)c->ddesc.p, n, llr_floats, (gr80211_frame_info *)c->info.p, s);
    mark(c, 1, s);
    int rc = run_decode_stages(c, s, n, false, nullptr, nullptr, 0, nullptr, d_llr, d_out, out_stride, d_out_len, d_n_pdu, d_bits,
                               bits_stride, true, nullptr, nullptr);
    if (rc) return rc;
    return finish_timing(c, s);
}

int gr80211_rx_resume(gr80211_ctx *c, const void *d_iq0, const void *d_iq1, uint64_t arena_len, const gr80211_frame_desc *h_desc,
                      const int32_t *h_index, const gr80211_frame_info *h_info, int n, uint8_t *d_out, int out_stride,
                      int32_t *d_out_len, int32_t *d_n_pdu, gr80211_frame_info *d_info, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_rx_resume")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_iq0 || !h_desc || !h_index || !h_info || n < 0 || !d_out || !d_out_len || !d_n_pdu || out_stride < 8)
        return fail(c, GR80211_ERR_ARG, "gr80211_rx_resume: bad argument");
    if (c->signal_n <= 0) return fail(c, GR80211_ERR_ARG, "gr80211_rx_resume: the context holds no results of gr80211_signal_batch");
    if (c->signal_two != (d_iq1 != nullptr)) return fail(c, GR80211_ERR_ARG, "gr80211_rx_resume: antenna count differs from gr80211_signal_batch");
    if (n > c->signal_n) return fail(c, GR80211_ERR_ARG, "gr80211_rx_resume: more frames than gr80211_signal_batch saw");
    for (int i = 0; i < n; i++)
        if (h_index[i] < 0 || h_index[i] >= c->signal_n) return fail(c, GR80211_ERR_ARG, "gr80211_rx_resume: index outside the signal batch");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    int rc;
    // the scheduler's totals from the records the caller holds: exact group and soft-bit counts, bounds for the rest
    const SchedTotals known = host_schedule(n, [&](int i, bool *admitted, int *cr, int *trellis, long long *soft) {
        const gr80211_frame_info &fi = h_info[h_index[i]];
        const bool ok = fi.status == GR80211_ST_OK;
        *admitted = ok && !(fi.len > GR80211_MAX_PSDU || fi.trellis > GR80211_MAX_TRELLIS || fi.trellis <= 0 || fi.cr < 0 || fi.cr > 3);
        *cr = fi.cr;
        *trellis = fi.trellis;
        *soft = ok ? (fi.total > 0 ? fi.total : 0) : 0;
    });
    if ((rc = ensure(c, c->desc, (size_t)n * sizeof(gr80211_frame_desc)))) return rc;
    if ((rc = ensure(c, c->sel_index, (size_t)n * sizeof(int32_t)))) return rc;
    if ((rc = ensure(c, c->info_alt, (size_t)n * sizeof(gr80211_frame_info)))) return rc;
    if ((rc = ensure(c, c->chan1_alt, (size_t)n * 64 * 8))) return rc;
    if (d_iq1 && (rc = ensure(c, c->chan2_alt, (size_t)n * 520 * 8))) return rc;
    if ((rc = upload_small(c, s, c->desc.p, h_desc, (size_t)n * sizeof(gr80211_frame_desc)))) return rc;
    if ((rc = upload_small(c, s, c->sel_index.p, h_index, (size_t)n * sizeof(int32_t)))) return rc;
    mark(c, 0, s);
    // the preamble has run over these frames already: their records and channels move to the front of the second
    // set of buffers, which takes the first set's place
    launch_pre_gather((const int32_t *)c->sel_index.p, n, (const gr80211_frame_info *)c->info.p, c->chan1.p,
                      d_iq1 ? c->chan2.p : nullptr, (gr80211_frame_info *)c->info_alt.p, c->chan1_alt.p, c->chan2_alt.p, s);
    std::swap(c->info, c->info_alt);
    std::swap(c->chan1, c->chan1_alt);
    if (d_iq1) std::swap(c->chan2, c->chan2_alt);
    c->signal_n = 0;
    mark(c, 1, s);
    PoolLease lease(c, s);
    rc = run_decode_stages(c, s, n, true, d_iq0, d_iq1, arena_len, (const gr80211_frame_desc *)c->desc.p, nullptr, d_out, out_stride,
                           d_out_len, d_n_pdu, nullptr, 0, true, nullptr, nullptr, &known);
    if (rc) return rc;
    if (d_info)
        HIP_TRY(c, hipMemcpyAsync(d_info, c->info.p, (size_t)n * sizeof(gr80211_frame_info), hipMemcpyDeviceToDevice, s));
    return finish_timing(c, s);
}

int gr80211_pack_pdus(gr80211_ctx *c, const uint8_t *d_out, int out_stride, const int32_t *d_out_len, int n, uint8_t *d_packed,
                      uint64_t packed_cap, uint64_t *d_total, uint64_t *d_off, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (n < 0 || !d_total || (n > 0 && (!d_out || !d_out_len || !d_packed)) || out_stride < 16 || (out_stride & 15) ||
        ((uintptr_t)d_out & 15) || ((uintptr_t)d_packed & 15))
        return fail(c, GR80211_ERR_ARG, "gr80211_pack_pdus: bad argument (slots and packed run 16-byte aligned, stride a multiple of 16)");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_pack_pdus: at most 1048576 frames per call");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    const int nblk = (n + PACK_BLOCK - 1) / PACK_BLOCK;
    int rc = ensure(c, c->pack_ws, ((size_t)n + 1024 + 8) * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long *off = (unsigned long long *)c->pack_ws.p, *blk = off + n;
    if (nblk > 0) hipLaunchKernelGGL(k_pack_sizes, dim3(nblk), dim3(PACK_BLOCK), 0, s, d_out_len, n, out_stride, off, blk);
    hipLaunchKernelGGL(k_pack_scan_blocks, dim3(1), dim3(1024), 0, s, blk, nblk, (unsigned long long *)d_total);
    if (n > 0)
        hipLaunchKernelGGL(k_pack_copy, dim3((n + 3) / 4), dim3(256), 0, s, d_out, out_stride, d_out_len, n, off, blk, d_packed,
                           (unsigned long long)packed_cap, (unsigned long long *)d_off);
    HIP_TRY(c, hipGetLastError());
    return GR80211_OK;
}

int gr80211_last_batch_refused(gr80211_ctx *c)
{
    // the scheduler's record of the last batch, as its copy to the host left it (the stream must have run it)
    if (!c || !c->totals_h) return GR80211_ERR_ARG;
    return c->totals_h->n_groups < 0 ? 1 : 0;
}

int gr80211_copy_h2d_async(gr80211_ctx *c, void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, pick(c, stream)));
    return GR80211_OK;
}

long long gr80211_pull_h2d(gr80211_ctx *c, void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (!bytes || !d_dst || !h_src) return fail(c, GR80211_ERR_ARG, "gr80211_pull_h2d: bad argument");
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipSetDevice");
    if (!c->pull_flag_h) {
        if (hipHostMalloc((void **)&c->pull_flag_h, 64) != hipSuccess || hipMalloc((void **)&c->pull_ctr_d, 64) != hipSuccess ||
            hipMemsetAsync(c->pull_ctr_d, 0, 64, pick(c, stream)) != hipSuccess)
            return fail(c, GR80211_ERR_NOMEM, "gr80211_pull_h2d: set-up failed");
        *c->pull_flag_h = 0;
        if (hipHostGetDevicePointer(&c->pull_flag_view, c->pull_flag_h, 0) != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipHostGetDevicePointer");
    }
    // one finished-workgroup counter and one flag word per context: two pull kernels running at once on different
    // streams would share them, so a context's pulls all go to the stream the first one went to
    if (!c->pull_stream_set) {
        c->pull_stream = pick(c, stream);
        c->pull_stream_set = true;
    } else if (c->pull_stream != pick(c, stream)) {
        return fail(c, GR80211_ERR_ARG, "gr80211_pull_h2d: a context's pull copies all go to one stream (use another context for another stream)");
    }
    // the last range that matched is tried first (a stream buffer is used call after call)
    void *dev_view = nullptr;
    const uintptr_t a = (uintptr_t)h_src;
    const uint64_t gen = g_host_gen.load(std::memory_order_acquire);
    if (gen != c->pull_gen || !(a >= c->pull_lo && a + bytes <= c->pull_hi)) {
        HostRange r{0, 0, 0};
        (void)host_range_of(h_src, bytes, &r);
        c->pull_lo = r.lo, c->pull_hi = r.hi, c->pull_to_dev = r.to_dev;
        c->pull_gen = gen;
    }
    if (c->pull_hi) {
        dev_view = (void *)(a + c->pull_to_dev);
    } else {
        const hipError_t e = hipHostGetDevicePointer(&dev_view, const_cast<void *>(h_src), 0);
        if (e != hipSuccess || !dev_view) {
            (void)hipGetLastError();
            return fail(c, GR80211_ERR_ARG, "gr80211_pull_h2d: the source is not page-locked host memory the GPU can read", e);
        }
    }
    void *flag_view = c->pull_flag_view;
    const long long number = ++c->pull_seq;
    launch_link_copy(d_dst, dev_view, bytes, c->pull_ctr_d, (uint32_t *)flag_view, (uint32_t)number, pick(c, stream));
    if (hipGetLastError() != hipSuccess) return fail(c, GR80211_ERR_HIP, "gr80211_pull_h2d: launch failed");
    return number;
}

int gr80211_pull_done(gr80211_ctx *c, long long number, int wait)
{
    if (!c || number < 1 || number > c->pull_seq || !c->pull_flag_h) return GR80211_ERR_ARG;
    volatile uint32_t *f = c->pull_flag_h;
    // the numbers of one stream's copies arrive in order; (int32_t) difference: the 32-bit word may wrap
    if ((int32_t)(*f - (uint32_t)number) >= 0) return 1;
    if (!wait) return 0;
    for (long spins = 0; (int32_t)(*f - (uint32_t)number) < 0; spins++) {
        if (spins > 2000 && (spins & 1023) == 0) {
            // a copy that takes this long may be behind work that has failed: ask the runtime
            const hipError_t e = hipStreamQuery(c->pull_stream_set ? c->pull_stream : c->stream);
            if (e != hipSuccess && e != hipErrorNotReady) return fail(c, GR80211_ERR_HIP, "gr80211_pull_done", e);
            if (e == hipSuccess && (int32_t)(*f - (uint32_t)number) < 0 && spins > 200000000)
                return fail(c, GR80211_ERR_HIP, "gr80211_pull_done: the copy's number never arrived");
        }
        __builtin_ia32_pause();
    }
    return 1;
}

int gr80211_push_d2h(gr80211_ctx *c, void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (!bytes) return GR80211_OK;
    if (!h_dst || !d_src) return fail(c, GR80211_ERR_ARG, "gr80211_push_d2h: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    void *dev_view = nullptr;
    HostRange r{0, 0, 0};
    if (host_range_of(h_dst, bytes, &r)) {
        dev_view = (void *)((uintptr_t)h_dst + r.to_dev);
    } else {
        const hipError_t e = hipHostGetDevicePointer(&dev_view, h_dst, 0);
        if (e != hipSuccess || !dev_view) {
            (void)hipGetLastError();
            return fail(c, GR80211_ERR_ARG, "gr80211_push_d2h: the destination is not page-locked host memory the GPU can write", e);
        }
    }
    launch_link_copy(dev_view, d_src, bytes, nullptr, nullptr, 0, pick(c, stream));
    if (hipGetLastError() != hipSuccess) return fail(c, GR80211_ERR_HIP, "gr80211_push_d2h: launch failed");
    return GR80211_OK;
}

int gr80211_copy_d2h_async(gr80211_ctx *c, void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, pick(c, stream)));
    return GR80211_OK;
}

int gr80211_copy_d2d_async(gr80211_ctx *c, void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, pick(c, stream)));
    return GR80211_OK;
}

int gr80211_stream_idle(gr80211_ctx *c, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    const hipError_t e = hipStreamQuery(pick(c, stream));
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return fail(c, GR80211_ERR_HIP, "hipStreamQuery", e);
}

int gr80211_stream_wait(gr80211_ctx *c, void *stream, gr80211_ctx *other, void *other_stream)
{
    if (!c || !other) return GR80211_ERR_ARG;
    if (c->device != other->device) return fail(c, GR80211_ERR_ARG, "gr80211_stream_wait: contexts of different devices");
    HIP_TRY(c, hipSetDevice(c->device));
    // ev[6] is not one of the profiling events
    HIP_TRY(c, hipEventRecord(other->ev[6], pick(other, other_stream)));
    HIP_TRY(c, hipStreamWaitEvent(pick(c, stream), other->ev[6], 0));
    return GR80211_OK;
}

long long gr80211_marker_record(gr80211_ctx *c, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipSetDevice");
    const long long id = c->mark_next;
    hipEvent_t &e = c->mark_ev[id % gr80211_ctx::N_MARK];
    if (!e) {
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipEventCreate");
    } else if (hipEventSynchronize(e) != hipSuccess) {  // the marker N_MARK records back gives up its place
        return fail(c, GR80211_ERR_HIP, "hipEventSynchronize");
    }
    if (hipEventRecord(e, pick(c, stream)) != hipSuccess) return fail(c, GR80211_ERR_HIP, "hipEventRecord");
    c->mark_next = id + 1;
    return id;
}

int gr80211_marker_done(gr80211_ctx *c, long long id, int wait)
{
    if (!c || id < 0 || id >= c->mark_next) return GR80211_ERR_ARG;
    if (id < c->mark_next - gr80211_ctx::N_MARK) return 1;  // its place went to a later marker, which waited for it first
    hipEvent_t e = c->mark_ev[id % gr80211_ctx::N_MARK];
    if (wait) {
        HIP_TRY(c, hipEventSynchronize(e));
        return 1;
    }
    const hipError_t r = hipEventQuery(e);
    if (r == hipSuccess) return 1;
    if (r == hipErrorNotReady) return 0;
    return fail(c, GR80211_ERR_HIP, "hipEventQuery", r);
}

int gr80211_host_register(gr80211_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // a system that cannot lock user pages says so here; the error must not stick to later calls
        return fail(c, GR80211_ERR_HIP, "hipHostRegister", e);
    }
    return GR80211_OK;
}

int gr80211_host_unregister(gr80211_ctx *c, void *p)
{
    if (!c || !p) return GR80211_ERR_ARG;
    HIP_TRY(c, hipHostUnregister(p));
    return GR80211_OK;
}

int gr80211_tx_samples(int format, int mcs, int nss, int psdu_len)
{
    TxMod m;
    return tx_mod(format, mcs, nss, psdu_len, m) ? m.nsamp : -1;
}

int gr80211_tx_batch(gr80211_ctx *c, const uint8_t *d_psdu, uint64_t psdu_bytes, const gr80211_tx_desc *desc, int n,
                     unsigned flags, void *d_iq0, void *d_iq1, uint64_t arena_len, int32_t *d_n_samples, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_tx_batch")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_psdu || !desc || n < 0 || !d_iq0) return fail(c, GR80211_ERR_ARG, "gr80211_tx_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_tx_batch: at most 1048576 frames per call");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    const void *dd = nullptr;
    int rc = stage_desc(c, s, desc, (size_t)n * sizeof(gr80211_tx_desc), flags, c->tx_desc, &dd);
    if (rc) return rc;
    launch_tx(d_psdu, psdu_bytes, (const gr80211_tx_desc *)dd, n, d_iq0, d_iq1, arena_len, d_n_samples, s);
    HIP_TRY(c, hipGetLastError());
    return GR80211_OK;
}

int gr80211_tx_mu_batch(gr80211_ctx *c, const uint8_t *d_psdu, uint64_t psdu_bytes, const gr80211_tx_mu_desc *desc, int n,
                        unsigned flags, const void *d_q, int n_q, void *d_iq0, void *d_iq1, uint64_t arena_len,
                        int32_t *d_n_samples, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    if (int rc_busy = refuse_if_pending(c, "gr80211_tx_mu_batch")) return rc_busy;
    if (n == 0) return GR80211_OK;
    if (!d_psdu || !desc || n < 0 || !d_q || n_q < 1 || !d_iq0 || !d_iq1)
        return fail(c, GR80211_ERR_ARG, "gr80211_tx_mu_batch: bad argument");
    if (n > (1 << 20)) return fail(c, GR80211_ERR_ARG, "gr80211_tx_mu_batch: at most 1048576 frames per call");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    const void *dd = nullptr;
    int rc = stage_desc(c, s, desc, (size_t)n * sizeof(gr80211_tx_mu_desc), flags, c->tx_desc, &dd);
    if (rc) return rc;
    launch_tx_mu(d_psdu, psdu_bytes, (const gr80211_tx_mu_desc *)dd, n, d_q, n_q, d_iq0, d_iq1, arena_len, d_n_samples, s);
    HIP_TRY(c, hipGetLastError());
    return GR80211_OK;
}

void *gr80211_dev_alloc(gr80211_ctx *c, size_t bytes)
{
    if (!c) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}

void gr80211_dev_free(gr80211_ctx *c, void *p)
{
    if (c && p) (void)hipFree(p);
}

void *gr80211_host_alloc(gr80211_ctx *c, size_t bytes)
{
    if (!c) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    void *view = nullptr;
    if (hipHostGetDevicePointer(&view, p, 0) == hipSuccess && view) {
        std::lock_guard<std::mutex> lk(g_host_mutex);
        g_host_ranges.push_back({(uintptr_t)p, (uintptr_t)p + (bytes ? bytes : 1), (intptr_t)((uintptr_t)view - (uintptr_t)p)});
        g_host_gen.fetch_add(1, std::memory_order_release);
    } else {
        (void)hipGetLastError();
    }
    return p;
}

void gr80211_host_free(gr80211_ctx *c, void *p)
{
    if (!c || !p) return;
    {
        std::lock_guard<std::mutex> lk(g_host_mutex);
        for (size_t i = 0; i < g_host_ranges.size(); i++)
            if (g_host_ranges[i].lo == (uintptr_t)p) {
                g_host_ranges.erase(g_host_ranges.begin() + (long)i);
                g_host_gen.fetch_add(1, std::memory_order_release);
                break;
            }
    }
    (void)hipHostFree(p);
}

int gr80211_copy_h2d(gr80211_ctx *c, void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    HIP_TRY(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return GR80211_OK;
}

int gr80211_copy_d2h(gr80211_ctx *c, void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    HIP_TRY(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return GR80211_OK;
}

int gr80211_copy_d2d(gr80211_ctx *c, void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!c) return GR80211_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return GR80211_OK;
}

int gr80211_use_work_pool(gr80211_ctx *c, int max_sets)
{
    if (!c || max_sets < 0 || max_sets > 64) return GR80211_ERR_ARG;
    if (c->pending.active) return fail(c, GR80211_ERR_ARG, "gr80211_use_work_pool: a submitted batch is waiting for gr80211_rx_finish");
    if (max_sets == 0) {
        pool_leave(c);
        return GR80211_OK;
    }
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    DevicePool &dp = g_pool[c->device];
    if (c->pool_sets <= 0) dp.users++;
    c->pool_sets = max_sets;
    // max_sets survivor buffers and two soft-bit buffers more, so that the demodulators of the batches after next need not
    // wait for the oldest decoder (GR80211_POOL_SOFT_EXTRA=<n>: another number, for experiments)
    int extra = 2;
    if (const char *e = getenv("GR80211_POOL_SOFT_EXTRA")) extra = atoi(e);
    if (max_sets > dp.max_items[POOL_SURV]) dp.max_items[POOL_SURV] = max_sets;
    if (max_sets + extra > dp.max_items[POOL_SOFT]) dp.max_items[POOL_SOFT] = max_sets + extra > 1 ? max_sets + extra : 1;
    return GR80211_OK;
}

int gr80211_work_memory(gr80211_ctx *c, uint64_t *own_bytes, uint64_t *pool_bytes, int *pool_sets)
{
    if (!c) return GR80211_ERR_ARG;
    if (own_bytes) *own_bytes = (uint64_t)c->llr.cap + c->dec.cap + c->bitsw.cap;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    const DevicePool &dp = g_pool[c->device];
    uint64_t b = 0;
    for (int k = 0; k < POOL_KINDS; k++)
        for (const PoolItem *w : dp.items[k]) b += (uint64_t)w->a.cap + w->b.cap;
    if (pool_bytes) *pool_bytes = b;
    if (pool_sets) *pool_sets = (int)dp.items[POOL_SURV].size();
    return GR80211_OK;
}

int gr80211_set_mu_pos(gr80211_ctx *c, int mu_pos)
{
    if (!c || mu_pos < 0 || mu_pos > 1) return GR80211_ERR_ARG;
    c->mu_pos = (uint32_t)mu_pos;
    return GR80211_OK;
}

int gr80211_set_ampdu_strict(gr80211_ctx *c, int on)
{
    if (!c) return GR80211_ERR_ARG;
    c->ampdu_strict = on != 0;
    return GR80211_OK;
}

int gr80211_set_lane_state_max(gr80211_ctx *c, int max_frames)
{
    if (!c) return GR80211_ERR_ARG;
    c->lane_state_max = max_frames < 0 ? c->lane_state_default : max_frames;
    return GR80211_OK;
}

int gr80211_set_profiling(gr80211_ctx *c, int on)
{
    if (!c) return GR80211_ERR_ARG;
    c->profiling = on != 0;
    return GR80211_OK;
}

int gr80211_last_timing(gr80211_ctx *c, float *ms, int n)
{
    if (!c || !ms) return GR80211_ERR_ARG;
    for (int i = 0; i < n && i < 6; i++) ms[i] = c->ms[i];
    return GR80211_OK;
}

#ifdef GR80211_TIMELINE
// Debug build only (not declared in the public header): per-group time stamps of the last k_decode launch.
int gr80211_debug_timeline(unsigned long long *dst, int n_groups) { return gr80211::read_timeline(dst, n_groups); }
#endif

}  // extern "C"
