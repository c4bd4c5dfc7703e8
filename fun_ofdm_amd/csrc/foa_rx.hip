// foa_rx.hip -- the C ABI of include/fun_ofdm_amd.h over the gfx950 kernels.
// Built by fun_ofdm_amd/csrc/Makefile:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>

#include "frontend_kernels.h"
#include "frontend_lps.h"
#include "frontend_q4.h"
#include "viterbi_v1.h"
#include "viterbi_v3.h"
#if FOA_XCHECK
#include "viterbi_v4.h"      // four states per lane: an exact alternative forward pass, measured and not adopted
#endif
#include "stage_kernels.h"
#include "sync_host.h"
#include "sync_kernels.h"
#include "tx_kernels.h"
#include "probe_kernels.h"

using namespace foa;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? FOA_E_NOMEM : FOA_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Every entry point that queues work starts here: select the handle's device and FORGET whatever error an earlier HIP call on this
// thread left behind -- a failed call of ours that was already reported, a polled hipEventQuery, or another library's probing (PyTorch
// asks about peers a one-GPU box does not have: "invalid device ordinal").  The launch checks below (hipGetLastError after the kernels
// are queued) must report THIS call's errors only; round 4 found a transmit-side test failing on a stale one.
static inline hipError_t enter_device(int device)
{
    (void)hipGetLastError();
    return hipSetDevice(device);
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t want)
    {
        if (want <= n) return FOA_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        HIP_TRY(hipMalloc((void **)&p, want * sizeof(T)));
        n = want;
        return FOA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

// ---- host-built constant tables -------------------------------------------------------------------
void foa::build_tables(DeviceTables *t)
{
    memset(t, 0, sizeof *t);
    // rates.h:52-196
    static const int rows[kNumRates][5] = {
        { 0xD, 48, 24, 1, 0 }, { 0xE, 48, 32, 1, 1 }, { 0xF, 48, 36, 1, 2 }, { 0x5, 96, 48, 2, 0 }, { 0x6, 96, 64, 2, 1 }, { 0x7, 96, 72, 2, 2 },
        { 0x9, 192, 96, 4, 0 }, { 0xA, 192, 128, 4, 1 }, { 0xB, 192, 144, 4, 2 }, { 0x1, 288, 192, 6, 1 }, { 0x3, 288, 216, 6, 2 } };
    for (int r = 0; r < kNumRates; r++) {
        RateRow &x = t->rates[r];
        x.rate_field = rows[r][0]; x.cbps = rows[r][1]; x.dbps = rows[r][2]; x.bpsc = rows[r][3]; x.punct = rows[r][4];
        // qam.h:35-51: NumBits = bits per axis, power 1.0 for BPSK else 0.5 (modulator.cpp:117-157)
        int nb = x.bpsc == 1 ? 1 : x.bpsc / 2;
        double power = x.bpsc == 1 ? 1.0 : 0.5;
        int nn = 1 << (nb - 1), sum2 = (4 * nn * nn * nn - nn) / 3;
        double sf = std::sqrt(power * (double)nn / (double)sum2);
        x.numbits = nb;
        x.scale_d = (double)(1 << (8 - nb)) / sf;
    }
    for (int k = 0; k < 64; k++) {
        double a = -2.0 * M_PI * (double)k / 64.0;
        t->tw_re[k] = std::cos(a); t->tw_im[k] = std::sin(a);
    }
    t->tw_re[0] = 1; t->tw_im[0] = 0; t->tw_re[16] = 0; t->tw_im[16] = -1; t->tw_re[32] = -1; t->tw_im[32] = 0; t->tw_re[48] = 0; t->tw_im[48] = 1;
    // 802.11a-1999 17.3.3 long training sequence L(-26..26); preamble.h:363 stores it at index k+32
    static const signed char lts[53] = { 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                                         1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1 };
    for (int i = 0; i < 53; i++) t->lts_freq[i + 6] = lts[i];
    // pilot polarity p_0..126 (17.3.5.9): scrambler sequence for the all-ones seed, 0 -> +1, 1 -> -1
    int st = 0x7F;
    for (int i = 0; i < 127; i++) {
        int fb = ((st >> 6) ^ (st >> 3)) & 1;
        st = ((st << 1) & 0x7E) | fb;
        t->polarity[i] = fb ? -1 : 1;
    }
    // phase_tracker.cpp:37-50
    int n = 0;
    for (int s = 0; s < 64; s++) {
        t->data_index[s] = -1;
        if (s < 6 || s > 58 || s == 32) t->carrier_kind[s] = 0;
        else if (s == 11 || s == 25 || s == 39 || s == 53) t->carrier_kind[s] = 2;
        else { t->carrier_kind[s] = 1; t->data_index[s] = (int8_t)n++; }
    }
    // ppdu.cpp:256-264: per-byte feedback bit of the 7-bit LFSR seeded with 93 (period 127)
    st = 93;
    for (int i = 0; i < 128; i++) {
        int fb = ((st >> 6) & 1) ^ ((st >> 3) & 1);
        t->scramble[i] = (uint8_t)fb;
        st = ((st << 1) & 0x7E) | fb;
    }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        t->crc_table[i] = c;
    }
    std::complex<double> ltc[64];
    foa::make_lts_time_conj(ltc);
    for (int i = 0; i < 64; i++) { t->lts_conj_re[i] = ltc[i].real(); t->lts_conj_im[i] = ltc[i].imag(); }
    std::complex<double> pre[320];
    foa::make_preamble(pre);
    for (int i = 0; i < 320; i++) { t->preamble_re[i] = pre[i].real(); t->preamble_im[i] = pre[i].imag(); }
    // qam.h:110-125 with NumBits = 3 (fewer bits = a prefix of the same loop); |pt| >= 320 gives the value of +-320
    for (int p = -320; p <= 320; p++) {
        uint32_t pt = (uint32_t)p, word = 0;
        int flip = 1, amp = 128;
        for (int i = 0; i < 3; i++) {
            int v = (int)((uint32_t)flip * pt + 128u);
            word |= (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)) << (8 * i);
            int bit = ((int)pt < 0) ? -1 : 1;
            pt -= (uint32_t)(bit * amp);
            flip = -bit;
            amp >>= 1;
        }
        t->qam_lut[p + 320] = word;
    }
    for (int r = 0; r < kNumRates; r++)
        for (int c = 0; c < t->rates[r].cbps; c++) {
            const int w = c % 48, dd = 48 * (c / 48) + 16 * (w % 3) + w / 3, punct = t->rates[r].punct;     // interleaver.h:66-75 inverse
            static const int k34[4] = { 0, 1, 3, 5 }, k23[3] = { 0, 2, 3 };
            t->sym_pos[r][c] = (uint16_t)(punct == 2 ? 6 * (dd >> 2) + k34[dd & 3] : punct == 1 ? 4 * (dd / 3) + k23[dd % 3] : dd);
        }
}

// Everything one decode call writes between its header kernel and its finish kernel.  Several sets rotate (kSets) so that the
// chain-back of call k can run under the forward pass of call k+1 while call k+2's front end is already filling the next one.
struct WorkSet {
    DevBuf<FrameInfo> info;
    DevBuf<double2> hinv;
    DevBuf<int32_t> sym2frame, seg2frame;
    DevBuf<uint16_t> tb_state;
    DevBuf<uint64_t> dec;
    DevBuf<uint64_t> dec4;        // option "forward" = 4 while its decisions are still converted for viterbi_v3.h's chain-back: the forward pass's own layout
    DevBuf<uint16_t> sp;          // depunctured soft pairs, one per trellis step (front end -> forward pass, taps)
    DevBuf<uint32_t> decoded;
    DevBuf<int64_t> totals;
    DevBuf<double2> eq_sig, eq_data;
    size_t sym_cap = 0, dec_cap = 0;
    hipEvent_t ev[8] = {};       // start, after header, after scan, after symbols, end, after the forward pass, start of the finish,
                                 // start of the forward pass (pipelined path)
    WorkSet *before = nullptr;   // the set of the call queued before this one (pipelined path)
    hipEvent_t done = nullptr, walk_done = nullptr;
    bool used = false, have_timing = false, piped = false;
    void release_all()
    {
        info.release(); hinv.release(); sym2frame.release(); seg2frame.release(); tb_state.release(); dec.release(); dec4.release(); sp.release();
        decoded.release(); totals.release(); eq_sig.release(); eq_data.release();
    }
};

// One asynchronous host-pointer call (foa_rx_submit_host): device staging for inputs and outputs, page-locked host mirrors
// of both (the caller's buffers are free again when submit returns; the results wait in ours until they are collected).
struct HostJob {
    bool busy = false;
    uint64_t ticket = 0;
    DevBuf<uint8_t> dev;
    uint8_t *pin = nullptr;          // page-locked host buffer, same layout as dev
    size_t pin_cap = 0;              // its size
    size_t total = 0, o_psdu = 0, o_res = 0, n_frames = 0, slot_bytes = 0;      // layout of the call in flight
    hipEvent_t done = nullptr;
    bool copy_queued = false;
};
constexpr int kMaxJobs = 8;

constexpr int kSets = 6;

struct foa_stream;
void foa_stream_shutdown(foa_stream *s);

struct foa_rx {
    int device = 0;
    hipStream_t stream = nullptr;      // everything when calls run in line; the first lane of pipelined calls
    hipStream_t stream2 = nullptr;     // pipelined path: the stitch / CRC kernel behind a walk (lanes), or walk + finish (lanes off)
    hipStream_t stream3 = nullptr;     // pipelined path: copies of the host-pointer entry points and the pre-sync stage (lanes), or header + scan +
                                       // data symbols (lanes off)
    hipStream_t stream4 = nullptr;     // the second lane of pipelined calls (the first is `stream`)
    hipStream_t stream5 = nullptr, stream6 = nullptr;      // third and fourth lane, used for small grids (option "depth")
    int hw_queues = 4, max_depth = 4;  // hardware queues the runtime was started with, as far as the environment tells (foa_rx_create), and the depth they allow
    std::string notes;                 // non-fatal remarks about how the handle is set up (foa_rx_notes)
    int64_t sync_origin = 0;           // one-shot device pre-sync: stream index of d_iq[0] (option "sync_origin")
    int64_t stream_longest = 0;        // foa_stream_create: longest frame (samples, preamble to last symbol + 192) the stream will hold; 0 = any frame the format allows
    int depth = 0;                     // lanes: how many calls' loops are in flight; 0 = by grid size (2, or 4 below kDeepBelow frames)
    int depth_saved = -1;              // (the stream engine pins 2 while a stream is open and restores this)
    unsigned n_calls = 0;              // pipelined decode calls made so far (a call's lane is n_calls mod depth)
    int fwd_calls = 0;
    int fwd_kind = 3;            // forward pass: 3 viterbi_v3.h (a state per lane, two frames per wave), 4 viterbi_v4.h (four states per lane, four frames per wave)
    int viterbi_kind = 2;        // 0: lane per state (viterbi_v1.h), 1: packed, serial chain-back (v2), 2: packed, segment chain-back (v3)
    int tb_segment = 960, tb_overlap = 96;   // v3 chain-back: data steps per segment / run-in steps (multiples of 96)
    bool pipeline = true;        // v3: finish of one call overlaps the next call's front end (two work sets, two streams)
    bool record_eq = false;
    bool record_soft = true;     // (the soft bytes are the front end's output and always there; the option is accepted for compatibility)
    hipEvent_t in_ready = nullptr;     // inputs copied by a host-pointer entry point are on the device (recorded on the copy stream)
    bool in_wait = false;              // ... and the next decode call's front end has to wait for that
    bool lanes = true;           // pipelined path: everything on a call's critical loop -- front end, forward pass, chain-back walk -- on ONE stream per
                                 // call parity (below); false: front end on the third stream, walk + finish on the second (the round-1 arrangement)
    int fe_hold = 1;             // pipelined path: 1 = header, scan and data symbols of call k+1 wait for the chain-back walk of call k-1;
                                 // 2 = only the data-symbol kernel does; 0 = nothing is held back (A/B measurement)
    int64_t sync_call = kSyncCallDefault;    // pre-sync: the reference receiver's call size to decide timing_sync.cpp:99 by (csrc/sync_host.h); 0 = as one call
    int walk_on_lane = 1;        // lanes: the chain-back walk runs on the call's lane (1) or with the finish on the second stream, behind the forward pass's event (0; A/B)
    int sync_flags_kind = 1;     // k_sync_flags (1) or, in the cross-check build, k_sync_flags_direct (0)
    int frontend_kind = -1;      // 1: one lane per symbol (frontend_lps.h), 0: one wave per symbol, 2: four lanes per symbol (frontend_q4.h);
                                 // -1: the default, 2
    WorkSet sets[kSets];         // up to depth + 1 are in use at any time (below); one more keeps the call before them readable (timings)
    WorkSet *w = &sets[0];       // the set of the most recent decode call
    WorkSet *prev = nullptr;     // the set of the call before it (kernel times of a call that is certainly complete)
    // Pipelined path: the chain-back + finish of a call is queued (on stream2) only when the NEXT call has queued its
    // front end, so that it runs under that call's forward pass (memory-bound next to issue-bound) rather than under its
    // latency-bound front end; foa_rx_sync and everything that needs results queue it at once.
    struct Pending {
        bool valid = false, lanes = false;
        hipStream_t lane = nullptr;  // lanes: the stream of the call's forward pass, where its walk follows
        WorkSet *w = nullptr;
        int nf = 0, S = 0, L = 0;
        size_t max_segs = 0, slot_bytes = 0;
        uint8_t *psdu = nullptr;
        foa_frame_result *results = nullptr;
        HostJob *job = nullptr;  // submit_host: copy the outputs back once the finish is queued
    } pending;
    HostJob jobs[kMaxJobs];
    uint64_t next_ticket = 1;
    HostJob *attach_job = nullptr;   // set by submit_host around its decode call
    DevBuf<uint8_t> scratch;     // staging for the host-pointer entry points
    DevBuf<uint32_t> sy_flags;   // device pre-sync workspace
    DevBuf<int32_t> sy_cnt, sy_off, sy_keep, sy_n;
    DevBuf<int64_t> sy_x;
    DevBuf<SyncCand> sy_cand;
    size_t last_frames = 0;
    int32_t *sy_pin = nullptr;       // page-locked { STS_END candidates, -, -, alignments found } of the pre-sync begun last (foa_rx_sync_dev_begin)
    hipEvent_t sy_done = nullptr;
    bool sy_open = false;
    int32_t sy_ccap = 0;
    size_t sy_cap = 0;
    struct foa_stream *open_stream = nullptr;      // the stream engine that owns this handle right now (stream_engine.h), if any
    int64_t ns_wait_set = 0;     // host time spent waiting for a work set to come free (the GPU is more than kSets - 1 calls behind)
};

namespace {

// A machine that a call fills (config 2: five forward-pass waves per SIMD) is best served by two calls' loops in flight; a call of a few
// thousand frames leaves most SIMDs one wave or none, its forward pass lasts as long as ONE wave needs for its frames' trellis steps
// whatever the batch, and more loops in flight are what raises the throughput then (1 000 frames x 4 092 bytes at 54 Mbps: 1.43 ms per
// batch with two, 0.94 with four).  Needs as many hardware queues as streams in use:
// GPU_MAX_HW_QUEUES >= 6 (the runtime's default of 4 makes two lanes share a queue, i.e. run one after the other).
constexpr int kDeepBelow = 2049;                 // frames: up to one forward-pass wave per SIMD (measured: 1 000-frame batches gain 40-50 % with four
                                                 // loops in flight, a 4 000-frame stream of mixed rates loses 5 %, 10 000 frames lose 9 %)
hipStream_t lane_stream(foa_rx *rx, int lane) { return lane == 0 ? rx->stream : lane == 1 ? rx->stream4 : lane == 2 ? rx->stream5 : rx->stream6; }

// Host-pointer entry points copy their inputs (and the pre-sync stage runs) on the third stream when calls are pipelined, off the
// lanes, so that a copy never sits behind a forward pass; the decode call that follows waits for the event.
hipStream_t side_stream(foa_rx *rx) { return (rx->pipeline && rx->viterbi_kind == 2) ? rx->stream3 : rx->stream; }
int inputs_queued(foa_rx *rx, hipStream_t cs)
{
    if (cs == rx->stream) return FOA_OK;
    HIP_TRY(hipEventRecord(rx->in_ready, cs));
    rx->in_wait = true;
    return FOA_OK;
}

// the forward pass of a work set's frames (option "forward")
int launch_forward(foa_rx *rx, hipStream_t st, WorkSet *w, int nf)
{
#if FOA_XCHECK
    if (rx->fwd_kind == 5 && rx->fwd_calls++ < 24) {
        // timing experiment only (FOA_FORWARD=5, tools/exp_forward_v4.sh pipelined): the first calls run viterbi_v3.h's pass, so that every work set holds valid decisions of
        // the batch (the bench decodes the same batch every step); from then on viterbi_v4.h's pass runs WITHOUT the conversion and the chain-back reads
        // those -- the step as it would be with a chain-back of the same cost for the new layout
        launch_fwd3(st, w->info.p, nf, w->sp.p, w->dec.p);
        return FOA_OK;
    }
    if (rx->fwd_kind >= 4) {
        int rc = w->dec4.ensure(w->dec.n);
        if (rc) return rc;
        const size_t cap = w->dec.n < w->sp.n ? w->dec.n : w->sp.n;
        launch_fwd4(st, w->info.p, nf, w->sp.p, w->dec4.p, cap);
        if (rx->fwd_kind == 4) launch_dec4_to_dec3(st, w->info.p, nf, w->dec4.p, w->dec.p);
        return FOA_OK;
    }
#endif
    launch_fwd3(st, w->info.p, nf, w->sp.p, w->dec.p);
    return FOA_OK;
}

int workspace(foa_rx *rx, size_t n_samples, size_t n_frames)
{
    // (room for an eighth more frames than asked for, in steps of 1024: a stream's batches differ by a few frames, and a buffer that
    // grows by one element costs a hipFree, which waits for the whole device)
    n_frames = ((n_frames + (n_frames >> 3) + 1024) & ~(size_t)1023);
    size_t sym_cap = n_samples / 80 + 4;
    size_t dec_cap = 216 * sym_cap + 192 * (n_frames + 1);
    int rc;
    if ((rc = rx->w->info.ensure(n_frames + 1)) || (rc = rx->w->hinv.ensure((n_frames + 1) * 64)) || (rc = rx->w->sym2frame.ensure(sym_cap)) ||
        (rc = rx->w->dec.ensure(dec_cap)) || (rc = rx->w->sp.ensure(dec_cap)) || (rc = rx->w->decoded.ensure(dec_cap)) ||
        (rc = rx->w->totals.ensure(8 + 4 * ((n_frames + kScanBlock - 1) / kScanBlock + 1))))
        return rc;
    if (rx->record_eq && ((rc = rx->w->eq_sig.ensure((n_frames + 1) * 48)) || (rc = rx->w->eq_data.ensure(sym_cap * 48)))) return rc;
    // chain-back segments: every frame has at most dec_words/segment + 1 of them
    const size_t seg_cap = dec_cap / 96 + n_frames + 64;
    if ((rc = rx->w->seg2frame.ensure(seg_cap)) || (rc = rx->w->tb_state.ensure(seg_cap))) return rc;
    // capacities handed to the scan are those of the buffers actually allocated
    rx->w->sym_cap = rx->w->sym2frame.n; rx->w->dec_cap = rx->w->dec.n < rx->w->sp.n ? rx->w->dec.n : rx->w->sp.n;
    if (rx->record_eq && rx->w->eq_data.n / 48 < rx->w->sym_cap) rx->w->sym_cap = rx->w->eq_data.n / 48;
    return FOA_OK;
}

// queue the deferred chain-back + finish; after_front_end: the event of the call it should run under (or null: now)
int flush_pending(foa_rx *rx, hipEvent_t after_front_end)
{
    foa_rx::Pending &p = rx->pending;
    if (!p.valid) return FOA_OK;
    hipStream_t sb = rx->stream2;
    if (p.lanes && rx->walk_on_lane) {
        // the walk follows its forward pass on the call's own lane -- no event between them -- and the next call of that lane
        // queues its front end behind it; the stitch/CRC kernel, which nothing on the loop waits for, goes to the second stream
        if (after_front_end) HIP_TRY(hipStreamWaitEvent(p.lane, after_front_end, 0));
        HIP_TRY(hipEventRecord(p.w->ev[6], p.lane));
        launch_finish3(p.lane, sb, p.w->info.p, p.nf, p.w->dec.p, p.w->decoded.p, p.w->seg2frame.p, p.w->totals.p, p.w->tb_state.p, p.max_segs, p.S, p.L,
                       p.psdu, p.slot_bytes, p.results, p.w->walk_done);
    } else {
        HIP_TRY(hipStreamWaitEvent(sb, p.w->ev[5], 0));      // its forward pass (the timing event doubles as the dependency:
                                                              // every packet between two forward passes costs their streams microseconds)
        if (after_front_end) HIP_TRY(hipStreamWaitEvent(sb, after_front_end, 0));
        HIP_TRY(hipEventRecord(p.w->ev[6], sb));
        launch_finish3(sb, sb, p.w->info.p, p.nf, p.w->dec.p, p.w->decoded.p, p.w->seg2frame.p, p.w->totals.p, p.w->tb_state.p, p.max_segs, p.S, p.L,
                       p.psdu, p.slot_bytes, p.results, p.w->walk_done);
    }
    HIP_TRY(hipEventRecord(p.w->ev[4], sb));
    HIP_TRY(hipEventRecord(p.w->done, sb));
    if (p.job) {
        HostJob &j = *p.job;
        HIP_TRY(hipMemcpyAsync(j.pin + j.o_psdu, j.dev.p + j.o_psdu, j.total - j.o_psdu, hipMemcpyDeviceToHost, sb));
        HIP_TRY(hipEventRecord(j.done, sb));
        j.copy_queued = true;
        p.job = nullptr;
    }
    HIP_TRY(hipGetLastError());
    p.valid = false;
    return FOA_OK;
}

int drain(foa_rx *rx)
{
    int rc = flush_pending(rx, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream2));
    HIP_TRY(hipStreamSynchronize(rx->stream3));
    HIP_TRY(hipStreamSynchronize(rx->stream4));
    HIP_TRY(hipStreamSynchronize(rx->stream5));
    HIP_TRY(hipStreamSynchronize(rx->stream6));
    return FOA_OK;
}

}  // namespace

extern "C" {

int foa_version(void) { return FOA_VERSION; }
const char *foa_last_error(void) { return g_err.c_str(); }

int foa_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(FOA_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int foa_recommended_hw_queues(void) { return 8; }

const char *foa_rx_notes(foa_rx *rx) { return rx ? rx->notes.c_str() : ""; }

int foa_rx_create(foa_rx **out, int device)
{
    if (!out) return fail(FOA_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(FOA_E_NO_DEVICE, "no HIP device (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(FOA_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(FOA_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    foa_rx *rx = new foa_rx();
    rx->device = device;
    // Six streams want six hardware queues (lane_stream above).  The runtime reads GPU_MAX_HW_QUEUES once, when it starts, so it is the
    // HOST PROCESS that sets it (foa_recommended_hw_queues(); bench.py and examples/foa_sim.cpp do) -- a library does not edit its
    // host's environment.  What the variable says now is all that can be known here: with fewer than six queues four lanes would
    // share queues and run one after the other, so small grids then keep two loops in flight and the handle says so (foa_rx_notes).
    {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        rx->hw_queues = (q && atoi(q) > 0) ? atoi(q) : 4;             // (the runtime's default)
        if (rx->hw_queues < 6) {
            rx->max_depth = 2;
            char buf[320];
            snprintf(buf, sizeof buf, "GPU_MAX_HW_QUEUES is %s (%d hardware queues): decode calls of fewer than %d frames keep 2 loops in flight instead of 4 "
                     "(20-30 %% slower for such batches); set GPU_MAX_HW_QUEUES=%d in the environment before the HIP runtime starts. ",
                     q ? "set low" : "unset", rx->hw_queues, kDeepBelow, foa_recommended_hw_queues());
            rx->notes += buf;
        }
    }
    HIP_TRY(hipStreamCreateWithFlags(&rx->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&rx->stream2, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&rx->stream3, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&rx->stream4, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&rx->stream5, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&rx->stream6, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&rx->in_ready, hipEventDisableTiming));
    for (auto &ws : rx->sets) {
        for (auto &e : ws.ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ws.walk_done, hipEventDisableTiming));
    }
    DeviceTables tab;
    build_tables(&tab);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &tab, sizeof tab));
#if FOA_XCHECK
    if (const char *e = getenv("FOA_FORWARD")) { const int v = atoi(e); if (v >= 3 && v <= 5) rx->fwd_kind = v; }      // (A/B runs of tools/exp_forward_v4.sh; the option "forward" is the interface)
#endif
    *out = rx;
    return FOA_OK;
}

void foa_rx_destroy(foa_rx *rx)
{
    if (!rx) return;
    if (rx->open_stream) foa_stream_shutdown(rx->open_stream);       // (joins the engine's threads: they use the handle; the owner still frees the shell)
    (void)hipSetDevice(rx->device);
    (void)drain(rx);
    for (auto &ws : rx->sets) {
        ws.release_all();
        for (auto &e : ws.ev) if (e) (void)hipEventDestroy(e);
        if (ws.done) (void)hipEventDestroy(ws.done);
        if (ws.walk_done) (void)hipEventDestroy(ws.walk_done);
    }
    if (rx->in_ready) (void)hipEventDestroy(rx->in_ready);
    if (rx->sy_done) (void)hipEventDestroy(rx->sy_done);
    if (rx->sy_pin) (void)hipHostFree(rx->sy_pin);
    for (auto &j : rx->jobs) {
        j.dev.release();
        if (j.pin) (void)hipHostFree(j.pin);
        if (j.done) (void)hipEventDestroy(j.done);
    }
    rx->scratch.release();
    rx->sy_flags.release(); rx->sy_cnt.release(); rx->sy_off.release(); rx->sy_keep.release(); rx->sy_n.release(); rx->sy_x.release(); rx->sy_cand.release();
    if (rx->stream) (void)hipStreamDestroy(rx->stream);
    if (rx->stream2) (void)hipStreamDestroy(rx->stream2);
    if (rx->stream3) (void)hipStreamDestroy(rx->stream3);
    if (rx->stream4) (void)hipStreamDestroy(rx->stream4);
    if (rx->stream5) (void)hipStreamDestroy(rx->stream5);
    if (rx->stream6) (void)hipStreamDestroy(rx->stream6);
    delete rx;
}

int foa_rx_reserve(foa_rx *rx, size_t n_samples, size_t n_frames)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    WorkSet *keep = rx->w;
    int rc = FOA_OK;
    for (auto &ws : rx->sets) {                                        // both work sets (the second only matters when pipelining)
        rx->w = &ws;
        if ((rc = workspace(rx, n_samples, n_frames))) break;
        if (!(rx->pipeline && rx->viterbi_kind == 2)) break;
    }
    rx->w = keep;
    return rc;
}

int foa_rx_set_option(foa_rx *rx, const char *name, int64_t value)
{
    if (!rx || !name) return fail(FOA_E_INVALID, "NULL argument");
    if (!strcmp(name, "viterbi")) {
        if (value < 0 || value > 2) return fail(FOA_E_INVALID, "viterbi must be 0 (lane per state), 1 (packed, serial chain-back) or 2 (packed, segment chain-back)");
        if (!FOA_XCHECK && value != 2) return fail(FOA_E_INVALID, "viterbi %d is a cross-check kernel: it is only in libfun_ofdm_amd_xcheck.so (make xcheck)", (int)value);
        rx->viterbi_kind = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "forward")) {
        if (value != 3 && value != 4) return fail(FOA_E_INVALID, "forward must be 3 (viterbi_v3.h) or 4 (viterbi_v4.h)");
        if (!FOA_XCHECK && value != 3) return fail(FOA_E_INVALID, "forward 4 is a cross-check kernel: it is only in libfun_ofdm_amd_xcheck.so (make xcheck)");
        rx->fwd_kind = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "tb_segment")) {
        if (value < 96 || value > kTbMaxSeg || value % 96) return fail(FOA_E_INVALID, "tb_segment must be a multiple of 96 in [96, %d]", kTbMaxSeg);
        rx->tb_segment = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "tb_overlap")) {
        if (value < 0 || value > 3072 || value % 96) return fail(FOA_E_INVALID, "tb_overlap must be a multiple of 96 in [0, 3072]");
        rx->tb_overlap = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "pipeline")) { rx->pipeline = value != 0; return FOA_OK; }
    if (!strcmp(name, "record_eq")) { rx->record_eq = value != 0; return FOA_OK; }
    if (!strcmp(name, "fe_hold")) {
        if (value < 0 || value > 2) return fail(FOA_E_INVALID, "fe_hold must be 0, 1 or 2");
        rx->fe_hold = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "record_soft")) { rx->record_soft = value != 0; return FOA_OK; }
    if (!strcmp(name, "depth")) {
        if (value != 0 && (value < 2 || value > 4)) return fail(FOA_E_INVALID, "depth must be 0 (by grid size), 2, 3 or 4");
        rx->depth = (int)value;
        if (value > rx->max_depth && rx->notes.find("option depth") == std::string::npos)
            rx->notes += "option depth exceeds what the hardware queues the runtime started with can run side by side: lanes will share queues. ";
        return FOA_OK;
    }
    if (!strcmp(name, "lanes")) { int rc0 = drain(rx); if (rc0) return rc0; rx->lanes = value != 0; return FOA_OK; }
    if (!strcmp(name, "frontend")) {
        if (value < -1 || value > 2) return fail(FOA_E_INVALID, "frontend must be -1 (by context), 0 (wave per symbol), 1 (lane per symbol) or 2 (quad per symbol)");
        if (!FOA_XCHECK && (value == 0 || value == 1)) return fail(FOA_E_INVALID, "frontend %d is a cross-check kernel: it is only in libfun_ofdm_amd_xcheck.so (make xcheck)", (int)value);
        rx->frontend_kind = (int)value;
        return FOA_OK;
    }
    if (!strcmp(name, "walk_lane")) { int rc0 = drain(rx); if (rc0) return rc0; rx->walk_on_lane = value != 0; return FOA_OK; }
    if (!strcmp(name, "sync_call")) {
        if (value != 0 && value <= 160) return fail(FOA_E_INVALID, "sync_call must be 0 (decide as one call over the whole stream) or > 160 (timing_sync.cpp:55)");
        if (rx->open_stream) return fail(FOA_E_STATE, "sync_call cannot change while a stream engine is open on the handle (its submitter thread reads it)");
        rx->sync_call = value;
        return FOA_OK;
    }
    if (!strcmp(name, "stream_longest")) {
        if (value != 0 && (value < 1024 || value > 110592)) return fail(FOA_E_INVALID, "stream_longest must be 0 (any frame: 110 592 samples) or lie in [1024, 110592]");
        if (rx->open_stream) return fail(FOA_E_STATE, "stream_longest is read when a stream is created");
        rx->stream_longest = value;
        return FOA_OK;
    }
    if (!strcmp(name, "sync_origin")) {
        if (value < 0) return fail(FOA_E_INVALID, "sync_origin is a stream index (>= 0)");
        if (rx->open_stream) return fail(FOA_E_STATE, "sync_origin cannot change while a stream engine is open on the handle");
        rx->sync_origin = value;
        return FOA_OK;
    }
    if (!strcmp(name, "sync_flags")) {
        if (value < 0 || value > 1) return fail(FOA_E_INVALID, "sync_flags must be 1 (grouped tail / head sums) or 0 (direct sums)");
        if (!FOA_XCHECK && value == 0) return fail(FOA_E_INVALID, "sync_flags 0 is a cross-check kernel: it is only in libfun_ofdm_amd_xcheck.so (make xcheck)");
        rx->sync_flags_kind = (int)value;
        return FOA_OK;
    }
    return fail(FOA_E_INVALID, "unknown option '%s'", name);
}

void *foa_rx_stream(foa_rx *rx) { return rx ? (void *)rx->stream : nullptr; }

int foa_rx_sync(foa_rx *rx)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    return drain(rx);
}

int foa_rx_wait_age(foa_rx *rx, int age)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (age < 0 || age > 4) return fail(FOA_E_INVALID, "age must lie in 0 .. 4");
    WorkSet *w = rx->w;
    if (age == 1 && !(rx->pipeline && rx->viterbi_kind == 2)) w = rx->prev;      // calls in line: the same set again
    else for (int i = 0; i < age && w; i++) w = w->before;
    if (!w || !w->used || (age > 0 && w == rx->w)) return FOA_OK;       // no such call: nothing to wait for
    if (rx->pending.valid && rx->pending.w == w) { int rc = flush_pending(rx, nullptr); if (rc) return rc; }
    HIP_TRY(hipEventSynchronize(w->done));
    return FOA_OK;
}

int foa_rx_wait_previous(foa_rx *rx) { return foa_rx_wait_age(rx, 1); }

int foa_rx_decode_frames_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                             size_t n_frames, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames == 0) { rx->last_frames = 0; return FOA_OK; }
    if (!d_iq || !d_descs || !d_ends || !d_psdu || !d_results) return fail(FOA_E_INVALID, "NULL device pointer");
    if (n_frames > 0x7FFFFFF0u) return fail(FOA_E_INVALID, "too many frames");
    HIP_TRY(enter_device(rx->device));
    // Two work sets take turns when the finish runs on its own stream: this call's front end and forward pass may then
    // start while the previous call's chain-back is still reading the other set.
    const bool piped = rx->pipeline && rx->viterbi_kind == 2;
    if (!piped) { int rc0 = flush_pending(rx, nullptr); if (rc0) return rc0; }
    rx->prev = rx->w;
    if (piped) rx->w = &rx->sets[(int)((rx->w - rx->sets) + 1) % kSets];     // in use: front end k+1 | forward pass k | finish k-1
    if (rx->w->used) {                                                 // the call that last used this set is complete
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipEventSynchronize(rx->w->done));
        rx->ns_wait_set += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    int rc = workspace(rx, n_samples, n_frames);
    if (rc) return rc;
    // st: header, scan, data symbols; st_fwd: the forward pass.  Forward passes of consecutive calls take turns on two (or four)
    // streams: nothing orders one behind the other (each has its own work set), so the next one starts the moment its front end is
    // done, into the tail of the one before.
    // lanes (default): the call's front end runs on the stream of its own forward pass, behind the chain-back walk of the call two
    // back (queued there when the call before this one was made).  The loop that sets the step -- forward pass k, walk k, header,
    // scan and data symbols of call k+2, forward pass k+2 -- is then one in-order stream with no event packet in it; with the front
    // end on the third stream and the walk on the second, every hand-over between them cost 20-26 us, about 90 us per loop.
    const bool lanes = piped && rx->lanes;
    const int depth = !lanes ? 2 : rx->depth > 0 ? rx->depth : (n_frames < (size_t)kDeepBelow ? std::min(4, rx->max_depth) : 2);
    hipStream_t st_fwd = piped ? lane_stream(rx, (int)(rx->n_calls++ % (unsigned)depth)) : rx->stream;
    hipStream_t st = piped ? (lanes ? st_fwd : rx->stream3) : rx->stream;
    // Under one forward pass first the chain-back walk of the call before, then the front end of the call after: the two
    // heavy guests at once slow each other and the forward pass down more than they gain (measured: 1.50 against 1.43 ms per
    // step; letting only the light header and scan run alongside the walk is no better: 1.52).  The stitch/CRC kernel
    // behind the walk is light and latency-bound, so the front end does not wait for that one.
    const bool can_hold = piped && rx->prev->before && rx->prev->before->used && rx->prev->before->piped && rx->prev->before != rx->w;
    if (can_hold && !lanes && rx->fe_hold == 1) HIP_TRY(hipStreamWaitEvent(st, rx->prev->before->walk_done, 0));
    if (!piped && rx->prev->used && rx->prev != rx->w) HIP_TRY(hipStreamWaitEvent(st, rx->prev->done, 0));
    const int nf = (int)n_frames;
    const float2 *iq = (const float2 *)d_iq;
    double2 *eq_sig = rx->record_eq ? rx->w->eq_sig.p : nullptr, *eq_data = rx->record_eq ? rx->w->eq_data.p : nullptr;

    if (rx->in_wait) { HIP_TRY(hipStreamWaitEvent(st, rx->in_ready, 0)); rx->in_wait = false; }
    HIP_TRY(hipEventRecord(rx->w->ev[0], st));
    const bool exp_skip_fe = (FOA_EXP & 8) && rx->w->used;          // timing experiment only (tools/exp_chainback.sh): what is the front end on the loop worth?
    const bool exp_skip_hdr = (FOA_EXP & 16) && rx->w->used;        // timing experiment only: the bound on what a cheaper header kernel could gain (the work set keeps the last call's results)
    if (!exp_skip_fe && !exp_skip_hdr)
    hipLaunchKernelGGL(k_header, dim3(nf), dim3(64), 0, st, iq, d_descs, d_ends, (int64_t)n_samples, nf, rx->w->info.p, rx->w->hinv.p, eq_sig);
    HIP_TRY(hipEventRecord(rx->w->ev[1], st));
    // segments of this call: at most (total data steps)/S + one per frame; lanes beyond the real total idle
    const size_t max_segs = std::min(rx->w->seg2frame.n, rx->w->dec_cap / (size_t)rx->tb_segment + n_frames + 1);
    const int n_sb = (nf + kScanBlock - 1) / kScanBlock;
    int64_t *blk = rx->w->totals.p + 8;
    if (!exp_skip_fe) {
    hipLaunchKernelGGL(k_scan_sums, dim3(n_sb), dim3(kScanBlock), 0, st, rx->w->info.p, nf, rx->tb_segment, blk);
    if (n_sb <= 4096) hipLaunchKernelGGL(k_scan_blocks_w, dim3(1), dim3(64), 0, st, blk, n_sb, (int64_t)rx->w->sym_cap, rx->w->totals.p);
    else hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, blk, n_sb, (int64_t)rx->w->sym_cap, rx->w->totals.p);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_sb), dim3(kScanBlock), 0, st, rx->w->info.p, nf, (int64_t)rx->w->sym_cap,
                       (int64_t)rx->w->dec_cap, rx->tb_segment, (int64_t)rx->w->seg2frame.n, blk, rx->w->sym2frame.p, rx->w->seg2frame.p);
    }
    HIP_TRY(hipEventRecord(rx->w->ev[2], st));
    // upper bound on data symbols in n_samples samples; waves beyond the real total exit at once
    const size_t max_sym = rx->w->sym_cap;
    if (can_hold && !lanes && rx->fe_hold == 2) HIP_TRY(hipStreamWaitEvent(st, rx->prev->before->walk_done, 0));
    const int frontend = rx->frontend_kind >= 0 ? rx->frontend_kind : 2;
#if FOA_XCHECK
    if (frontend == 1) {
        hipLaunchKernelGGL(k_data_symbols_lps, dim3((unsigned)((max_sym + 63) / 64)), dim3(64), 0, st, iq, d_descs, rx->w->info.p, rx->w->sym2frame.p,
                           rx->w->totals.p, rx->w->hinv.p, rx->w->sp.p, eq_data);
    } else if (frontend == 0) {
        hipLaunchKernelGGL(k_data_symbols, dim3((unsigned)((max_sym + kSymWaves - 1) / kSymWaves)), dim3(64 * kSymWaves), 0, st, iq, d_descs,
                           rx->w->info.p, rx->w->sym2frame.p, rx->w->totals.p, rx->w->hinv.p, rx->w->sp.p, eq_data);
    } else
#endif
    if (!exp_skip_fe) {
        (void)frontend;
        hipLaunchKernelGGL(k_data_symbols_q4, dim3((unsigned)((max_sym + 16 * kQ4Waves - 1) / (16 * kQ4Waves))), dim3(64 * kQ4Waves), 0, st, iq,
                           d_descs, rx->w->info.p, rx->w->sym2frame.p, rx->w->totals.p, rx->w->hinv.p, rx->w->sp.p, eq_data);
    }
    HIP_TRY(hipEventRecord(rx->w->ev[3], st));
    if (piped) {
        // the previous call's chain-back + finish goes under this call's forward pass
        if ((rc = flush_pending(rx, rx->w->ev[3]))) return rc;
        if (!lanes) HIP_TRY(hipStreamWaitEvent(st_fwd, rx->w->ev[3], 0));
        HIP_TRY(hipEventRecord(rx->w->ev[7], st_fwd));          // start of the forward pass (this stream idles every other step: free)
        if ((rc = launch_forward(rx, st_fwd, rx->w, nf))) return rc;
        HIP_TRY(hipEventRecord(rx->w->ev[5], st_fwd));
        foa_rx::Pending &p = rx->pending;
        p.valid = true; p.w = rx->w; p.nf = nf; p.S = rx->tb_segment; p.L = rx->tb_overlap; p.max_segs = max_segs; p.slot_bytes = slot_bytes;
        p.psdu = d_psdu; p.results = d_results; p.job = rx->attach_job; p.lanes = lanes; p.lane = st_fwd;
    } else {
#if FOA_XCHECK
        if (rx->viterbi_kind == 0)
            hipLaunchKernelGGL(k_viterbi_v1, dim3(nf), dim3(64), 0, st, rx->w->info.p, nf, rx->w->sp.p, rx->w->dec.p, d_psdu, slot_bytes, d_results);
        else if (rx->viterbi_kind == 1)
            launch_viterbi_v2(st, rx->w->info.p, nf, rx->w->sp.p, rx->w->dec.p, rx->w->decoded.p, d_psdu, slot_bytes, d_results, rx->w->ev[5]);
        else
#endif
        {
            if ((rc = launch_forward(rx, st, rx->w, nf))) return rc;
            HIP_TRY(hipEventRecord(rx->w->ev[5], st));
            launch_finish3(st, st, rx->w->info.p, nf, rx->w->dec.p, rx->w->decoded.p, rx->w->seg2frame.p, rx->w->totals.p, rx->w->tb_state.p, max_segs, rx->tb_segment,
                           rx->tb_overlap, d_psdu, slot_bytes, d_results);
        }
        if (rx->viterbi_kind == 0) HIP_TRY(hipEventRecord(rx->w->ev[5], st));
        HIP_TRY(hipEventRecord(rx->w->ev[6], st));                     // (not separable from the forward pass on one stream)
        HIP_TRY(hipEventRecord(rx->w->ev[4], st));
        HIP_TRY(hipEventRecord(rx->w->done, st));
    }
    HIP_TRY(hipGetLastError());
    rx->w->used = true;
    rx->w->have_timing = true;
    rx->w->piped = piped;
    rx->w->before = piped ? rx->prev : nullptr;
    rx->last_frames = n_frames;
    return FOA_OK;
}

int foa_rx_decode_frames_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends,
                              size_t n_frames, uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames == 0) return FOA_OK;
    if (!iq || !descs || !ends || !psdu || !results) return fail(FOA_E_INVALID, "NULL pointer");
    HIP_TRY(enter_device(rx->device));
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_iq = 0, o_desc = o_iq + up(n_samples * 8), o_end = o_desc + up(n_frames * sizeof(foa_frame_desc)),
           o_psdu = o_end + up(n_frames * 8), o_res = o_psdu + up(n_frames * slot_bytes), total = o_res + up(n_frames * sizeof(foa_frame_result));
    int rc = rx->scratch.ensure(total);
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    hipStream_t st = side_stream(rx);
    HIP_TRY(hipMemcpyAsync(b + o_iq, iq, n_samples * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_desc, descs, n_frames * sizeof(foa_frame_desc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_end, ends, n_frames * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b + o_psdu, 0, n_frames * slot_bytes, st));
    if ((rc = inputs_queued(rx, st))) return rc;
    rc = foa_rx_decode_frames_dev(rx, (const float *)(b + o_iq), n_samples, (const foa_frame_desc *)(b + o_desc), (const int64_t *)(b + o_end),
                                  n_frames, b + o_psdu, slot_bytes, (foa_frame_result *)(b + o_res));
    if (rc) return rc;
    if ((rc = flush_pending(rx, nullptr))) return rc;                   // the finish runs on the second stream
    HIP_TRY(hipStreamWaitEvent(st, rx->w->done, 0));
    HIP_TRY(hipMemcpyAsync(psdu, b + o_psdu, n_frames * slot_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(results, b + o_res, n_frames * sizeof(foa_frame_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FOA_OK;
}

int foa_rx_submit_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                       size_t slot_bytes, uint64_t *ticket)
{
    if (!rx || !ticket) return fail(FOA_E_INVALID, "NULL argument");
    if (n_frames == 0 || !iq || !descs || !ends) return fail(FOA_E_INVALID, "empty call or NULL pointer");
    HIP_TRY(enter_device(rx->device));
    HostJob *job = nullptr;
    for (auto &j : rx->jobs) if (!j.busy) { job = &j; break; }
    if (!job) return fail(FOA_E_STATE, "%d calls are in flight: collect the oldest first", kMaxJobs);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_iq = 0, o_desc = o_iq + up(n_samples * 8), o_end = o_desc + up(n_frames * sizeof(foa_frame_desc)), o_psdu = o_end + up(n_frames * 8),
                 o_res = o_psdu + up(n_frames * slot_bytes), total = o_res + up(n_frames * sizeof(foa_frame_result));
    int rc = job->dev.ensure(total);
    if (rc) return rc;
    if (job->pin_cap < total) {
        if (job->pin) (void)hipHostFree(job->pin);
        job->pin = nullptr; job->pin_cap = 0;
        const size_t want = total + total / 2;
        HIP_TRY(hipHostMalloc((void **)&job->pin, want, hipHostMallocDefault));
        job->pin_cap = want;
    }
    if (!job->done) HIP_TRY(hipEventCreateWithFlags(&job->done, hipEventDisableTiming));
    // the caller's buffers are ours only until we return: mirror them, then everything else is asynchronous
    memcpy(job->pin + o_iq, iq, n_samples * 8);
    memcpy(job->pin + o_desc, descs, n_frames * sizeof(foa_frame_desc));
    memcpy(job->pin + o_end, ends, n_frames * 8);
    const bool piped = rx->pipeline && rx->viterbi_kind == 2;
    hipStream_t st = side_stream(rx);
    uint8_t *b = job->dev.p;
    HIP_TRY(hipMemcpyAsync(b, job->pin, o_psdu, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b + o_psdu, 0, n_frames * slot_bytes, st));
    if ((rc = inputs_queued(rx, st))) return rc;
    job->total = total; job->o_psdu = o_psdu; job->o_res = o_res; job->n_frames = n_frames; job->slot_bytes = slot_bytes; job->copy_queued = false;
    rx->attach_job = piped ? job : nullptr;
    rc = foa_rx_decode_frames_dev(rx, (const float *)(b + o_iq), n_samples, (const foa_frame_desc *)(b + o_desc), (const int64_t *)(b + o_end), n_frames,
                                  b + o_psdu, slot_bytes, (foa_frame_result *)(b + o_res));
    rx->attach_job = nullptr;
    if (rc) return rc;
    if (!piped) {
        HIP_TRY(hipMemcpyAsync(job->pin + o_psdu, b + o_psdu, total - o_psdu, hipMemcpyDeviceToHost, rx->stream));
        HIP_TRY(hipEventRecord(job->done, rx->stream));
        job->copy_queued = true;
    }
    job->busy = true;
    job->ticket = rx->next_ticket++;
    *ticket = job->ticket;
    return FOA_OK;
}

}  // extern "C"

// The job behind a ticket once its outputs are in its page-locked mirror: 1 = *out is complete (the caller reads job->pin and then
// clears job->busy), 0 = not yet (wait == false), < 0 = error.  Shared by foa_rx_collect and the stream engine.
static int job_ready(foa_rx *rx, uint64_t ticket, bool wait, HostJob **out)
{
    HostJob *job = nullptr;
    for (auto &j : rx->jobs) if (j.busy && j.ticket == ticket) { job = &j; break; }
    if (!job) return fail(FOA_E_INVALID, "unknown ticket");
    HIP_TRY(enter_device(rx->device));
    if (!job->copy_queued) {
        // its chain-back + finish is still the pending one: queue it (it would otherwise wait for the next call)
        if (!(rx->pending.valid && rx->pending.job == job)) return fail(FOA_E_STATE, "internal: job without a pending finish");
        int rc = flush_pending(rx, nullptr);
        if (rc) return rc;
    }
    if (wait) {
        HIP_TRY(hipEventSynchronize(job->done));
    } else {
        hipError_t e = hipEventQuery(job->done);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) return fail(FOA_E_HIP, "hipEventQuery: %s", hipGetErrorString(e));
    }
    *out = job;
    return 1;
}

extern "C" {

int foa_rx_collect(foa_rx *rx, uint64_t ticket, int wait, uint8_t *psdu, foa_frame_result *results)
{
    if (!rx || !psdu || !results) return fail(FOA_E_INVALID, "NULL argument");
    HostJob *job = nullptr;
    const int rc = job_ready(rx, ticket, wait != 0, &job);
    if (rc <= 0) return rc;
    memcpy(psdu, job->pin + job->o_psdu, job->n_frames * job->slot_bytes);
    memcpy(results, job->pin + job->o_res, job->n_frames * sizeof(foa_frame_result));
    job->busy = false;
    return 1;
}

static int kernel_ms_of(foa_rx *rx, WorkSet *w, float out_ms[6])
{
    if (!w || !w->have_timing) return fail(FOA_E_STATE, "no such decode call has been made on this handle");
    if (rx->pending.valid && rx->pending.w == w) { int rc = flush_pending(rx, nullptr); if (rc) return rc; }
    HIP_TRY(hipEventSynchronize(w->ev[4]));
    HIP_TRY(hipEventSynchronize(w->ev[5]));
    for (int i = 0; i < 3; i++) HIP_TRY(hipEventElapsedTime(&out_ms[i], w->ev[i], w->ev[i + 1]));
    // forward pass (or the fused v1 kernel).  Pipelined, it has its own start event: consecutive forward passes overlap by
    // design (two streams), so this is the launch's own duration, like a kernel trace reports it, not the step's share.
    if (w->piped) HIP_TRY(hipEventElapsedTime(&out_ms[3], w->ev[7], w->ev[5]));
    else HIP_TRY(hipEventElapsedTime(&out_ms[3], w->ev[3], w->ev[5]));
    // chain-back + descramble + CRC (0 for v1); on the pipelined path from where the walk is queued behind its forward pass
    if (w->piped) HIP_TRY(hipEventElapsedTime(&out_ms[4], w->ev[6], w->ev[4]));
    else HIP_TRY(hipEventElapsedTime(&out_ms[4], w->ev[5], w->ev[4]));
    HIP_TRY(hipEventElapsedTime(&out_ms[5], w->ev[0], w->ev[4]));      // whole call, first kernel to last (includes the deferral)
    return FOA_OK;
}

int foa_rx_last_kernel_ms(foa_rx *rx, float out_ms[6])
{
    if (!rx || !out_ms) return fail(FOA_E_INVALID, "NULL argument");
    return kernel_ms_of(rx, rx->w, out_ms);
}

int foa_rx_prev_kernel_ms(foa_rx *rx, float out_ms[6])
{
    if (!rx || !out_ms) return fail(FOA_E_INVALID, "NULL argument");
    return kernel_ms_of(rx, rx->prev, out_ms);
}

int foa_rx_kernel_ms_age(foa_rx *rx, int age, float out_ms[6])
{
    if (!rx || !out_ms) return fail(FOA_E_INVALID, "NULL argument");
    if (age < 0 || age > 4) return fail(FOA_E_INVALID, "age must lie in 0 .. 4");
    WorkSet *w = rx->w;
    for (int i = 0; i < age && w; i++) w = w->before;                // the pipelined calls link their work sets
    if (age > 0 && (!w || w == rx->w)) return fail(FOA_E_STATE, "no decode call of that age (calls must be pipelined)");
    return kernel_ms_of(rx, w, out_ms);
}

int foa_rx_probe_issue(foa_rx *rx, double out[6])
{
    if (!rx || !out) return fail(FOA_E_INVALID, "NULL argument");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, rx->device));
    const int n_simd = prop.multiProcessorCount * 4, W = 8, nw = n_simd * W, window_k = 600;      // ~0.26 ms per launch
    DevBuf<unsigned long long> buf;
    int rc = buf.ensure((size_t)3 * nw);
    if (rc) return rc;
    std::vector<unsigned long long> h((size_t)3 * nw);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; kind++) {
        float ms = 0;
        for (int it = 0; it < 3; it++) {                       // the first launches bring the clock to where a busy chip holds it
            HIP_TRY(hipEventRecord(e0, rx->stream));
            if (kind == 0) hipLaunchKernelGGL(k_probe_issue<0>, dim3(nw / 4), dim3(256), 0, rx->stream, buf.p, 7u, window_k);
            else hipLaunchKernelGGL(k_probe_issue<1>, dim3(nw / 4), dim3(256), 0, rx->stream, buf.p, 7u, window_k);
            HIP_TRY(hipEventRecord(e1, rx->stream));
            HIP_TRY(hipStreamSynchronize(rx->stream));
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        }
        HIP_TRY(hipMemcpy(h.data(), buf.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double ticks = 0, instr = 0;
        for (int i = 0; i < nw; i++) { ticks += (double)h[3 * i]; instr += (double)h[3 * i + 1] * 64.0; }
        const double window = ticks / nw;                      // shader clocks every wave was issuing for
        out[3 * kind + 0] = window / (instr / n_simd);         // SIMD clocks per wave64 instruction
        out[3 * kind + 1] = window / (ms * 1e6);               // GHz sustained during the launch
        out[3 * kind + 2] = instr / (ms * 1e-3);               // wave-instructions per second, whole chip
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    buf.release();
    return FOA_OK;
}

int foa_rx_get_taps(foa_rx *rx, size_t n_frames, double *hinv, double *eq, size_t eq_cap, uint64_t *eq_off, uint8_t *soft, size_t soft_cap,
                    uint64_t *soft_off)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames != rx->last_frames || n_frames == 0) return fail(FOA_E_STATE, "n_frames does not match the last decode call");
    if (eq && !rx->record_eq) return fail(FOA_E_STATE, "set option record_eq=1 before the decode call to get eq");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    std::vector<FrameInfo> info(n_frames);
    HIP_TRY(hipMemcpy(info.data(), rx->w->info.p, n_frames * sizeof(FrameInfo), hipMemcpyDeviceToHost));
    if (hinv) HIP_TRY(hipMemcpy(hinv, rx->w->hinv.p, n_frames * 64 * sizeof(double2), hipMemcpyDeviceToHost));
    DeviceTables tab;
    build_tables(&tab);
    size_t eo = 0, so = 0;
    for (size_t f = 0; f < n_frames; f++) {
        const FrameInfo &fi = info[f];
        if (eq_off) eq_off[f] = eo;
        if (soft_off) soft_off[f] = so;
        if (fi.rate < 0) continue;
        const int nsym = fi.nsym > 0 ? fi.nsym : 0;
        if (eq) {
            if (eo + (size_t)(1 + nsym) * 48 > eq_cap) return fail(FOA_E_INVALID, "eq_cap too small");
            HIP_TRY(hipMemcpy(eq + 2 * eo, rx->w->eq_sig.p + f * 48, 48 * sizeof(double2), hipMemcpyDeviceToHost));
            if (nsym) HIP_TRY(hipMemcpy(eq + 2 * (eo + 48), rx->w->eq_data.p + (size_t)fi.sym_off * 48, (size_t)nsym * 48 * sizeof(double2), hipMemcpyDeviceToHost));
        }
        eo += (size_t)(1 + nsym) * 48;
        const size_t sb = nsym ? (size_t)2 * fi.nsteps : 0;
        if (soft && sb) {
            if (so + sb > soft_cap) return fail(FOA_E_INVALID, "soft_cap too small");
            HIP_TRY(hipMemcpy(soft + so, rx->w->sp.p + fi.dec_off, sb, hipMemcpyDeviceToHost));
        }
        so += sb;
    }
    if (eq_off) eq_off[n_frames] = eo;
    if (soft_off) soft_off[n_frames] = so;
    return FOA_OK;
}

int foa_rx_get_decisions(foa_rx *rx, size_t frame, uint64_t *out, size_t cap, size_t *n_steps)
{
    if (!rx || !out || !n_steps) return fail(FOA_E_INVALID, "NULL argument");
    if (frame >= rx->last_frames) return fail(FOA_E_STATE, "frame index beyond the last decode call");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    FrameInfo fi;
    HIP_TRY(hipMemcpy(&fi, rx->w->info.p + frame, sizeof fi, hipMemcpyDeviceToHost));
    const size_t n = fi.nsym > 0 ? (size_t)fi.nsteps : 0;
    *n_steps = n;
    if (n > cap) return fail(FOA_E_INVALID, "cap too small (%zu steps)", n);
    if (n) HIP_TRY(hipMemcpy(out, rx->w->dec.p + fi.dec_off, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FOA_OK;
}

}  // extern "C"

// The launching half of foa_rx_sync_dev: every kernel of the pre-sync stage queued on the side stream, nothing waited for.  The counts
// stay on the device in rx->sy_n ([0] STS_END candidates, [3] alignments found); *ccap_out = the candidate capacity they are checked
// against.  (The stream engine queues this for batch k+1 while batch k decodes and reads the counts later.)
// origin: stream index of d_iq[0] (the stream engine's buffers start inside the stream; a one-shot call's starts it)
static int sync_dev_issue(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap, int32_t *ccap_out, int64_t origin = 0)
{
    const int64_t n = (int64_t)n_samples, n_words = (n + 31) / 32;
    const int n_blocks = (int)((n_words + kSyncBlockWords - 1) / kSyncBlockWords);
    const int32_t ccap = (int32_t)std::min<size_t>(n_samples / 64 + 64, 0x7FFFFFF0u);
    int rc;
    if ((rc = rx->sy_flags.ensure((size_t)n_words)) || (rc = rx->sy_cnt.ensure((size_t)std::max(n_blocks, (ccap + 255) / 256))) ||
        (rc = rx->sy_off.ensure((size_t)std::max(n_blocks, (ccap + 255) / 256))) ||
        (rc = rx->sy_x.ensure((size_t)ccap)) || (rc = rx->sy_cand.ensure((size_t)ccap)) || (rc = rx->sy_keep.ensure((size_t)ccap)) || (rc = rx->sy_n.ensure(8)))
        return rc;
    // With calls pipelined this stage runs on the third stream, under the forward pass of the decode call in flight (its own
    // scratch is touched by nothing else; the descriptors it writes are read by the header and data-symbol kernels of the next
    // decode call, which is made after this call has waited for its stream).
    hipStream_t st = side_stream(rx);
    // (with the front ends on the lanes nothing else orders this stage behind the decode call before it, whose header and data-symbol
    // kernels may still be reading the descriptor buffers a caller reuses from round to round)
    if (st != rx->stream && rx->lanes && rx->w->used && rx->w->piped) HIP_TRY(hipStreamWaitEvent(st, rx->w->ev[3], 0));
    const float2 *iq = (const float2 *)d_iq;
#if FOA_XCHECK
    if (rx->sync_flags_kind == 0)
        hipLaunchKernelGGL(k_sync_flags_direct, dim3((unsigned)((n + kFlagSamples - 1) / kFlagSamples)), dim3(256), 0, st, iq, n, rx->sy_flags.p);
    else
#endif
    hipLaunchKernelGGL(k_sync_flags, dim3((unsigned)((n + kFlagSamples - 1) / kFlagSamples)), dim3(64), 0, st, iq, n, rx->sy_flags.p, n_words);
    hipLaunchKernelGGL(k_sync_sts_end, dim3(n_blocks), dim3(kSyncBlockWords), 0, st, rx->sy_flags.p, n_words, 0, rx->sy_cnt.p, rx->sy_off.p, rx->sy_x.p, ccap);
    hipLaunchKernelGGL(k_sync_scan, dim3(1), dim3(1024), 0, st, rx->sy_cnt.p, n_blocks, rx->sy_off.p, rx->sy_n.p);
    hipLaunchKernelGGL(k_sync_sts_end, dim3(n_blocks), dim3(kSyncBlockWords), 0, st, rx->sy_flags.p, n_words, 1, rx->sy_cnt.p, rx->sy_off.p, rx->sy_x.p, ccap);
    // one wave per candidate; the count stays on the device: fixed grids stride over it (k_sync_finish reports overflow)
    const int lts_grid = (int)std::min<int64_t>(ccap, 16384);
    hipLaunchKernelGGL(k_sync_lts, dim3(lts_grid), dim3(64), 0, st, iq, n, rx->sy_x.p, rx->sy_n.p, ccap, rx->sy_cand.p, origin, rx->sync_call);
    const int kb = (ccap + 255) / 256;       // blocks of the keep / emit stage; their counts reuse the STS_END stage's count buffers
    hipLaunchKernelGGL(k_sync_keep, dim3(kb), dim3(256), 0, st, rx->sy_cand.p, rx->sy_n.p, ccap, rx->sy_keep.p, rx->sy_cnt.p);
    hipLaunchKernelGGL(k_sync_scan, dim3(1), dim3(1024), 0, st, rx->sy_cnt.p, kb, rx->sy_off.p, rx->sy_n.p + 3);
    hipLaunchKernelGGL(k_sync_emit, dim3(kb), dim3(256), 0, st, rx->sy_cand.p, rx->sy_keep.p, rx->sy_n.p, ccap, rx->sy_off.p, rx->sy_n.p + 3, n, d_descs, d_ends,
                       (int32_t)std::min<size_t>(cap, 0x7FFFFFF0u));
    *ccap_out = ccap;
    return FOA_OK;
}

extern "C" {

int foa_rx_sync_dev_begin(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap)
{
    if (!rx || !d_iq || !d_descs || !d_ends) return fail(FOA_E_INVALID, "NULL argument");
    if (rx->open_stream) return fail(FOA_E_STATE, "a stream engine owns this handle (and its pre-sync scratch): destroy the stream first");
    if (rx->sy_open) return fail(FOA_E_STATE, "a pre-sync is already in flight on this handle: foa_rx_sync_dev_end first");
    if (n_samples > 0x7FFFFFFFull * 16) return fail(FOA_E_INVALID, "stream too long for one call");
    HIP_TRY(enter_device(rx->device));
    if (!rx->sy_pin) HIP_TRY(hipHostMalloc((void **)&rx->sy_pin, 4 * sizeof(int32_t), hipHostMallocDefault));
    if (!rx->sy_done) HIP_TRY(hipEventCreateWithFlags(&rx->sy_done, hipEventDisableTiming));
    rx->sy_pin[0] = rx->sy_pin[1] = rx->sy_pin[2] = rx->sy_pin[3] = 0;
    rx->sy_cap = cap; rx->sy_ccap = 0;
    if (n_samples == 0 || cap == 0) { rx->sy_cap = 0; rx->sy_open = true; return FOA_OK; }          // (nothing queued; _end reports 0)
    { int rc = sync_dev_issue(rx, d_iq, n_samples, d_descs, d_ends, cap, &rx->sy_ccap, rx->sync_origin); if (rc) return rc; }
    hipStream_t st = side_stream(rx);
    HIP_TRY(hipMemcpyAsync(rx->sy_pin, rx->sy_n.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(rx->sy_done, st));
    HIP_TRY(hipGetLastError());
    rx->sy_open = true;              // only now: a pre-sync whose launch failed half-way is not "in flight" (its _end would report an empty batch as a success)
    return FOA_OK;
}

int foa_rx_sync_dev_end(foa_rx *rx, size_t *n_found)
{
    if (!rx || !n_found) return fail(FOA_E_INVALID, "NULL argument");
    *n_found = 0;
    if (!rx->sy_open) return fail(FOA_E_STATE, "no pre-sync in flight (foa_rx_sync_dev_begin first)");
    rx->sy_open = false;
    if (rx->sy_cap == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    HIP_TRY(hipEventSynchronize(rx->sy_done));
    if (rx->sy_pin[0] > rx->sy_ccap) return fail(FOA_E_NOMEM, "too many STS_END candidates (%d)", rx->sy_pin[0]);
    if ((size_t)rx->sy_pin[3] > rx->sy_cap) return fail(FOA_E_INVALID, "cap too small: %d alignments found", rx->sy_pin[3]);
    *n_found = (size_t)rx->sy_pin[3];
    return FOA_OK;
}

int foa_rx_sync_dev(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap, size_t *n_found)
{
    if (!n_found) return fail(FOA_E_INVALID, "NULL argument");
    *n_found = 0;
    int rc = foa_rx_sync_dev_begin(rx, d_iq, n_samples, d_descs, d_ends, cap);
    if (rc) return rc;
    return foa_rx_sync_dev_end(rx, n_found);
}

void foa_stream_destroy(struct foa_stream *s);
int foa_stream_push_f64_owned(struct foa_stream *s, const double *iq, size_t n_samples, void (*release)(void *), void *ctx);

}  // extern "C" (reopened below)

#include "stream_engine.h"
#include "shard_engine.h"

// ---- host-side pre-sync ---------------------------------------------------------------------------
struct foa_sync {
    foa::SyncHost impl;
    std::vector<foa_frame_desc> pending;
};

template <typename T>
static int sync_push(foa_sync *s, const T *iq, size_t n, foa_frame_desc *out, size_t cap, size_t *n_out)
{
    if (!s || (n && !iq) || (cap && !out) || !n_out) return fail(FOA_E_INVALID, "NULL argument");
    s->impl.push(iq, n, s->pending);
    size_t k = s->pending.size() < cap ? s->pending.size() : cap;
    if (k) memcpy(out, s->pending.data(), k * sizeof(foa_frame_desc));
    s->pending.erase(s->pending.begin(), s->pending.begin() + k);
    *n_out = k;
    return FOA_OK;
}

extern "C" {

int foa_sync_create(foa_sync **out)
{
    if (!out) return fail(FOA_E_INVALID, "out is NULL");
    *out = new foa_sync();
    return FOA_OK;
}
void foa_sync_destroy(foa_sync *s) { delete s; }
int foa_sync_push_f32(foa_sync *s, const float *iq, size_t n, foa_frame_desc *out, size_t cap, size_t *n_out) { return sync_push(s, iq, n, out, cap, n_out); }
int foa_sync_push_f64(foa_sync *s, const double *iq, size_t n, foa_frame_desc *out, size_t cap, size_t *n_out) { return sync_push(s, iq, n, out, cap, n_out); }
int64_t foa_sync_settled(const foa_sync *s) { return s ? s->impl.settled() : 0; }
int foa_sync_set_call(foa_sync *s, int64_t call)
{
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    if (call != 0 && call <= 160) return fail(FOA_E_INVALID, "call must be 0 (decide as one call over the whole stream) or > 160 (timing_sync.cpp:55)");
    s->impl.set_call(call);
    return FOA_OK;
}

// ---- stage-level entry points ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fft_vectors(double2 *__restrict__ v, int n_vec)
{
    __shared__ cpx lds_all[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n_vec) return;
    double2 x = v[(size_t)i * 64 + lane];
    cpx y = fft64_lane(cpx{ x.x, x.y }, lds_all[wave], lane);
    v[(size_t)i * 64 + lane_subcarrier(lane)] = make_double2(y.x, y.y);
}

int foa_fft_forward_f64(foa_rx *rx, double *vectors, size_t n_vec)
{
    if (!rx || !vectors) return fail(FOA_E_INVALID, "NULL argument");
    if (n_vec == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    size_t bytes = n_vec * 64 * sizeof(double2);
    int rc = rx->scratch.ensure(bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rx->scratch.p, vectors, bytes, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_fft_vectors, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, rx->stream, (double2 *)rx->scratch.p, (int)n_vec);
    HIP_TRY(hipMemcpyAsync(vectors, rx->scratch.p, bytes, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_conv_decode(foa_rx *rx, const uint8_t *symbols, uint8_t *data, int data_bits, size_t n_blocks)
{
    if (!rx || !symbols || !data) return fail(FOA_E_INVALID, "NULL argument");
    if (data_bits < 1 || data_bits > 8 * (kMaxDecodedBytes - 8)) return fail(FOA_E_INVALID, "data_bits out of range");
    if (n_blocks == 0) return FOA_OK;
    if (n_blocks > 0xFFFFu) return fail(FOA_E_INVALID, "at most 65535 blocks per call");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // this entry point reuses the work set of the batch calls
    const size_t nsteps = (size_t)data_bits + 6, sym_bytes = n_blocks * 2 * nsteps, nbytes = (size_t)((data_bits + 7) / 8), out_bytes = n_blocks * nbytes;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    int rc = rx->scratch.ensure(up(sym_bytes) + up(out_bytes));
    if (rc) return rc;
    uint8_t *d_sym = rx->scratch.p, *d_out = rx->scratch.p + up(sym_bytes);
    hipStream_t st = rx->stream;
    HIP_TRY(hipMemcpyAsync(d_sym, symbols, sym_bytes, hipMemcpyHostToDevice, st));
#if FOA_XCHECK
    if (rx->viterbi_kind == 0) {
        const size_t stride = (nsteps + 63) & ~(size_t)63;
        if ((rc = rx->w->dec.ensure(n_blocks * stride))) return rc;
        hipLaunchKernelGGL(k_conv_decode, dim3((unsigned)n_blocks), dim3(64), 0, st, d_sym, d_out, data_bits, (int)n_blocks, rx->w->dec.p, (int)stride);
    } else
#endif
    {
        // The kernels of the batch path (option "viterbi" = 1: k_viterbi_fwd2 + k_viterbi_finish2; 2: k_viterbi_fwd3 + k_tb_walk +
        // k_tb_finish), fed the way the front end feeds them: one frame record and one region of branch-metric words per block.
        // viterbi.cpp:209 drops an odd last step: its decision word stays zero (viterbi.cpp:193-194), so the chain-back reads
        // bit data_bits-1 as 0 and stays in state 0 -- the same as decoding one bit less and appending a zero.
        const int T = 2 * (int)(nsteps / 2), N = T - 6;
        HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, st));
        if (N > 0) {
            std::vector<FrameInfo> info(n_blocks);
            std::vector<int32_t> seg2frame;
            const int64_t words = dec_words(T);
            const int nseg = tb_segments(T, rx->tb_segment);
            for (size_t b = 0; b < n_blocks; b++) {
                FrameInfo &fi = info[b];
                fi.status = FOA_ST_CRC_FAIL; fi.rate = 0; fi.length = 0; fi.nsym = 1; fi.sym_off = 0; fi.nsteps = T; fi.soft_off = 0;
                fi.dec_off = (int64_t)b * words; fi.seg_off = (int32_t)(b * (size_t)nseg); fi.reserved_ = 0;
                seg2frame.insert(seg2frame.end(), (size_t)nseg, (int32_t)b);
            }
            const size_t total = n_blocks * (size_t)words + 64;
            if ((rc = rx->w->info.ensure(n_blocks + 1)) || (rc = rx->w->dec.ensure(total)) || (rc = rx->w->sp.ensure(total)) || (rc = rx->w->decoded.ensure(total)) ||
                (rc = rx->w->seg2frame.ensure(seg2frame.size() + 64)) || (rc = rx->w->tb_state.ensure(seg2frame.size() + 64)) || (rc = rx->w->totals.ensure(8)))
                return rc;
            HIP_TRY(hipMemcpyAsync(rx->w->info.p, info.data(), n_blocks * sizeof(FrameInfo), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(rx->w->seg2frame.p, seg2frame.data(), seg2frame.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            const int64_t n_segs = (int64_t)seg2frame.size();
            HIP_TRY(hipMemcpyAsync(rx->w->totals.p + 4, &n_segs, sizeof n_segs, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_conv_sp, dim3((unsigned)((T + 255) / 256), (unsigned)n_blocks), dim3(256), 0, st, d_sym, 2 * nsteps, T, rx->w->info.p, rx->w->sp.p);
#if FOA_XCHECK
            if (rx->viterbi_kind == 1)
                launch_viterbi_v2(st, rx->w->info.p, (int)n_blocks, rx->w->sp.p, rx->w->dec.p, rx->w->decoded.p, nullptr, 0, nullptr, nullptr);
            else
#endif
            {
                if ((rc = launch_forward(rx, st, rx->w, (int)n_blocks))) return rc;
                launch_finish3(st, st, rx->w->info.p, (int)n_blocks, rx->w->dec.p, rx->w->decoded.p, rx->w->seg2frame.p, rx->w->totals.p, rx->w->tb_state.p, seg2frame.size(),
                               rx->tb_segment, rx->tb_overlap, nullptr, 0, nullptr);
            }
            hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)((nbytes + 255) / 256), (unsigned)n_blocks), dim3(256), 0, st, rx->w->decoded.p, rx->w->info.p,
                               (N + 7) / 8, (int)nbytes, d_out);
            HIP_TRY(hipStreamSynchronize(st));        // the host vectors above are the copies' sources
            rx->last_frames = 0;                      // the workspace no longer describes a decode_frames call
        }
    }
    HIP_TRY(hipMemcpyAsync(data, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return FOA_OK;
}

int foa_channel_estimate_f64(foa_rx *rx, const double *lts_pairs, double *hinv, size_t n)
{
    if (!rx || !lts_pairs || !hinv) return fail(FOA_E_INVALID, "NULL argument");
    if (n == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    const size_t in_b = n * 128 * sizeof(double2), out_b = n * 64 * sizeof(double2);
    int rc = rx->scratch.ensure(in_b + out_b);
    if (rc) return rc;
    double2 *d_in = (double2 *)rx->scratch.p, *d_out = (double2 *)(rx->scratch.p + in_b);
    HIP_TRY(hipMemcpyAsync(d_in, lts_pairs, in_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_chanest, dim3((unsigned)n), dim3(64), 0, rx->stream, d_in, d_out, (int)n);
    HIP_TRY(hipMemcpyAsync(hinv, d_out, out_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_equalize_f64(foa_rx *rx, double *vectors, size_t n_vec, const double *hinv, size_t n_hinv, const int32_t *hinv_index)
{
    if (!rx || !vectors || !hinv || !hinv_index) return fail(FOA_E_INVALID, "NULL argument");
    if (n_vec == 0) return FOA_OK;
    for (size_t i = 0; i < n_vec; i++)
        if (hinv_index[i] < 0 || (size_t)hinv_index[i] >= n_hinv) return fail(FOA_E_INVALID, "hinv_index[%zu] out of range", i);
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t v_b = n_vec * 64 * sizeof(double2), h_b = n_hinv * 64 * sizeof(double2), i_b = n_vec * sizeof(int32_t);
    int rc = rx->scratch.ensure(up(v_b) + up(h_b) + up(i_b));
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    HIP_TRY(hipMemcpyAsync(b, vectors, v_b, hipMemcpyHostToDevice, rx->stream));
    HIP_TRY(hipMemcpyAsync(b + up(v_b), hinv, h_b, hipMemcpyHostToDevice, rx->stream));
    HIP_TRY(hipMemcpyAsync(b + up(v_b) + up(h_b), hinv_index, i_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_equalize, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, rx->stream, (double2 *)b, (int)n_vec,
                       (const double2 *)(b + up(v_b)), (const int32_t *)(b + up(v_b) + up(h_b)));
    HIP_TRY(hipMemcpyAsync(vectors, b, v_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_phase_track_f64(foa_rx *rx, const double *vectors, const int32_t *symbol_count, size_t n_vec, double *out48)
{
    if (!rx || !vectors || !symbol_count || !out48) return fail(FOA_E_INVALID, "NULL argument");
    if (n_vec == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t v_b = n_vec * 64 * sizeof(double2), c_b = n_vec * sizeof(int32_t), o_b = n_vec * 48 * sizeof(double2);
    int rc = rx->scratch.ensure(up(v_b) + up(c_b) + up(o_b));
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    HIP_TRY(hipMemcpyAsync(b, vectors, v_b, hipMemcpyHostToDevice, rx->stream));
    HIP_TRY(hipMemcpyAsync(b + up(v_b), symbol_count, c_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_phase, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, rx->stream, (const double2 *)b,
                       (const int32_t *)(b + up(v_b)), (int)n_vec, (double2 *)(b + up(v_b) + up(c_b)));
    HIP_TRY(hipMemcpyAsync(out48, b + up(v_b) + up(c_b), o_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_decode_header_f64(foa_rx *rx, const double *carriers48, size_t n, foa_frame_result *results)
{
    if (!rx || !carriers48 || !results) return fail(FOA_E_INVALID, "NULL argument");
    if (n == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t c_b = n * 48 * sizeof(double2), r_b = n * sizeof(foa_frame_result);
    int rc = rx->scratch.ensure(up(c_b) + up(r_b));
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    HIP_TRY(hipMemcpyAsync(b, carriers48, c_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_header, dim3((unsigned)n), dim3(64), 0, rx->stream, (const double2 *)b, (int)n, (foa_frame_result *)(b + up(c_b)));
    HIP_TRY(hipMemcpyAsync(results, b + up(c_b), r_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_decode_data_f64(foa_rx *rx, const double *carriers, const uint64_t *carrier_off, size_t n_frames, foa_frame_result *results,
                        uint8_t *psdu, size_t slot_bytes)
{
    if (!rx || !carriers || !carrier_off || !results || !psdu) return fail(FOA_E_INVALID, "NULL argument");
    if (n_frames == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // this entry point reuses the work set of the batch calls
    DeviceTables tab;
    build_tables(&tab);
    // frame records and offsets on the host (what k_header + the k_scan_* kernels produce in the fused path)
    std::vector<FrameInfo> info(n_frames);
    std::vector<int32_t> sym2frame, seg2frame;
    std::vector<int64_t> coff(n_frames + 1);
    int64_t dec_off = 0;
    for (size_t f = 0; f < n_frames; f++) {
        const int rate = results[f].rate, len = results[f].length;
        if (rate < 0 || rate >= kNumRates || len < 0 || len > 4095) return fail(FOA_E_INVALID, "frame %zu: bad rate/length", f);
        const int dbps = tab.rates[rate].dbps, nsym = (16 + 8 * (len + 4) + 6 + dbps - 1) / dbps;
        if (carrier_off[f + 1] - carrier_off[f] != (uint64_t)nsym * 48) return fail(FOA_E_INVALID, "frame %zu: needs %d carriers", f, nsym * 48);
        FrameInfo &fi = info[f];
        fi.status = FOA_ST_CRC_FAIL; fi.rate = rate; fi.length = len; fi.nsym = nsym; fi.sym_off = (int32_t)sym2frame.size();
        fi.nsteps = nsym * dbps; fi.soft_off = 2 * dec_off; fi.dec_off = dec_off;
        fi.seg_off = (int32_t)seg2frame.size(); fi.reserved_ = 0;
        seg2frame.insert(seg2frame.end(), (size_t)tb_segments(fi.nsteps, rx->tb_segment), (int32_t)f);
        dec_off += dec_words(fi.nsteps);
        coff[f] = (int64_t)carrier_off[f];
        sym2frame.insert(sym2frame.end(), (size_t)nsym, (int32_t)f);
        results[f].num_symbols = nsym;
    }
    coff[n_frames] = (int64_t)carrier_off[n_frames];
    const size_t n_sym = sym2frame.size(), n_car = (size_t)carrier_off[n_frames];
    int rc;
    if ((rc = rx->w->info.ensure(n_frames + 1)) || (rc = rx->w->sym2frame.ensure(n_sym + 1)) ||
        (rc = rx->w->dec.ensure((size_t)dec_off + 64)) || (rc = rx->w->sp.ensure((size_t)dec_off + 64)) || (rc = rx->w->decoded.ensure((size_t)dec_off + 64)) ||
        (rc = rx->w->seg2frame.ensure(seg2frame.size() + 64)) || (rc = rx->w->tb_state.ensure(seg2frame.size() + 64)) || (rc = rx->w->totals.ensure(8)))
        return rc;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t c_b = n_car * sizeof(double2), o_b = (n_frames + 1) * sizeof(int64_t), p_b = n_frames * slot_bytes, r_b = n_frames * sizeof(foa_frame_result);
    if ((rc = rx->scratch.ensure(up(c_b) + up(o_b) + up(p_b) + up(r_b)))) return rc;
    uint8_t *b = rx->scratch.p;
    uint8_t *d_psdu = b + up(c_b) + up(o_b);
    foa_frame_result *d_res = (foa_frame_result *)(d_psdu + up(p_b));
    hipStream_t st = rx->stream;
    HIP_TRY(hipMemcpyAsync(b, carriers, c_b, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + up(c_b), coff.data(), o_b, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rx->w->info.p, info.data(), n_frames * sizeof(FrameInfo), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rx->w->sym2frame.p, sym2frame.data(), n_sym * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const int64_t n_segs = (int64_t)seg2frame.size();
    HIP_TRY(hipMemcpyAsync(rx->w->seg2frame.p, seg2frame.data(), seg2frame.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rx->w->totals.p + 4, &n_segs, sizeof n_segs, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_psdu, 0, p_b, st));
    hipLaunchKernelGGL(k_stage_demap, dim3((unsigned)((n_sym + kSymWaves - 1) / kSymWaves)), dim3(64 * kSymWaves), 0, st, (const double2 *)b,
                       (const int64_t *)(b + up(c_b)), rx->w->info.p, rx->w->sym2frame.p, (int)n_sym, rx->w->sp.p);
#if FOA_XCHECK
    if (rx->viterbi_kind == 0)
        hipLaunchKernelGGL(k_viterbi_v1, dim3((unsigned)n_frames), dim3(64), 0, st, rx->w->info.p, (int)n_frames, rx->w->sp.p, rx->w->dec.p, d_psdu, slot_bytes, d_res);
    else if (rx->viterbi_kind == 1)
        launch_viterbi_v2(st, rx->w->info.p, (int)n_frames, rx->w->sp.p, rx->w->dec.p, rx->w->decoded.p, d_psdu, slot_bytes, d_res, nullptr);
    else
#endif
    {
        int rcf = launch_forward(rx, st, rx->w, (int)n_frames);
        if (rcf) return rcf;
        launch_finish3(st, st, rx->w->info.p, (int)n_frames, rx->w->dec.p, rx->w->decoded.p, rx->w->seg2frame.p, rx->w->totals.p, rx->w->tb_state.p, seg2frame.size(),
                       rx->tb_segment, rx->tb_overlap, d_psdu, slot_bytes, d_res);
    }
    HIP_TRY(hipMemcpyAsync(psdu, d_psdu, p_b, hipMemcpyDeviceToHost, st));
    std::vector<foa_frame_result> out(n_frames);
    HIP_TRY(hipMemcpyAsync(out.data(), d_res, r_b, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (size_t f = 0; f < n_frames; f++) results[f].status = out[f].status;
    rx->last_frames = 0;      // the workspace no longer describes a decode_frames call
    return FOA_OK;
}

int foa_tx_build_frames_dev(foa_rx *rx, const uint8_t *d_payloads, size_t payload_pitch, int length, int rate, size_t n_frames,
                            double *d_frames, size_t *frame_samples)
{
    if (!rx || !frame_samples) return fail(FOA_E_INVALID, "NULL argument");
    if (rate < 0 || rate >= kNumRates || length < 0 || length > 4095) return fail(FOA_E_INVALID, "bad rate/length");
    DeviceTables tab;
    build_tables(&tab);
    const int dbps = tab.rates[rate].dbps, nsym = (16 + 8 * (length + 4) + 6 + dbps - 1) / dbps, nbytes = nsym * dbps / 8;
    *frame_samples = 320 + (size_t)80 * (nsym + 1);
    if (n_frames == 0) return FOA_OK;
    if ((!d_payloads && length > 0) || !d_frames) return fail(FOA_E_INVALID, "NULL device pointer");
    if (payload_pitch < (size_t)length) return fail(FOA_E_INVALID, "payload_pitch smaller than length");
    if (n_frames > 0x7FFFFFF0u / (size_t)(nsym + 1)) return fail(FOA_E_INVALID, "too many frames for one call");
    HIP_TRY(enter_device(rx->device));
    const size_t stride = ((size_t)nbytes + 1 + 15) & ~(size_t)15;
    int rc = rx->scratch.ensure(n_frames * stride);
    if (rc) return rc;
    hipStream_t st = rx->stream;
    const int nf = (int)n_frames;
    hipLaunchKernelGGL(k_tx_prepare, dim3((nf + 63) / 64), dim3(64), 0, st, d_payloads, payload_pitch, length, nf, nbytes, rx->scratch.p, stride);
    const int64_t threads = (int64_t)nf * (nsym + 1);
    hipLaunchKernelGGL(k_tx_symbols, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, rx->scratch.p, stride, length, rate, nsym, nf,
                       (double2 *)d_frames, *frame_samples);
    HIP_TRY(hipGetLastError());
    return FOA_OK;
}

int foa_tx_channel_dev(foa_rx *rx, const double *d_frames, size_t n_frames, size_t frame_samples, size_t pitch, size_t lead, double snr_db,
                       double cfo_hz, uint64_t seed, float *d_iq)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames == 0) return FOA_OK;
    if (!d_frames || !d_iq) return fail(FOA_E_INVALID, "NULL device pointer");
    if (lead + frame_samples > pitch) return fail(FOA_E_INVALID, "lead + frame_samples exceeds the pitch");
    HIP_TRY(enter_device(rx->device));
    // sigma^2 per real component = P_ref / (2 10^(SNR/10)), P_ref = 0.0124
    const double sigma = std::sqrt(0.0124 / (2.0 * std::pow(10.0, snr_db / 10.0)));
    const int64_t total = (int64_t)n_frames * (int64_t)pitch;
    hipLaunchKernelGGL(k_tx_channel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, rx->stream, (const double2 *)d_frames, (int64_t)n_frames,
                       (int64_t)frame_samples, (int64_t)pitch, (int64_t)lead, sigma, cfo_hz, seed, (float2 *)d_iq);
    HIP_TRY(hipGetLastError());
    return FOA_OK;
}

}  // extern "C"
