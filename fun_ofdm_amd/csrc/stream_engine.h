// stream_engine.h -- receiver_chain::process_samples() with EVERYTHING on the device (SURVEY 8f #1 + #3):
// foa_stream_* of include/fun_ofdm_amd.h.  Host code only (rx_stream.hip); it drives the handle's streams, work sets and job slots.
//
// The reference runs frame_detector -> timing_sync -> fft_symbols -> ... -> frame_decoder inside every
// process_samples() call (src/receiver_chain.cpp:106-126) on 4096-sample chunks, carrying 16 + 160 samples and the frame in
// progress from call to call.  A GPU wants millions of samples per launch, so the engine cuts the stream into BATCHES of
// `batch_samples` and lets consecutive batches OVERLAP instead of carrying state:
//
//   stream      ....|<------------- batch k-1 ------------->|<-------------- batch k --------------->|....
//   device buf k              |<-- carry C -->|<------------- batch k (H2D) -------------------------->|
//                             ^ start_k = pos_k - C                      decisions up to pos_k + B - 192 ^ pos_k + B
//
//   * device buffer k = the last C samples before the batch (a device-to-device copy out of buffer k-1) + the batch;
//   * the pre-sync kernels run over the whole buffer.  Of the alignments they find, batch k DECIDES, in stream order from the first one
//     not decided yet, those that can be decided with the samples there are: a look-ahead (LTS + SIGNAL of the candidates, k_header_range,
//     then k_stream_resolve) finds the first alignment whose frame would run into the end of the buffer -- FOA_ST_TRUNCATED -- and it and
//     everything behind it wait for batch k+1.  A frame is therefore handed out by the first batch that holds its last sample (the
//     reference: five 4096-sample calls after it, receiver_chain.cpp:118-125), whatever the longest frame the format allows;
//   * tags within kStreamSettle samples of the buffer's end are not final (timing_sync looks 160 samples ahead): the batch's decisions
//     are made as if the stream ended there, and alignments whose STS_END lies beyond are left to the next batch;
//   * C = L + 2048 with L = the longest frame (+ look-ahead), so the first undecided alignment and everything within 2048 samples before
//     it are inside buffer k+1 as well -- the detector's 16-sample windows, the plateau that ends in an STS_END, the earlier hit that can
//     overwrite a tag (timing_sync.cpp:105-106) all see the same samples as they would in one pass over the whole stream;
//   * the state that crosses batches stays on the device (StreamState): where the first undecided alignment's STS_END lies, and the phasor
//     timing_sync left in force before it (m_phase_acc, timing_sync.cpp:113-125), patched into the batch's first descriptor.
// Batches are queued through the same asynchronous job slots as foa_rx_submit_host (pinned mirrors, D2H behind the
// finish kernel), so H2D of batch k+1, the kernels of batch k and the D2H of batch k-1 overlap, and results come out in
// stream order.  The double -> float narrowing of process_samples' complex<double> input is the only per-sample work the
// host does.  Threads (stream_core.h): the caller only assigns samples their place in page-locked staging memory; helper
// threads narrow them (a caller that hands its buffer over, as process_samples' by-value vector allows, does not even wait
// for that); one submitter thread makes every GPU call while the stream is open.
#pragma once

#include <algorithm>
#include <atomic>
#include <deque>
#include <mutex>

#include "rx_handle.h"
#include "stream_core.h"

using namespace foa;

namespace foa {

constexpr int64_t kStreamLongest = 110592;          // >= 320 + 80 * 1369 (4095 bytes at 6 Mbps) + 32 + timing_sync's 160-sample look-ahead
constexpr int64_t kStreamCarry = kStreamLongest + 2048;
constexpr int kStreamBufs = 12;                     // at most this many device sample buffers / pinned staging buffers in rotation (= StreamCore::kSlots); a stream uses n_bufs of them

}  // namespace foa

// Queue the decode of m alignments of a batch buffer through a job slot of the asynchronous host entry (page-locked mirror, D2H behind the
// finish kernel); *ticket identifies the job for stream_collect_job.  Shared by the single-device engine below and the multi-device one
// (shard_engine.h).  t_prep (may be null): ns spent finding and sizing the job slot.
// n_ctx: alignments behind the m decoded ones that serve them as context (foa_rx_decode_frames_ctx_dev): the rest of the buffer's.
// descs / ends: ALL tags of the buffer; the first n_lead were decided by earlier batches (foa_rx_decode_frames_lead_ctx_dev: looked at only for where they sit).
static int stream_decode_batch(foa_rx *rx, const float *d, size_t n_buf, const foa_frame_desc *descs, const int64_t *ends, size_t n_lead, size_t m, size_t n_ctx, size_t slot_bytes,
                               uint64_t *ticket, int64_t *t_prep)
{
    const auto t0 = std::chrono::steady_clock::now();
    const bool piped = rx->pipeline;
    HostJob *job = nullptr;
    for (auto &j : rx->jobs) if (!j.busy) { job = &j; break; }
    if (!job) return fail(FOA_E_STATE, "internal: no free job slot");
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_res = up(m * slot_bytes), total = o_res + up(m * sizeof(foa_frame_result));
    // (twice what this batch needs: the number of frames differs a little from batch to batch, and growing a buffer means a
    // hipFree, which waits for the whole device -- with exact sizes that was 0.4-0.9 ms of the submitter's time per batch)
    const size_t roomy = (2 * total + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    int rc;
    if (job->dev.n < total && (rc = job->dev.ensure(roomy))) return rc;
    if (job->pin_cap < total) {
        if (job->pin) (void)hipHostFree(job->pin);
        job->pin = nullptr; job->pin_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&job->pin, roomy, hipHostMallocDefault));
        job->pin_cap = roomy;
    }
    if (!job->done) HIP_TRY(hipEventCreateWithFlags(&job->done, hipEventDisableTiming));
    // (the slots are not cleared: collect reads the payload of a frame only where the result says it passed)
    job->total = total; job->o_psdu = 0; job->o_res = o_res; job->n_frames = m; job->slot_bytes = slot_bytes; job->copy_queued = false;
    rx->attach_job = piped ? job : nullptr;
    if (t_prep) *t_prep += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    rc = foa_rx_decode_frames_lead_ctx_dev(rx, d, n_buf, descs, ends, n_lead, m, n_ctx, job->dev.p, slot_bytes, (foa_frame_result *)(job->dev.p + o_res));
    rx->attach_job = nullptr;
    if (rc) return rc;
    if (!piped) {
        HIP_TRY(hipMemcpyAsync(job->pin, job->dev.p, total, hipMemcpyDeviceToHost, rx->stream));
        HIP_TRY(hipEventRecord(job->done, rx->stream));
        job->copy_queued = true;
    }
    job->busy = true;
    job->ticket = rx->next_ticket++;
    *ticket = job->ticket;
    return FOA_OK;
}

// The job behind `ticket`: 1 = complete, its CRC-passing payloads appended to *out in frame order and the slot released; 0 = not yet
// (only with wait false); < 0 error.  by_status[5]: alignments per FOA_ST_* of the batch.
static int stream_collect_job(foa_rx *rx, uint64_t ticket, size_t n_frames, bool wait, foa::StreamReady *out, uint64_t by_status[5])
{
    HostJob *job = nullptr;
    const int rc = job_ready(rx, ticket, wait, &job);
    if (rc <= 0) return rc;
    // straight out of the job's page-locked mirror: only the payload bytes of the frames that passed move again
    const foa_frame_result *res = (const foa_frame_result *)(job->pin + job->o_res);
    const uint8_t *ps = job->pin + job->o_psdu;
    size_t bytes = 0, n_ok = 0;
    for (size_t i = 0; i < n_frames; i++) {
        const int st = res[i].status;
        if (st >= 0 && st < 5) by_status[st]++;
        else if (st == FOA_ST_SUPERSEDED) by_status[FOA_ST_TRUNCATED]++;          // (counted together: foa_stream_stats)
        if (st == FOA_ST_OK) { bytes += (size_t)res[i].length; n_ok++; }
    }
    const size_t at = out->bytes.size(), at_len = out->len.size();
    out->bytes.resize(at + bytes);
    out->len.resize(at_len + n_ok);
    uint8_t *dst = out->bytes.data() + at;
    uint32_t *dl = out->len.data() + at_len;
    for (size_t i = 0; i < n_frames; i++) {
        const foa_frame_result &r = res[i];
        if (r.status != FOA_ST_OK) continue;
        *dl++ = (uint32_t)r.length;
        memcpy(dst, ps + i * job->slot_bytes, (size_t)r.length);
        dst += r.length;
    }
    job->busy = false;
    return 1;
}

// The GPU side of one stream: everything here except staging() runs on the core's submitter thread.
struct StreamGpu {
    foa_rx *rx = nullptr;
    int64_t B = 0;                                   // batch_samples
    int n_bufs = 6;                                  // buffers in rotation (foa_stream_create)
    bool copies_in_line = false;                     // small batches: upload and carry copy on the pre-sync's own stream (stage_impl)
    bool fill_by_kernel = false;                     // ... and done by one kernel that reads the page-locked staging memory itself
    int64_t L = foa::kStreamLongest, C = foa::kStreamCarry;      // longest frame the stream may hold (+ look-ahead) and the carry it implies (option "stream_longest")
    size_t slot_bytes = 4096;
    float *pin[foa::kStreamBufs] = {};               // page-locked staging, B float2 each
    DevBuf<float> dev[foa::kStreamBufs];             // (C + B) float2 each
    hipEvent_t in_done[foa::kStreamBufs] = {};       // H2D + carry copy of the buffer's batch are through
    hipStream_t st_in = nullptr;                     // carry copies and H2D
    DevBuf<uint8_t> d_desc[foa::kStreamBufs];
    DevBuf<int64_t> d_ends[foa::kStreamBufs];
    size_t desc_cap = 0;
    int64_t submitted_samples = 0;                   // samples in the batches submitted so far
    int64_t n_batches = 0;
    int64_t n_staged = 0;                            // batches whose upload has been queued
    int64_t staged_samples = 0;                      // samples in the batches staged so far
    DevBuf<StreamState> state;                       // first undecided alignment + phasor in force before it, kept on the device from batch to batch
    DevBuf<FrameInfo> la_info;                       // the look-ahead's own alignment records and channel estimates (one batch at a time: the side stream is in order)
    DevBuf<double2> la_hinv;
    DevBuf<int32_t> la_range;                        // per buffer: the candidates' index range
    DevBuf<int32_t> sel_dev;                         // per buffer, 8 ints: the look-ahead's verdict, copied out behind it
    int32_t *sel = nullptr;                          // page-locked: per buffer { STS_END candidates, alignments found, first of the batch, how many it decides, context behind them }
    int64_t n_eff[foa::kStreamBufs] = {};            // per buffer: samples up to which its tags are final (the "end of the stream" for its decisions)
    hipEvent_t sel_done[foa::kStreamBufs] = {};      // pre-sync + selection of the buffer's batch are through and `sel` is written
    int32_t ccap[foa::kStreamBufs] = {};
    struct InFlight { uint64_t handle, ticket; size_t n_frames; };
    std::deque<InFlight> flight;
    uint64_t next_handle = 1;
    std::atomic<uint64_t> status_count[5], alignments;
    std::mutex err_m;
    std::string err_text;                            // text of the first error raised on the submitter thread

    // where the submitter thread's time goes (ns; printed by foa_stream_destroy when FOA_STREAM_STATS is set)
    int64_t t_sync = 0, t_desc = 0, t_decode = 0, t_collect = 0, t_collect_wait = 0, t_prep = 0;
    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    StreamGpu() { for (auto &c : status_count) c.store(0); alignments.store(0); }
    float *staging(int slot) { return pin[slot]; }

    int keep_error(int rc)
    {
        std::lock_guard<std::mutex> lk(err_m);
        if (err_text.empty()) err_text = last_error_text();
        return rc;
    }

    // queue the upload of the next batch: n_new samples wait in staging slot k.  Device buffer k = the last C samples before the
    // batch (out of buffer k-1, whose own upload is ahead of this copy on the same stream) + the batch.  Never blocks.
    int stage(int k, int64_t n_new, bool final)
    {
        const int rc = stage_impl(k, n_new, final);
        return rc ? keep_error(rc) : 0;
    }
    int stage_impl(int k, int64_t n_new, bool final)
    {
        const int kp = (k + n_bufs - 1) % n_bufs;
        HIP_TRY(enter_device(rx->device));
        float *d = dev[k].p;
        // ... and right behind it, on the side stream, the pre-sync over the whole buffer and the selection of this batch's alignments:
        // by the time the batch is submitted the host has nothing to wait for but four integers
        hipStream_t st = side_stream(rx);
        // (the copies of a SMALL batch go to the side stream itself: a stream of their own lets a large upload run under the batch before's
        // pre-sync, but the hop from one stream to the other is ~25 us -- more than a 32 KB upload and its carry copy take)
        hipStream_t cs = copies_in_line ? st : st_in;
        if (copies_in_line && fill_by_kernel) {
            // (... and by ONE kernel that reads the staging memory itself: no DMA engine in the batch's way, sync_kernels.h)
            launch_stream_fill(cs, d, n_staged == 0 ? nullptr : dev[kp].p + 2 * B, C, pin[k], n_new);
        } else {
        if (n_staged == 0) HIP_TRY(hipMemsetAsync(d, 0, (size_t)C * 8, cs));                        // silence before the stream
        else HIP_TRY(hipMemcpyAsync(d, dev[kp].p + 2 * B, (size_t)C * 8, hipMemcpyDeviceToDevice, cs));      // (every batch but the last is full)
        if (n_new) HIP_TRY(hipMemcpyAsync(d + 2 * C, pin[k], (size_t)n_new * 8, hipMemcpyHostToDevice, cs));
        }
        HIP_TRY(hipEventRecord(in_done[k], cs));
        if (cs != st) HIP_TRY(hipStreamWaitEvent(st, in_done[k], 0));
        const int64_t n_buf = C + n_new, pushed = staged_samples + n_new;
        const int64_t start = pushed - n_new - C;                    // stream index of the buffer's first sample
        n_eff[k] = final ? n_buf : n_buf - foa::kStreamSettle;       // tags are final up to here: the batch decides as if the stream ended there
        const int64_t hz_abs = start + n_eff[k];
        int rc = sync_dev_issue(rx, d, (size_t)n_buf, (foa_frame_desc *)d_desc[k].p, d_ends[k].p, desc_cap, &ccap[k], start);
        if (rc) return rc;
        launch_stream_range(st, (foa_frame_desc *)d_desc[k].p, rx->sy_n.p, (int32_t)desc_cap, start, hz_abs, state.p, la_range.p + 2 * k);
        launch_stream_resolve(st, d, n_eff[k], (const foa_frame_desc *)d_desc[k].p, d_ends[k].p, la_range.p + 2 * k, rx->sy_n.p, start, hz_abs, final, state.p,
                              la_info.p, la_hinv.p, fill_by_kernel ? sel + 8 * k : sel_dev.p + 8 * k, (unsigned)desc_cap);
        // (a small batch's look-ahead writes its five integers into the page-locked array itself: one copy launch less in front of the host's look at them)
        if (!fill_by_kernel) HIP_TRY(hipMemcpyAsync(sel + 8 * k, sel_dev.p + 8 * k, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(sel_done[k], st));
        HIP_TRY(hipGetLastError());
        staged_samples = pushed;
        n_staged++;
        return FOA_OK;
    }
    // (the core asks this before it submits: the batch's samples are on the device AND its alignments are known)
    bool uploaded(int k) { return hipEventQuery(sel_done[k]) != hipErrorNotReady; }

    // queue the kernels of batch `n_batches`, whose samples stage() put into device buffer k
    int submit(int slot, int64_t n_new, bool final, uint64_t *handle)
    {
        const int rc = submit_impl(slot, n_new, final, handle);
        return rc ? keep_error(rc) : 0;
    }
    int submit_impl(int k, int64_t n_new, bool final, uint64_t *handle)
    {
        const int64_t pushed = submitted_samples + n_new;
        (void)final;
        HIP_TRY(enter_device(rx->device));
        float *d = dev[k].p;
        int64_t t0 = now_ns();
        HIP_TRY(hipEventSynchronize(sel_done[k]));                   // (through already when the core asked uploaded(); the staging slot is free again)
        t_sync += now_ns() - t0;
        const int32_t *q = sel + 8 * k;
        if (q[0] > ccap[k]) return fail(FOA_E_NOMEM, "too many STS_END candidates (%d)", q[0]);
        if ((size_t)q[1] > desc_cap) return fail(FOA_E_INVALID, "internal: %d alignments in one batch buffer", q[1]);
        const size_t i0 = (size_t)q[2], m = (size_t)q[3], n_ctx = (size_t)q[4];
        int rc;
        t0 = now_ns();
        InFlight fl;
        fl.handle = next_handle++; fl.n_frames = m; fl.ticket = 0;
        if (m) {
            // (the candidates behind the batch's own go along as context: a frame cut short by a later LTS1 may fill on with their vectors,
            // fft_symbols.cpp:41-50 / frame_decoder.cpp:52-88 -- the look-ahead decided with them, and they are decided by the next batch)
            rc = stream_decode_batch(rx, d, (size_t)n_eff[k], (const foa_frame_desc *)d_desc[k].p, d_ends[k].p, i0, m, n_ctx, slot_bytes, &fl.ticket, &t_prep);
            if (rc) return rc;
            alignments.fetch_add(m);
        }
        t_decode += now_ns() - t0;
        flight.push_back(fl);
        *handle = fl.handle;
        submitted_samples = pushed;
        n_batches++;
        return FOA_OK;
    }

    // the batch behind `handle` (always the oldest in flight): 1 = complete, payloads unpacked; 0 = not yet
    int collect(uint64_t handle, bool wait, foa::StreamReady *out)
    {
        if (flight.empty() || flight.front().handle != handle) return keep_error(fail(FOA_E_STATE, "internal: batches collected out of order"));
        const InFlight f = flight.front();
        if (f.n_frames) {
            (void)hipSetDevice(rx->device);
            const int64_t t0 = now_ns();
            uint64_t by_status[5] = { 0, 0, 0, 0, 0 };
            const int rc = stream_collect_job(rx, f.ticket, f.n_frames, wait, out, by_status);
            if (rc < 0) { flight.pop_front(); return keep_error(rc); }
            if (rc == 0) { t_collect_wait += now_ns() - t0; return 0; }
            for (int k = 0; k < 5; k++) if (by_status[k]) status_count[k].fetch_add(by_status[k], std::memory_order_relaxed);
            t_collect += now_ns() - t0;
        }
        flight.pop_front();
        return 1;
    }
};

static_assert(foa::kStreamBufs == foa::StreamCore<StreamGpu>::kSlots, "one device buffer and one staging buffer per slot of the core");

struct foa_stream {
    StreamGpu gpu;
    foa::StreamCore<StreamGpu> *core = nullptr;
    foa::StreamReady ready;                          // the batch foa_stream_ready reported and foa_stream_take has not yet taken
    bool have_ready = false;
};

namespace {

// an error raised on the submitter thread, reported on the caller's
int stream_fail(foa_stream *s, int rc)
{
    std::lock_guard<std::mutex> lk(s->gpu.err_m);
    return fail(rc, "foa_stream: %s", s->gpu.err_text.empty() ? "call sequence error (push after flush?)" : s->gpu.err_text.c_str());
}

}  // namespace

extern "C" {

int foa_stream_create(foa_rx *rx, size_t batch_samples, int narrow_threads, foa_stream **out)
{
    if (!rx || !out) return fail(FOA_E_INVALID, "NULL argument");
    *out = nullptr;
    if (batch_samples < 4096 || batch_samples > ((size_t)1 << 28)) return fail(FOA_E_INVALID, "batch_samples must lie in [4096, 2^28]");
    if (narrow_threads < 0 || narrow_threads > 64) return fail(FOA_E_INVALID, "narrow_threads must lie in [0, 64]");
    // one stream per handle: its submitter thread owns the handle's streams, work sets and job slots until the stream is destroyed
    if (rx->open_stream) return fail(FOA_E_STATE, "a stream is already open on this handle: destroy it first (one engine per handle)");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }
    foa_stream *s = new foa_stream();
    StreamGpu &g = s->gpu;
    g.rx = rx;
    g.B = (int64_t)batch_samples;
    if (rx->stream_longest > 0) { g.L = rx->stream_longest; g.C = g.L + 2048; }
    g.desc_cap = (size_t)((g.C + g.B) / 300 + 64);
    // Batches in rotation: a batch is about a millisecond on its way whatever its size (upload, pre-sync, the host's look at the count, a
    // decode call whose forward pass walks the longest frame's trellis at the lone-wave rate), so batches that arrive every 0.2-0.8 ms
    // -- 4 .. 16 Ki samples at 20 Msample/s -- need more than six of them on their way at once; with six -- five staged or in flight -- the engine's capacity at 4 Ki
    // samples is 23 Msample/s, and a backlog, once there, drains at 3 while the caller waits for staging slots (profiles/r06_latency_stages.txt).  FOA_STREAM_BUFS overrides (A/B).
    g.n_bufs = batch_samples <= ((size_t)1 << 16) ? foa::kStreamBufs : 6;
    g.copies_in_line = batch_samples <= ((size_t)1 << 16) && !getenv("FOA_STREAM_COPY_STREAM");
    g.fill_by_kernel = g.copies_in_line && !getenv("FOA_STREAM_DMA_UPLOAD");
    if (const char *e = getenv("FOA_STREAM_BUFS")) { const int v = atoi(e); if (v >= 3 && v <= foa::kStreamBufs) g.n_bufs = v; }
    int rc = FOA_OK;
    for (int i = 0; i < g.n_bufs && !rc; i++) {
        if (hipHostMalloc((void **)&g.pin[i], (size_t)g.B * 8, hipHostMallocDefault) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc of a %zu-byte staging buffer failed", (size_t)g.B * 8);
        if (!rc) rc = g.dev[i].ensure((size_t)(g.C + g.B) * 2);
        if (!rc) rc = g.d_desc[i].ensure(g.desc_cap * sizeof(foa_frame_desc));
        if (!rc) rc = g.d_ends[i].ensure(g.desc_cap);
        if (!rc && hipEventCreateWithFlags(&g.in_done[i], hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
        if (!rc && hipEventCreateWithFlags(&g.sel_done[i], hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
    }
    if (!rc && hipStreamCreateWithPriority(&g.st_in, hipStreamNonBlocking, high_priority()) != hipSuccess) rc = fail(FOA_E_HIP, "hipStreamCreate failed");
    if (!rc && hipHostMalloc((void **)&g.sel, (size_t)foa::kStreamBufs * 8 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc failed");
    if (!rc) rc = g.sel_dev.ensure((size_t)foa::kStreamBufs * 8);
    if (!rc) rc = g.la_range.ensure((size_t)foa::kStreamBufs * 2);
    if (!rc) rc = g.la_info.ensure(g.desc_cap + 1);
    if (!rc) rc = g.la_hinv.ensure((g.desc_cap + 1) * 64);
    if (!rc) rc = g.state.ensure(1);
    if (!rc) {
        const StreamState first = { 0, 1.0, 0.0 };                  // nothing decided yet; timing_sync's m_phase_acc before the first frame: 0
        if (hipMemcpy(g.state.p, &first, sizeof first, hipMemcpyHostToDevice) != hipSuccess) rc = fail(FOA_E_HIP, "hipMemcpy failed");
    }
    if (rc) { foa_stream_destroy(s); return rc; }
    // Everything a batch will need is allocated HERE, not by the first batches that need it: a first-use hipMalloc (and every growth, whose
    // hipFree waits for the device) inside the stream cost a 53-batch capture a fifth of its time.  Work sets for a buffer's worth of
    // samples and as many frames as its shortest-plausible spacing gives, job slots for twice a 54 Mbps-dense batch (both grow if a stream
    // turns out denser).
    if (!rc) rc = foa_rx_reserve(rx, (size_t)(g.C + g.B), (size_t)((g.C + g.B) / 1200 + 64));
    for (auto &j : rx->jobs) {
        if (rc) break;
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t m = (size_t)(g.B / 3600 + 64), total = up(m * g.slot_bytes) + up(m * sizeof(foa_frame_result));
        const size_t roomy = (2 * total + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
        if (j.busy) continue;
        if (j.dev.n < roomy) rc = j.dev.ensure(roomy);
        if (!rc && j.pin_cap < roomy) {
            if (j.pin) (void)hipHostFree(j.pin);
            j.pin = nullptr; j.pin_cap = 0;
            if (hipHostMalloc((void **)&j.pin, roomy, hipHostMallocDefault) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc of a %zu-byte result mirror failed", roomy);
            else j.pin_cap = roomy;
        }
        if (!rc && !j.done && hipEventCreateWithFlags(&j.done, hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
    }
    if (rc) { foa_stream_destroy(s); return rc; }
    // How many decode calls' loops in flight?  A batch of a million samples and more fills the machine with its forward pass: two (2.2
    // against 2.0 Gsample/s through process_samples with four).  A SMALL batch is a handful of forward-pass waves, and one wave walks its
    // frames' trellis at the lone-wave rate (0.65 ms for a 1024-byte frame at 54 Mbps) whatever the batch: what sets the rate of 64 Ki-sample
    // batches is loops in flight over that latency -- 0.36 ms per batch with two -- so batches below a million samples get four where the
    // runtime has the hardware queues for them (foa_rx_create).  FOA_STREAM_DEPTH overrides (A/B).
    rx->finish_in_line = batch_samples >= 8192 && batch_samples <= ((size_t)1 << 16) && !getenv("FOA_STREAM_FINISH_STITCH");
    rx->timing = false;                              // (nobody reads per-kernel times of a stream's batches: seven runtime calls less per batch)
    rx->depth_saved = rx->depth;
    rx->depth = (batch_samples <= ((size_t)1 << 20) && rx->max_depth >= 4) ? 4 : 2;
    if (const char *e = getenv("FOA_STREAM_DEPTH")) { const int v = atoi(e); if (v >= 2 && v <= 4) rx->depth = v; }
    // (and the submitter thread of such a stream polls instead of sleeping while batches keep coming: stream_core.h; FOA_STREAM_SPIN_US overrides, 0 = it sleeps)
    int spin_us = batch_samples <= ((size_t)1 << 16) ? 2000 : 0;
    {   // (a process confined to a few CPUs needs them for the caller and the helpers: the submitter sleeps as it always did)
        cpu_set_t have;
        if (spin_us && sched_getaffinity(0, sizeof have, &have) == 0 && CPU_COUNT(&have) < 4) spin_us = 0;
    }
    if (const char *e = getenv("FOA_STREAM_SPIN_US")) { const int v = atoi(e); if (v >= 0 && v <= 1000000) spin_us = v; }
    s->core = new foa::StreamCore<StreamGpu>(&g, g.B, narrow_threads, g.n_bufs, spin_us);
    rx->open_stream = s;
    *out = s;
    return FOA_OK;
}

}  // extern "C"

// Everything the stream itself allocated, given back; idempotent (every pointer is cleared), so that it serves the normal shutdown, a
// creation that failed half-way -- whichever allocation it failed at -- and a destroy after either.
static void stream_free_buffers(StreamGpu &g)
{
    for (int i = 0; i < foa::kStreamBufs; i++) {
        if (g.pin[i]) (void)hipHostFree(g.pin[i]);
        g.pin[i] = nullptr;
        g.dev[i].release(); g.d_desc[i].release(); g.d_ends[i].release();
        if (g.in_done[i]) (void)hipEventDestroy(g.in_done[i]);
        if (g.sel_done[i]) (void)hipEventDestroy(g.sel_done[i]);
        g.in_done[i] = g.sel_done[i] = nullptr;
    }
    if (g.st_in) (void)hipStreamDestroy(g.st_in);
    g.st_in = nullptr;
    if (g.sel) (void)hipHostFree(g.sel);
    g.sel = nullptr;
    g.sel_dev.release(); g.state.release(); g.la_info.release(); g.la_hinv.release(); g.la_range.release();
}

// Stops the engine and gives everything it holds on the device back; the handle is the caller's again.  Idempotent: foa_rx_destroy
// calls it for a stream that is still open (its threads use the handle), and the owner's later foa_stream_destroy then only frees
// the shell -- every other foa_stream_* call on such a stream fails with FOA_E_STATE.
void foa::stream_shutdown(foa_stream *s)
{
    if (!s || !s->core) return;
    delete s->core;                                   // joins the helpers and the submitter: from here on this thread owns the handle
    s->core = nullptr;
    StreamGpu &g = s->gpu;
    if (getenv("FOA_STREAM_STATS"))
        fprintf(stderr, "foa_stream: %lld batches; submitter ms: waiting for upload + pre-sync %.1f, decode call %.1f (of which output slots %.1f, waiting for a work set %.1f), collect %.1f, polling %.1f\n",
                (long long)g.n_batches, g.t_sync * 1e-6, g.t_decode * 1e-6, g.t_prep * 1e-6, g.rx->ns_wait_set * 1e-6, g.t_collect * 1e-6, g.t_collect_wait * 1e-6);
    (void)hipSetDevice(g.rx->device);
    if (g.rx->open_stream == s) g.rx->open_stream = nullptr;
    (void)foa_rx_sync(g.rx);
    if (g.rx->depth_saved >= 0) { g.rx->depth = g.rx->depth_saved; g.rx->depth_saved = -1; }
    g.rx->timing = true;
    g.rx->finish_in_line = false;
    // the job slots of batches nobody took are released
    while (!g.flight.empty()) { foa::StreamReady r; if (g.collect(g.flight.front().handle, true, &r) <= 0) break; }
    stream_free_buffers(g);
}

extern "C" {

void foa_stream_destroy(foa_stream *s)
{
    if (!s) return;
    if (s->core) stream_shutdown(s);
    else {
        // no engine: creation failed half-way (buffers may exist, whichever allocation it failed at), or the stream was shut down already
        // (nothing is left: the call is a no-op then)
        if (s->gpu.rx) (void)hipSetDevice(s->gpu.rx->device);
        stream_free_buffers(s->gpu);
    }
    delete s;
}

#define FOA_STREAM_LIVE(s) do { if ((s) && !(s)->core) return fail(FOA_E_STATE, "the stream was shut down (its handle was destroyed)"); } while (0)

int foa_stream_push_f32(foa_stream *s, const float *iq, size_t n_samples)
{
    FOA_STREAM_LIVE(s);
    if (!s || (n_samples && !iq)) return fail(FOA_E_INVALID, "NULL argument");
    const int rc = s->core->push(iq, n_samples, nullptr, nullptr);
    return rc ? stream_fail(s, rc) : FOA_OK;
}
int foa_stream_push_f64(foa_stream *s, const double *iq, size_t n_samples)
{
    FOA_STREAM_LIVE(s);
    if (!s || (n_samples && !iq)) return fail(FOA_E_INVALID, "NULL argument");
    const int rc = s->core->push(iq, n_samples, nullptr, nullptr);
    return rc ? stream_fail(s, rc) : FOA_OK;
}
int foa_stream_push_f64_owned(foa_stream *s, const double *iq, size_t n_samples, void (*release)(void *), void *ctx)
{
    if (s && !s->core) { if (release) release(ctx); return fail(FOA_E_STATE, "the stream was shut down (its handle was destroyed)"); }
    if (!s || (n_samples && !iq) || !release) { if (release) release(ctx); return fail(FOA_E_INVALID, "NULL argument"); }
    const int rc = s->core->push(iq, n_samples, release, ctx);
    return rc ? stream_fail(s, rc) : FOA_OK;
}

int foa_stream_flush(foa_stream *s)
{
    FOA_STREAM_LIVE(s);
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    const int rc = s->core->flush();
    return rc ? stream_fail(s, rc) : FOA_OK;
}

int foa_stream_ready(foa_stream *s, int wait, size_t *n_payloads, size_t *n_bytes)
{
    FOA_STREAM_LIVE(s);
    if (!s || !n_payloads || !n_bytes) return fail(FOA_E_INVALID, "NULL argument");
    *n_payloads = 0; *n_bytes = 0;
    if (!s->have_ready) {
        s->ready = foa::StreamReady();
        const int rc = s->core->take(wait != 0, &s->ready);
        if (rc < 0) return stream_fail(s, rc);
        if (rc == 0) return 0;
        s->have_ready = true;
    }
    *n_payloads = s->ready.len.size();
    *n_bytes = s->ready.bytes.size();
    return 1;
}

int foa_stream_take(foa_stream *s, uint8_t *payloads, uint32_t *lengths)
{
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    if (!s->have_ready) return fail(FOA_E_STATE, "foa_stream_take without a batch reported by foa_stream_ready");
    const foa::StreamReady &r = s->ready;
    if (!r.len.empty() && (!payloads || !lengths)) return fail(FOA_E_INVALID, "NULL argument");
    if (!r.bytes.empty()) memcpy(payloads, r.bytes.data(), r.bytes.size());
    if (!r.len.empty()) memcpy(lengths, r.len.data(), r.len.size() * sizeof(uint32_t));
    s->have_ready = false;
    return FOA_OK;
}

int foa_stream_stats(const foa_stream *s, uint64_t out[8])
{
    FOA_STREAM_LIVE(s);
    if (!s || !out) return fail(FOA_E_INVALID, "NULL argument");
    for (int i = 0; i < 5; i++) out[i] = s->gpu.status_count[i].load();
    out[5] = s->gpu.alignments.load(); out[6] = (uint64_t)s->core->batches_closed(); out[7] = (uint64_t)s->core->pushed();
    return FOA_OK;
}

}  // extern "C"
