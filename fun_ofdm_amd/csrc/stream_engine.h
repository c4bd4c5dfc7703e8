// stream_engine.h -- receiver_chain::process_samples() with EVERYTHING on the device (SURVEY 8f #1 + #3):
// foa_stream_* of include/fun_ofdm_amd.h.  Included by foa_rx.hip (it drives the handle's streams and work sets).
//
// The reference runs frame_detector -> timing_sync -> fft_symbols -> ... -> frame_decoder inside every
// process_samples() call (src/receiver_chain.cpp:106-126) on 4096-sample chunks, carrying 16 + 160 samples and the frame in
// progress from call to call.  A GPU wants millions of samples per launch, so the engine cuts the stream into BATCHES of
// `batch_samples` and lets consecutive batches OVERLAP instead of carrying state:
//
//   stream      ....|<------------- batch k-1 ------------->|<-------------- batch k --------------->|....
//   device buf k              |<-- carry C -->|<------------- batch k (H2D) -------------------------->|
//                             ^ start_k = pos_k - C                          cut_k = pos_k + B - L    ^ pos_k + B
//
//   * device buffer k = the last C samples before the batch (a device-to-device copy out of buffer k-1) + the batch;
//   * foa_rx_sync_dev runs over the whole buffer; of the alignments it finds, batch k DECODES those whose STS_END
//     sample x lies in [cut_(k-1), cut_k): L = longest possible frame + slack, so every such frame is complete inside the
//     buffer, and C = L + 2048, so everything within 2048 samples before cut_(k-1) is inside buffer k as well -- the
//     detector's 16-sample windows, the plateau that ends in an STS_END, the earlier hit that can overwrite a tag
//     (timing_sync.cpp:105-106) all see the same samples as they would in one pass over the whole stream;
//   * the one piece of state that does cross batches, the phasor timing_sync left in force (m_phase_acc,
//     timing_sync.cpp:113-125), is the phasor of the last alignment decoded so far: the host patches it into the first
//     descriptor of the batch.
// Batches are queued through the same asynchronous job slots as foa_rx_submit_host (pinned mirrors, D2H behind the
// finish kernel), so H2D of batch k+1, the kernels of batch k and the D2H of batch k-1 overlap, and results come out in
// stream order.  The double -> float narrowing of process_samples' complex<double> input is the only per-sample work the
// host does; above 32 Ki samples per call it is spread over a few worker threads.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace foa {

constexpr int64_t kStreamLongest = 110592;          // >= 320 + 80 * 1369 (4095 bytes at 6 Mbps) + 32 + timing_sync's 160-sample look-ahead
constexpr int64_t kStreamCarry = kStreamLongest + 2048;
constexpr int kStreamBufs = 4;                      // device sample buffers / pinned staging buffers in rotation

// a few persistent threads that narrow double -> float (or copy floats) into the pinned staging buffer
class NarrowPool {
public:
    explicit NarrowPool(int threads) : stop_(false), gen_(0), left_(0)
    {
        for (int i = 0; i < threads; i++) th_.emplace_back([this, i] { loop(i); });
    }
    ~NarrowPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    // dst[0 .. 2n) = (float) src[0 .. 2n); the caller takes a share too
    void run(float *dst, const double *src, size_t n)
    {
        const int parts = size() + 1;
        const size_t per = ((n + parts - 1) / parts + 15) & ~(size_t)15;
        { std::lock_guard<std::mutex> lk(m_); dst_ = dst; src_ = src; n_ = n; per_ = per; left_ = size(); gen_++; }
        cv_.notify_all();
        narrow(dst, src, 0, std::min(per, n));
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return left_ == 0; });
    }
    static void narrow(float *dst, const double *src, size_t lo, size_t hi)
    {
        for (size_t i = 2 * lo; i < 2 * hi; i++) dst[i] = (float)src[i];
    }

private:
    void loop(int i)
    {
        uint64_t seen = 0;
        for (;;) {
            float *dst; const double *src; size_t n, per;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                dst = dst_; src = src_; n = n_; per = per_;
            }
            const size_t lo = std::min(n, per * (size_t)(i + 1)), hi = std::min(n, per * (size_t)(i + 2));
            narrow(dst, src, lo, hi);
            { std::lock_guard<std::mutex> lk(m_); left_--; }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    uint64_t gen_;
    int left_;
    float *dst_ = nullptr;
    const double *src_ = nullptr;
    size_t n_ = 0, per_ = 0;
};

}  // namespace foa

struct foa_stream {
    foa_rx *rx = nullptr;
    int64_t B = 0;                                   // batch_samples
    size_t slot = 4096;
    // rotating buffers
    float *pin[foa::kStreamBufs] = {};               // page-locked staging, B float2 each
    DevBuf<float> dev[foa::kStreamBufs];             // (C + B) float2 each
    hipEvent_t in_done[foa::kStreamBufs] = {};       // H2D + carry copy of the buffer's batch are through
    hipStream_t st_in = nullptr;                     // carry copies and H2D
    DevBuf<uint8_t> d_desc[foa::kStreamBufs];
    DevBuf<int64_t> d_ends[foa::kStreamBufs];
    size_t desc_cap = 0;
    // stream state
    int64_t pushed = 0;                              // samples accepted so far
    int64_t fill = 0;                                // samples in the staging buffer of the batch being filled
    int64_t n_batches = 0;                           // batches submitted
    int64_t cut_prev = 0;                            // STS_END positions below this have been dealt with
    double prev_c = 1.0, prev_s = 0.0;               // phasor of the last alignment decoded (timing_sync's m_phase_acc)
    struct InFlight { uint64_t ticket; size_t n_frames; int buf; };
    std::deque<InFlight> flight;                     // batches whose results have not been taken yet (stream order)
    // finished batches, unpacked (CRC-passing payloads back to back), oldest first
    struct Ready { std::vector<uint8_t> bytes; std::vector<uint32_t> len; };
    std::deque<Ready> ready;
    std::vector<uint8_t> tmp_psdu;
    std::vector<foa_frame_result> tmp_res;
    bool finished = false;                           // foa_stream_flush has been called: the stream is over
    uint64_t status_count[5] = { 0, 0, 0, 0, 0 };
    uint64_t alignments = 0;
    foa::NarrowPool *pool = nullptr;
    std::vector<foa_frame_desc> h_desc;
};

namespace {

// Take the oldest batch in flight out of its job slot (wait = block until it is complete).  1 = taken, 0 = not yet.
int stream_collect_oldest(foa_stream *s, int wait)
{
    if (s->flight.empty()) return 0;
    const foa_stream::InFlight f = s->flight.front();
    foa_stream::Ready out;
    if (f.n_frames) {
        s->tmp_psdu.resize(f.n_frames * s->slot);
        s->tmp_res.resize(f.n_frames);
        const int rc = foa_rx_collect(s->rx, f.ticket, wait, s->tmp_psdu.data(), s->tmp_res.data());
        if (rc <= 0) return rc;
        for (size_t i = 0; i < f.n_frames; i++) {
            const foa_frame_result &r = s->tmp_res[i];
            if (r.status >= 0 && r.status < 5) s->status_count[r.status]++;
            if (r.status != FOA_ST_OK) continue;
            out.len.push_back((uint32_t)r.length);
            out.bytes.insert(out.bytes.end(), s->tmp_psdu.begin() + i * s->slot, s->tmp_psdu.begin() + i * s->slot + r.length);
        }
    }
    s->flight.pop_front();
    s->ready.push_back(std::move(out));
    return 1;
}

// Queue the batch in the staging buffer (fill samples; final = the stream ends here: decode everything that is left).
int stream_submit(foa_stream *s, bool final)
{
    foa_rx *rx = s->rx;
    const int64_t C = foa::kStreamCarry, L = foa::kStreamLongest;
    const int k = (int)(s->n_batches % foa::kStreamBufs), kp = (int)((s->n_batches + foa::kStreamBufs - 1) % foa::kStreamBufs);
    const int64_t n_new = s->fill, n_buf = C + n_new;
    const int64_t start = s->pushed - n_new - C;                   // stream index of the buffer's first sample
    // the buffer (and its descriptor arrays) may still be read by the batch that used it four batches ago
    while ((int)s->flight.size() >= foa::kStreamBufs - 1 || (int)s->flight.size() >= kMaxJobs - 1) {
        const int rc = stream_collect_oldest(s, 1);                // (its payloads wait in s->ready until they are taken)
        if (rc < 0) return rc;
    }
    HIP_TRY(hipSetDevice(rx->device));
    float *d = s->dev[k].p;
    if (s->n_batches == 0) HIP_TRY(hipMemsetAsync(d, 0, (size_t)C * 8, s->st_in));                 // silence before the stream
    else HIP_TRY(hipMemcpyAsync(d, s->dev[kp].p + 2 * s->B, (size_t)C * 8, hipMemcpyDeviceToDevice, s->st_in));   // (every batch but the last is full)
    if (n_new) HIP_TRY(hipMemcpyAsync(d + 2 * C, s->pin[k], (size_t)n_new * 8, hipMemcpyHostToDevice, s->st_in));
    HIP_TRY(hipEventRecord(s->in_done[k], s->st_in));
    const bool piped = rx->pipeline && rx->viterbi_kind == 2;
    HIP_TRY(hipStreamWaitEvent(piped ? rx->stream3 : rx->stream, s->in_done[k], 0));                // pre-sync and front end follow there
    size_t found = 0;
    int rc = foa_rx_sync_dev(rx, d, (size_t)n_buf, (foa_frame_desc *)s->d_desc[k].p, s->d_ends[k].p, s->desc_cap, &found);
    if (rc) return rc;
    // which of them are this batch's: STS_END sample in [cut_prev, cut)
    const int64_t cut = final ? s->pushed + 1 : s->pushed - L;
    size_t i0 = 0, i1 = 0;
    if (found) {
        s->h_desc.resize(found);
        HIP_TRY(hipMemcpy(s->h_desc.data(), s->d_desc[k].p, found * sizeof(foa_frame_desc), hipMemcpyDeviceToHost));
        while (i0 < found && start + s->h_desc[i0].rot_start < s->cut_prev) i0++;
        i1 = i0;
        while (i1 < found && start + s->h_desc[i1].rot_start < cut) i1++;
    }
    const size_t m = i1 - i0;
    foa_stream::InFlight fl;
    fl.n_frames = m; fl.buf = k; fl.ticket = 0;
    if (m) {
        // timing_sync's phasor before the first alignment of the batch: the one the last decoded alignment set
        foa_frame_desc &first = s->h_desc[i0];
        first.c_prev = s->prev_c; first.s_prev = s->prev_s;
        HIP_TRY(hipMemcpyAsync(s->d_desc[k].p + i0 * sizeof(foa_frame_desc), &first, sizeof first, hipMemcpyHostToDevice, piped ? rx->stream3 : rx->stream));
        s->prev_c = s->h_desc[i1 - 1].c; s->prev_s = s->h_desc[i1 - 1].s;
        // outputs go through a job slot of the asynchronous host entry (page-locked mirror, D2H behind the finish kernel)
        HostJob *job = nullptr;
        for (auto &j : rx->jobs) if (!j.busy) { job = &j; break; }
        if (!job) return fail(FOA_E_STATE, "internal: no free job slot");
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_res = up(m * s->slot), total = o_res + up(m * sizeof(foa_frame_result));
        if ((rc = job->dev.ensure(total))) return rc;
        if (job->pin_cap < total) {
            if (job->pin) (void)hipHostFree(job->pin);
            job->pin = nullptr; job->pin_cap = 0;
            const size_t want = total + total / 2;
            HIP_TRY(hipHostMalloc((void **)&job->pin, want, hipHostMallocDefault));
            job->pin_cap = want;
        }
        if (!job->done) HIP_TRY(hipEventCreateWithFlags(&job->done, hipEventDisableTiming));
        HIP_TRY(hipMemsetAsync(job->dev.p, 0, m * s->slot, piped ? rx->stream3 : rx->stream));
        job->total = total; job->o_psdu = 0; job->o_res = o_res; job->n_frames = m; job->slot_bytes = s->slot; job->copy_queued = false;
        rx->attach_job = piped ? job : nullptr;
        rc = foa_rx_decode_frames_dev(rx, d, (size_t)n_buf, (const foa_frame_desc *)s->d_desc[k].p + i0, s->d_ends[k].p + i0, m, job->dev.p, s->slot,
                                      (foa_frame_result *)(job->dev.p + o_res));
        rx->attach_job = nullptr;
        if (rc) return rc;
        if (!piped) {
            HIP_TRY(hipMemcpyAsync(job->pin, job->dev.p, total, hipMemcpyDeviceToHost, rx->stream));
            HIP_TRY(hipEventRecord(job->done, rx->stream));
            job->copy_queued = true;
        }
        job->busy = true;
        job->ticket = rx->next_ticket++;
        fl.ticket = job->ticket;
        s->alignments += m;
    }
    s->flight.push_back(fl);
    s->cut_prev = cut;
    s->n_batches++;
    s->fill = 0;
    return FOA_OK;
}

template <typename T>
int stream_push(foa_stream *s, const T *iq, size_t n)
{
    if (!s || (n && !iq)) return fail(FOA_E_INVALID, "NULL argument");
    if (s->finished) return fail(FOA_E_STATE, "foa_stream: the stream was flushed; create a new one");
    while (n) {
        const int k = (int)(s->n_batches % foa::kStreamBufs);
        const size_t room = (size_t)(s->B - s->fill), take = n < room ? n : room;
        float *dst = s->pin[k] + 2 * s->fill;
        if (sizeof(T) == sizeof(float)) memcpy(dst, iq, take * 8);
        else if (s->pool && take >= 32768) s->pool->run(dst, (const double *)iq, take);
        else foa::NarrowPool::narrow(dst, (const double *)iq, 0, take);
        s->fill += (int64_t)take; s->pushed += (int64_t)take;
        iq += 2 * take; n -= take;
        if (s->fill == s->B) {
            // the staging buffer of the NEXT batch must be free again: its last H2D was four batches ago and is long done,
            // but make that explicit rather than assumed
            const int rc = stream_submit(s, false);
            if (rc) return rc;
            HIP_TRY(hipEventSynchronize(s->in_done[(int)(s->n_batches % foa::kStreamBufs)]));
        }
    }
    return FOA_OK;
}

}  // namespace

extern "C" {

int foa_stream_create(foa_rx *rx, size_t batch_samples, int narrow_threads, foa_stream **out)
{
    if (!rx || !out) return fail(FOA_E_INVALID, "NULL argument");
    *out = nullptr;
    if (batch_samples < 4096 || batch_samples > ((size_t)1 << 28)) return fail(FOA_E_INVALID, "batch_samples must lie in [4096, 2^28]");
    if (narrow_threads < 0 || narrow_threads > 64) return fail(FOA_E_INVALID, "narrow_threads must lie in [0, 64]");
    HIP_TRY(hipSetDevice(rx->device));
    foa_stream *s = new foa_stream();
    s->rx = rx;
    s->B = (int64_t)batch_samples;
    s->desc_cap = (size_t)((foa::kStreamCarry + s->B) / 300 + 64);
    int rc = FOA_OK;
    for (int i = 0; i < foa::kStreamBufs && !rc; i++) {
        if (hipHostMalloc((void **)&s->pin[i], (size_t)s->B * 8, hipHostMallocDefault) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc of a %zu-byte staging buffer failed", (size_t)s->B * 8);
        if (!rc) rc = s->dev[i].ensure((size_t)(foa::kStreamCarry + s->B) * 2);
        if (!rc) rc = s->d_desc[i].ensure(s->desc_cap * sizeof(foa_frame_desc));
        if (!rc) rc = s->d_ends[i].ensure(s->desc_cap);
        if (!rc && hipEventCreateWithFlags(&s->in_done[i], hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
        if (!rc && hipEventRecord(s->in_done[i], rx->stream) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventRecord failed");
    }
    if (!rc && hipStreamCreateWithFlags(&s->st_in, hipStreamNonBlocking) != hipSuccess) rc = fail(FOA_E_HIP, "hipStreamCreate failed");
    if (rc) { foa_stream_destroy(s); return rc; }
    if (narrow_threads > 0) s->pool = new foa::NarrowPool(narrow_threads);
    *out = s;
    return FOA_OK;
}

void foa_stream_destroy(foa_stream *s)
{
    if (!s) return;
    (void)hipSetDevice(s->rx->device);
    (void)foa_rx_sync(s->rx);
    // the job slots of batches nobody took are released
    while (!s->flight.empty()) if (stream_collect_oldest(s, 1) <= 0) break;
    delete s->pool;
    for (int i = 0; i < foa::kStreamBufs; i++) {
        if (s->pin[i]) (void)hipHostFree(s->pin[i]);
        s->dev[i].release(); s->d_desc[i].release(); s->d_ends[i].release();
        if (s->in_done[i]) (void)hipEventDestroy(s->in_done[i]);
    }
    if (s->st_in) (void)hipStreamDestroy(s->st_in);
    delete s;
}

int foa_stream_push_f32(foa_stream *s, const float *iq, size_t n_samples) { return stream_push(s, iq, n_samples); }
int foa_stream_push_f64(foa_stream *s, const double *iq, size_t n_samples) { return stream_push(s, iq, n_samples); }

int foa_stream_flush(foa_stream *s)
{
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    if (s->finished) return FOA_OK;
    s->finished = true;
    // (also with an empty staging buffer: the alignments after the last cut are still undecoded)
    return stream_submit(s, true);
}

int foa_stream_ready(foa_stream *s, int wait, size_t *n_payloads, size_t *n_bytes)
{
    if (!s || !n_payloads || !n_bytes) return fail(FOA_E_INVALID, "NULL argument");
    *n_payloads = 0; *n_bytes = 0;
    if (s->ready.empty()) {
        HIP_TRY(hipSetDevice(s->rx->device));
        const int rc = stream_collect_oldest(s, wait);
        if (rc <= 0) return rc;
    }
    *n_payloads = s->ready.front().len.size();
    *n_bytes = s->ready.front().bytes.size();
    return 1;
}

int foa_stream_take(foa_stream *s, uint8_t *payloads, uint32_t *lengths)
{
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    if (s->ready.empty()) return fail(FOA_E_STATE, "foa_stream_take without a batch reported by foa_stream_ready");
    const foa_stream::Ready &r = s->ready.front();
    if (!r.len.empty() && (!payloads || !lengths)) return fail(FOA_E_INVALID, "NULL argument");
    if (!r.len.empty()) {
        if (!r.bytes.empty()) memcpy(payloads, r.bytes.data(), r.bytes.size());
        memcpy(lengths, r.len.data(), r.len.size() * sizeof(uint32_t));
    }
    s->ready.pop_front();
    return FOA_OK;
}

int foa_stream_stats(const foa_stream *s, uint64_t out[8])
{
    if (!s || !out) return fail(FOA_E_INVALID, "NULL argument");
    for (int i = 0; i < 5; i++) out[i] = s->status_count[i];
    out[5] = s->alignments; out[6] = (uint64_t)s->n_batches; out[7] = (uint64_t)s->pushed;
    return FOA_OK;
}

}  // extern "C"
