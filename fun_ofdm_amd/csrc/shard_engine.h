// shard_engine.h -- foa_shard_*: one stream dealt over several devices (include/fun_ofdm_amd.h; the logic is shard_core.h, this is its GPU
// side).  Included by rx_stream.hip behind stream_engine.h, whose batch geometry (carry, longest frame, job slots) it shares.
//
// Every device has its own receiver handle -- its own streams, work sets, pre-sync scratch and job slots -- created and destroyed by
// the shard.  Batch k of the stream goes to device k mod N: two host-to-device copies (the C samples before the batch out of the host's
// page-locked carry, the batch out of its staging slot), the pre-sync kernels over the buffer, the selection of the batch's alignments
// with the phasor the batch before left (handed on by the host: sixteen bytes), the decode call, the copy back of its PSDUs.  No device
// reads another's memory, nothing is gathered on a device: payloads are merged in stream order on the host, where the caller wants them.
#pragma once

#include "shard_core.h"

struct ShardDev {
    foa_rx *rx = nullptr;
    int64_t B = 0;
    size_t slot_bytes = 4096, desc_cap = 0;
    DevBuf<float> dev[foa::kStreamBufs];
    DevBuf<uint8_t> d_desc[foa::kStreamBufs];
    DevBuf<int64_t> d_ends[foa::kStreamBufs];
    hipEvent_t in_done[foa::kStreamBufs] = {}, sel_done[foa::kStreamBufs] = {}, sync_done[foa::kStreamBufs] = {};
    hipStream_t st_in = nullptr;
    hipStream_t st_sel = nullptr;                    // the chain's own stream: a batch's look-ahead waits for that batch's pre-sync and nothing else -- on the
                                                     // side stream it would sit behind the pre-syncs of the device's NEXT batches, which are staged ahead
    DevBuf<StreamState> state;                       // the chain state as handed in, moved on by the look-ahead
    DevBuf<FrameInfo> la_info;                       // the look-ahead's own alignment records / channel estimates (the chain is serial: one batch at a time)
    DevBuf<double2> la_hinv;
    DevBuf<int32_t> la_range;
    DevBuf<int32_t> sel_dev;
    DevBuf<int32_t> syn_dev;                         // per buffer: the pre-sync's counts (rx->sy_n) as they stood when ITS kernels finished -- the
                                                     // look-ahead is queued later, possibly after the pre-sync of the handle's next batch, which reuses the scratch
    int32_t *sel = nullptr;                          // page-locked: per buffer { STS_END candidates, alignments found, first of the batch, how many it decides, context behind them }
    StreamState *st_pin = nullptr;                   // page-locked: per buffer { state handed in, state after the batch }
    int32_t ccap[foa::kStreamBufs] = {};
    int64_t n_buf[foa::kStreamBufs] = {}, n_eff[foa::kStreamBufs] = {}, start_abs[foa::kStreamBufs] = {};
    struct Fl { uint64_t handle, ticket; size_t n_frames; };
    std::deque<Fl> flight;
    uint64_t next_handle = 1;
    std::atomic<uint64_t> status_count[5], alignments;
    std::mutex *err_m = nullptr;
    std::string *err_text = nullptr;

    ShardDev() { for (auto &c : status_count) c.store(0); alignments.store(0); }
    int keep(int rc)
    {
        if (rc) { std::lock_guard<std::mutex> lk(*err_m); if (err_text->empty()) *err_text = last_error_text(); }
        return rc;
    }
    int init(foa_rx *handle, int64_t batch)
    {
        rx = handle; B = batch;
        desc_cap = (size_t)((foa::kStreamCarry + B) / 300 + 64);
        HIP_TRY(enter_device(rx->device));
        int rc = FOA_OK;
        for (int i = 0; i < foa::kStreamBufs && !rc; i++) {
            rc = dev[i].ensure((size_t)(foa::kStreamCarry + B) * 2);
            if (!rc) rc = d_desc[i].ensure(desc_cap * sizeof(foa_frame_desc));
            if (!rc) rc = d_ends[i].ensure(desc_cap);
            if (!rc && hipEventCreateWithFlags(&in_done[i], hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
            if (!rc && hipEventCreateWithFlags(&sel_done[i], hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
            if (!rc && hipEventCreateWithFlags(&sync_done[i], hipEventDisableTiming) != hipSuccess) rc = fail(FOA_E_HIP, "hipEventCreate failed");
        }
        // (the look-ahead stream first: made right behind the handle's six it lands on the pipe of the handle's copy / pre-sync stream, the
        // staging stream on that of lane 2, whose forward pass holds its pipe for a millisecond at a time -- rx_handle.hip, foa_rx_create)
        if (!rc && hipStreamCreateWithPriority(&st_sel, hipStreamNonBlocking, foa::high_priority()) != hipSuccess) rc = fail(FOA_E_HIP, "hipStreamCreate failed");
        if (!rc && hipStreamCreateWithPriority(&st_in, hipStreamNonBlocking, foa::high_priority()) != hipSuccess) rc = fail(FOA_E_HIP, "hipStreamCreate failed");
        if (!rc && hipHostMalloc((void **)&sel, (size_t)foa::kStreamBufs * 8 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc failed");
        if (!rc && hipHostMalloc((void **)&st_pin, (size_t)foa::kStreamBufs * 2 * sizeof(StreamState), hipHostMallocDefault) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc failed");
        if (!rc) rc = sel_dev.ensure((size_t)foa::kStreamBufs * 8);
        if (!rc) rc = syn_dev.ensure((size_t)foa::kStreamBufs * 8);
        if (!rc) rc = la_range.ensure((size_t)foa::kStreamBufs * 2);
        if (!rc) rc = la_info.ensure(desc_cap + 1);
        if (!rc) rc = la_hinv.ensure((desc_cap + 1) * 64);
        if (!rc) rc = state.ensure(1);
        // everything a batch will need is allocated here, not by the first batches that need it (stream_engine.h)
        if (!rc) rc = foa_rx_reserve(rx, (size_t)(foa::kStreamCarry + B), (size_t)((foa::kStreamCarry + B) / 1200 + 64));
        if (!rc) { rx->depth_saved = rx->depth; rx->depth = (B <= ((int64_t)1 << 20) && rx->max_depth >= 4) ? 4 : 2; rx->timing = false; }      // (as the single-device engine: small batches want four loops in flight, nobody reads per-kernel times)
        return rc;
    }
    void release()
    {
        if (!rx) return;
        (void)hipSetDevice(rx->device);
        (void)foa_rx_sync(rx);
        while (!flight.empty()) { foa::StreamReady r; if (collect(flight.front().handle, true, &r) <= 0) break; }
        for (int i = 0; i < foa::kStreamBufs; i++) {
            dev[i].release(); d_desc[i].release(); d_ends[i].release();
            if (in_done[i]) (void)hipEventDestroy(in_done[i]);
            if (sel_done[i]) (void)hipEventDestroy(sel_done[i]);
            if (sync_done[i]) (void)hipEventDestroy(sync_done[i]);
            in_done[i] = sel_done[i] = sync_done[i] = nullptr;
        }
        if (st_in) (void)hipStreamDestroy(st_in);
        if (st_sel) (void)hipStreamDestroy(st_sel);
        st_in = st_sel = nullptr;
        if (sel) (void)hipHostFree(sel);
        if (st_pin) (void)hipHostFree(st_pin);
        sel = nullptr; st_pin = nullptr;
        sel_dev.release(); state.release(); syn_dev.release(); la_info.release(); la_hinv.release(); la_range.release();
    }

    // ---- the Dev interface of shard_core.h (submitter thread only) ----
    int upload(int k, const float *carry, const float *batch, int64_t n_new, int64_t start) { return keep(upload_impl(k, carry, batch, n_new, start)); }
    int upload_impl(int k, const float *carry, const float *batch, int64_t n_new, int64_t start)
    {
        const int64_t C = foa::kStreamCarry;
        HIP_TRY(enter_device(rx->device));
        float *d = dev[k].p;
        HIP_TRY(hipMemcpyAsync(d, carry, (size_t)C * 8, hipMemcpyHostToDevice, st_in));
        if (n_new) HIP_TRY(hipMemcpyAsync(d + 2 * C, batch, (size_t)n_new * 8, hipMemcpyHostToDevice, st_in));
        HIP_TRY(hipEventRecord(in_done[k], st_in));
        hipStream_t st = side_stream(rx);
        HIP_TRY(hipStreamWaitEvent(st, in_done[k], 0));
        n_buf[k] = C + n_new; start_abs[k] = start;
        int rc = sync_dev_issue(rx, d, (size_t)n_buf[k], (foa_frame_desc *)d_desc[k].p, d_ends[k].p, desc_cap, &ccap[k], start);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(syn_dev.p + 8 * k, rx->sy_n.p, 8 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipEventRecord(sync_done[k], st));
        HIP_TRY(hipGetLastError());
        return FOA_OK;
    }
    int select(int k, int64_t n_eff_k, bool final, const foa::ChainState &in) { return keep(select_impl(k, n_eff_k, final, in)); }
    int select_impl(int k, int64_t n_eff_k, bool final, const foa::ChainState &in)
    {
        HIP_TRY(enter_device(rx->device));
        hipStream_t st = st_sel;
        HIP_TRY(hipStreamWaitEvent(st, sync_done[k], 0));            // this buffer's pre-sync (and its counts in syn_dev): nothing else
        n_eff[k] = n_eff_k;
        st_pin[2 * k].lo_abs = in.lo_abs; st_pin[2 * k].c = in.c; st_pin[2 * k].s = in.s;
        HIP_TRY(hipMemcpyAsync(state.p, st_pin + 2 * k, sizeof(StreamState), hipMemcpyHostToDevice, st));
        const int64_t hz_abs = start_abs[k] + n_eff_k;
        launch_stream_range(st, (foa_frame_desc *)d_desc[k].p, syn_dev.p + 8 * k, (int32_t)desc_cap, start_abs[k], hz_abs, state.p, la_range.p + 2 * k);
        launch_stream_resolve(st, dev[k].p, n_eff_k, (const foa_frame_desc *)d_desc[k].p, d_ends[k].p, la_range.p + 2 * k, syn_dev.p + 8 * k, start_abs[k], hz_abs, final,
                              state.p, la_info.p, la_hinv.p, sel_dev.p + 8 * k, (unsigned)desc_cap);
        HIP_TRY(hipMemcpyAsync(sel + 8 * k, sel_dev.p + 8 * k, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(st_pin + 2 * k + 1, state.p, sizeof(StreamState), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(sel_done[k], st));
        HIP_TRY(hipGetLastError());
        return FOA_OK;
    }
    int selected(int k, foa::ChainState *out)
    {
        (void)hipSetDevice(rx->device);                  // (the submitter thread's current device is whichever handle it served last)
        const hipError_t e = hipEventQuery(sel_done[k]);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) return keep(fail(FOA_E_HIP, "hipEventQuery: %s", hipGetErrorString(e)));
        out->lo_abs = st_pin[2 * k + 1].lo_abs; out->c = st_pin[2 * k + 1].c; out->s = st_pin[2 * k + 1].s;
        return 1;
    }
    int decode(int k, int64_t n_new, uint64_t *handle) { (void)n_new; return keep(decode_impl(k, handle)); }
    int decode_impl(int k, uint64_t *handle)
    {
        HIP_TRY(enter_device(rx->device));
        const int32_t *q = sel + 8 * k;
        if (q[0] > ccap[k]) return fail(FOA_E_NOMEM, "too many STS_END candidates (%d)", q[0]);
        if ((size_t)q[1] > desc_cap) return fail(FOA_E_INVALID, "internal: %d alignments in one batch buffer", q[1]);
        const size_t i0 = (size_t)q[2], m = (size_t)q[3], n_ctx = (size_t)q[4];
        Fl fl;
        fl.handle = next_handle++; fl.n_frames = m; fl.ticket = 0;
        if (m) {
            // (context for the batch's cut frames, and the stream's end for its decisions, as in stream_engine.h.  The host has SEEN the
            // look-ahead's event -- selected() -- so the descriptor it patched is in place)
            int rc = stream_decode_batch(rx, dev[k].p, (size_t)n_eff[k], (const foa_frame_desc *)d_desc[k].p, d_ends[k].p, i0, m, n_ctx, slot_bytes, &fl.ticket, nullptr);
            if (rc) return rc;
            alignments.fetch_add(m);
        }
        flight.push_back(fl);
        *handle = fl.handle;
        return FOA_OK;
    }
    int collect(uint64_t handle, bool wait, foa::StreamReady *out)
    {
        if (flight.empty() || flight.front().handle != handle) return keep(fail(FOA_E_STATE, "internal: batches collected out of order"));
        const Fl f = flight.front();
        if (f.n_frames) {
            (void)hipSetDevice(rx->device);
            uint64_t by_status[5] = { 0, 0, 0, 0, 0 };
            const int rc = stream_collect_job(rx, f.ticket, f.n_frames, wait, out, by_status);
            if (rc < 0) { flight.pop_front(); return keep(rc); }
            if (rc == 0) return 0;
            for (int i = 0; i < 5; i++) if (by_status[i]) status_count[i].fetch_add(by_status[i], std::memory_order_relaxed);
        }
        flight.pop_front();
        return 1;
    }
};

typedef foa::ShardBackend<ShardDev> ShardBe;
static_assert(ShardBe::kSlots <= foa::StreamCore<ShardBe>::kSlots && ShardBe::kSlots <= foa::kStreamBufs, "one staging slot, one carry and one device buffer per slot of the core");
static_assert(sizeof(foa::ChainState) == sizeof(StreamState) && foa::kShardSettle == foa::kStreamSettle, "shard_core.h mirrors the device code's chain state");

struct foa_shard {
    std::vector<foa_rx *> rx;                        // one handle per entry of the device list (the same device may appear more than once)
    std::vector<ShardDev *> devs;
    float *staging[foa::kStreamBufs] = {}, *carry[foa::kStreamBufs] = {};
    ShardBe *be = nullptr;
    foa::StreamCore<ShardBe> *core = nullptr;
    foa::StreamReady ready;
    bool have_ready = false;
    std::mutex err_m;
    std::string err_text;
};

static int shard_fail(foa_shard *s, int rc)
{
    std::lock_guard<std::mutex> lk(s->err_m);
    return fail(rc, "foa_shard: %s", s->err_text.empty() ? "call sequence error (push after flush?)" : s->err_text.c_str());
}

extern "C" {

void foa_shard_destroy(foa_shard *s)
{
    if (!s) return;
    delete s->core;                                   // joins the helpers and the submitter
    s->core = nullptr;
    for (auto *d : s->devs) { d->release(); delete d; }
    s->devs.clear();
    delete s->be;
    for (auto *h : s->rx) foa_rx_destroy(h);
    for (int i = 0; i < foa::kStreamBufs; i++) {
        if (s->staging[i]) (void)hipHostFree(s->staging[i]);
        if (s->carry[i]) (void)hipHostFree(s->carry[i]);
    }
    delete s;
}

int foa_shard_create(const int *devices, int n_devices, size_t batch_samples, int narrow_threads, foa_shard **out)
{
    if (!devices || !out) return fail(FOA_E_INVALID, "NULL argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(FOA_E_INVALID, "n_devices must lie in [1, 64]");
    if (batch_samples < 4096 || batch_samples > ((size_t)1 << 28)) return fail(FOA_E_INVALID, "batch_samples must lie in [4096, 2^28]");
    if (narrow_threads < 0 || narrow_threads > 64) return fail(FOA_E_INVALID, "narrow_threads must lie in [0, 64]");
    foa_shard *s = new foa_shard();
    int rc = FOA_OK;
    for (int i = 0; i < n_devices && !rc; i++) {
        foa_rx *h = nullptr;
        rc = foa_rx_create(&h, devices[i]);
        if (!rc) s->rx.push_back(h);
    }
    // page-locked memory every device reads from: portable, so that it is registered with all of them
    for (int i = 0; i < foa::kStreamBufs && !rc; i++) {
        if (hipHostMalloc((void **)&s->staging[i], batch_samples * 8, hipHostMallocPortable) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc of a %zu-byte staging buffer failed", batch_samples * 8);
        if (!rc && hipHostMalloc((void **)&s->carry[i], (size_t)foa::kStreamCarry * 8, hipHostMallocPortable) != hipSuccess) rc = fail(FOA_E_NOMEM, "hipHostMalloc of a carry buffer failed");
    }
    for (size_t i = 0; i < s->rx.size() && !rc; i++) {
        ShardDev *d = new ShardDev();
        d->err_m = &s->err_m; d->err_text = &s->err_text;
        s->devs.push_back(d);
        rc = d->init(s->rx[i], (int64_t)batch_samples);      // (the handles are the shard's own and never leave it: nothing else can call into them)
    }
    if (rc) { foa_shard_destroy(s); return rc; }
    s->be = new ShardBe(s->devs, (int64_t)batch_samples, foa::kStreamCarry, foa::kStreamLongest, s->staging, s->carry);
    s->core = new foa::StreamCore<ShardBe>(s->be, (int64_t)batch_samples, narrow_threads, ShardBe::kSlots);
    *out = s;
    return FOA_OK;
}

int foa_shard_devices(const foa_shard *s) { return s ? (int)s->rx.size() : fail(FOA_E_INVALID, "NULL argument"); }

int foa_shard_push_f32(foa_shard *s, const float *iq, size_t n_samples)
{
    if (!s || (n_samples && !iq)) return fail(FOA_E_INVALID, "NULL argument");
    const int rc = s->core->push(iq, n_samples, nullptr, nullptr);
    return rc ? shard_fail(s, rc) : FOA_OK;
}
int foa_shard_push_f64(foa_shard *s, const double *iq, size_t n_samples)
{
    if (!s || (n_samples && !iq)) return fail(FOA_E_INVALID, "NULL argument");
    const int rc = s->core->push(iq, n_samples, nullptr, nullptr);
    return rc ? shard_fail(s, rc) : FOA_OK;
}
int foa_shard_push_f64_owned(foa_shard *s, const double *iq, size_t n_samples, void (*release)(void *), void *ctx)
{
    if (!s || (n_samples && !iq) || !release) { if (release) release(ctx); return fail(FOA_E_INVALID, "NULL argument"); }
    const int rc = s->core->push(iq, n_samples, release, ctx);
    return rc ? shard_fail(s, rc) : FOA_OK;
}
int foa_shard_flush(foa_shard *s)
{
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    const int rc = s->core->flush();
    return rc ? shard_fail(s, rc) : FOA_OK;
}
int foa_shard_ready(foa_shard *s, int wait, size_t *n_payloads, size_t *n_bytes)
{
    if (!s || !n_payloads || !n_bytes) return fail(FOA_E_INVALID, "NULL argument");
    *n_payloads = 0; *n_bytes = 0;
    if (!s->have_ready) {
        s->ready = foa::StreamReady();
        const int rc = s->core->take(wait != 0, &s->ready);
        if (rc < 0) return shard_fail(s, rc);
        if (rc == 0) return 0;
        s->have_ready = true;
    }
    *n_payloads = s->ready.len.size();
    *n_bytes = s->ready.bytes.size();
    return 1;
}
int foa_shard_take(foa_shard *s, uint8_t *payloads, uint32_t *lengths)
{
    if (!s) return fail(FOA_E_INVALID, "NULL argument");
    if (!s->have_ready) return fail(FOA_E_STATE, "foa_shard_take without a batch reported by foa_shard_ready");
    const foa::StreamReady &r = s->ready;
    if (!r.len.empty() && (!payloads || !lengths)) return fail(FOA_E_INVALID, "NULL argument");
    if (!r.bytes.empty()) memcpy(payloads, r.bytes.data(), r.bytes.size());
    if (!r.len.empty()) memcpy(lengths, r.len.data(), r.len.size() * sizeof(uint32_t));
    s->have_ready = false;
    return FOA_OK;
}
int foa_shard_stats(const foa_shard *s, uint64_t out[8], uint64_t *per_device_alignments, int n_devices)
{
    if (!s || !out) return fail(FOA_E_INVALID, "NULL argument");
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (size_t d = 0; d < s->devs.size(); d++) {
        for (int i = 0; i < 5; i++) out[i] += s->devs[d]->status_count[i].load();
        out[5] += s->devs[d]->alignments.load();
        if (per_device_alignments && (int)d < n_devices) per_device_alignments[d] = s->devs[d]->alignments.load();
    }
    out[6] = (uint64_t)s->core->batches_closed(); out[7] = (uint64_t)s->core->pushed();
    return FOA_OK;
}

}  // extern "C"
