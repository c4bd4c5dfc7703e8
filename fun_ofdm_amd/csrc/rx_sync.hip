// rx_sync.hip -- pre-sync: frame_detector + timing_sync on the device over a resident stream (foa_rx_sync_dev*),
// and their streaming host restatement (foa_sync_*, sync_host.h).
#include <algorithm>

#include "rx_handle.h"
#include "sync_kernels.h"

using namespace foa;

int foa::upload_tables_sync(const DeviceTables &t)
{
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &t, sizeof t));
    return FOA_OK;
}

void foa::launch_stream_range(hipStream_t st, foa_frame_desc *descs, const int32_t *sy_n, int32_t cap, int64_t start_abs, int64_t hz_abs, const StreamState *state,
                              int32_t *range)
{
    hipLaunchKernelGGL(k_stream_range, dim3(1), dim3(64), 0, st, descs, sy_n, cap, start_abs, hz_abs, state, range);
}

void foa::launch_stream_fill(hipStream_t st, float *dst, const float *carry_src, int64_t carry, const float *host_src, int64_t n_new)
{
    const int64_t n = carry + n_new;
    if (n <= 0) return;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 1023) / 1024, 1024);
    hipLaunchKernelGGL(k_stream_fill, dim3(blocks), dim3(256), 0, st, (float2 *)dst, (const float2 *)carry_src, carry, (const float2 *)host_src, n_new);
}

int foa::sync_dev_issue(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap, int32_t *ccap_out, int64_t origin, bool behind_walk)
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
    // (nothing else orders this stage behind the decode calls in flight, whose header and data-symbol kernels may still be reading the
    // descriptor buffers a caller reuses from round to round: wait for the front end of every one that reads what this stage is about to
    // write -- and of no other.  A caller that alternates two descriptor sets has the pre-sync of batch k + 1 running while batch k's front
    // end still reads the other set: round 5 waited for the LATEST call's front end whatever it read, which put the pre-sync behind the
    // data-symbol kernel and the next front end behind the pre-sync -- one chain per step, 0.2 ms longer than the forward pass's)
    if (st != rx->stream) {
        const uint8_t *w0 = (const uint8_t *)d_descs, *w1 = w0 + cap * sizeof(foa_frame_desc), *e0 = (const uint8_t *)d_ends, *e1 = e0 + cap * sizeof(int64_t);
        WorkSet *w = rx->w;
        for (int i = 0; i < kSets - 1 && w && w->used && w->piped; i++, w = w->before) {
            const uint8_t *r0 = (const uint8_t *)w->in_descs, *r1 = r0 + w->in_count * sizeof(foa_frame_desc);
            const uint8_t *q0 = (const uint8_t *)w->in_ends, *q1 = q0 + w->in_count * sizeof(int64_t);
            if ((r0 < w1 && w0 < r1) || (q0 < e1 && e0 < q1)) HIP_TRY(hipStreamWaitEvent(st, w->ev[3], 0));
        }
    }
    // The flag kernel streams the whole capture through HBM, and so does the chain-back walk of the decode call two back, which is running
    // about now and sits on the loop that sets the step (forward pass k, walk k, front end k + 2 on one lane): side by side they share the
    // memory pipe and the walk -- the one somebody waits for -- takes longer.  Behind the walk the flags run under the second half of a
    // forward pass, when nothing else wants HBM: BASELINE config 2 with the pre-sync in the loop 1.16 -> 1.08 ms per step
    // (profiles/r06_ab_presync_behind_walk.txt).  (The stream engines' batches are small and latency-bound: they do not wait.)
    if (behind_walk && st != rx->stream && rx->last_walk_done) HIP_TRY(hipStreamWaitEvent(st, rx->last_walk_done, 0));
    if ((rc = wait_after(rx, st))) return rc;                          // foa_rx_after: the caller's producers, on the device
    const float2 *iq = (const float2 *)d_iq;
    hipLaunchKernelGGL(k_sync_flags, dim3((unsigned)((n + kFlagSamples - 1) / kFlagSamples)), dim3(64), 0, st, iq, n, rx->sy_flags.p, n_words);
    hipLaunchKernelGGL(k_sync_sts_end, dim3(n_blocks), dim3(kSyncBlockWords), 0, st, rx->sy_flags.p, n_words, 0, rx->sy_cnt.p, rx->sy_off.p, rx->sy_x.p, ccap);
    hipLaunchKernelGGL(k_sync_scan, dim3(1), dim3(64), 0, st, rx->sy_cnt.p, n_blocks, rx->sy_off.p, rx->sy_n.p);
    hipLaunchKernelGGL(k_sync_sts_end, dim3(n_blocks), dim3(kSyncBlockWords), 0, st, rx->sy_flags.p, n_words, 1, rx->sy_cnt.p, rx->sy_off.p, rx->sy_x.p, ccap);
    // one wave per candidate; the count stays on the device: fixed grids stride over it (k_sync_finish reports overflow)
    const int lts_grid = (int)std::min<int64_t>(ccap, 16384);
    hipLaunchKernelGGL(k_sync_lts, dim3(lts_grid), dim3(64), 0, st, iq, n, rx->sy_x.p, rx->sy_n.p, ccap, rx->sy_cand.p, origin, rx->sync_call);
    const int kb = (ccap + 255) / 256;       // blocks of the keep / emit stage; their counts reuse the STS_END stage's count buffers
    hipLaunchKernelGGL(k_sync_keep, dim3(kb), dim3(256), 0, st, rx->sy_cand.p, rx->sy_n.p, ccap, rx->sy_keep.p, rx->sy_cnt.p);
    hipLaunchKernelGGL(k_sync_scan, dim3(1), dim3(64), 0, st, rx->sy_cnt.p, kb, rx->sy_off.p, rx->sy_n.p + 3);
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
    if (n_samples == 0 || cap == 0) { rx->sy_cap = 0; rx->sy_open = true; rx->after.clear(); return FOA_OK; }          // (nothing queued; _end reports 0)
    { int rc = sync_dev_issue(rx, d_iq, n_samples, d_descs, d_ends, cap, &rx->sy_ccap, rx->sync_origin, true); if (rc) return rc; }
    hipStream_t st = side_stream(rx);
    HIP_TRY(hipMemcpyAsync(rx->sy_pin, rx->sy_n.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(rx->sy_done, st));
    HIP_TRY(hipGetLastError());
    rx->sy_used = true;
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

}  // extern "C"

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

}  // extern "C"
