// rx_decode.hip -- the batch decode call of include/fun_ofdm_amd.h (foa_rx_decode_frames_dev / _host, foa_rx_submit_host /
// foa_rx_collect) and the kernels it is made of: k_header, k_scan_*, k_data_symbols_q4, k_viterbi_fwd3, k_tb_walk, k_tb_finish.
#include <algorithm>

#include "rx_handle.h"
#include "frontend_kernels.h"
#include "frontend_q4.h"
#include "viterbi_tb.h"

using namespace foa;

int foa::upload_tables_decode(const DeviceTables &t)
{
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &t, sizeof t));
    return FOA_OK;
}

void foa::launch_fwd3(hipStream_t st, const FrameInfo *info, int nf, const uint16_t *sp, uint64_t *dec)
{
    // A call of up to 2 048 alignments -- two four-wave workgroups per CU at most -- takes ONE frame per wave: half the renormalisation events per
    // wave, and a step whose test is a compare and a branch (viterbi_fwd.h).  Against two frames per wave, 1 024-byte frames at 54 Mbps
    // (profiles/r06_forward_one_frame_crossover.txt): alone 0.47 against 0.59 ms at 1 200 - 2 000 frames, pipelined calls 9 % / 4 % / 0 % faster
    // at 1 200 / 1 500 / 2 000; from 2 600 frames on two frames per wave win the pipelined loop by 6-9 % (alone still 7 % behind).
    // (Round 5, before the one-frame form had its own test and read its increments ahead: the crossover was at 1 024.)
    if (nf < kSingleBelow) hipLaunchKernelGGL(k_viterbi_fwd3<1>, dim3((nf + kFwdWaves - 1) / kFwdWaves), dim3(64 * kFwdWaves), 0, st, info, nf, sp, dec);
    else hipLaunchKernelGGL(k_viterbi_fwd3<2>, dim3(((nf + 1) / 2 + kFwdWaves - 1) / kFwdWaves), dim3(64 * kFwdWaves), 0, st, info, nf, sp, dec);
}

void foa::launch_finish3(hipStream_t st, hipStream_t st_fin, const FrameInfo *info, int nf, const uint64_t *dec, uint32_t *decoded, const int32_t *seg2frame,
                         const int64_t *totals, uint16_t *tb_state, size_t max_segs, int S, int L, uint8_t *psdu, size_t slot_bytes,
                         foa_frame_result *results, hipEvent_t walk_done)
{
    hipLaunchKernelGGL(k_tb_walk, dim3((unsigned)((max_segs + 63) / 64)), dim3(64), 0, st, info, seg2frame, totals, dec, decoded, tb_state, S, L);
    if (walk_done) (void)hipEventRecord(walk_done, st);
    if (st_fin != st) (void)hipStreamWaitEvent(st_fin, walk_done, 0);
    hipLaunchKernelGGL(k_tb_finish, dim3((nf + 63) / 64), dim3(64), 0, st_fin, info, nf, dec, decoded, tb_state, S, psdu, slot_bytes, results);
}

void foa::launch_stream_resolve(hipStream_t st, const float *d_iq, int64_t n_eff, const foa_frame_desc *descs, const int64_t *ends, const int32_t *range,
                                const int32_t *sy_n, int64_t start_abs, int64_t hz_abs, bool final, StreamState *state, FrameInfo *info, double2 *hinv, int32_t *sel,
                                unsigned grid)
{
    hipLaunchKernelGGL(k_header_range, dim3(grid), dim3(64), 0, st, (const float2 *)d_iq, descs, ends, n_eff, range, info, hinv);
    hipLaunchKernelGGL(k_stream_resolve, dim3(1), dim3(64), 0, st, info, descs, range, sy_n, start_abs, hz_abs, final ? 1 : 0, state, sel);
}

static hipStream_t lane_stream(foa_rx *rx, int lane) { return lane == 0 ? rx->stream : lane == 1 ? rx->stream4 : lane == 2 ? rx->stream5 : rx->stream6; }

int foa::inputs_queued(foa_rx *rx, hipStream_t cs)
{
    if (cs == rx->stream) return FOA_OK;
    HIP_TRY(hipEventRecord(rx->in_ready, cs));
    rx->in_wait = true;
    return FOA_OK;
}

int foa::workspace(foa_rx *rx, size_t n_samples, size_t n_frames)
{
    // (room for an eighth more frames than asked for, in steps of 1024: a stream's batches differ by a few frames, and a buffer that
    // grows by one element costs a hipFree, which waits for the whole device)
    n_frames = ((n_frames + (n_frames >> 3) + 1024) & ~(size_t)1023);
    size_t sym_cap = n_samples / 80 + 4;
    // Per-step buffers (soft pairs 2 B, decisions 8 B, decoded bits 1/4 B per step of capacity): a stream of n_samples samples holds at most
    // n_samples / 80 data symbols of at most max_dbps trellis steps each -- 216 (54 Mbps) unless the caller has promised less (option
    // "max_dbps": a 6 Mbps capture needs a ninth of it, and the room decides how many frames one call can take)
    size_t dec_cap = (size_t)rx->max_dbps * sym_cap + 192 * (n_frames + 1);
    int rc;
    if ((rc = rx->w->info.ensure(n_frames + 1)) || (rc = rx->w->hinv.ensure((n_frames + 1) * 64)) || (rc = rx->w->sym2frame.ensure(sym_cap)) ||
        (rc = rx->w->spec.ensure(sym_cap)) ||
        (rc = rx->w->dec.ensure(dec_cap)) || (rc = rx->w->sp.ensure(dec_cap)) || (rc = rx->w->decoded.ensure(decoded_words_for(dec_cap))) ||
        (rc = rx->w->totals.ensure(8 + 4 * ((n_frames + kScanBlock - 1) / kScanBlock + 1))))
        return rc;
    if (rx->record_eq && ((rc = rx->w->eq_sig.ensure((n_frames + 1) * 48)) || (rc = rx->w->eq_data.ensure(sym_cap * 48)))) return rc;
    // chain-back segments: every frame has at most dec_words/segment + 1 of them
    const size_t seg_cap = dec_cap / 96 + n_frames + 64;
    if ((rc = rx->w->seg2frame.ensure(seg_cap)) || (rc = rx->w->tb_state.ensure(seg_cap))) return rc;
    // capacities handed to the scan are those of the buffers actually allocated
    rx->w->sym_cap = std::min(rx->w->sym2frame.n, rx->w->spec.n); rx->w->dec_cap = rx->w->dec.n < rx->w->sp.n ? rx->w->dec.n : rx->w->sp.n;
    if (rx->record_eq && rx->w->eq_data.n / 48 < rx->w->sym_cap) rx->w->sym_cap = rx->w->eq_data.n / 48;
    return FOA_OK;
}

// queue the deferred chain-back + finish; after_front_end: the event of the call it should run under (or null: now)
int foa::flush_pending(foa_rx *rx, hipEvent_t after_front_end)
{
    foa_rx::Pending &p = rx->pending;
    if (!p.valid) return FOA_OK;
    // (Batches of a live stream that arrive 0.4 ms apart and more leave the lanes mostly idle: their finish and copy back follow the walk on the
    // lane itself and save the hop to the stitch stream -- ~35 us of a batch's way, the tail unchanged.  At 4 Ki samples, a call every 0.2 ms, the
    // four lanes are 3/4 occupied and holding one 60 us longer costs the tail more than the hop: profiles/r06_exp_finish_in_line.txt.)
    hipStream_t sb = (rx->finish_in_line && p.deep && p.nf <= 256) ? p.lane : rx->stream2;
    // the walk follows its forward pass on the call's own lane -- no event between them -- and the next call of that lane
    // queues its front end behind it; the stitch/CRC kernel, which nothing on the loop waits for, goes to the second stream
    if (after_front_end) HIP_TRY(hipStreamWaitEvent(p.lane, after_front_end, 0));
    if (p.w->have_timing) HIP_TRY(hipEventRecord(p.w->ev[6], p.lane));
    launch_finish3(p.lane, sb, p.w->info.p, p.nf, p.w->dec.p, p.w->decoded.p, p.w->seg2frame.p, p.w->totals.p, p.w->tb_state.p, p.max_segs, p.S, p.L,
                   p.psdu, p.slot_bytes, p.results, p.w->walk_done);
    if (p.w->have_timing) HIP_TRY(hipEventRecord(p.w->ev[4], sb));
    HIP_TRY(hipEventRecord(p.w->done, sb));
    // (the one-shot pre-sync goes behind this walk -- rx_sync.hip -- when two loops are in flight: the walk is then about to run.  With
    // three or four loops it waits for a forward pass that has two or three others in front of it, and a pre-sync behind it would hold up
    // the host, which needs the count to queue the next call: BASELINE config 5, 55 -> 52 Gsample/s)
    rx->last_walk_done = p.deep ? nullptr : p.w->walk_done;
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

int foa::drain(foa_rx *rx)
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

// The job behind a ticket once its outputs are in its page-locked mirror: 1 = *out is complete (the caller reads job->pin and then
// clears job->busy), 0 = not yet (wait == false), < 0 = error.  Shared by foa_rx_collect and the stream engines.
int foa::job_ready(foa_rx *rx, uint64_t ticket, bool wait, HostJob **out)
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

static int decode_frames_any(foa_rx *rx, const void *d_iq, bool f64, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                             size_t n_frames, size_t n_context, size_t n_lead, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results);

extern "C" {

int foa_rx_decode_frames_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                             size_t n_frames, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results)
{
    return foa_rx_decode_frames_ctx_dev(rx, d_iq, n_samples, d_descs, d_ends, n_frames, 0, d_psdu, slot_bytes, d_results);
}

int foa_rx_decode_frames_ctx_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                                 size_t n_frames, size_t n_context, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results)
{
    return decode_frames_any(rx, d_iq, false, n_samples, d_descs, d_ends, n_frames, n_context, 0, d_psdu, slot_bytes, d_results);
}

int foa_rx_decode_frames_lead_ctx_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                                      size_t n_lead, size_t n_frames, size_t n_context, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results)
{
    if (n_lead > 0x7FFFFFF0u) return fail(FOA_E_INVALID, "too many frames");
    if ((n_lead || n_frames) && (!d_descs || !d_ends)) return fail(FOA_E_INVALID, "NULL device pointer");
    return decode_frames_any(rx, d_iq, false, n_samples, d_descs + n_lead, d_ends + n_lead, n_frames, n_context, n_lead, d_psdu, slot_bytes, d_results);
}

}  // extern "C"

// f64: d_iq holds complex<double> samples that timing_sync has rotated already (the fused stage block of blocks.hpp); else complex<float>
static int decode_frames_any(foa_rx *rx, const void *d_iq, bool f64, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                             size_t n_frames, size_t n_context, size_t n_lead, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames == 0) { rx->last_frames = 0; rx->after.clear(); return FOA_OK; }      // (nothing is queued: nothing waits)
    if (!d_iq || !d_descs || !d_ends || !d_psdu || !d_results) return fail(FOA_E_INVALID, "NULL device pointer");
    if (n_frames > 0x7FFFFFF0u || n_context > 0x7FFFFFF0u - n_frames) return fail(FOA_E_INVALID, "too many frames");
    HIP_TRY(enter_device(rx->device));
    // Work sets take turns when calls are pipelined: this call's front end and forward pass may then start while the previous call's
    // chain-back is still reading another set.
    const bool piped = rx->pipeline;
    if (!piped) { int rc0 = flush_pending(rx, nullptr); if (rc0) return rc0; }
    rx->prev = rx->w;
    if (piped) rx->w = &rx->sets[(int)((rx->w - rx->sets) + 1) % kSets];     // in use: front end k+1 | forward pass k | finish k-1
    if (rx->w->used) {                                                 // the call that last used this set is complete
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipEventSynchronize(rx->w->done));
        rx->ns_wait_set += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    int rc = workspace(rx, n_samples, n_frames + n_context);
    if (rc) return rc;
    // Pipelined calls take turns on two (or four) streams ("lanes"): a call's front end, forward pass and chain-back walk run on ONE
    // stream, the front end behind the walk of the call `depth` back (queued there when the call before this one was made).  The loop
    // that sets the step -- forward pass k, walk k, header, scan and data symbols of call k+2, forward pass k+2 -- is then one in-order
    // stream with no event packet in it, and nothing orders one lane behind the other (each call has its own work set), so a forward
    // pass starts the moment its front end is done, into the tail of the one before.
    // (big grids of BPSK-only captures -- the caller has said so: option "max_dbps" <= 36 -- spend a third of their time in the data-symbol
    // kernel, which shares the vector pipes with the forward pass: three loops keep a forward pass on the machine while two front ends
    // run, + 8-11 % at 6 / 9 Mbps; from QPSK on it is within 3 % either way and config 2 loses 12 % to a third loop: profiles/r06_depth_by_frames.txt)
    const int depth = rx->depth > 0 ? rx->depth : (n_frames < (size_t)kDeepBelow ? std::min(4, rx->max_depth) : (rx->max_dbps <= 36 ? std::min(3, rx->max_depth) : 2));
    hipStream_t st = piped ? lane_stream(rx, (int)(rx->n_calls++ % (unsigned)depth)) : rx->stream;
    if (!piped && rx->prev->used && rx->prev != rx->w) HIP_TRY(hipStreamWaitEvent(st, rx->prev->done, 0));
    const int nf = (int)n_frames, n_total = (int)(n_frames + n_context);      // context alignments: header kernel only
    const float2 *iq = (const float2 *)d_iq;
    const double2 *iq64 = (const double2 *)d_iq;
    double2 *eq_sig = rx->record_eq ? rx->w->eq_sig.p : nullptr, *eq_data = rx->record_eq ? rx->w->eq_data.p : nullptr;

    if (rx->in_wait) { HIP_TRY(hipStreamWaitEvent(st, rx->in_ready, 0)); rx->in_wait = false; }
    if ((rc = wait_after(rx, st))) return rc;                          // foa_rx_after: the caller's producers, on the device
    const bool tm = rx->timing;                                       // (the per-kernel timing events: seven runtime calls per decode call, off while a stream engine drives the handle)
    if (tm) HIP_TRY(hipEventRecord(rx->w->ev[0], st));
    if (f64) hipLaunchKernelGGL(k_header<double2>, dim3(n_total), dim3(64), 0, st, iq64, d_descs, d_ends, (int64_t)n_samples, (int)n_lead, n_total, rx->w->info.p, rx->w->hinv.p, eq_sig);
    else hipLaunchKernelGGL(k_header<float2>, dim3(n_total), dim3(64), 0, st, iq, d_descs, d_ends, (int64_t)n_samples, (int)n_lead, n_total, rx->w->info.p, rx->w->hinv.p, eq_sig);
    if (tm) HIP_TRY(hipEventRecord(rx->w->ev[1], st));
    // segments of this call: at most (total data steps)/S + one per frame; lanes beyond the real total idle
    // (960 data steps per lane keep the run-in at a tenth of the work when a call fills the machine.  A call of a few frames has lanes to spare
    // and waits for ONE lane's walk: 192 + 96 steps instead of 960 + 96, 45 -> ~15 us; FOA_TB_SMALL_SEG=0 in the environment: off, for A/B)
    static const int small_seg = [] { const char *e = getenv("FOA_TB_SMALL_SEG"); const int v = e ? atoi(e) : kSmallCallSegment; return (v >= 96 && v % 96 == 0) ? v : 0; }();
    const int seg = (!rx->tb_segment_set && small_seg && nf <= 256) ? small_seg : rx->tb_segment;
    const size_t max_segs = std::min(rx->w->seg2frame.n, rx->w->dec_cap / (size_t)seg + n_frames + 1);
    const int n_sb = (nf + kScanBlock - 1) / kScanBlock;
    int64_t *blk = rx->w->totals.p + 8;
    if (n_sb == 1) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(kScanBlock), 0, st, rx->w->info.p, nf, (int64_t)rx->w->sym_cap, (int64_t)rx->w->dec_cap, seg,
                           (int64_t)rx->w->seg2frame.n, rx->w->totals.p, rx->w->sym2frame.p, rx->w->seg2frame.p, rx->w->spec.p);
    } else {
        hipLaunchKernelGGL(k_scan_sums, dim3(n_sb), dim3(kScanBlock), 0, st, rx->w->info.p, nf, seg, blk);
        if (n_sb <= 4096) hipLaunchKernelGGL(k_scan_blocks_w, dim3(1), dim3(64), 0, st, blk, n_sb, (int64_t)rx->w->sym_cap, rx->w->totals.p);
        else hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, blk, n_sb, (int64_t)rx->w->sym_cap, rx->w->totals.p);
        hipLaunchKernelGGL(k_scan_apply, dim3(n_sb), dim3(kScanBlock), 0, st, rx->w->info.p, nf, (int64_t)rx->w->sym_cap,
                           (int64_t)rx->w->dec_cap, seg, (int64_t)rx->w->seg2frame.n, blk, rx->w->sym2frame.p, rx->w->seg2frame.p, rx->w->spec.p);
    }
    if (tm) HIP_TRY(hipEventRecord(rx->w->ev[2], st));
    // upper bound on data symbols in n_samples samples; waves beyond the real total exit at once
    const size_t max_sym = rx->w->sym_cap;
    // (persistent workgroups: as many as the device holds at once walk over the groups of 64 symbols, frontend_q4.h)
    if (rx->q4_resident == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_data_symbols_q4<float2>, 64 * kQ4Waves, 0));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, rx->device));
        rx->q4_resident = std::max(1, per_cu) * std::max(1, cus);
    }
    const dim3 q4_grid((unsigned)std::min<size_t>((max_sym + 16 * kQ4Waves - 1) / (16 * kQ4Waves), (size_t)rx->q4_resident));
    if (f64) hipLaunchKernelGGL(k_data_symbols_q4<double2>, q4_grid, dim3(64 * kQ4Waves), 0, st, iq64, d_descs, rx->w->info.p, rx->w->sym2frame.p, rx->w->spec.p,
                                rx->w->totals.p, rx->w->hinv.p, rx->w->sp.p, eq_data);
    else hipLaunchKernelGGL(k_data_symbols_q4<float2>, q4_grid, dim3(64 * kQ4Waves), 0, st, iq, d_descs, rx->w->info.p, rx->w->sym2frame.p, rx->w->spec.p,
                            rx->w->totals.p, rx->w->hinv.p, rx->w->sp.p, eq_data);
    HIP_TRY(hipEventRecord(rx->w->ev[3], st));
    if (piped) {
        // the previous call's chain-back + finish goes under this call's forward pass
        if ((rc = flush_pending(rx, rx->w->ev[3]))) return rc;
        if (tm) HIP_TRY(hipEventRecord(rx->w->ev[7], st));       // start of the forward pass
        launch_fwd3(st, rx->w->info.p, nf, rx->w->sp.p, rx->w->dec.p);
        if (tm) HIP_TRY(hipEventRecord(rx->w->ev[5], st));
        foa_rx::Pending &p = rx->pending;
        p.valid = true; p.w = rx->w; p.nf = nf; p.S = seg; p.L = rx->tb_overlap; p.max_segs = max_segs; p.slot_bytes = slot_bytes;
        p.psdu = d_psdu; p.results = d_results; p.job = rx->attach_job; p.lane = st; p.deep = depth > 2;
    } else {
        launch_fwd3(st, rx->w->info.p, nf, rx->w->sp.p, rx->w->dec.p);
        if (tm) HIP_TRY(hipEventRecord(rx->w->ev[5], st));
        launch_finish3(st, st, rx->w->info.p, nf, rx->w->dec.p, rx->w->decoded.p, rx->w->seg2frame.p, rx->w->totals.p, rx->w->tb_state.p, max_segs, seg,
                       rx->tb_overlap, d_psdu, slot_bytes, d_results);
        if (tm) HIP_TRY(hipEventRecord(rx->w->ev[6], st));             // (not separable from the forward pass on one stream)
        if (tm) HIP_TRY(hipEventRecord(rx->w->ev[4], st));
        HIP_TRY(hipEventRecord(rx->w->done, st));
    }
    HIP_TRY(hipGetLastError());
    rx->w->used = true;
    rx->w->in_descs = d_descs - n_lead; rx->w->in_ends = d_ends - n_lead; rx->w->in_count = n_lead + n_frames + n_context;
    rx->w->have_timing = tm;
    rx->w->piped = piped;
    rx->w->before = piped ? rx->prev : nullptr;
    rx->last_frames = n_frames;
    return FOA_OK;
}

static int decode_frames_host_any(foa_rx *rx, const void *iq, bool f64, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends,
                                  size_t n_frames, uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames == 0) return FOA_OK;
    if (!iq || !descs || !ends || !psdu || !results) return fail(FOA_E_INVALID, "NULL pointer");
    HIP_TRY(enter_device(rx->device));
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t sample_bytes = f64 ? 16 : 8;
    size_t o_iq = 0, o_desc = o_iq + up(n_samples * sample_bytes), o_end = o_desc + up(n_frames * sizeof(foa_frame_desc)),
           o_psdu = o_end + up(n_frames * 8), o_res = o_psdu + up(n_frames * slot_bytes), total = o_res + up(n_frames * sizeof(foa_frame_result));
    int rc = rx->scratch.ensure(total);
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    hipStream_t st = side_stream(rx);
    if ((rc = wait_after(rx, st))) return rc;
    HIP_TRY(hipMemcpyAsync(b + o_iq, iq, n_samples * sample_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_desc, descs, n_frames * sizeof(foa_frame_desc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_end, ends, n_frames * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b + o_psdu, 0, n_frames * slot_bytes, st));
    if ((rc = inputs_queued(rx, st))) return rc;
    rc = decode_frames_any(rx, b + o_iq, f64, n_samples, (const foa_frame_desc *)(b + o_desc), (const int64_t *)(b + o_end),
                           n_frames, 0, 0, b + o_psdu, slot_bytes, (foa_frame_result *)(b + o_res));
    if (rc) return rc;
    if ((rc = flush_pending(rx, nullptr))) return rc;                   // the finish runs on the second stream
    HIP_TRY(hipStreamWaitEvent(st, rx->w->done, 0));
    HIP_TRY(hipMemcpyAsync(psdu, b + o_psdu, n_frames * slot_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(results, b + o_res, n_frames * sizeof(foa_frame_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FOA_OK;
}

extern "C" {

int foa_rx_decode_frames_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends,
                              size_t n_frames, uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    return decode_frames_host_any(rx, iq, false, n_samples, descs, ends, n_frames, psdu, slot_bytes, results);
}

int foa_rx_decode_frames_f64_host(foa_rx *rx, const double *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends,
                                  size_t n_frames, uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    return decode_frames_host_any(rx, iq, true, n_samples, descs, ends, n_frames, psdu, slot_bytes, results);
}

int foa_rx_submit_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                       size_t slot_bytes, uint64_t *ticket)
{
    return foa_rx_submit_host_ctx(rx, iq, n_samples, descs, ends, n_frames, 0, slot_bytes, ticket);
}

int foa_rx_submit_host_ctx(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                           size_t n_context, size_t slot_bytes, uint64_t *ticket)
{
    if (!rx || !ticket) return fail(FOA_E_INVALID, "NULL argument");
    if (n_frames == 0 || !iq || !descs || !ends) return fail(FOA_E_INVALID, "empty call or NULL pointer");
    HIP_TRY(enter_device(rx->device));
    HostJob *job = nullptr;
    int n_busy = 0;
    for (auto &j : rx->jobs) { if (j.busy) n_busy++; else if (!job) job = &j; }
    // (the documented limit of this entry point; the slots beyond it are the stream engines', which keep more small batches in flight)
    if (!job || n_busy >= kMaxHostCalls) return fail(FOA_E_STATE, "%d calls are in flight: collect the oldest first", kMaxHostCalls);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n_all = n_frames + n_context;
    const size_t o_iq = 0, o_desc = o_iq + up(n_samples * 8), o_end = o_desc + up(n_all * sizeof(foa_frame_desc)), o_psdu = o_end + up(n_all * 8),
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
    memcpy(job->pin + o_desc, descs, n_all * sizeof(foa_frame_desc));
    memcpy(job->pin + o_end, ends, n_all * 8);
    const bool piped = rx->pipeline;
    hipStream_t st = side_stream(rx);
    uint8_t *b = job->dev.p;
    if ((rc = wait_after(rx, st))) return rc;
    HIP_TRY(hipMemcpyAsync(b, job->pin, o_psdu, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b + o_psdu, 0, n_frames * slot_bytes, st));
    if ((rc = inputs_queued(rx, st))) return rc;
    job->total = total; job->o_psdu = o_psdu; job->o_res = o_res; job->n_frames = n_frames; job->slot_bytes = slot_bytes; job->copy_queued = false;
    rx->attach_job = piped ? job : nullptr;
    rc = foa_rx_decode_frames_ctx_dev(rx, (const float *)(b + o_iq), n_samples, (const foa_frame_desc *)(b + o_desc), (const int64_t *)(b + o_end), n_frames,
                                      n_context, b + o_psdu, slot_bytes, (foa_frame_result *)(b + o_res));
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

}  // extern "C"
