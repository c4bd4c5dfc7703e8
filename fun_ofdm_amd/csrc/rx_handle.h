// rx_handle.h -- what the translation units of libfun_ofdm_amd.so share on the HOST side: the receiver handle (foa_rx), its work
// sets and job slots, and the functions one unit offers the others.
//
//   rx_handle.hip   handle life cycle, options, constant tables, timings / taps / probe          (no kernels of the receive path)
//   rx_decode.hip   the batch decode call: k_header, k_scan_*, k_data_symbols_q4, k_viterbi_fwd3, k_tb_walk, k_tb_finish
//   rx_sync.hip     pre-sync: frame_detector + timing_sync on the device (and their host restatement, foa_sync_*)
//   rx_stage.hip    one entry point per replaced fun::block (fft, channel estimate, equalise, phase track, header, data)
//   rx_tx.hip       frame_builder and the synthetic channel on the device
//   rx_stream.hip   foa_stream_* / foa_shard_*: process_samples() on one or several devices (host code only)
// Each unit with kernels keeps its own __constant__ copy of the tables (device_math.h) and uploads it through upload_tables_*.
#pragma once

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "foa_common.h"

namespace foa {

// per-thread error text behind foa_last_error(); returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const std::string &last_error_text();

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return foa::fail(e_ == hipErrorOutOfMemory ? FOA_E_NOMEM : FOA_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Every entry point that queues work starts here: select the handle's device and FORGET whatever error an earlier HIP call on this
// thread left behind -- a failed call of ours that was already reported, a polled hipEventQuery, or another library's probing (PyTorch
// asks about peers a one-GPU box does not have: "invalid device ordinal").  The launch checks (hipGetLastError after the kernels are
// queued) must report THIS call's errors only.
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

struct SyncCand;     // sync_kernels.h

// Everything one decode call writes between its header kernel and its finish kernel.  Several sets rotate (kSets) so that the
// chain-back of call k can run under the forward pass of call k+1 while call k+2's front end is already filling the next one.
struct WorkSet {
    DevBuf<FrameInfo> info;
    DevBuf<double2> hinv;
    DevBuf<int32_t> sym2frame, seg2frame;
    DevBuf<SpecSym> spec;         // symbols that are not plain windows of their frame's own alignment (k_scan_apply -> k_data_symbols_q4)
    DevBuf<uint16_t> tb_state;
    DevBuf<uint64_t> dec;
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
    const void *in_descs = nullptr, *in_ends = nullptr;      // the descriptor / end buffers the call's front end reads (until ev[3]) ...
    size_t in_count = 0;                                      // ... and how many entries of them
    void release_all()
    {
        info.release(); hinv.release(); sym2frame.release(); seg2frame.release(); spec.release(); tb_state.release(); dec.release(); sp.release();
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
constexpr int kMaxHostCalls = 8;                  // foa_rx_submit_host: calls in flight (include/fun_ofdm_amd.h)
constexpr int kMaxJobs = 12;                      // >= kStreamBufs - 1: a stream of small batches keeps that many decode calls in flight
constexpr int kSets = 6;

}  // namespace foa

struct foa_stream;

struct foa_rx {
    int device = 0;
    hipStream_t stream = nullptr;      // everything when calls run in line; the first lane of pipelined calls
    hipStream_t stream2 = nullptr;     // pipelined path: the stitch / CRC kernel behind a call's chain-back walk
    hipStream_t stream3 = nullptr;     // pipelined path: copies of the host-pointer entry points and the pre-sync stage
    hipStream_t stream4 = nullptr;     // the second lane of pipelined calls (the first is `stream`)
    hipStream_t stream5 = nullptr, stream6 = nullptr;      // third and fourth lane, used for small grids (option "depth")
    int max_depth = 4;                 // lanes that can run side by side (2 if the host cut the runtime down to one hardware queue per priority: foa_rx_create)
    std::string notes;                 // non-fatal remarks about how the handle is set up (foa_rx_notes)
    int64_t sync_origin = 0;           // one-shot device pre-sync: stream index of d_iq[0] (option "sync_origin")
    int64_t stream_longest = 0;        // foa_stream_create: longest frame (samples, preamble to last symbol + 192) the stream will hold; 0 = any frame the format allows
    int depth = 0;                     // how many calls' loops are in flight; 0 = by grid size (2, or 4 below kDeepBelow frames)
    int depth_saved = -1;              // (the stream engine pins its own while a stream is open and restores this)
    unsigned n_calls = 0;              // pipelined decode calls made so far (a call's lane is n_calls mod depth)
    int q4_resident = 0;               // workgroups of the data-symbol kernel the device holds at once (its grid: rx_decode.hip)
    int max_dbps = 216;                // work sets hold this many trellis steps per 80 samples (option "max_dbps": the highest rate the caller's frames carry)
    int tb_segment = 960, tb_overlap = 96;   // chain-back: data steps per segment / run-in steps (multiples of 96)
    bool tb_segment_set = false;             // option "tb_segment" was given: also calls of a few frames use it (otherwise kSmallCallSegment)
    bool finish_in_line = false; // a small pipelined call's finish + copy back follow its walk on the call's own lane (a stream engine of 8 .. 64 Ki-sample batches sets it: flush_pending)
    bool timing = true;          // per-kernel HIP events around every decode call (foa_rx_*_kernel_ms); the stream engines switch them off while they own the handle
    bool pipeline = true;        // the finish of one call overlaps the next calls' front end and forward pass (rotating work sets, several streams)
    bool record_eq = false;
    bool record_soft = true;     // (the soft bytes are the front end's output and always there; the option is accepted for compatibility)
    hipEvent_t in_ready = nullptr;     // inputs copied by a host-pointer entry point are on the device (recorded on the copy stream)
    bool in_wait = false;              // ... and the next decode call's front end has to wait for that
    int64_t sync_call = 4096;    // pre-sync: the reference receiver's call size to decide timing_sync.cpp:99 by (sync_host.h kSyncCallDefault); 0 = as one call
    foa::WorkSet sets[foa::kSets];   // up to depth + 1 are in use at any time; one more keeps the call before them readable (timings)
    foa::WorkSet *w = &sets[0];      // the set of the most recent decode call
    foa::WorkSet *prev = nullptr;    // the set of the call before it (kernel times of a call that is certainly complete)
    // Pipelined path: the chain-back + finish of a call is queued only when the NEXT call has queued its front end, so that it runs
    // under that call's forward pass (memory-bound next to issue-bound) rather than under its latency-bound front end; foa_rx_sync
    // and everything that needs results queue it at once.
    struct Pending {
        bool valid = false;
        hipStream_t lane = nullptr;  // the stream of the call's forward pass, where its walk follows
        bool deep = false;           // more than two loops in flight when the call was made
        foa::WorkSet *w = nullptr;
        int nf = 0, S = 0, L = 0;            // (nf: the alignments that are decoded -- context alignments have no results)
        size_t max_segs = 0, slot_bytes = 0;
        uint8_t *psdu = nullptr;
        foa_frame_result *results = nullptr;
        foa::HostJob *job = nullptr; // submit_host: copy the outputs back once the finish is queued
    } pending;
    hipEvent_t last_walk_done = nullptr;   // behind the most recent chain-back walk queued (the one-shot pre-sync goes behind it: rx_sync.hip)
    foa::HostJob jobs[foa::kMaxJobs];
    uint64_t next_ticket = 1;
    foa::HostJob *attach_job = nullptr;   // set by submit_host around its decode call
    foa::DevBuf<uint8_t> scratch;     // staging for the host-pointer entry points
    foa::DevBuf<uint32_t> sy_flags;   // device pre-sync workspace
    foa::DevBuf<int32_t> sy_cnt, sy_off, sy_keep, sy_n;
    foa::DevBuf<int64_t> sy_x;
    foa::DevBuf<foa::SyncCand> sy_cand;
    size_t last_frames = 0;
    int32_t *sy_pin = nullptr;        // page-locked { STS_END candidates, -, -, alignments found } of the pre-sync begun last (foa_rx_sync_dev_begin)
    hipEvent_t sy_done = nullptr;
    bool sy_open = false;
    int32_t sy_ccap = 0;
    size_t sy_cap = 0;
    // Device-side ordering against the caller's own streams (foa_rx_after / foa_rx_record_consumed / foa_rx_record_done):
    std::vector<hipEvent_t> after;     // events the NEXT call that queues work makes its first stream wait for (one-shot)
    hipStream_t join = nullptr;        // made on first use, high priority: carries event waits and the caller's record, never a kernel
    hipEvent_t tx_done = nullptr;      // behind the most recent foa_tx_* call (they read caller buffers on `stream`)
    bool tx_used = false, sy_used = false;
    foa_stream *open_stream = nullptr;      // the stream engine that owns this handle right now (stream_engine.h), if any
    int64_t ns_wait_set = 0;     // host time spent waiting for a work set to come free (the GPU is more than kSets - 1 calls behind)
};

namespace foa {

// A machine that a call fills (config 2: five forward-pass waves per SIMD) is best served by two calls' loops in flight; a call of a few
// thousand frames leaves most SIMDs one wave or none, its forward pass lasts as long as ONE wave needs for its frames' trellis steps
// whatever the batch, and more loops in flight are what raises the throughput then (1 000 frames x 4 092 bytes at 54 Mbps: 1.43 ms per
// batch with two, 0.94 with four).  The lanes sit on hardware queues of their own (stream priorities: foa_rx_create).
constexpr int kDeepBelow = 4609;                 // frames: up to 2.25 forward-pass waves per SIMD.  (Round 5 drew the line at 2049; measured since: 3 000 frames per call
                                                 // +9 % with four loops, 4 000 mixed-rate alignments -- BASELINE config 5 -- +14 %, 5 000 frames the same
                                                 // either way, 6 000 and more 2-8 % better with two: profiles/r06_depth_by_frames.txt)
constexpr int kSmallCallSegment = 192;           // chain-back segment of a call of up to 256 alignments: a lane walks segment + run-in steps one after the other, and such a call waits for exactly that
constexpr int kSingleBelow = 2049;               // alignments: below this -- up to two four-wave workgroups per CU -- the forward pass takes one frame per wave (launch_fwd3)

inline bool piped(const foa_rx *rx) { return rx->pipeline; }
// Host-pointer entry points copy their inputs (and the pre-sync stage runs) on the third stream when calls are pipelined, off the
// lanes, so that a copy never sits behind a forward pass; the decode call that follows waits for the event.
// the priority level of copy / pre-sync / look-ahead streams (foa_rx_create: hardware queues come in a pool per priority)
inline int high_priority() { int least = 0, greatest = 0; (void)hipDeviceGetStreamPriorityRange(&least, &greatest); return greatest; }
inline hipStream_t side_stream(foa_rx *rx) { return rx->pipeline ? rx->stream3 : rx->stream; }

// foa_rx_after: whatever the caller registered is waited for by `st`, the stream the call's first kernel or copy goes to -- everything else
// of the call is ordered behind that (lane -> walk -> stitch stream; side stream -> in_ready -> lane)
inline int wait_after(foa_rx *rx, hipStream_t st)
{
    for (hipEvent_t e : rx->after) HIP_TRY(hipStreamWaitEvent(st, e, 0));
    rx->after.clear();
    return FOA_OK;
}

// ---- rx_decode.hip ----
int upload_tables_decode(const DeviceTables &t);
int workspace(foa_rx *rx, size_t n_samples, size_t n_frames);           // sizes rx->w
int flush_pending(foa_rx *rx, hipEvent_t after_front_end);              // queue the deferred chain-back + finish
int drain(foa_rx *rx);                                                  // ... and wait for everything on the handle's streams
int inputs_queued(foa_rx *rx, hipStream_t cs);
int job_ready(foa_rx *rx, uint64_t ticket, bool wait, HostJob **out);
void launch_fwd3(hipStream_t st, const FrameInfo *info, int nf, const uint16_t *sp, uint64_t *dec);
// chain-back walk on st, stitch + descramble + CRC on st_fin (the same stream, or another one that then waits for walk_done)
void launch_finish3(hipStream_t st, hipStream_t st_fin, const FrameInfo *info, int nf, const uint64_t *dec, uint32_t *decoded, const int32_t *seg2frame,
                    const int64_t *totals, uint16_t *tb_state, size_t max_segs, int S, int L, uint8_t *psdu, size_t slot_bytes,
                    foa_frame_result *results, hipEvent_t walk_done = nullptr);

// ... (rx_decode.hip) and how many of them, in order, can be DECIDED with the samples up to n_eff: LTS + SIGNAL of each (k_header_range into
// the engine's own records info / hinv, `grid` >= the range's length) and the first whose frame would run into the end (k_stream_resolve).
// sel (device, 5 ints): { STS_END candidates, alignments found, first of the batch, how many it decides, context alignments behind them };
// *state moves on to the first undecided alignment.
void launch_stream_resolve(hipStream_t st, const float *d_iq, int64_t n_eff, const foa_frame_desc *descs, const int64_t *ends, const int32_t *range,
                           const int32_t *sy_n, int64_t start_abs, int64_t hz_abs, bool final, StreamState *state, FrameInfo *info, double2 *hinv, int32_t *sel,
                           unsigned grid);

// ---- rx_sync.hip ----
int upload_tables_sync(const DeviceTables &t);
// The launching half of foa_rx_sync_dev: every kernel of the pre-sync stage queued on the side stream, nothing waited for.  The counts
// stay on the device in rx->sy_n ([0] STS_END candidates, [3] alignments found); *ccap_out = the candidate capacity they are checked
// against.  origin: stream index of d_iq[0].
int sync_dev_issue(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap, int32_t *ccap_out, int64_t origin,
                   bool behind_walk = false);
// The stream engines' look-ahead for one batch buffer, queued on st behind the buffer's pre-sync: which alignments are still to be
// decided and final in their tags (k_stream_range: STS_END in [state->lo_abs, hz_abs), first one's c_prev patched from the state) ...
void launch_stream_range(hipStream_t st, foa_frame_desc *descs, const int32_t *sy_n, int32_t cap, int64_t start_abs, int64_t hz_abs, const StreamState *state,
                         int32_t *range);
// (a small batch buffer filled by one kernel: `carry` samples out of carry_src -- device memory, null = zeros --, then n_new out of page-locked host memory)
void launch_stream_fill(hipStream_t st, float *dst, const float *carry_src, int64_t carry, const float *host_src, int64_t n_new);

// ---- rx_stage.hip / rx_tx.hip ----
int upload_tables_stage(const DeviceTables &t);
int upload_tables_tx(const DeviceTables &t);

// ---- rx_stream.hip ----
void stream_shutdown(foa_stream *s);      // joins the engine's threads (they use the handle); the owner still frees the shell

}  // namespace foa
