/*
 * fun_ofdm_amd.h -- C ABI of the MI355X-native 802.11a-like receive hot path.
 *
 * The library (fun_ofdm_amd/csrc/libfun_ofdm_amd.so) replaces, for bmorgan5/fun_ofdm, the work done
 * behind fun::receiver_chain::process_samples() from fft_symbols on:
 *
 *   fft_symbols::work   (src/fft_symbols.cpp:33-79)   + fft::forward (src/fft.cpp:50-59)
 *   channel_est::work   (src/channel_est.cpp:36-85)
 *   phase_tracker::work (src/phase_tracker.cpp:70-104)
 *   frame_decoder::work (src/frame_decoder.cpp:45-91) -> ppdu::decode_header / decode_data
 *                       (src/ppdu.cpp:168-295): modulator::demodulate, interleaver::deinterleave,
 *                       puncturer::depuncture, viterbi::conv_decode, descrambler, CRC-32
 *
 * Conventions: plain C types only; every call returns 0 on success and a negative FOA_E_* code on
 * error (foa_last_error() gives the text, per thread); the library never frees caller memory; a handle
 * is used from one thread at a time; each handle owns its HIP streams (below); there is NO CPU fallback -- a
 * call fails with FOA_E_NO_DEVICE when no gfx950 device/HIP runtime is usable.
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef FUN_OFDM_AMD_H
#define FUN_OFDM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOA_VERSION 110

enum {
    FOA_OK = 0,
    FOA_E_INVALID = -1,     /* bad argument */
    FOA_E_NO_DEVICE = -2,   /* no usable HIP device */
    FOA_E_HIP = -3,         /* a HIP runtime call failed */
    FOA_E_NOMEM = -4,       /* device allocation failed */
    FOA_E_STATE = -5        /* call sequence error (e.g. taps requested before any decode) */
};

/* Per-frame outcome; the reference drops failed frames silently (src/frame_decoder.cpp:63-68,
 * src/ppdu.cpp:187-203,272-279), here they are reported. */
enum {
    FOA_ST_OK = 0,           /* CRC-32 matched: psdu slot holds `length` payload bytes */
    FOA_ST_HEADER_FAIL = 1,  /* SIGNAL parity or rate check failed (ppdu.cpp:187-203) */
    FOA_ST_CRC_FAIL = 2,     /* ppdu.cpp:272-279 */
    FOA_ST_TRUNCATED = 3,    /* the samples handed over END before the alignment's LTS windows, its SIGNAL symbol or the last symbol its
                              * SIGNAL announces: with more of the stream the outcome would be another (a caller that cuts a stream into
                              * pieces decodes the alignment again when it has them) */
    FOA_ST_NO_SPACE = 4,     /* workspace exhausted (overlapping frame ranges; a rate above option "max_dbps"), or the CRC matched but the payload is longer
                              * than slot_bytes (nothing is copied) */
    FOA_ST_SUPERSEDED = 5    /* a later alignment took the stream over before this one's frame was complete: its LTS or SIGNAL window is
                              * cut by the next LTS1 tag, or a valid SIGNAL arrived before the frame's last symbol (frame_decoder.cpp:52-76
                              * abandons the frame being collected).  Final: the reference never delivers such a frame. */
};

/* One alignment = one LTS1/LTS2 tag pair produced by timing_sync (src/timing_sync.cpp:98-113).
 * Symbol windows relative to lts1_pos follow fft_symbols.cpp:41-73: LTS1 [0,64), LTS2 [64,128),
 * SIGNAL [144,208), data symbol k (1-based) [144+80k, 208+80k).
 * timing_sync rotates every sample by exp(+j*m_phase_acc) (timing_sync.cpp:114-125); the phase changes
 * at the STS_END sample that found the LTS, which may lie up to 8 samples after lts1_pos, hence the
 * second phasor. */
typedef struct foa_frame_desc {
    int64_t lts1_pos;       /* stream index of the sample tagged LTS1 */
    int64_t rot_start;      /* samples with index >= rot_start use (c,s), earlier ones (c_prev,s_prev) */
    double c, s;            /* cos/sin of m_phase_acc in force from rot_start on */
    double c_prev, s_prev;  /* cos/sin of m_phase_acc before */
} foa_frame_desc;

typedef struct foa_frame_result {
    int32_t status;         /* FOA_ST_* */
    int32_t rate;           /* fun::Rate enum value 0..10 (src/rates.h:31-44), -1 if header failed */
    int32_t length;         /* payload bytes announced by SIGNAL */
    int32_t num_symbols;    /* data OFDM symbols (ppdu.cpp:40-44) */
} foa_frame_result;

typedef struct foa_rx foa_rx;

int foa_version(void);
const char *foa_last_error(void);
/* number of HIP devices visible, or a negative FOA_E_* */
int foa_device_count(void);

/* The library runs a call's stages on up to six HIP streams (two to four lanes of pipelined decode calls, one for the stitch / CRC
 * kernels, one for copies and the pre-sync).  Streams overlap when they sit on different hardware queues, and big grids only when those
 * queues sit on different dispatch pipes.  The HIP runtime keeps a pool of GPU_MAX_HW_QUEUES queues (default 4, fixed when it starts) per
 * stream PRIORITY, and a process's queues go round the four pipes in the order they were made; so the library makes its lanes at the LOW
 * priority level and its stitch and copy streams at the HIGH one, in an order that puts any four in a row on four pipes -- none at the
 * normal level, where the host's own streams live.  It runs at full speed with the runtime's defaults and whatever streams the host has
 * made (profiles/r05_ab_stream_layout_host_queues.txt); the host sets nothing (rounds 3-4 asked for GPU_MAX_HW_QUEUES=8, and lost a
 * third of their speed behind two host streams).  The decode therefore yields to normal-priority work of the host on the same GPU.
 * Only a host that sets GPU_MAX_HW_QUEUES below 4 is told so by foa_rx_notes(). */

/* Create a receiver on HIP device `device` (its own non-blocking streams). */
int foa_rx_create(foa_rx **out, int device);
/* Non-fatal remarks about how the handle is set up ("" if none), e.g. a runtime cut down to fewer hardware queues per priority than the library has lanes.  Valid until the handle
 * is destroyed. */
const char *foa_rx_notes(foa_rx *rx);
void foa_rx_destroy(foa_rx *rx);

/* Pre-size the device workspace for streams of up to n_samples samples holding up to n_frames
 * alignments, so that later decode calls allocate nothing. */
int foa_rx_reserve(foa_rx *rx, size_t n_samples, size_t n_frames);

/* Options.  Results are identical for every setting of the first group (they shape scheduling and memory; the diagnostic options are in fun_ofdm_amd_diag.h); the second
 * group says which reference behaviour the pre-sync reproduces.  Unknown names and out-of-range values: FOA_E_INVALID.
 *   "tb_segment"  data steps per chain-back segment, a multiple of 96 in [96, 3072] (default 960; calls of up to 256 alignments take 192 unless this option was set)
 *   "tb_overlap"  run-in steps above a segment, a multiple of 96 in [0, 3072] (default 96); any value gives the same result as a
 *                 serial chain-back, small values cost re-walks (fun_ofdm_amd/csrc/viterbi_tb.h)
 *   "pipeline"    consecutive decode calls form a three-stage pipeline (front end | forward pass | chain-back and finish) over several
 *                 streams and rotating work sets (default 1).  Results and their order are unchanged; the outputs of a call are final
 *                 after foa_rx_sync (or, for the call before the most recent one, after foa_rx_wait_previous), and the INPUTS of a call
 *                 must be complete when it is made and stay untouched until then.  0 = every call runs start to end on the handle's
 *                 stream.  (Waits for everything in flight before it switches.)
 *   "depth"       pipelined calls: how many calls' loops are in flight -- 0 (default) = by grid size: 2 (3 for BPSK-only captures: option "max_dbps" <= 36), or 4 for calls of up to 4608 frames,
 *                 whose forward pass leaves most SIMDs a single wave (1 000-frame batches decode 40-50 % faster in steady state); 2, 3, 4 =
 *                 fixed.  (the lanes sit on hardware queues of their own: stream priorities, see above)
 *   "max_dbps"    what the work sets of later calls (and foa_rx_reserve) are sized for: the data bits per OFDM symbol of the HIGHEST rate the
 *                 caller's frames carry, 24 (6 Mbps) .. 216 (54 Mbps, the default: any rate).  A call's per-step buffers -- 10.25 bytes per
 *                 trellis step, several sets in rotation -- hold n_samples / 80 x max_dbps steps, so a capture of 6 Mbps frames needs a
 *                 ninth of the default, and a call can take nine times the frames (machine-filling calls at 6-18 Mbps: DESIGN.md 4).  A
 *                 promise, checked on the device: a frame that does not fit in what is left of its call's work set is reported
 *                 FOA_ST_NO_SPACE, never decoded wrongly.  Results are otherwise identical for every value.
 *
 *   "sync_call"   foa_rx_sync_dev / the stream engine: the reference call size by which timing_sync.cpp:99 is decided (foa_sync_set_call, below);
 *                 default 4096 = receiver.h:16, 0 = as one call over the whole stream.  FOA_E_STATE while a stream engine is open.
 *   "sync_origin" foa_rx_sync_dev / _begin: the stream index of d_iq[0] (default 0).  The rule above is decided by ABSOLUTE stream index
 *                 ((index of STS_END + 160) mod sync_call), so a caller that pre-synchronises one stream in consecutive batches sets this
 *                 to each batch's offset in the stream (or starts every batch on a multiple of sync_call, or sets sync_call 0);
 *                 otherwise frames are dropped at batch-relative positions the reference never drops them at
 *   "stream_longest"  read by foa_stream_create: the longest frame the stream will hold, in samples from the first preamble sample to the last data
 *                 sample, plus 192 (timing_sync's look-ahead and the window offset).  Every batch buffer starts this far (+ 2048) before its batch, so
 *                 that a frame still undecided when a batch ends is whole inside the next buffer; the default 0 = 110 592 covers the longest frame the
 *                 format allows (4095 bytes at 6 Mbps).  What it buys is a shorter carry (stream_longest + 2048 samples re-synchronised with every
 *                 batch: most of a small batch's cost), not latency: a frame is decoded by the first batch that holds its last sample whatever
 *                 this value.  A frame LONGER than it is delivered only if some batch buffer (carry + batch) happens to hold all of it -- a
 *                 deviation from the reference the caller has asked for, and one that depends on where the batch boundaries fall. */
int foa_rx_set_option(foa_rx *rx, const char *name, int64_t value);

/*
 * Batch decode with DEVICE pointers; asynchronous (call foa_rx_sync before reading the outputs; with option
 * "pipeline" = 0 the work is on the handle's stream alone and may be ordered against foa_rx_stream).  Replaces the per-frame work of fft_symbols, channel_est,
 * phase_tracker and frame_decoder (files and lines above) for n_frames alignments.
 *   d_iq       n_samples interleaved (re,im) float pairs: the raw stream handed to process_samples,
 *              before timing_sync's rotation (the kernel applies it from the descriptor)
 *   d_descs    n_frames descriptors, in stream order
 *   d_ends     per alignment: exclusive end index of the samples that belong to it: the next alignment's lts1_pos where the stream
 *              goes on into it, anything else (n_samples, the end of a capture's slot) where it does not
 *   d_psdu     n_frames slots of slot_bytes bytes (slot_bytes >= longest payload, <= 4095 needed)
 *   d_results  n_frames results
 * What an alignment's end means.  fft_symbols emits a vector every 80 samples behind an LTS2 tag until the next LTS1 tag re-aligns it,
 * pushing the partly filled vector it holds at that moment (fft_symbols.cpp:41-50); channel_est equalises each with the estimate in
 * force and frame_decoder copies the vectors behind a valid SIGNAL into its frame wherever they come from, dropping the frame when
 * another valid SIGNAL arrives first (frame_decoder.cpp:52-88).  An alignment whose end IS the next descriptor's lts1_pos is therefore
 * LINKED to it: a frame cut short by the next LTS1 takes the partial vector, the next alignment's SIGNAL vector and -- if that SIGNAL is
 * invalid -- its symbols, exactly as the reference's blocks do (FOA_ST_SUPERSEDED if a valid SIGNAL intervenes).  Any other end is where
 * the stream ends for that alignment and everything linked in front of it (FOA_ST_TRUNCATED if a frame needs more).  Frames that fit in
 * front of their alignment's end -- every frame of an undisturbed stream -- do not depend on any of this.
 * Pile-ups.  An alignment whose end comes before its second LTS window is complete (less than 128 samples on) gives no vector at all
 * (FOA_ST_SUPERSEDED).  An alignment less than 64 samples behind another one it is linked to has that one's LTS2 tag inside its own first
 * LTS window: fft_symbols restarts the vector there and again at the alignment's own LTS2 tag, so the first vector it completes -- still
 * tagged LTS_START -- is the window at lts1_pos + 64, channel_est takes the window 80 samples on as the second LTS vector, and SIGNAL and
 * every symbol are read one symbol (80 samples) later than in an undisturbed alignment (fft_symbols.cpp:53-56, channel_est.cpp:44-58).
 * The call reproduces that from the descriptors it is given, so a caller that decodes a stream in pieces hands the alignments of such a
 * pile-up over TOGETHER (the adaptors of blocks.hpp keep them together; foa_rx_decode_frames_lead_ctx_dev takes decided ones along; the timing_sync restated here does not produce such tags on
 * anything but pathological input).  With these rules a call returns, for any list of descriptors in stream order, what the reference's
 * blocks return when fed the tags those descriptors stand for.
 */
int foa_rx_decode_frames_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs,
                             const int64_t *d_ends, size_t n_frames, uint8_t *d_psdu, size_t slot_bytes,
                             foa_frame_result *d_results);
/* The same for a caller that decodes ONE stream in pieces: d_descs / d_ends hold n_frames + n_context alignments; the last n_context
 * are CONTEXT -- their LTS and SIGNAL are looked at, and their vectors serve the frames in front of them (as sources, as superseding
 * SIGNALs), but they are not decoded and get no result or PSDU slot (the caller decodes them with its next piece). */
int foa_rx_decode_frames_ctx_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs,
                                 const int64_t *d_ends, size_t n_frames, size_t n_context, uint8_t *d_psdu, size_t slot_bytes,
                                 foa_frame_result *d_results);

/* ... and with n_lead alignments IN FRONT of the n_frames decoded ones: d_descs / d_ends hold n_lead + n_frames + n_context alignments in
 * stream order; the first n_lead were decided by an earlier piece and are looked at only for where they sit (an alignment less than 64
 * samples behind one of them is read one symbol late: "Pile-ups" above).  Results and PSDU slots: the n_frames in the middle. */
int foa_rx_decode_frames_lead_ctx_dev(foa_rx *rx, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs, const int64_t *d_ends,
                                      size_t n_lead, size_t n_frames, size_t n_context, uint8_t *d_psdu, size_t slot_bytes,
                                      foa_frame_result *d_results);

/* Same with HOST pointers: copies in, decodes, copies out, synchronises. */
int foa_rx_decode_frames_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs,
                              const int64_t *ends, size_t n_frames, uint8_t *psdu, size_t slot_bytes,
                              foa_frame_result *results);

/* The same for a stream timing_sync has ALREADY rotated: iq holds n_samples complex<double> samples (interleaved re, im) exactly as
 * fun::timing_sync::work leaves them in its output_buffer (src/timing_sync.cpp:114-125), and the descriptors' phasors are not applied
 * (lts1_pos and the ends are what counts).  This is what the fused stage block fun_amd::rx_backend
 * (fun::block<fun::tagged_sample, std::vector<unsigned char>>, include/fun_ofdm_amd/blocks.hpp) calls once per work(): the reference's
 * own frame_detector and timing_sync threads in front, everything behind them in one device call. */
int foa_rx_decode_frames_f64_host(foa_rx *rx, const double *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends,
                                  size_t n_frames, uint8_t *psdu, size_t slot_bytes, foa_frame_result *results);

/* Block until everything queued on the handle's stream has finished. */
int foa_rx_sync(foa_rx *rx);

/* Blocks until the decode call BEFORE the most recent one is complete (its PSDUs and results are final), without
 * waiting for the most recent one: with option "pipeline" a caller can consume batch k-1 while batch k is in flight. */
int foa_rx_wait_previous(foa_rx *rx);
/* The same for the call `age` calls back (0 = the most recent, i.e. everything; 1 = previous; 2 = the one before: in a
 * pipelined sequence that one is complete by the time the most recent call has been queued, so this does not hold the
 * host up -- and a host that is held up queues the next call's front end late). */
int foa_rx_wait_age(foa_rx *rx, int age);
/* The handle's hipStream_t (as void*) so callers can order their own work against it. */
void *foa_rx_stream(foa_rx *rx);

/* The same with nothing left to wait for: foa_rx_submit_host copies the caller's buffers, queues the transfers and the decode
 * and returns a ticket; up to 8 calls may be in flight.  foa_rx_collect(ticket, wait, ...) returns 1 and fills psdu
 * (n_frames * slot_bytes) and results once that call is complete, 0 if it is not yet (only with wait = 0), < 0 on error.
 * Calls complete in the order they were submitted.  For a stream front end that must not stall on the GPU:
 * H2D of batch k+1, compute of batch k and D2H of batch k-1 overlap on the library's streams. */
int foa_rx_submit_host(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                       size_t slot_bytes, uint64_t *ticket);
/* ... with n_context context alignments behind the n_frames decoded ones (foa_rx_decode_frames_ctx_dev) */
int foa_rx_submit_host_ctx(foa_rx *rx, const float *iq, size_t n_samples, const foa_frame_desc *descs, const int64_t *ends, size_t n_frames,
                           size_t n_context, size_t slot_bytes, uint64_t *ticket);
int foa_rx_collect(foa_rx *rx, uint64_t ticket, int wait, uint8_t *psdu, foa_frame_result *results);

/* ---- device-side ordering against the caller's own HIP streams -------------------------------------------------------------------------
 *
 * The plain entry points above keep SURVEY 8(b)'s contract in its simplest form: buffers handed to a call must be COMPLETE when the call
 * is made (and stay untouched until it has read them), and outputs are final after foa_rx_sync -- the host waits.  A caller whose samples
 * are produced on the device by a stream of its own (an SDR's DMA engine, a channel simulator, another library) orders the two on the
 * DEVICE instead: every `event` below is a hipEvent_t (passed as void * so that this header needs no HIP header), recorded or waited for
 * with the caller's own HIP calls; the library never makes the host wait in any of these.
 *
 *   foa_rx_after(rx, e)             the NEXT call on the handle that queues device work (decode, pre-sync, transmit) starts only after
 *                                   e -- recorded by the caller behind whatever produces that call's inputs and prepares its outputs (a
 *                                   fill of the PSDU slots, say).  One-shot; up to 16 events may be registered for one call.  The event must
 *                                   have been recorded, and must stay alive until that call has returned (the library keeps the handle, not
 *                                   a reference); a call that queues nothing (n_frames = 0) takes the registered events with it.
 *   foa_rx_record_consumed(rx, e)   records e so that it completes when everything queued so far has READ what it reads of the caller's
 *                                   input buffers: the caller's stream waits for e (hipStreamWaitEvent) before it refills them.  Cheap.
 *   foa_rx_record_done(rx, e)       records e so that it completes when everything queued so far is complete (PSDU slots, results,
 *                                   descriptors): the caller's stream waits for e before it reads them.  Queues the deferred chain-back of
 *                                   a pipelined call at once (option "pipeline"), like foa_rx_sync does.
 *   foa_rx_decode_frames_dev_after / foa_rx_sync_dev_begin_after = foa_rx_after(rx, inputs_ready) + the call (inputs_ready may be NULL).
 *
 * A capture loop over two device buffers with no host synchronisation in it: INTEGRATION.md 2; tests/cpp/test_ordering.cpp runs one
 * against the oracle.  The library's lanes run BELOW the normal stream priority and its short stitch / copy streams above (see the
 * top of this file), so an un-ordered producer on a normal-priority stream is overtaken in both directions: order with these calls, or
 * complete the buffers before the call. */
int foa_rx_after(foa_rx *rx, void *event);
int foa_rx_record_consumed(foa_rx *rx, void *event);
int foa_rx_record_done(foa_rx *rx, void *event);
int foa_rx_decode_frames_dev_after(foa_rx *rx, void *inputs_ready, const float *d_iq, size_t n_samples, const foa_frame_desc *d_descs,
                                   const int64_t *d_ends, size_t n_frames, uint8_t *d_psdu, size_t slot_bytes, foa_frame_result *d_results);

/* ---- pre-sync: on the host (streaming) and on the device (whole resident streams), SURVEY 8f #1 ---- */

/* Streaming frame_detector + timing_sync (src/frame_detector.cpp:41-93, src/timing_sync.cpp:51-139):
 * consumes the raw stream in chunks of any size and reports one descriptor per LTS1/LTS2 tag pair,
 * with stream-absolute positions.  No GPU needed. */
typedef struct foa_sync foa_sync;
int foa_sync_create(foa_sync **out);
void foa_sync_destroy(foa_sync *s);
/* Push n_samples interleaved (re,im) samples; up to cap finished descriptors are written to out and
 * *n_out is set.  Descriptors that did not fit stay queued: call again with n_samples = 0 to drain. */
int foa_sync_push_f32(foa_sync *s, const float *iq, size_t n_samples, foa_frame_desc *out, size_t cap, size_t *n_out);
int foa_sync_push_f64(foa_sync *s, const double *iq, size_t n_samples, foa_frame_desc *out, size_t cap, size_t *n_out);
/* Stream index up to which timing_sync has looked (it trails the input by 160 samples). */
int64_t foa_sync_settled(const foa_sync *s);
/* One line of timing_sync depends on how the stream is cut into CALLS: `if(lts_offset < 0) break;` (timing_sync.cpp:99) drops an
 * alignment whose LTS guard interval would start before the working buffer of the call that walks over its STS_END -- with the STS_END
 * tag usually a sample or two late, that is a frame whose STS ends exactly 160 samples before a call boundary.  The reference's receiver
 * cuts the stream every NUM_RX_SAMPLES = 4096 (receiver.h:16), and that is what the pre-sync decides by -- on the host and on the device,
 * whatever the chunk sizes it is fed with -- unless told otherwise: call = the reference call size to reproduce (> 160), or 0 = decide
 * as ONE call over the whole stream would (no alignment is ever dropped for that reason).  Device side: option "sync_call". */
int foa_sync_set_call(foa_sync *s, int64_t call);

/* The same two blocks on the DEVICE, over a stream that is already resident in HBM: fills d_descs / d_ends (device
 * pointers, capacity cap) in stream order and returns the number of alignments in *n_found (synchronises).  The decisions
 * are the reference's up to floating-point ties: windowed sums are formed directly instead of by the reference's
 * ever-drifting running sums, so a threshold decision can differ when the normalised correlation is within ~1e-15 of
 * 0.9 -- and the reference's running sums never quite forget a huge sample: after a glitch of >= ~1e10 times the
 * signal amplitude its frame_detector tags residue artefacts for the rest of the stream (circular_accumulator.h:88-95), which
 * this stage does not reproduce (it finds the real frames only; foa_sync_push_* reproduces the reference there as well;
 * tests/test_gpu_parity.py::test_device_sync_large_dynamic_range).  A NaN sample is taken in as the reference takes it -- the two
 * products it is part of count as zero (circular_accumulator.h:91) -- so descriptors agree wherever it falls; an INFINITE sample
 * blinds the reference's detector for the rest of the stream (Inf - Inf = NaN stays in its running sum), this stage for the 32
 * samples whose windows hold it (test_device_sync_non_finite_samples).  d_iq must be complete when the call is made (like the input of foa_rx_decode_frames_dev: with
 * calls pipelined the stage runs on the library's third stream, under the forward pass of a decode call in flight, and
 * is not ordered behind foa_rx_stream()); d_descs / d_ends may be handed to the next decode call straight away.
 * FOA_E_INVALID if more than cap alignments are found, FOA_E_NOMEM if the stream holds more STS_END candidates than
 * one per 64 samples. */
int foa_rx_sync_dev(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap, size_t *n_found);
/* The same in two halves, for a caller that pipelines pre-sync and decode over consecutive batches: _begin queues the kernels
 * (nothing is waited for; one pre-sync in flight per handle), _end waits for them and returns the count.  Between the two the caller
 * may queue decode calls -- e.g. _end(k), _begin(k+1), foa_rx_decode_frames_dev(k) per step: the host then never waits for a
 * pre-sync it has only just queued, and the pre-sync of batch k+1 runs under the forward pass of batch k.  The descriptor and end
 * buffers of a batch must stay untouched from its _begin to the completion of the decode call that reads them. */
int foa_rx_sync_dev_begin(foa_rx *rx, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap);
int foa_rx_sync_dev_end(foa_rx *rx, size_t *n_found);
/* ... whose kernels start only after `inputs_ready` (a hipEvent_t of the caller's: "device-side ordering" above) */
int foa_rx_sync_dev_begin_after(foa_rx *rx, void *inputs_ready, const float *d_iq, size_t n_samples, foa_frame_desc *d_descs, int64_t *d_ends, size_t cap);

/* ---- process_samples() entirely on the device (SURVEY 8f #1 + #3): a stream engine over the calls above ------------
 *
 * What fun::receiver_chain::process_samples() does call by call (src/receiver_chain.cpp:106-126: frame_detector ->
 * timing_sync -> fft_symbols -> channel_est -> phase_tracker -> frame_decoder on a 4096-sample chunk, state carried from
 * call to call), done batch by batch on the GPU: pushed samples collect in page-locked memory; every `batch_samples`
 * samples one batch goes out -- H2D, the pre-sync kernels over the batch plus the 112 640 samples before it, a look-ahead that finds
 * the alignments that can be DECIDED with the samples there are (everything in front of the first frame that would run into the
 * buffer's end), foa_rx_decode_frames_ctx_dev for those, D2H of the PSDUs -- asynchronously, several batches in flight; finished
 * batches hand out their CRC-passing payloads in stream order.  Consecutive buffers overlap by more than the longest frame, so a
 * frame is decoded exactly once, by the first batch that holds its last sample, and the pre-sync sees the same samples around every
 * decision as a single pass over the whole stream would (fun_ofdm_amd/csrc/stream_engine.h).  The payload list equals the reference
 * chain's (tests/test_gpu_stream.py); a payload comes back one batch period + the device's ~1 ms after its frame's last sample.
 * One engine per handle at a time; the handle's other entry points must not be used while a stream is open (the engine's
 * submitter thread makes the GPU calls; push / flush / ready / take / stats belong to ONE caller thread). */
typedef struct foa_stream foa_stream;
/* batch_samples in [4096, 2^28]; narrow_threads: helper threads for the double -> float narrowing of
 * foa_stream_push_f64 (0 = the calling thread alone; used for pushes of >= 65536 samples and for every buffer handed over with
 * foa_stream_push_f64_owned).  The engine's threads are confined to the block of eight consecutive CPUs the creating thread runs
 * on (the cores that share its last-level cache on the hosts measured; several times faster than threads spread over two
 * sockets); the environment variable FOA_STREAM_AFFINITY=0 leaves them to the scheduler.  Measured on a 2 x EPYC 9575F host:
 * 4 Mi-sample batches carry 3.2-3.4 Gsample/s of complex<double> through process_samples with four helpers and 4.0-4.8 with eight (more than
 * four alternate between the caller's block of eight CPUs and its neighbour; FOA_STREAM_AFFINITY=1 keeps all on one) (profiles/).
 * Streams of batches up to 65536 samples -- a live radio's -- keep twelve batch buffers in rotation (larger batches: six), and their submitter
 * thread POLLS instead of sleeping while batches keep coming (until 2 ms pass without one): one CPU busy for the stream's lifetime buys the
 * tail of the payload latency (4 Ki batches at 20 Msample/s: a payload four calls of 4096 samples after its frame's last sample, p50 0.82 ms,
 * p99 1.0-1.2 instead of 1.4-3; profiles/r06_latency_stages.txt).  The environment
 * variable FOA_STREAM_SPIN_US sets that interval (0: the thread sleeps, as it does by itself in a process confined to fewer than four CPUs).
 * One stream per handle: a second create while one is open fails with FOA_E_STATE (the engine's submitter thread owns the handle's
 * streams and work sets); destroying the HANDLE first stops the engine -- every later call on the stream then fails with FOA_E_STATE
 * and foa_stream_destroy only frees it.  Samples pushed with foa_stream_push_f64_owned into a batch that is still open may stay
 * un-narrowed (and their buffers unreleased) until a few more pushes, a flush or the destroy -- or, if the caller makes no further call
 * (a producer that waits for its pool of buffers to come back), for at most about half a millisecond: the engine's submitter thread
 * then publishes and releases what is left.  Buffers of 128 KB and more are released on the helper thread that narrowed their last
 * slice (they are the allocator's own mappings: giving them back is a munmap, which would otherwise be the caller's time). */
int foa_stream_create(foa_rx *rx, size_t batch_samples, int narrow_threads, foa_stream **out);
void foa_stream_destroy(foa_stream *s);
/* The next n_samples of the stream (interleaved re,im).  Returns when they are copied; submits a batch whenever one is full
 * (which may wait for the oldest batch in flight when all buffers are in use). */
int foa_stream_push_f32(foa_stream *s, const float *iq, size_t n_samples);
int foa_stream_push_f64(foa_stream *s, const double *iq, size_t n_samples);
/* The same, handing the buffer over: the call returns at once, the engine's helper threads narrow the samples later and
 * call release(ctx) (on one of their threads, or on this one) when the buffer is no longer needed.  For callers that own
 * their buffer anyway -- receiver_chain::process_samples takes its vector by value (src/receiver_chain.h:56). */
int foa_stream_push_f64_owned(foa_stream *s, const double *iq, size_t n_samples, void (*release)(void *), void *ctx);
/* End of the stream: submits what is left, the frames after the last batch boundary included.  No pushes afterwards. */
int foa_stream_flush(foa_stream *s);
/* 1 if the oldest submitted batch is complete (then *n_payloads / *n_bytes describe its CRC-passing payloads), 0 if it is
 * not (wait = 0) or if no batch is outstanding, < 0 on error.  wait = 1 blocks until the oldest batch is complete. */
int foa_stream_ready(foa_stream *s, int wait, size_t *n_payloads, size_t *n_bytes);
/* The payloads of the batch foa_stream_ready reported, back to back in stream order, and their lengths; releases it. */
int foa_stream_take(foa_stream *s, uint8_t *payloads, uint32_t *lengths);
/* out[0..4]: alignments per FOA_ST_* status so far (taken batches; FOA_ST_SUPERSEDED is counted with FOA_ST_TRUNCATED in [3]), [5] alignments
 * submitted, [6] batches, [7] samples pushed */
int foa_stream_stats(const foa_stream *s, uint64_t out[8]);

/* ---- the same stream engine over SEVERAL devices (BASELINE config 4's frame sharding, behind the C ABI) -----------------------------
 *
 * foa_shard_* is foa_stream_* with a list of devices: batch k of the stream goes to device k mod n_devices -- upload, pre-sync,
 * decode and copy back on that device, through a receiver handle the shard creates for it -- so n_devices consecutive batches are in
 * work at once, and the CRC-passing payloads still come out in stream order (foa_shard_ready / _take, batch by batch).  Every batch
 * buffer is filled from the host (the carry of 112 640 samples before the batch included), so no device reads another device's memory and
 * there is no collective: what crosses from one batch to the next is where the first undecided alignment begins and the phasor timing_sync
 * left in force (timing_sync.cpp:113-125), 24 bytes the host hands from device to device in stream order (fun_ofdm_amd/csrc/shard_core.h; its ordering logic runs against
 * device doubles at 1, 2, 3 and 8 devices in tests/cpp/shard_core_test.cpp).  The payload list equals foa_stream_*'s and the reference
 * chain's (tests/test_gpu_stream.py runs it with the one device a test box has, listed once and twice).  A device may be listed more
 * than once (two handles on one device share it).  NOT measured on a multi-GPU node: none was available to the builder.
 * Threads, ownership of handed-over buffers and the meaning of every call are those of the foa_stream_* functions of the same name. */
typedef struct foa_shard foa_shard;
int foa_shard_create(const int *devices, int n_devices, size_t batch_samples, int narrow_threads, foa_shard **out);
void foa_shard_destroy(foa_shard *s);
int foa_shard_devices(const foa_shard *s);
int foa_shard_push_f32(foa_shard *s, const float *iq, size_t n_samples);
int foa_shard_push_f64(foa_shard *s, const double *iq, size_t n_samples);
int foa_shard_push_f64_owned(foa_shard *s, const double *iq, size_t n_samples, void (*release)(void *), void *ctx);
int foa_shard_flush(foa_shard *s);
int foa_shard_ready(foa_shard *s, int wait, size_t *n_payloads, size_t *n_bytes);
int foa_shard_take(foa_shard *s, uint8_t *payloads, uint32_t *lengths);
/* out[0..7] as foa_stream_stats (summed over the devices); per_device_alignments (may be NULL): alignments decoded by each of the
 * first n_devices entries of the device list */
int foa_shard_stats(const foa_shard *s, uint64_t out[8], uint64_t *per_device_alignments, int n_devices);

/* ---- stage-level entry points (one per replaced fun::block, for the per-block adaptors) ---- */

/* fft::forward over n_vec vectors of 64 complex doubles (host pointers, in place): unscaled DFT with
 * the reference's index shift (src/fft.cpp:50-59), i.e. what fft_symbols::work applies per vector. */
int foa_fft_forward_f64(foa_rx *rx, double *vectors, size_t n_vec);

/* viterbi::conv_decode (src/viterbi.cpp:31-37) on host buffers: symbols[2*(data_bits+6)] soft bytes ->
 * data[(data_bits+7)/8] bytes.  n_blocks independent blocks of identical data_bits, packed. */
int foa_conv_decode(foa_rx *rx, const uint8_t *symbols, uint8_t *data, int data_bits, size_t n_blocks);

/* channel_est::work, estimation half (src/channel_est.cpp:44-58): n pairs of FFT'd LTS vectors
 * (lts_pairs[n][2][64] complex doubles) -> hinv[n][64] = sum over the pair of (LTS_FREQ_DOMAIN / Y) / 2. */
int foa_channel_estimate_f64(foa_rx *rx, const double *lts_pairs, double *hinv, size_t n);

/* channel_est::work, correction half (src/channel_est.cpp:77-81), in place: vectors[i][j] *= hinv[hinv_index[i]][j]. */
int foa_equalize_f64(foa_rx *rx, double *vectors, size_t n_vec, const double *hinv, size_t n_hinv, const int32_t *hinv_index);

/* phase_tracker::work (src/phase_tracker.cpp:70-104): vectors[n][64] equalised symbols with their symbol counters
 * (0 = SIGNAL) -> out48[n][48] derotated data carriers. */
int foa_phase_track_f64(foa_rx *rx, const double *vectors, const int32_t *symbol_count, size_t n_vec, double *out48);

/* ppdu::decode_header (src/ppdu.cpp:168-218) on n SIGNAL symbols of 48 carriers: results[i].status is FOA_ST_OK or
 * FOA_ST_HEADER_FAIL, rate/length/num_symbols filled on success. */
int foa_decode_header_f64(foa_rx *rx, const double *carriers48, size_t n, foa_frame_result *results);

/* ppdu::decode_data (src/ppdu.cpp:223-295) on n_frames frames: frame i has results[i].rate / .length set by the
 * caller and its num_symbols*48 carriers at carriers[2*carrier_off[i]] (carrier_off has n_frames+1 entries, in
 * complex elements).  On return results[i].status is FOA_ST_OK (payload in psdu slot i) or FOA_ST_CRC_FAIL. */
int foa_decode_data_f64(foa_rx *rx, const double *carriers, const uint64_t *carrier_off, size_t n_frames, foa_frame_result *results,
                        uint8_t *psdu, size_t slot_bytes);

/* ---- transmit side on the device: synthetic workloads and loop-back tests in HBM --------------------- */

/* frame_builder::build_frame (src/frame_builder.cpp:53-82: ppdu::encode src/ppdu.cpp:65-165, symbol_mapper::map
 * src/symbol_mapper.cpp:81-119, fft::inverse src/fft.cpp:68-96, preamble src/preamble.h:24) for n_frames payloads of the
 * same length and rate (Rate enum 0..10): payload i at d_payloads + i * payload_pitch; frame i as *frame_samples =
 * 320 + 80 (num_symbols + 1) complex<double> samples at d_frames + 2 * i * *frame_samples (interleaved re, im).
 * Asynchronous on the handle's stream.  With n_frames = 0 only *frame_samples is computed. */
int foa_tx_build_frames_dev(foa_rx *rx, const uint8_t *d_payloads, size_t payload_pitch, int length, int rate, size_t n_frames,
                            double *d_frames, size_t *frame_samples);
/* The synthetic channel the benchmark workloads use, not a reference component: frame i starts at sample
 * i * pitch + lead of the output stream, gets a uniformly random carrier phase and, if cfo_hz > 0, a constant frequency
 * offset uniform in +-cfo_hz (20 Msample/s); complex white Gaussian noise with sigma^2 = 0.0124 / (2 10^(snr_db/10)) per
 * real component is added everywhere; the result is rounded to complex<float>.  Deterministic in (seed, sample index). */
int foa_tx_channel_dev(foa_rx *rx, const double *d_frames, size_t n_frames, size_t frame_samples, size_t pitch, size_t lead, double snr_db,
                       double cfo_hz, uint64_t seed, float *d_iq);

#ifdef __cplusplus
}
#endif
#endif /* FUN_OFDM_AMD_H */
