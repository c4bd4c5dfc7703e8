/*
 * fun_ofdm_amd_diag.h -- diagnostics and measurement surface of libfun_ofdm_amd.so: kernel timings, intermediates of a decode call,
 * an issue-rate probe.  NOT part of the drop-in boundary (include/fun_ofdm_amd.h is what INTEGRATION.md binds, and all a host of the
 * receive path needs): the tests, bench.py and the tools under tools/ use these to check parity of intermediates (FFT / equaliser
 * output within 1e-4, soft bytes bit for bit) and to price the kernels against their rooflines.  Same library, no stability promise:
 * entry points here may change with the kernels they look into.
 */
#ifndef FUN_OFDM_AMD_DIAG_H
#define FUN_OFDM_AMD_DIAG_H

#include "fun_ofdm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Options of foa_rx_set_option that exist for the entry points below (results of a decode are identical for every setting):
 *   "record_soft" keep the depunctured soft bytes for foa_rx_get_taps (default 1; 0 saves their HBM writes)
 *   "record_eq"   keep the equalised carriers for foa_rx_get_taps (default 0)
 */

/* HIP-event durations (ms) of the kernels of the most recent decode call, measured on the handle's
 * stream: [0] header (LTS+SIGNAL), [1] offset scan, [2] data-symbol FFT/equalise/demap,
 * [3] Viterbi forward pass, [4] chain-back + descramble + CRC, [5] whole call.  Synchronises. */
int foa_rx_last_kernel_ms(foa_rx *rx, float out_ms[6]);
/* The same for the decode call before the most recent one: that call is complete (or nearly) while the most recent one
 * may still be running, so reading it does not stall a pipelined sequence of calls. */
int foa_rx_prev_kernel_ms(foa_rx *rx, float out_ms[6]);
/* ... and for the call `age` calls back (0 = most recent, 1 = previous, 2 = the one before: with pipelined calls that one is
 * certainly complete, so asking never delays the host, which matters because the next call's front end runs under the
 * forward pass that is on the GPU now). */
int foa_rx_kernel_ms_age(foa_rx *rx, int age, float out_ms[6]);

/* Pipelined calls: how the forward pass of the call `age` calls back (1 .. 3) lies against the one of the call before it, from the same HIP
 * events: out[0] = start to start, out[1] = how long the earlier pass was still running after this one had started (> 0: they overlapped;
 * consecutive passes run on different streams by design), out[2] = this pass's own duration (ms).  A launch that shares the machine with its
 * neighbour lasts longer than the step: bench.py reports both next to its roofline fraction. */
int foa_rx_forward_spacing_ms(foa_rx *rx, int age, float out[3]);

/* Issue-rate probe of the device the handle lives on (measurement aid for bench.py's roofline; nothing in the receive path
 * uses it): out[0..2] = SIMD clocks per wave64 `v_pk_add_u16 ... clamp`, shader clock (GHz) sustained meanwhile, wave-instructions
 * per second over the whole chip; out[3..5] the same for the plain 32-bit VOP2 `v_add_u32`.  Eight waves per SIMD issue from
 * independent chains for a fixed window of shader clocks (csrc/probe_kernels.h).  Synchronises; about a millisecond. */
int foa_rx_probe_issue(foa_rx *rx, double out[6]);

/* Host-to-device rate the way the stream engines move a capture (page-locked hipHostMalloc staging, hipMemcpyAsync on the library's copy
 * stream, in_flight (1..16) pieces of piece_bytes queued at once, `rounds` times after one warm-up pass): the ceiling of every leg that
 * hands over host buffers, 8 bytes per sample.  Synchronises; allocates and frees its own buffers. */
int foa_rx_probe_h2d(foa_rx *rx, size_t piece_bytes, int in_flight, int rounds, double *gbytes_per_s);

/* Intermediates of the most recent decode call, copied to HOST memory (parity tests).
 *   hinv    64 complex doubles (re,im) per frame: channel_est's m_chan_est (channel_est.cpp:53-58)
 *   eq      per frame (1 + num_symbols) * 48 complex doubles: phase_tracker output incl. SIGNAL
 *   soft    per frame 2 * num_symbols * dbps depunctured soft bytes (puncturer.cpp:78-123 output)
 * Any pointer may be NULL.  eq/soft are packed frame after frame in frame order for frames whose header
 * decoded; eq_off/soft_off (n_frames+1 entries each, may be NULL) receive the element offsets. */
int foa_rx_get_taps(foa_rx *rx, size_t n_frames, double *hinv, double *eq, size_t eq_cap, uint64_t *eq_off,
                    uint8_t *soft, size_t soft_cap, uint64_t *soft_off);
/* Raw decision words of one frame of the most recent decode call (debugging / unit parity of the forward
 * kernel): n_steps = num_symbols * dbps words, the raw region of the forward pass's transposed layout (16-bit words
 * [block of 16 data steps][63 - slot], complemented bits, trellis steps 6.. only; see fun_ofdm_amd/csrc/viterbi_fwd.h). */
int foa_rx_get_decisions(foa_rx *rx, size_t frame, uint64_t *out, size_t cap, size_t *n_steps);

#ifdef __cplusplus
}
#endif
#endif /* FUN_OFDM_AMD_DIAG_H */
