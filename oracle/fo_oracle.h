/*
 * fo_oracle.h -- CPU restatement of the fun_ofdm 802.11a-like receive path (+ the TX needed to
 * make inputs for it).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path (fun_ofdm_amd/) never does.
 *
 * Every function restates -- it does not copy -- the behaviour of the reference file:line it
 * cites (paths relative to the reference's src/).  Where the reference's arithmetic lives in a
 * third-party library that is not in the reference tree the published algorithm is restated:
 *   - FFTW3 (unpinned, "fftw3 >= 3.0"): unscaled 64-point DFT by definition (fft.cpp:30-59);
 *   - Boost.CRC crc_32_type (Boost >= 1.59): IEEE 802.3 CRC-32, check value 0xCBF43926.
 *
 * Pinning: the restatement is checked (tests/test_oracle_vs_ref.py, in the dev container) against
 * oracle/_ref/libfun_ofdm_ref.so, which is built from the reference's own self-contained sources
 * where they lie (viterbi, parity, interleaver, puncturer, modulator, channel_est, phase_tracker,
 * frame_detector, timing_sync, symbol_mapper).  fft.cpp, fft_symbols.cpp, ppdu.cpp,
 * frame_decoder.cpp, frame_builder.cpp and receiver_chain.cpp need FFTW3/Boost, which this image
 * lacks, so they are unbuildable here: for the logic of those six files parity is pinned only by
 * definition (DFT, CRC-32 check value), by the reference's documented loop-back result
 * (README.md:169-183) and by round trips through the real reference encoder pieces.
 */
#ifndef FO_ORACLE_H
#define FO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* std::complex<double> layout (tagged_vector.h:46,84) */
typedef struct { double re, im; } fo_c64;

/* tagged_vector.h:25-34 */
enum fo_tag {
    FO_NONE = 0, FO_STS_START, FO_STS_END, FO_LTS_START, FO_LTS1, FO_LTS2, FO_START_OF_FRAME
};

/* tagged_vector.h:82-95 (sizeof 24) / :43-76 (sizeof 1032, 776) */
typedef struct { fo_c64 sample; int32_t tag; int32_t _pad; } fo_tagged_sample;
typedef struct { fo_c64 samples[64]; int32_t tag; int32_t _pad; } fo_tagged_vec64;
typedef struct { fo_c64 samples[48]; int32_t tag; int32_t _pad; } fo_tagged_vec48;

/* rates.h:31-44, :52-196 */
enum { FO_NUM_RATES = 11 };
typedef struct {
    int rate_field, cbps, dbps, bpsc, rate;
    int punct; /* 0: 1/2, 1: 2/3, 2: 3/4 */
} fo_rate_params;

int fo_rate_params_get(int rate, fo_rate_params *out);   /* 0 ok, -1 bad rate */
int fo_rate_from_field(int rate_field);                  /* rate enum or -1 (rates.h:21,208-249) */
int fo_num_symbols(int rate, int length);                /* ppdu.cpp:40-44 */
double fo_demod_scale(int rate);                         /* qam.h:35-51 d_scale_d */

/* ---- tables (preamble.h, phase_tracker.cpp:23-50) ---- */
const fo_c64 *fo_preamble_samples(void);        /* 320, preamble.h:24   */
const fo_c64 *fo_lts_freq_domain(void);         /* 64,  preamble.h:363  */
const fo_c64 *fo_lts_time_domain_conj(void);    /* 64,  preamble.h:432  */
const double *fo_polarity(void);                /* 127, phase_tracker.cpp:23-32 */
const int *fo_data_subcarriers(void);           /* 48,  phase_tracker.cpp:46-50 */
const int *fo_pilot_subcarriers(void);          /* 4 indices; signs {1,1,1,-1} */

/* ---- bit-level codec ---- */
int fo_parity(unsigned int x);                                              /* parity.h:43-48 */
uint32_t fo_crc32(const uint8_t *data, size_t n);                           /* ppdu.cpp:134-137,267-271 */
void fo_scramble(const uint8_t *in, uint8_t *out, size_t n);                /* ppdu.cpp:141-147,256-264 */
void fo_conv_encode(const uint8_t *data, uint8_t *symbols, int data_bits);  /* viterbi.cpp:39-62 */
/* viterbi.cpp:31-37,71-78,108-146,208-457. symbols: 2*(data_bits+6) soft bytes; data: (data_bits+7)/8 bytes
 * written exactly where viterbi_chainback writes them. */
void fo_conv_decode(const uint8_t *symbols, uint8_t *data, int data_bits);
/* forward pass only: nsteps trellis steps (an odd trailing step is dropped like viterbi.cpp:209),
 * decisions[nsteps] as decision_t words (viterbi.h:36-41), final metrics[64]; returns number of
 * saturating adds that clipped and of renormalisations in stats[2] (may be NULL). */
void fo_viterbi_forward(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics,
                        uint64_t *stats);
void fo_viterbi_chainback(const uint64_t *decisions, uint8_t *data, int data_bits);

size_t fo_puncture(const uint8_t *in, size_t n, int rate, uint8_t *out);    /* puncturer.cpp:19-71 */
size_t fo_depuncture(const uint8_t *in, size_t n, int rate, uint8_t *out);  /* puncturer.cpp:78-123 */
void fo_interleave(const uint8_t *in, size_t n, uint8_t *out);              /* interleaver.cpp:15-25 */
void fo_deinterleave(const uint8_t *in, size_t n, uint8_t *out);            /* interleaver.cpp:28-38 */
/* modulator.cpp:14-105: n coded bits (0/1 bytes) -> n/bpsc carriers */
size_t fo_modulate(const uint8_t *bits, size_t n, int rate, fo_c64 *out);
/* modulator.cpp:108-164 + qam.h:110-125: n carriers -> n*bpsc soft bytes */
size_t fo_demodulate(const fo_c64 *in, size_t n, int rate, uint8_t *out);

/* ---- PPDU ---- */
/* ppdu.cpp:76-113: 48 BPSK carriers of the SIGNAL symbol */
void fo_encode_header(int rate, int length, fo_c64 *out48);
/* ppdu.cpp:168-218: returns 1 and fills rate/length/num_symbols on success, 0 on parity/rate failure */
int fo_decode_header(const fo_c64 *in48, int *rate, int *length, int *num_symbols);
/* ppdu.cpp:115-165: returns number of carriers written (num_symbols*48) */
size_t fo_encode_data(const uint8_t *payload, int length, int rate, fo_c64 *out);
/* ppdu.cpp:223-295: returns 1 (CRC ok; payload[length] filled) or 0. Optional taps (may be NULL):
 * soft = depunctured soft bytes (2*num_symbols*dbps), decoded = descrambled bytes (num_data_bytes). */
int fo_decode_data(const fo_c64 *in, int rate, int length, uint8_t *payload, uint8_t *soft, uint8_t *decoded);

/* ---- TX ---- */
/* symbol_mapper.cpp:81-119: n48 data carriers (multiple of 48) -> n48/48*64 bins */
size_t fo_symbol_map(const fo_c64 *in, size_t n48, fo_c64 *out);
void fo_ifft64(fo_c64 *data);      /* fft.cpp:68-96 incl. fft_map and the 1/64 scale */
void fo_fft64(fo_c64 *data);       /* fft.cpp:50-59 incl. fft_map */
size_t fo_frame_samples(int rate, int length);   /* 320 + 80*(num_symbols+1) */
/* frame_builder.cpp:53-82 */
size_t fo_build_frame(const uint8_t *payload, int length, int rate, fo_c64 *out);

/* ---- RX blocks (block.h:68-112 semantics: work() consumes all input, replaces output) ---- */
typedef struct fo_frame_detector fo_frame_detector;
typedef struct fo_timing_sync fo_timing_sync;
typedef struct fo_fft_symbols fo_fft_symbols;
typedef struct fo_channel_est fo_channel_est;
typedef struct fo_phase_tracker fo_phase_tracker;
typedef struct fo_frame_decoder fo_frame_decoder;

fo_frame_detector *fo_frame_detector_new(void);
void fo_frame_detector_free(fo_frame_detector *);
/* frame_detector.cpp:41-93 (+circular_accumulator.h:88-95). out[n]. n must be >= 16. */
void fo_frame_detector_work(fo_frame_detector *, const fo_c64 *in, size_t n, fo_tagged_sample *out);

fo_timing_sync *fo_timing_sync_new(void);
void fo_timing_sync_free(fo_timing_sync *);
/* timing_sync.cpp:51-139. out[n]; n must be > 160. */
void fo_timing_sync_work(fo_timing_sync *, const fo_tagged_sample *in, size_t n, fo_tagged_sample *out);
double fo_timing_sync_phase_acc(const fo_timing_sync *);

fo_fft_symbols *fo_fft_symbols_new(void);
void fo_fft_symbols_free(fo_fft_symbols *);
/* fft_symbols.cpp:33-79. out capacity must be >= n/64 + 2; returns vectors written. */
size_t fo_fft_symbols_work(fo_fft_symbols *, const fo_tagged_sample *in, size_t n, fo_tagged_vec64 *out);

fo_channel_est *fo_channel_est_new(void);
void fo_channel_est_free(fo_channel_est *);
/* channel_est.cpp:36-85. out capacity n; returns vectors written. */
size_t fo_channel_est_work(fo_channel_est *, const fo_tagged_vec64 *in, size_t n, fo_tagged_vec64 *out);
const fo_c64 *fo_channel_est_state(const fo_channel_est *);   /* m_chan_est[64] */

fo_phase_tracker *fo_phase_tracker_new(void);
void fo_phase_tracker_free(fo_phase_tracker *);
/* phase_tracker.cpp:70-104. out[n]. */
void fo_phase_tracker_work(fo_phase_tracker *, const fo_tagged_vec64 *in, size_t n, fo_tagged_vec48 *out);

/* A list of payloads (std::vector<std::vector<unsigned char>>) */
typedef struct fo_payloads fo_payloads;
fo_payloads *fo_payloads_new(void);
void fo_payloads_free(fo_payloads *);
void fo_payloads_clear(fo_payloads *);
size_t fo_payloads_count(const fo_payloads *);
size_t fo_payloads_len(const fo_payloads *, size_t i);
const uint8_t *fo_payloads_data(const fo_payloads *, size_t i);

fo_frame_decoder *fo_frame_decoder_new(void);
void fo_frame_decoder_free(fo_frame_decoder *);
/* frame_decoder.cpp:45-91. Replaces the content of out (output_buffer.resize(0)) when n > 0. */
void fo_frame_decoder_work(fo_frame_decoder *, const fo_tagged_vec48 *in, size_t n, fo_payloads *out);
/* counters since creation: [0] headers ok, [1] headers rejected, [2] CRC ok, [3] CRC failed */
const uint64_t *fo_frame_decoder_stats(const fo_frame_decoder *);

/* ---- receiver_chain (receiver_chain.cpp:29-126) ---- */
typedef struct fo_receiver_chain fo_receiver_chain;
fo_receiver_chain *fo_receiver_chain_new(void);
void fo_receiver_chain_free(fo_receiver_chain *);
/* One process_samples() call: all six blocks run on the buffers they hold (the reference runs them
 * concurrently on six threads; they touch disjoint buffers, so running them in sequence gives the
 * same result), then the five swaps.  Returns the decoder's output_buffer (a frame surfaces 5 calls
 * after the call that delivered its last sample).  The result stays valid until the next call. */
const fo_payloads *fo_receiver_chain_process_samples(fo_receiver_chain *, const fo_c64 *in, size_t n);
/* Same but every block works on a dedicated thread like receiver_chain.cpp:58-95 (used for timing). */
fo_receiver_chain *fo_receiver_chain_new_threaded(void);
const uint64_t *fo_receiver_chain_decoder_stats(const fo_receiver_chain *);

/* ---- the hot path in isolation (what the GPU batch entry point computes) ---- */
typedef struct {
    int64_t lts1_pos;      /* stream index of the sample tagged LTS1 (timing_sync.cpp:105) */
    int64_t rot_start;     /* samples at index >= rot_start are rotated by (c,s), earlier ones by (c_prev,s_prev) */
    double c, s;           /* cos/sin of m_phase_acc after the LTS was found (timing_sync.cpp:114-125) */
    double c_prev, s_prev; /* same for the phase in force before */
} fo_frame_desc;

typedef struct {
    int32_t status;        /* FO_ST_* */
    int32_t rate, length, num_symbols;
} fo_frame_result;

enum { FO_ST_OK = 0, FO_ST_HEADER_FAIL = 1, FO_ST_CRC_FAIL = 2, FO_ST_TRUNCATED = 3, FO_ST_SUPERSEDED = 5 };     /* (4 is the device's NO_SPACE) */

/* Decode one alignment: fft_symbols -> channel_est -> phase_tracker -> frame_decoder on the float
 * samples iq[2*n] (interleaved re,im; widened to double exactly like the CPU receiver would), where
 * the LTS1/LTS2 tags sit at d->lts1_pos / +64 and samples end at index `end` (exclusive; the next
 * alignment's LTS1 or the end of the buffer).  psdu must hold 4095 bytes.
 * taps (any may be NULL): hinv[64], eq[(1+num_symbols)*48] derotated carriers incl. SIGNAL,
 * soft[2*num_symbols*dbps], fftout[(3+num_symbols)*64] incl. both LTS. */
void fo_decode_alignment_f32(const float *iq, int64_t end, const fo_frame_desc *d, uint8_t *psdu,
                             fo_frame_result *res, fo_c64 *hinv, fo_c64 *eq, uint8_t *soft, fo_c64 *fftout);

/* Host-side sync over a whole float stream: runs frame_detector + timing_sync (chunk = 4096 like
 * receiver.h:16) and returns up to cap alignment descriptors in stream order. */
size_t fo_find_alignments_f32(const float *iq, int64_t n, fo_frame_desc *out, size_t cap);

/* The batch path's checker: decode n_frames alignments of one stream (`threads` worker threads, each alignment on its own as above),
 * then decide the alignments that are cut short by the next one with what follows them (fo_decode_batch_v2_f32's rules) -- the same
 * results as fo_decode_batch_v2_f32 with n_ctx = 0. */
void fo_decode_batch_f32(const float *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends,
                         size_t n_frames, uint8_t *psdu, size_t slot_bytes, fo_frame_result *res, int threads);

/* The blocks behind timing_sync (fft_symbols .. frame_decoder) fed with the tagged, rotated stream timing_sync WOULD produce for a
 * given list of alignments: the ground truth for a batch decoder handed the same descriptors -- the partial-vector flush of
 * fft_symbols.cpp:41-50 and frame_decoder's frame-in-progress logic (frame_decoder.cpp:52-88) included. */
void fo_chain_from_tags_f32(const float *iq, int64_t n, const fo_frame_desc *descs, size_t n_al, fo_payloads *out);
/* ... and those blocks restated per alignment, the way the device's batch path is organised: every alignment's vectors, the partial
 * one included, form one sequence with those of the alignments it is LINKED to (ends[j] == descs[j+1].lts1_pos: the stream goes on into
 * the next alignment); a frame takes the nsym vectors behind its SIGNAL from wherever they come and is dropped by a valid SIGNAL among
 * them.  ends may be NULL (alignment j ends at alignment j+1's lts1_pos, the last one at n).  descs holds n_al + n_ctx alignments: the
 * last n_ctx are CONTEXT -- sources of vectors and of superseding SIGNALs for the frames before them, not decoded themselves (what a
 * caller that cuts a stream into batches hands over behind a batch).  Status: FO_ST_TRUNCATED = the samples ended (an unlinked end)
 * before the LTS windows, the SIGNAL vector or the frame's last vector; FO_ST_SUPERSEDED = a later alignment took the stream over (LTS
 * or SIGNAL window cut by the next LTS1, or a valid SIGNAL before the frame's last vector); rate / length / num_symbols as SIGNAL
 * announced them wherever it decoded. */
void fo_decode_batch_v2_f32(const float *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_al, size_t n_ctx, uint8_t *psdu,
                            size_t slot_bytes, fo_frame_result *res);
/* The same on complex<double> samples that timing_sync has rotated already (its output_buffer: timing_sync.cpp:114-125); the
 * descriptors' phasors are not applied. */
void fo_decode_batch_v2_f64(const double *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_al, size_t n_ctx, uint8_t *psdu,
                            size_t slot_bytes, fo_frame_result *res);

/* ---- the TIMED CPU baseline of bench.py (never the checker: the functions above are) ----
 * fo_viterbi_forward_simd: fo_viterbi_forward on sixteen butterflies per SSE instruction, the way the reference's own decoder
 * (viterbi.cpp:208-457) is laid out; identical decision words and metrics (asserted in tests/).  metrics may be NULL.
 * fo_pool: pre-spawned workers with per-thread scratch decoding a batch of alignments like fo_decode_batch_f32, with the SIMD
 * forward pass and no allocation per frame; identical results (asserted in tests/ and by bench.py on every timed sample). */
void fo_viterbi_forward_simd(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics);
const char *fo_viterbi_simd_kind(void);
/* 1: fo_conv_decode (and with it the block chain's frame_decoder) uses the SSE forward pass -- for TIMED legs only; 0 (default): the scalar model */
void fo_set_timed_simd_viterbi(int on);
typedef struct fo_pool fo_pool;
fo_pool *fo_pool_new(int threads);
void fo_pool_free(fo_pool *);
int fo_pool_threads(const fo_pool *);
void fo_pool_decode(fo_pool *, const float *iq, const fo_frame_desc *descs, const int64_t *ends, size_t n_frames,
                    uint8_t *psdu, size_t slot_bytes, fo_frame_result *res);

#ifdef __cplusplus
}
#endif
#endif
