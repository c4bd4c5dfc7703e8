// foa_common.h -- shared declarations of the gfx950 receive path (device + host side).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fun_ofdm_amd.h"
#include "../../include/fun_ofdm_amd_diag.h"

namespace foa {

constexpr int kNumRates = 11;
constexpr int kWave = 64;
constexpr int kMaxDecodedBytes = 4128;   // num_data_bytes <= 4104 for length <= 4095 at any rate

// Per-alignment working record kept in HBM between the kernels of one decode call.
struct FrameInfo {
    int32_t status;      // FOA_ST_*; alignments that decode nothing keep nsym = 0
    int32_t rate;        // Rate enum the SIGNAL vector announced, -1 if it did not decode (or there was none)
    int32_t length;
    int32_t nsym;        // data symbols to process (0 if nothing to do)
    int32_t sym_off;     // first data symbol's index in the call-wide symbol numbering
    int32_t nsteps;      // trellis steps = nsym * dbps
    int64_t dec_off;     // offset (in per-step elements) of this frame's region in the soft-pair, decision and decoded buffers
    int32_t seg_off;     // first chain-back segment's index in the call-wide segment numbering (viterbi_tb.h)
    int32_t hdr_nsym;    // data symbols SIGNAL announced (0 if it did not decode): what foa_frame_result::num_symbols reports
    // what fft_symbols emits for this alignment behind its two LTS vectors (fft_symbols.cpp:41-73): the vectors of the complete symbol
    // windows from SIGNAL on, and -- when the next alignment's LTS1 arrives past a window's cyclic prefix -- one partly filled one
    int32_t nvec;        // vectors in all: complete windows k = 0 (SIGNAL) .. + the partial one
    int32_t fresh;       // the partial vector's own samples (the rest still holds the window before): 0 .. 63, -1 = no partial vector
    int32_t flags;       // kInfoLink | kInfoCross | kInfoLate
    int32_t spec_off;    // first entry of this frame in the table of symbols that are not plain windows of its own alignment
    int32_t n_own;       // data symbols that ARE plain windows of its own alignment: min(hdr_nsym, complete windows - 1)
    int32_t pad_;
};
constexpr int kInfoLink = 1;     // the stream goes on into the next alignment of the call (ends[f] is its LTS1): one vector sequence
constexpr int kInfoLate = 4;     // an earlier alignment's LTS2 tag sits inside this one's first LTS window: its vectors come one symbol later (frontend_kernels.h)
constexpr int kInfoCross = 2;    // valid SIGNAL, frame longer than the alignment's own complete windows, linked: decided by the scan kernels

// A data symbol of a frame that is not window k of the frame's own alignment: the partial vector, or a vector of a later alignment
// the frame fills on with (frame_decoder.cpp:52-88).
struct SpecSym {
    int32_t frame;       // the frame it belongs to (rate, output position)
    int32_t src;         // the alignment whose window, rotation, channel estimate and symbol count it takes
    int32_t k;           // vector k of that alignment (0 = its SIGNAL window)
    int32_t fresh;       // low byte 64: a complete window; < 64: samples fresh .. 63 come from the window before (fft_symbols.cpp:46-50); bit 8: src is a late alignment (kInfoLate)
};

// The stream engines' state between batches (stream_engine.h), carried on the device: the stream index of the STS_END sample of the first
// alignment not decided yet, and the phasor timing_sync had in force before it (timing_sync.cpp:113-125: m_phase_acc).
struct StreamState { int64_t lo_abs; double c, s; };
constexpr int64_t kStreamSettle = 192;     // an STS_END within this many samples of a buffer's end is left to the next batch: timing_sync looks
                                           // 160 samples ahead of it (timing_sync.cpp:69-113), frame_detector's windows 32 behind

// rates.h:52-196 as a device table
struct RateRow {
    int32_t rate_field, cbps, dbps, bpsc, punct, numbits;
    double scale_d;      // qam.h:35-51 d_scale_d
};

struct DeviceTables {
    RateRow rates[kNumRates];
    double tw_re[64], tw_im[64];   // exp(-2*pi*j*k/64)
    int8_t lts_freq[64];           // preamble.h:363 (index = subcarrier + 32)
    int8_t polarity[128];          // phase_tracker.cpp:23-32 (+1/-1), entry 127 unused
    int8_t carrier_kind[64];       // 0 null, 1 data, 2 pilot (by subcarrier index)
    int8_t data_index[64];         // subcarrier index -> 0..47 (phase_tracker.cpp:46-50), -1 otherwise
    uint8_t scramble[128];         // ppdu.cpp:256-264 feedback bit per byte index mod 127
    uint32_t crc_table[256];       // IEEE 802.3 CRC-32, reflected
    double lts_conj_re[64], lts_conj_im[64];   // preamble.h:432 LTS_TIME_DOMAIN_CONJ
    uint32_t qam_lut[641];         // qam.h:110-125 for pt = -320..320 (constant outside): soft byte i in bits 8i..8i+7
    // interleaver.cpp:28-38 + puncturer.cpp:94-118 as one map per rate: demodulated byte c = carrier * bpsc + bit of a
    // symbol -> its position among the symbol's 2 * dbps depunctured soft bytes (frontend_q4.h)
    uint16_t sym_pos[kNumRates][288];
    double preamble_re[320], preamble_im[320];   // preamble.h:24 PREAMBLE_SAMPLES (transmit side, tx_kernels.h)
};

// Filled once on the host (rx_handle.hip); every translation unit that holds kernels keeps its own copy in __constant__ memory
// (device_math.h) and uploads it when a handle is created.
void build_tables(DeviceTables *t);

}  // namespace foa
