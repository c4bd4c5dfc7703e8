// foa_common.h -- shared declarations of the gfx950 receive path (device + host side).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fun_ofdm_amd.h"

namespace foa {

constexpr int kNumRates = 11;
constexpr int kWave = 64;
constexpr int kMaxDecodedBytes = 4128;   // num_data_bytes <= 4104 for length <= 4095 at any rate

// Per-frame working record kept in HBM between the kernels of one decode call.
struct FrameInfo {
    int32_t status;      // FOA_ST_*; frames whose header failed keep nsym = 0
    int32_t rate;
    int32_t length;
    int32_t nsym;        // data symbols to process (0 if nothing to do)
    int32_t sym_off;     // first data symbol's index in the call-wide symbol numbering
    int32_t nsteps;      // trellis steps = nsym * dbps
    int64_t soft_off;    // byte offset of this frame's depunctured soft bytes (= 2 * dec_off: two per trellis step)
    int64_t dec_off;     // offset (in per-step elements) of this frame's region in the soft-pair, decision and decoded buffers
    int32_t seg_off;     // first chain-back segment's index in the call-wide segment numbering (viterbi_tb.h)
    int32_t reserved_;
};

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
