// frontend_kernels.h -- the first two stages of a decode call, for gfx950 (wave64):
//   k_header      per alignment: the two LTS windows -> channel estimate (channel_est.cpp:44-58), the SIGNAL window -> equalised,
//                 derotated, demapped, decoded (fft_symbols.cpp:33-79 + fft.cpp:50-59, channel_est.cpp:77-81, phase_tracker.cpp:70-104,
//                 ppdu.cpp:168-218) -> one FrameInfo record.  One wave per alignment, lane = FFT point.
//   k_scan_*      exclusive scans over the records: where each frame's symbols, soft pairs, decisions and chain-back segments live
//                 in the call's work set, and the symbol -> frame and segment -> frame maps.
// The data symbols themselves are frontend_q4.h.
#pragma once

#include "signal_decode.h"

namespace foa {

// =================================================================================================
// K1: per alignment: LTS1 + LTS2 + SIGNAL.  Channel estimate -> hinv, SIGNAL decode -> FrameInfo.
// One wave (64 threads) per block, one block per frame.
// =================================================================================================
__global__ __launch_bounds__(64) void k_header(const float2 *__restrict__ iq, const foa_frame_desc *__restrict__ descs,
                                               const int64_t *__restrict__ ends, int64_t n_samples, int n_frames,
                                               FrameInfo *__restrict__ info, double2 *__restrict__ hinv, double2 *__restrict__ eq_tap)
{
    __shared__ cpx lds[64];
    __shared__ uint8_t dem[48];
    __shared__ uint64_t decs[24];
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= n_frames) return;
    const foa_frame_desc d = descs[f];
    const int64_t end = min(ends[f], n_samples), p = d.lts1_pos;      // nothing is read beyond the stream, whatever the caller's ends say
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0;
    fi.soft_off = 0; fi.dec_off = 0;
    if (p < 0 || p + 208 > end) {                     // LTS or SIGNAL window cut off
        fi.status = FOA_ST_TRUNCATED;
        if (lane == 0) info[f] = fi;
        return;
    }
    const int s = lane_subcarrier(lane);
    // channel_est.cpp:44-58: est = sum over the two LTS of (LTS_FREQ_DOMAIN / Y) / 2
    cpx est = { 0.0, 0.0 };
    const cpx ref = { (double)g_tab.lts_freq[s], 0.0 };
#pragma unroll
    for (int w = 0; w < 2; w++) {
        cpx y = fft64_lane(load_rotated(iq, p + 64 * w + lane, d), lds, lane);
        cpx q = cdiv(ref, y);
        est.x += q.x / 2.0;
        est.y += q.y / 2.0;
    }
    hinv[(size_t)f * 64 + s] = make_double2(est.x, est.y);
    // SIGNAL: equalise (channel_est.cpp:77-81), pilot phase with polarity[0] (phase_tracker.cpp:74-99)
    cpx y = fft64_lane(load_rotated(iq, p + 144 + lane, d), lds, lane);
    cpx z = pilot_derotate(cmul(est, y), (int)g_tab.polarity[0]);
    const int di = g_tab.data_index[s];
    if (eq_tap && di >= 0) eq_tap[(size_t)f * 48 + di] = make_double2(z.x, z.y);   // SIGNAL tap: one row per frame
    // ppdu.cpp:171-175: BPSK demap, deinterleave
    if (di >= 0) {
        uint8_t b;
        qam_decode(z.x, 1, 128.0, &b);
        dem[deinterleaved_pos(di)] = b;
    }
    __syncthreads();
    int rate, length, nsym;
    decode_signal_bits(dem, decs, lane, rate, length, nsym);
    if (lane == 0) {
        if (rate >= 0) {
            fi.rate = rate; fi.length = length;
            if (p + 144 + 80 * (int64_t)nsym + 64 > end) { fi.status = FOA_ST_TRUNCATED; fi.nsteps = -nsym; }   // nsym still reported
            else { fi.status = FOA_ST_CRC_FAIL; fi.nsym = nsym; fi.nsteps = nsym * g_tab.rates[rate].dbps; }   // pending until the CRC is checked
        }
        info[f] = fi;
    }
}

// =================================================================================================
// K2: exclusive scans over frames -> sym_off / soft_off / dec_off / seg_off, and the symbol / segment maps.
// =================================================================================================

// Three small kernels instead of one block: a single CU's memory pipeline (one 64-line request per wave instruction)
// made the one-block version the 50 us tail of a 2 ms call.  One thread per frame throughout.
constexpr int kScanBlock = 256;

struct ScanQ { int64_t v[4]; };          // data symbols, (unused), per-step words (soft pairs / decisions / decoded), chain-back segments

__device__ __forceinline__ ScanQ scan_quantities(const FrameInfo *info, int f, int n_frames, int seg_steps)
{
    ScanQ q = { { 0, 0, 0, 0 } };
    if (f < n_frames) {
        const int nsym = info[f].nsym, nsteps = nsym > 0 ? info[f].nsteps : 0;
        q.v[0] = nsym; q.v[1] = 0; q.v[2] = dec_words(nsteps); q.v[3] = tb_segments(nsteps, seg_steps);
    }
    return q;
}

// exclusive scan over the block's threads; totals in tot[] (valid in every thread)
__device__ __forceinline__ ScanQ block_exclusive_scan(const ScanQ &q, int64_t (&tot)[4], int64_t (*part)[kScanBlock / 64])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ScanQ r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int64_t x = q.v[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) part[i][wv] = x;
        r.v[i] = x - q.v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int64_t run = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; w++) {
            if (w == wv) r.v[i] += run;
            run += part[i][w];
        }
        tot[i] = run;
    }
    return r;
}

// pass 1: per-block sums -> blk[4][n_blocks]
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(const FrameInfo *__restrict__ info, int n_frames, int seg_steps, int64_t *__restrict__ blk)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    int64_t tot[4];
    block_exclusive_scan(scan_quantities(info, blockIdx.x * kScanBlock + threadIdx.x, n_frames, seg_steps), tot, part);
    if (threadIdx.x < 4) blk[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = tot[threadIdx.x];
}

// pass 2: exclusive scan of the block sums in place, grand totals -> totals[0..2] and totals[4]  (one block)
__global__ __launch_bounds__(1024) void k_scan_blocks(int64_t *__restrict__ blk, int n_blocks, int64_t sym_cap, int64_t *__restrict__ totals)
{
    if (threadIdx.x == 0) totals[3] = sym_cap;            // read by the data-symbol kernels next to totals[0]
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = 0; i < 4; i++) {
        int64_t carry = 0;
        for (int base = 0; base < n_blocks; base += 1024) {
            const int j = base + tid;
            const int64_t v = j < n_blocks ? blk[(size_t)i * n_blocks + j] : 0;
            int64_t x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int64_t y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            __syncthreads();
            if (lane == 63) wsum[wv] = x;
            __syncthreads();
            int64_t before = 0, all = 0;
            for (int w = 0; w < 16; w++) {
                if (w < wv) before += wsum[w];
                all += wsum[w];
            }
            if (j < n_blocks) blk[(size_t)i * n_blocks + j] = carry + before + x - v;
            carry += all;
        }
        if (tid == 0) totals[i < 3 ? i : 4] = carry;          // totals[3] is the symbol capacity (above)
    }
}

// pass 2 for up to 64 x 64 blocks (a million frames) as ONE wave: under a forward pass a 1024-thread block waits for sixteen
// free wave slots on one CU (5.8 us alone, 100 us on average and up to 750 us in the pipelined bench), a lone wave for one
__global__ __launch_bounds__(64) void k_scan_blocks_w(int64_t *__restrict__ blk, int n_blocks, int64_t sym_cap, int64_t *__restrict__ totals)
{
    const int lane = threadIdx.x, per = (n_blocks + 63) / 64, lo = lane * per, hi = min(lo + per, n_blocks);
    if (lane == 0) totals[3] = sym_cap;
    for (int i = 0; i < 4; i++) {
        int64_t s = 0;
        for (int j = lo; j < hi; j++) s += blk[(size_t)i * n_blocks + j];
        int64_t x = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        int64_t run = x - s;                                   // exclusive prefix of this lane's run
        for (int j = lo; j < hi; j++) { const int64_t v = blk[(size_t)i * n_blocks + j]; blk[(size_t)i * n_blocks + j] = run; run += v; }
        if (lane == 63) totals[i < 3 ? i : 4] = x;
    }
}

// pass 3: offsets into the frame records (frames that do not fit are marked FOA_ST_NO_SPACE), and the symbol -> frame
// and segment -> frame maps (so that the data-symbol and chain-back kernels find their frame without a search;
// frames are short: <= 1368 symbols)
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(FrameInfo *__restrict__ info, int n_frames, int64_t sym_cap,
                                                           int64_t dec_cap, int seg_steps, int64_t seg_cap, const int64_t *__restrict__ blk,
                                                           int32_t *__restrict__ sym2frame, int32_t *__restrict__ seg2frame)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    const int f = blockIdx.x * kScanBlock + threadIdx.x;
    const ScanQ q = scan_quantities(info, f, n_frames, seg_steps);
    int64_t tot[4];
    const ScanQ ex = block_exclusive_scan(q, tot, part);
    if (f >= n_frames) return;
    const int64_t a = ex.v[0] + blk[blockIdx.x],
                  c = ex.v[2] + blk[(size_t)2 * gridDim.x + blockIdx.x], d = ex.v[3] + blk[(size_t)3 * gridDim.x + blockIdx.x];
    const int nsym = (int)q.v[0];
    info[f].seg_off = (int32_t)d;
    const int ns = (int)q.v[3];
    if (nsym > 0 && (a + nsym > sym_cap || c + q.v[2] > dec_cap)) {
        // keeps its slots in the numbering; they are marked unused (as far as the maps reach)
        info[f].status = FOA_ST_NO_SPACE; info[f].nsteps = -nsym; info[f].nsym = 0;
        for (int k = 0; k < nsym; k++) if (a + k < sym_cap) sym2frame[a + k] = -1;
        for (int k = 0; k < ns; k++) if (d + k < seg_cap) seg2frame[d + k] = -1;
        return;
    }
    info[f].sym_off = (int32_t)a; info[f].soft_off = 2 * c; info[f].dec_off = c;    // (soft_off: byte offset of the frame's soft pairs)
    for (int k = 0; k < nsym; k++) sym2frame[a + k] = f;
    for (int k = 0; k < ns; k++) if (d + k < seg_cap) seg2frame[d + k] = f;
}

}  // namespace foa
