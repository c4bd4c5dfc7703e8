// rx_stage.hip -- one entry point per replaced fun::block (include/fun_ofdm_amd.h, "stage-level entry points"), for the per-stage
// adaptors of include/fun_ofdm_amd/blocks.hpp: host buffers in the reference's own element order in and out, synchronous.
// foa_conv_decode and foa_decode_data_f64 run the batch path's own Viterbi kernels (rx_decode.hip) on records built on the host.
#include <algorithm>

#include "rx_handle.h"
#include "stage_kernels.h"

using namespace foa;

int foa::upload_tables_stage(const DeviceTables &t)
{
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &t, sizeof t));
    return FOA_OK;
}

extern "C" {

int foa_fft_forward_f64(foa_rx *rx, double *vectors, size_t n_vec)
{
    if (!rx || !vectors) return fail(FOA_E_INVALID, "NULL argument");
    if (n_vec == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    size_t bytes = n_vec * 64 * sizeof(double2);
    int rc = rx->scratch.ensure(bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rx->scratch.p, vectors, bytes, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_fft_vectors, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, rx->stream, (double2 *)rx->scratch.p, (int)n_vec);
    HIP_TRY(hipMemcpyAsync(vectors, rx->scratch.p, bytes, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_conv_decode(foa_rx *rx, const uint8_t *symbols, uint8_t *data, int data_bits, size_t n_blocks)
{
    if (!rx || !symbols || !data) return fail(FOA_E_INVALID, "NULL argument");
    if (data_bits < 1 || data_bits > 8 * (kMaxDecodedBytes - 8)) return fail(FOA_E_INVALID, "data_bits out of range");
    if (n_blocks == 0) return FOA_OK;
    if (n_blocks > 0xFFFFu) return fail(FOA_E_INVALID, "at most 65535 blocks per call");
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // this entry point reuses the work set of the batch calls
    const size_t nsteps = (size_t)data_bits + 6, sym_bytes = n_blocks * 2 * nsteps, nbytes = (size_t)((data_bits + 7) / 8), out_bytes = n_blocks * nbytes;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    int rc = rx->scratch.ensure(up(sym_bytes) + up(out_bytes));
    if (rc) return rc;
    uint8_t *d_sym = rx->scratch.p, *d_out = rx->scratch.p + up(sym_bytes);
    hipStream_t st = rx->stream;
    HIP_TRY(hipMemcpyAsync(d_sym, symbols, sym_bytes, hipMemcpyHostToDevice, st));
    {
        // The kernels of the batch path (k_viterbi_fwd3 + k_tb_walk + k_tb_finish), fed the way the front end feeds them: one frame
        // record and one region of soft pairs per block.
        // viterbi.cpp:209 drops an odd last step: its decision word stays zero (viterbi.cpp:193-194), so the chain-back reads
        // bit data_bits-1 as 0 and stays in state 0 -- the same as decoding one bit less and appending a zero.
        const int T = 2 * (int)(nsteps / 2), N = T - 6;
        HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, st));
        if (N > 0) {
            std::vector<FrameInfo> info(n_blocks);
            std::vector<int32_t> seg2frame;
            const int64_t words = dec_words(T);
            const int nseg = tb_segments(T, rx->tb_segment);
            for (size_t b = 0; b < n_blocks; b++) {
                FrameInfo &fi = info[b];
                memset(&fi, 0, sizeof fi);
                fi.status = FOA_ST_CRC_FAIL; fi.rate = 0; fi.length = 0; fi.nsym = 1; fi.hdr_nsym = 1; fi.nsteps = T; fi.fresh = -1;
                fi.dec_off = (int64_t)b * words; fi.seg_off = (int32_t)(b * (size_t)nseg);
                seg2frame.insert(seg2frame.end(), (size_t)nseg, (int32_t)b);
            }
            const size_t total = n_blocks * (size_t)words + 64;
            if ((rc = rx->w->info.ensure(n_blocks + 1)) || (rc = rx->w->dec.ensure(total)) || (rc = rx->w->sp.ensure(total)) || (rc = rx->w->decoded.ensure(decoded_words_for(total))) ||
                (rc = rx->w->seg2frame.ensure(seg2frame.size() + 64)) || (rc = rx->w->tb_state.ensure(seg2frame.size() + 64)) || (rc = rx->w->totals.ensure(8)))
                return rc;
            HIP_TRY(hipMemcpyAsync(rx->w->info.p, info.data(), n_blocks * sizeof(FrameInfo), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(rx->w->seg2frame.p, seg2frame.data(), seg2frame.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            const int64_t n_segs = (int64_t)seg2frame.size();
            HIP_TRY(hipMemcpyAsync(rx->w->totals.p + 4, &n_segs, sizeof n_segs, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_conv_sp, dim3((unsigned)((T + 255) / 256), (unsigned)n_blocks), dim3(256), 0, st, d_sym, 2 * nsteps, T, rx->w->info.p, rx->w->sp.p);
            launch_fwd3(st, rx->w->info.p, (int)n_blocks, rx->w->sp.p, rx->w->dec.p);
            launch_finish3(st, st, rx->w->info.p, (int)n_blocks, rx->w->dec.p, rx->w->decoded.p, rx->w->seg2frame.p, rx->w->totals.p, rx->w->tb_state.p, seg2frame.size(),
                           rx->tb_segment, rx->tb_overlap, nullptr, 0, nullptr);
            hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)((nbytes + 255) / 256), (unsigned)n_blocks), dim3(256), 0, st, rx->w->decoded.p, rx->w->info.p,
                               (N + 7) / 8, (int)nbytes, d_out);
            HIP_TRY(hipStreamSynchronize(st));        // the host vectors above are the copies' sources
            rx->last_frames = 0;                      // the workspace no longer describes a decode_frames call
        }
    }
    HIP_TRY(hipMemcpyAsync(data, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return FOA_OK;
}

int foa_channel_estimate_f64(foa_rx *rx, const double *lts_pairs, double *hinv, size_t n)
{
    if (!rx || !lts_pairs || !hinv) return fail(FOA_E_INVALID, "NULL argument");
    if (n == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    const size_t in_b = n * 128 * sizeof(double2), out_b = n * 64 * sizeof(double2);
    int rc = rx->scratch.ensure(in_b + out_b);
    if (rc) return rc;
    double2 *d_in = (double2 *)rx->scratch.p, *d_out = (double2 *)(rx->scratch.p + in_b);
    HIP_TRY(hipMemcpyAsync(d_in, lts_pairs, in_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_chanest, dim3((unsigned)n), dim3(64), 0, rx->stream, d_in, d_out, (int)n);
    HIP_TRY(hipMemcpyAsync(hinv, d_out, out_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_equalize_f64(foa_rx *rx, double *vectors, size_t n_vec, const double *hinv, size_t n_hinv, const int32_t *hinv_index)
{
    if (!rx || !vectors || !hinv || !hinv_index) return fail(FOA_E_INVALID, "NULL argument");
    if (n_vec == 0) return FOA_OK;
    for (size_t i = 0; i < n_vec; i++)
        if (hinv_index[i] < 0 || (size_t)hinv_index[i] >= n_hinv) return fail(FOA_E_INVALID, "hinv_index[%zu] out of range", i);
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t v_b = n_vec * 64 * sizeof(double2), h_b = n_hinv * 64 * sizeof(double2), i_b = n_vec * sizeof(int32_t);
    int rc = rx->scratch.ensure(up(v_b) + up(h_b) + up(i_b));
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    HIP_TRY(hipMemcpyAsync(b, vectors, v_b, hipMemcpyHostToDevice, rx->stream));
    HIP_TRY(hipMemcpyAsync(b + up(v_b), hinv, h_b, hipMemcpyHostToDevice, rx->stream));
    HIP_TRY(hipMemcpyAsync(b + up(v_b) + up(h_b), hinv_index, i_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_equalize, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, rx->stream, (double2 *)b, (int)n_vec,
                       (const double2 *)(b + up(v_b)), (const int32_t *)(b + up(v_b) + up(h_b)));
    HIP_TRY(hipMemcpyAsync(vectors, b, v_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_phase_track_f64(foa_rx *rx, const double *vectors, const int32_t *symbol_count, size_t n_vec, double *out48)
{
    if (!rx || !vectors || !symbol_count || !out48) return fail(FOA_E_INVALID, "NULL argument");
    if (n_vec == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t v_b = n_vec * 64 * sizeof(double2), c_b = n_vec * sizeof(int32_t), o_b = n_vec * 48 * sizeof(double2);
    int rc = rx->scratch.ensure(up(v_b) + up(c_b) + up(o_b));
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    HIP_TRY(hipMemcpyAsync(b, vectors, v_b, hipMemcpyHostToDevice, rx->stream));
    HIP_TRY(hipMemcpyAsync(b + up(v_b), symbol_count, c_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_phase, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, rx->stream, (const double2 *)b,
                       (const int32_t *)(b + up(v_b)), (int)n_vec, (double2 *)(b + up(v_b) + up(c_b)));
    HIP_TRY(hipMemcpyAsync(out48, b + up(v_b) + up(c_b), o_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_decode_header_f64(foa_rx *rx, const double *carriers48, size_t n, foa_frame_result *results)
{
    if (!rx || !carriers48 || !results) return fail(FOA_E_INVALID, "NULL argument");
    if (n == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // rx->scratch may still be read by a call in flight
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t c_b = n * 48 * sizeof(double2), r_b = n * sizeof(foa_frame_result);
    int rc = rx->scratch.ensure(up(c_b) + up(r_b));
    if (rc) return rc;
    uint8_t *b = rx->scratch.p;
    HIP_TRY(hipMemcpyAsync(b, carriers48, c_b, hipMemcpyHostToDevice, rx->stream));
    hipLaunchKernelGGL(k_stage_header, dim3((unsigned)n), dim3(64), 0, rx->stream, (const double2 *)b, (int)n, (foa_frame_result *)(b + up(c_b)));
    HIP_TRY(hipMemcpyAsync(results, b + up(c_b), r_b, hipMemcpyDeviceToHost, rx->stream));
    HIP_TRY(hipStreamSynchronize(rx->stream));
    return FOA_OK;
}

int foa_decode_data_f64(foa_rx *rx, const double *carriers, const uint64_t *carrier_off, size_t n_frames, foa_frame_result *results,
                        uint8_t *psdu, size_t slot_bytes)
{
    if (!rx || !carriers || !carrier_off || !results || !psdu) return fail(FOA_E_INVALID, "NULL argument");
    if (n_frames == 0) return FOA_OK;
    HIP_TRY(enter_device(rx->device));
    { int rc0 = drain(rx); if (rc0) return rc0; }                     // this entry point reuses the work set of the batch calls
    DeviceTables tab;
    build_tables(&tab);
    // frame records and offsets on the host (what k_header + the k_scan_* kernels produce in the fused path)
    std::vector<FrameInfo> info(n_frames);
    std::vector<int32_t> sym2frame, seg2frame;
    std::vector<int64_t> coff(n_frames + 1);
    int64_t dec_off = 0;
    for (size_t f = 0; f < n_frames; f++) {
        const int rate = results[f].rate, len = results[f].length;
        if (rate < 0 || rate >= kNumRates || len < 0 || len > 4095) return fail(FOA_E_INVALID, "frame %zu: bad rate/length", f);
        const int dbps = tab.rates[rate].dbps, nsym = (16 + 8 * (len + 4) + 6 + dbps - 1) / dbps;
        if (carrier_off[f + 1] - carrier_off[f] != (uint64_t)nsym * 48) return fail(FOA_E_INVALID, "frame %zu: needs %d carriers", f, nsym * 48);
        FrameInfo &fi = info[f];
        memset(&fi, 0, sizeof fi);
        fi.status = FOA_ST_CRC_FAIL; fi.rate = rate; fi.length = len; fi.nsym = nsym; fi.hdr_nsym = nsym; fi.n_own = nsym; fi.fresh = -1;
        fi.sym_off = (int32_t)sym2frame.size(); fi.nsteps = nsym * dbps; fi.dec_off = dec_off;
        fi.seg_off = (int32_t)seg2frame.size();
        seg2frame.insert(seg2frame.end(), (size_t)tb_segments(fi.nsteps, rx->tb_segment), (int32_t)f);
        dec_off += dec_words(fi.nsteps);
        coff[f] = (int64_t)carrier_off[f];
        sym2frame.insert(sym2frame.end(), (size_t)nsym, (int32_t)f);
        results[f].num_symbols = nsym;
    }
    coff[n_frames] = (int64_t)carrier_off[n_frames];
    const size_t n_sym = sym2frame.size(), n_car = (size_t)carrier_off[n_frames];
    int rc;
    if ((rc = rx->w->info.ensure(n_frames + 1)) || (rc = rx->w->sym2frame.ensure(n_sym + 1)) ||
        (rc = rx->w->dec.ensure((size_t)dec_off + 64)) || (rc = rx->w->sp.ensure((size_t)dec_off + 64)) || (rc = rx->w->decoded.ensure(decoded_words_for((size_t)dec_off + 64))) ||
        (rc = rx->w->seg2frame.ensure(seg2frame.size() + 64)) || (rc = rx->w->tb_state.ensure(seg2frame.size() + 64)) || (rc = rx->w->totals.ensure(8)))
        return rc;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t c_b = n_car * sizeof(double2), o_b = (n_frames + 1) * sizeof(int64_t), p_b = n_frames * slot_bytes, r_b = n_frames * sizeof(foa_frame_result);
    if ((rc = rx->scratch.ensure(up(c_b) + up(o_b) + up(p_b) + up(r_b)))) return rc;
    uint8_t *b = rx->scratch.p;
    uint8_t *d_psdu = b + up(c_b) + up(o_b);
    foa_frame_result *d_res = (foa_frame_result *)(d_psdu + up(p_b));
    hipStream_t st = rx->stream;
    HIP_TRY(hipMemcpyAsync(b, carriers, c_b, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + up(c_b), coff.data(), o_b, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rx->w->info.p, info.data(), n_frames * sizeof(FrameInfo), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rx->w->sym2frame.p, sym2frame.data(), n_sym * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const int64_t n_segs = (int64_t)seg2frame.size();
    HIP_TRY(hipMemcpyAsync(rx->w->seg2frame.p, seg2frame.data(), seg2frame.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(rx->w->totals.p + 4, &n_segs, sizeof n_segs, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_psdu, 0, p_b, st));
    hipLaunchKernelGGL(k_stage_demap, dim3((unsigned)((n_sym + kSymWaves - 1) / kSymWaves)), dim3(64 * kSymWaves), 0, st, (const double2 *)b,
                       (const int64_t *)(b + up(c_b)), rx->w->info.p, rx->w->sym2frame.p, (int)n_sym, rx->w->sp.p);
    {
        launch_fwd3(st, rx->w->info.p, (int)n_frames, rx->w->sp.p, rx->w->dec.p);
        launch_finish3(st, st, rx->w->info.p, (int)n_frames, rx->w->dec.p, rx->w->decoded.p, rx->w->seg2frame.p, rx->w->totals.p, rx->w->tb_state.p, seg2frame.size(),
                       rx->tb_segment, rx->tb_overlap, d_psdu, slot_bytes, d_res);
    }
    HIP_TRY(hipMemcpyAsync(psdu, d_psdu, p_b, hipMemcpyDeviceToHost, st));
    std::vector<foa_frame_result> out(n_frames);
    HIP_TRY(hipMemcpyAsync(out.data(), d_res, r_b, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (size_t f = 0; f < n_frames; f++) results[f].status = out[f].status;
    rx->last_frames = 0;      // the workspace no longer describes a decode_frames call
    return FOA_OK;
}

}  // extern "C"
