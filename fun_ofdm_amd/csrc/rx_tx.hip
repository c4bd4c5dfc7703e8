// rx_tx.hip -- the transmit side on the device: frame_builder::build_frame and the synthetic channel of SURVEY 8d,
// so that large synthetic workloads and loop-back tests live in HBM.
#include "rx_handle.h"
#include "tx_kernels.h"

using namespace foa;

int foa::upload_tables_tx(const DeviceTables &t)
{
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &t, sizeof t));
    return FOA_OK;
}

// (foa_rx_record_consumed / _done cover the transmit calls too: they read and write caller buffers on the handle's first stream)
static int tx_queued(foa_rx *rx)
{
    if (!rx->tx_done) HIP_TRY(hipEventCreateWithFlags(&rx->tx_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(rx->tx_done, rx->stream));
    rx->tx_used = true;
    return FOA_OK;
}

extern "C" {

int foa_tx_build_frames_dev(foa_rx *rx, const uint8_t *d_payloads, size_t payload_pitch, int length, int rate, size_t n_frames,
                            double *d_frames, size_t *frame_samples)
{
    if (!rx || !frame_samples) return fail(FOA_E_INVALID, "NULL argument");
    if (rate < 0 || rate >= kNumRates || length < 0 || length > 4095) return fail(FOA_E_INVALID, "bad rate/length");
    DeviceTables tab;
    build_tables(&tab);
    const int dbps = tab.rates[rate].dbps, nsym = (16 + 8 * (length + 4) + 6 + dbps - 1) / dbps, nbytes = nsym * dbps / 8;
    *frame_samples = 320 + (size_t)80 * (nsym + 1);
    if (n_frames == 0) return FOA_OK;
    if ((!d_payloads && length > 0) || !d_frames) return fail(FOA_E_INVALID, "NULL device pointer");
    if (payload_pitch < (size_t)length) return fail(FOA_E_INVALID, "payload_pitch smaller than length");
    if (n_frames > 0x7FFFFFF0u / (size_t)(nsym + 1)) return fail(FOA_E_INVALID, "too many frames for one call");
    HIP_TRY(enter_device(rx->device));
    const size_t stride = ((size_t)nbytes + 1 + 15) & ~(size_t)15;
    int rc = rx->scratch.ensure(n_frames * stride);
    if (rc) return rc;
    hipStream_t st = rx->stream;
    const int nf = (int)n_frames;
    if ((rc = wait_after(rx, st))) return rc;
    hipLaunchKernelGGL(k_tx_prepare, dim3((nf + 63) / 64), dim3(64), 0, st, d_payloads, payload_pitch, length, nf, nbytes, rx->scratch.p, stride);
    const int64_t threads = (int64_t)nf * (nsym + 1);
    hipLaunchKernelGGL(k_tx_symbols, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, rx->scratch.p, stride, length, rate, nsym, nf,
                       (double2 *)d_frames, *frame_samples);
    HIP_TRY(hipGetLastError());
    return tx_queued(rx);
}

int foa_tx_channel_dev(foa_rx *rx, const double *d_frames, size_t n_frames, size_t frame_samples, size_t pitch, size_t lead, double snr_db,
                       double cfo_hz, uint64_t seed, float *d_iq)
{
    if (!rx) return fail(FOA_E_INVALID, "rx is NULL");
    if (n_frames == 0) return FOA_OK;
    if (!d_frames || !d_iq) return fail(FOA_E_INVALID, "NULL device pointer");
    if (lead + frame_samples > pitch) return fail(FOA_E_INVALID, "lead + frame_samples exceeds the pitch");
    HIP_TRY(enter_device(rx->device));
    // sigma^2 per real component = P_ref / (2 10^(SNR/10)), P_ref = 0.0124
    const double sigma = std::sqrt(0.0124 / (2.0 * std::pow(10.0, snr_db / 10.0)));
    const int64_t total = (int64_t)n_frames * (int64_t)pitch;
    { int rc = wait_after(rx, rx->stream); if (rc) return rc; }
    hipLaunchKernelGGL(k_tx_channel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, rx->stream, (const double2 *)d_frames, (int64_t)n_frames,
                       (int64_t)frame_samples, (int64_t)pitch, (int64_t)lead, sigma, cfo_hz, seed, (float2 *)d_iq);
    HIP_TRY(hipGetLastError());
    return tx_queued(rx);
}

}  // extern "C"
