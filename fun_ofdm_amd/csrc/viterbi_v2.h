// viterbi_v2.h -- packed multi-frame-per-wave Viterbi (under construction: forwards to v1 for now).
#pragma once

#include "viterbi_v1.h"

namespace foa {

inline void launch_viterbi_v2(hipStream_t st, const FrameInfo *info, int nf, const uint8_t *soft, uint64_t *dec, uint8_t *psdu,
                              size_t slot_bytes, foa_frame_result *results)
{
    hipLaunchKernelGGL(k_viterbi_v1, dim3(nf), dim3(64), 0, st, info, nf, soft, dec, psdu, slot_bytes, results);
}

}  // namespace foa
