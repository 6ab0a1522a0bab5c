// postnet.h -- the PostNet in front of the vocoder, on device.
//
// Reference: PostNet.call, src/iris/postnet.py:48-67 (ctor :16-46): over a mel [B, n_mels, T]
//   h = x;  for the first L-1 layers: h = tanh(BatchNorm(Conv1D(k, 'same')(h)))   (dropout = identity at inference)
//   res = BatchNorm(Conv1D(n_mels, k, 'same')(h));   return x + res
// BatchNorm at inference is an affine map per channel and is folded into the conv on the host
// (iris/postnet.py), so each layer is one launch of the generic MFMA conv kernel with a tanh epilogue;
// the last launch leaves the residual channels-last and a small kernel adds it to the channels-first mel.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma_f32.h"

namespace iris {

// out[b][c][t] = mel[b][c][t] + res[b][t][c]
__global__ void __launch_bounds__(256) postnet_residual_kernel(const float* __restrict__ mel,
                                                               const float* __restrict__ res,
                                                               float* __restrict__ out, int C, int T) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float* r = res + ((size_t)b * T + t) * C;
    for (int c = 0; c < C; ++c) {
        const size_t off = ((size_t)b * C + c) * T + t;
        out[off] = mel[off] + r[c];
    }
}

}  // namespace iris
