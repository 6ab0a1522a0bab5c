// conv_post.h -- last layer of the generator: LeakyReLU -> Conv1d(C->1, k) -> tanh
// (reference: src/iris/hifigan_pretrained.py:139-141, conv ctor :119-121; Keras twin
// src/iris/vocoder.py:101,127-128), with the MRF mean of the last stage fused into the read
// (hifigan_pretrained.py:137).
//
// With channels-last activations the k taps x C channels that feed one output sample are ONE
// contiguous run of k*C floats, so the layer is a sliding dot product against a k*C weight
// vector: pure bandwidth (reads C floats per sample, writes 1).  A block stages
// (256 + k - 1) rows into LDS with the activation applied (row stride odd -> the per-lane column
// walk is conflict-free), each lane then owns one output sample.  Weights are wave-uniform and
// come through the scalar cache.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma_f32.h"

namespace iris {

struct ConvPostLaunch {
    const float* x[kMaxGroup];  // n_in inputs [B, L, C]; input = lrelu((x[0]+...)/n_in)
    int n_in;
    const float* w;             // [k][C]  (w_ref[0][ci][kap] transposed)
    const float* bias;          // [1]
    float* y;                   // [B, L]
    int B, L, C, k;
    float slope;
};

constexpr int kPostTile = 256;

__global__ void __launch_bounds__(256) conv_post_tanh_kernel(const ConvPostLaunch a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = a.C, k = a.k, pad = (k - 1) / 2;
    const int S = C | 1;
    const int b = blockIdx.y, t0 = blockIdx.x * kPostTile;
    const int R = kPostTile + k - 1;
    const int total = R * C;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int r = idx / C, c = idx - r * C;
        const int row = t0 - pad + r;
        float v = 0.f;
        if (row >= 0 && row < a.L) {
            const size_t off = ((size_t)b * a.L + row) * C + c;
            v = a.x[0][off];
            for (int j = 1; j < a.n_in; ++j) v += a.x[j][off];
            if (a.n_in > 1) v = v / (float)a.n_in;
            v = lrelu1(v, a.slope);
        }
        lds[r * S + c] = v;
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t >= a.L) return;
    const float* __restrict__ w = a.w;
    float acc = a.bias[0];
    for (int kap = 0; kap < k; ++kap) {
        const float* row = lds + (threadIdx.x + kap) * S;
        for (int c = 0; c < C; ++c) acc = fmaf(row[c], w[kap * C + c], acc);
    }
    a.y[(size_t)b * a.L + t] = tanhf(acc);
}

inline hipError_t launch_conv_post(const ConvPostLaunch& a, hipStream_t stream) {
    const size_t lds_bytes = (size_t)(kPostTile + a.k - 1) * (a.C | 1) * sizeof(float);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_post_tanh_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((a.L + kPostTile - 1) / kPostTile), (unsigned)a.B), block(256);
    hipLaunchKernelGGL(conv_post_tanh_kernel, grid, block, lds_bytes, stream, a);
    return hipGetLastError();
}

}  // namespace iris
