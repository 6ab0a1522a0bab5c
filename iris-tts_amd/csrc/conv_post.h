// conv_post.h -- last layer of the generator: LeakyReLU -> Conv1d(C->1, k) -> tanh
// (reference: src/iris/hifigan_pretrained.py:139-141, conv ctor :119-121; Keras twin
// src/iris/vocoder.py:101,127-128), with the MRF mean of the last stage fused into the read
// (hifigan_pretrained.py:137) when the last MRF step has not already formed it.
//
// With channels-last activations the k taps x C channels that feed one output sample are ONE
// contiguous run of k*C values, so the layer is a sliding dot product against a k*C weight vector:
// pure bandwidth (reads C values per sample, writes 1).
//
//   conv_post_rows_kernel<C, BF16_IN>  C in {8, 16, 32, 64} (the V1 generator: C = 32).  A block stages
//       (256 + k - 1) rows into LDS as fp32 with 16-byte buffer loads (hardware range check = the zero
//       padding; the batch index is folded into blockIdx.x), activation applied on the way; row stride
//       C + 4 floats = 4 * odd, so the ds_write_b128 staging writes and the per-lane ds_read_b128 reads
//       (lane = one output sample walking k rows) are conflict-free.  No integer division by runtime values.
//   conv_post_tanh_kernel              any C (scalar staging): the fallback for generic configurations.
// Both accumulate in the same order (bias, then taps ascending, channels ascending: one fmaf chain),
// so they produce identical bits.  Weights are wave-uniform and come through the scalar cache.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iris {
namespace post {      // self-contained: included by the fp32 and the bf16 translation unit

constexpr int kMaxIn = 8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lrelu1(float v, float slope) { return v > 0.f ? v : v * slope; }

struct ConvPostLaunch {
    const void* x[kMaxIn];   // n_in inputs [B, L, C] (fp32, or bf16 for the bf16-storage path);
                                // input = lrelu((x[0]+...)/n_in)   (fp32: true division; bf16 path: * inv_n)
    int n_in;
    const float* w;             // [k][C]  (w_ref[0][ci][kap] transposed)
    const float* bias;          // [1]
    float* y;                   // [B, L]
    int B, L, C, k;
    float slope;
    float inv_n;                // bf16 path: 1 / n_in
    int tiles_per_item;         // rows kernel: ceil(L / kPostTile)
};

constexpr int kPostTile = 256;

namespace detail {
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// (a function argument, not __builtin_bit_cast on a vector element: hipcc 7.2 reads element 0 for the latter)
__device__ __forceinline__ float as_f32(unsigned w) { return __builtin_bit_cast(float, w); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
constexpr unsigned kOob = 0x80000000u;
}  // namespace detail

template <int C, bool BF16_IN>
__global__ void __launch_bounds__(256) conv_post_rows_kernel(const ConvPostLaunch a) {
    using namespace detail;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = C + 4;                       // LDS row stride in floats (16 * odd bytes)
    constexpr int EPP = BF16_IN ? 8 : 4;           // elements per 16-byte piece
    constexpr int PPR = C / EPP;                   // pieces per row (power of two)
    constexpr int ESZ = BF16_IN ? 2 : 4;
    static_assert((PPR & (PPR - 1)) == 0 && PPR >= 1, "C must be 8, 16, 32 or 64");
    const int k = a.k, pad = (k - 1) / 2;
    const int b = (int)(blockIdx.x / (unsigned)a.tiles_per_item);            // wave-uniform scalar division
    const int t0 = ((int)blockIdx.x - b * a.tiles_per_item) * kPostTile;
    const int R = kPostTile + k - 1;
    const int total = R * PPR;
    const unsigned tensor_bytes = (unsigned)a.L * (unsigned)(C * ESZ);
    const size_t boff = (size_t)b * a.L * C * ESZ;
    const int n_in = a.n_in;
    const __amdgpu_buffer_rsrc_t r0 = rsrc((const char*)a.x[0] + boff, tensor_bytes);
    const __amdgpu_buffer_rsrc_t r1 = rsrc((const char*)(n_in > 1 ? a.x[1] : a.x[0]) + boff, n_in > 1 ? tensor_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r2 = rsrc((const char*)(n_in > 2 ? a.x[2] : a.x[0]) + boff, n_in > 2 ? tensor_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r3 = rsrc((const char*)(n_in > 3 ? a.x[3] : a.x[0]) + boff, n_in > 3 ? tensor_bytes : 0u);
    const float slope = a.slope;
    const float n_f = (float)n_in;
    auto act = [&](float v0, float v1, float v2, float v3) -> float {
        float v = v0;
        if (n_in > 1) {
            v = ((v0 + v1) + v2) + v3;                       // absent inputs read 0 (zero-length descriptors)
            v = BF16_IN ? v * a.inv_n : v / n_f;             // fp32 path: true division (hifigan_pretrained.py:137)
        }
        return lrelu1(v, slope);
    };
    constexpr int U = 4;                                     // 16-byte pieces in flight per thread and input
    for (int base = 0; base < total; base += 256 * U) {
        pu32x4 v0[U], v1[U], v2[U], v3[U];
        int ldso[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256 + (int)threadIdx.x;
            const int r = idx / PPR, pc = idx & (PPR - 1);   // PPR is a compile-time power of two
            const int row = t0 - pad + r;
            const unsigned voff = (idx < total && row >= 0 && row < a.L) ? (unsigned)(row * C + EPP * pc) * ESZ : kOob;
            ldso[u] = idx < total ? r * S + EPP * pc : -1;
            v0[u] = __builtin_amdgcn_raw_buffer_load_b128(r0, (int)voff, 0, 0);
            if (n_in > 1) {                                   // uniform branch
                v1[u] = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)voff, 0, 0);
                v2[u] = __builtin_amdgcn_raw_buffer_load_b128(r2, (int)voff, 0, 0);
                v3[u] = __builtin_amdgcn_raw_buffer_load_b128(r3, (int)voff, 0, 0);
            } else {
                v1[u] = v2[u] = v3[u] = pu32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ldso[u] < 0) continue;
            if constexpr (BF16_IN) {
                f32x4 lo4, hi4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = act(bf_lo(v0[u][e]), bf_lo(v1[u][e]), bf_lo(v2[u][e]), bf_lo(v3[u][e]));
                    const float hi = act(bf_hi(v0[u][e]), bf_hi(v1[u][e]), bf_hi(v2[u][e]), bf_hi(v3[u][e]));
                    if (e < 2) { lo4[2 * e] = lo; lo4[2 * e + 1] = hi; }
                    else       { hi4[2 * (e - 2)] = lo; hi4[2 * (e - 2) + 1] = hi; }
                }
                *reinterpret_cast<f32x4*>(lds + ldso[u]) = lo4;
                *reinterpret_cast<f32x4*>(lds + ldso[u] + 4) = hi4;
            } else {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = act(as_f32(v0[u][e]), as_f32(v1[u][e]), as_f32(v2[u][e]), as_f32(v3[u][e]));
                *reinterpret_cast<f32x4*>(lds + ldso[u]) = o;
            }
        }
    }
    __syncthreads();
    const int t = t0 + (int)threadIdx.x;
    if (t >= a.L) return;
    const float* __restrict__ w = a.w;
    float acc = a.bias[0];
    const float* row = lds + threadIdx.x * S;
    for (int kap = 0; kap < k; ++kap, row += S) {
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(v[e], w[kap * C + 4 * q + e], acc);
        }
    }
    a.y[(size_t)b * a.L + t] = tanhf(acc);
}

// Generic fallback (any C, fp32 input): scalar staging.  (A template only so that both translation units may include it.)
template <int UNUSED>
__global__ void __launch_bounds__(256) conv_post_tanh_kernel(const ConvPostLaunch a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = a.C, k = a.k, pad = (k - 1) / 2;
    const int S = C | 1;
    const int b = (int)(blockIdx.x / (unsigned)a.tiles_per_item);
    const int t0 = ((int)blockIdx.x - b * a.tiles_per_item) * kPostTile;
    const int R = kPostTile + k - 1;
    const int total = R * C;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int r = idx / C, c = idx - r * C;
        const int row = t0 - pad + r;
        float v = 0.f;
        if (row >= 0 && row < a.L) {
            const size_t off = ((size_t)b * a.L + row) * C + c;
            v = ((const float*)a.x[0])[off];
            for (int j = 1; j < a.n_in; ++j) v += ((const float*)a.x[j])[off];
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

// True when the 16-byte-staging kernel can take the launch (32-bit buffer offsets inside one batch item's tensor).
inline bool conv_post_rows_ok(const ConvPostLaunch& a, bool bf16_in) {
    const long long blocks = (long long)((a.L + kPostTile - 1) / kPostTile) * a.B;
    return (a.C == 8 || a.C == 16 || a.C == 32 || a.C == 64) && a.n_in <= 4 && blocks >= 1 && blocks <= 0x7fffffffLL &&
           (double)a.L * a.C * (bf16_in ? 2 : 4) < 2147483648.0;
}

template <bool BF16_IN>
inline hipError_t launch_conv_post_t(ConvPostLaunch a, hipStream_t stream) {
    a.tiles_per_item = (a.L + kPostTile - 1) / kPostTile;
    const long long blocks = (long long)a.tiles_per_item * a.B;
    if (blocks < 1 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const bool rows_ok = conv_post_rows_ok(a, BF16_IN);
    dim3 grid((unsigned)blocks), block(256);
    if (rows_ok) {
        const size_t lds_bytes = (size_t)(kPostTile + a.k - 1) * (a.C + 4) * sizeof(float);
        if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
#define IRIS_POST_CASE(C_)                                                                                   \
        if (a.C == C_) {                                                                                     \
            auto kfn = conv_post_rows_kernel<C_, BF16_IN>;                                                   \
            { const hipError_t e__ = ::iris::launch_kernel_named("conv_post_rows_kernel", kfn, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; } \
            return hipSuccess;                                                                               \
        }
        IRIS_POST_CASE(8) IRIS_POST_CASE(16) IRIS_POST_CASE(32) IRIS_POST_CASE(64)
#undef IRIS_POST_CASE
    }
    if (BF16_IN) return hipErrorInvalidValue;      // the bf16-storage path has no scalar fallback (C % 8 == 0 there)
    const size_t lds_bytes = (size_t)(kPostTile + a.k - 1) * (a.C | 1) * sizeof(float);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    { const hipError_t e__ = ::iris::launch_kernel(conv_post_tanh_kernel<0>, grid, block, lds_bytes, stream, a); if (e__ != hipSuccess) return e__; }
    return hipSuccess;       
}

inline hipError_t launch_conv_post(const ConvPostLaunch& a, hipStream_t stream) { return launch_conv_post_t<false>(a, stream); }

}  // namespace post
}  // namespace iris
