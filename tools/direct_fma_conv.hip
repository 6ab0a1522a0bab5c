// A/B microbenchmark (not product code) for north_star's clause "MFMA is used only where [it] beats the direct kernel under rocprof":
// ONE dilated Conv1d of an MRF ResBlock branch -- y[l][co] = b[co] + sum_{tap, ci} LeakyReLU(x[l + (tap - 3) d][ci]) w[tap][ci][co],
// C = 128, k = 7, d = 3, channels-last fp32 (reference src/iris/hifigan_pretrained.py:64-71) -- as a DIRECT kernel on the vector ALUs,
// built the way north_star describes the direct path: coalesced reads along the channel axis, the dilated tap window staged in LDS
// (LeakyReLU on the way in), register-tiled FMAs.  Each thread owns 8 rows x 8 channels (64 accumulators, held as float2 so that the
// compiler can use v_pk_fma_f32, the only way to the chip's 157 TFLOP/s vector fp32 peak); a 256-thread block owns 128 rows x all 128
// output channels; activations come from LDS as 16-byte reads (4 input channels of a row), weights as 16-byte loads that every thread
// of a channel group shares (L1 broadcast).  Per 256 FMAs a thread issues 8 LDS and 8 global 16-byte reads.
// The program checks the kernel against a one-thread-per-output reference on a short input, then times it at L = 64,000 (the
// stage-1 length of the 1 x 1000-frame headline: 14.7 GFLOP) and prints TFLOP/s; the library's MFMA kernel runs the same conv (as one
// of three branches of a grouped step) at 130-136 TFLOP/s (profiles/r04zz_bench.json: kernels.mrf_stage1_C128).
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/direct_fma_conv tools/direct_fma_conv.hip && tools/direct_fma_conv
//   rocprofv3 --kernel-trace --stats -d <dir> -- tools/direct_fma_conv        (the kernel's duration under rocprof)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

constexpr int C = 128, K = 7, TR = 128, SX = C + 4, DIL_MAX = 5;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) direct_conv(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ y, int L, int dil, float slope) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // (TR + (K-1) dil) rows x SX floats
    const int tid = threadIdx.x, l0 = blockIdx.x * TR, pad = (K - 1) / 2 * dil, R = TR + (K - 1) * dil;
    for (int i = tid; i < R * (C / 4); i += 256) {                         // window: coalesced 16-byte reads along the channel axis
        const int r = i / (C / 4), q = i - r * (C / 4), row = l0 - pad + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row >= 0 && row < L) v = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + 4 * q);
        v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
        *reinterpret_cast<f32x4*>(xs + r * SX + 4 * q) = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;                                // rows 8 ty .. 8 ty + 7, channels 8 tx .. 8 tx + 7
    f32x2 acc[8][4];
    {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 8 * tx), b1 = *reinterpret_cast<const f32x4*>(bias + 8 * tx + 4);
#pragma unroll
        for (int r = 0; r < 8; ++r) { acc[r][0] = f32x2{b0.x, b0.y}; acc[r][1] = f32x2{b0.z, b0.w}; acc[r][2] = f32x2{b1.x, b1.y}; acc[r][3] = f32x2{b1.z, b1.w}; }
    }
    for (int tap = 0; tap < K; ++tap) {
        const float* xrow = xs + (8 * ty + tap * dil) * SX;
        const float* wt = w + (size_t)tap * C * C + 8 * tx;
#pragma unroll 2
        for (int c4 = 0; c4 < C / 4; ++c4) {
            f32x4 xv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) xv[r] = *reinterpret_cast<const f32x4*>(xrow + r * SX + 4 * c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + (size_t)(4 * c4 + j) * C);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wt + (size_t)(4 * c4 + j) * C + 4);
                const f32x2 wa = {w0.x, w0.y}, wb = {w0.z, w0.w}, wc = {w1.x, w1.y}, wd = {w1.z, w1.w};
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float xr = xv[r][j];
                    const f32x2 xx = {xr, xr};
                    acc[r][0] = __builtin_elementwise_fma(xx, wa, acc[r][0]);
                    acc[r][1] = __builtin_elementwise_fma(xx, wb, acc[r][1]);
                    acc[r][2] = __builtin_elementwise_fma(xx, wc, acc[r][2]);
                    acc[r][3] = __builtin_elementwise_fma(xx, wd, acc[r][3]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = l0 + 8 * ty + r;
        if (row < L) {
            *reinterpret_cast<f32x4*>(y + (size_t)row * C + 8 * tx) = f32x4{acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y};
            *reinterpret_cast<f32x4*>(y + (size_t)row * C + 8 * tx + 4) = f32x4{acc[r][2].x, acc[r][2].y, acc[r][3].x, acc[r][3].y};
        }
    }
}

// one thread per output element, same order of summation (taps ascending, input channels ascending): the checker
__global__ void reference_conv(const float* x, const float* w, const float* bias, float* y, int L, int dil, float slope) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L * C) return;
    const int l = (int)(i / C), co = (int)(i % C);
    float a = bias[co];
    for (int tap = 0; tap < K; ++tap) {
        const int row = l + (tap - (K - 1) / 2) * dil;
        if (row < 0 || row >= L) continue;
        for (int ci = 0; ci < C; ++ci) {
            float v = x[(size_t)row * C + ci];
            v = fmaxf(v, v * slope);
            a = fmaf(v, w[((size_t)tap * C + ci) * C + co], a);
        }
    }
    y[i] = a;
}

int main() {
    const int L = 64000, Lc = 1000, dil = 3;
    const float slope = 0.1f;
    std::vector<float> hx((size_t)L * C), hw((size_t)K * C * C), hb(C);
    unsigned s = 2024u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) / 8388608.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.033f;          // ~ 1 / sqrt(C k)
    for (auto& v : hb) v = rnd();
    float *x, *w, *b, *y, *yr;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, hb.size() * 4); hipMalloc(&y, hx.size() * 4); hipMalloc(&yr, (size_t)Lc * C * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(TR + (K - 1) * dil) * SX * 4;          // 77 KB at d = 3: two blocks per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&direct_conv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // check on the first Lc rows (as a tensor of its own: zero padding at both ends)
    hipLaunchKernelGGL(direct_conv, dim3((Lc + TR - 1) / TR), dim3(256), lds, 0, x, w, b, y, Lc, dil, slope);
    hipLaunchKernelGGL(reference_conv, dim3((Lc * C + 255) / 256), dim3(256), 0, 0, x, w, b, yr, Lc, dil, slope);
    std::vector<float> h1((size_t)Lc * C), h2((size_t)Lc * C);
    hipMemcpy(h1.data(), y, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), yr, h2.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, mag = 0;
    for (size_t i = 0; i < h1.size(); ++i) { worst = fmax(worst, fabs((double)h1[i] - h2[i])); mag = fmax(mag, fabs((double)h2[i])); }
    printf("check on %d rows: max |direct - reference| = %.3e (max |y| = %.3f)\n", Lc, worst, mag);
    if (!(worst <= 1e-4)) { printf("MISMATCH\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 10; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(direct_conv, dim3((L + TR - 1) / TR), dim3(256), lds, 0, x, w, b, y, L, dil, slope);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * L * K * C * (double)C;
    printf("direct fp32 FMA kernel: L = %d, C = %d, k = %d, d = %d: %.3f ms (best of 9) = %.1f TFLOP/s (vector fp32 peak with v_pk_fma_f32: 157.3; "
           "the MFMA kernel runs this layer at 130-136)\n", L, C, K, dil, best, flop / best / 1e9);
    return 0;
}
