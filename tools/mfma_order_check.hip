// Does v_mfma_f32_16x16x4_f32 accumulate its four k-steps as a sequential fmaf chain in k order?  If so, a 16x16 tile
// computed with it can reproduce BIT FOR BIT the 32x32x2 chains of the MRF kernel (channel order 0,4,1,5,2,6,3,7 inside a
// group of 8 channels) by feeding K-quarter kq with channel {0,4,1,5}[kq] / {2,6,3,7}[kq].
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_order_check tools/mfma_order_check.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 256;   // channels (one tap)

// X [32 rows][K], W [32 co][K] -> D32[co][t] via 32x32x2 in the kernel's order; one wave
__global__ void k32(const float* X, const float* W, float* D) {
    const int lane = threadIdx.x, lo = lane & 31, hi = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.25f * r;          // a non-trivial C input
    for (int g = 0; g < K / 8; ++g)
        for (int e = 0; e < 4; ++e) {
            const float a = W[lo * K + 8 * g + 4 * hi + e];   // A[i = co = lo][k = hi]
            const float b = X[lo * K + 8 * g + 4 * hi + e];   // B[k = hi][j = t = lo]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo] = acc[r];   // row = co, col = t
}
// the same 32x32 result as four 16x16 tiles with 16x16x4, one wave, C input matched
__global__ void k16(const float* X, const float* W, float* D) {
    const int lane = threadIdx.x, t = lane & 15, kq = lane >> 4;
    const int chA = (kq & 1) * 4 + (kq >> 1);                 // {0,4,1,5}
    const int chB = chA + 2;                                  // {2,6,3,7}
    for (int ct = 0; ct < 2; ++ct)
        for (int tt = 0; tt < 2; ++tt) {
            f32x4 acc;
            for (int r = 0; r < 4; ++r) {
                const int co = ct * 16 + 4 * kq + r;          // D row
                const int r32 = ((co & 3) | ((co >> 3) << 2));   // register index of that row in the 32x32 layout: (co&3) + 4*(co>>3)
                acc[r] = 0.25f * r32;
            }
            for (int g = 0; g < K / 8; ++g) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(ct * 16 + t) * K + 8 * g + chA], X[(tt * 16 + t) * K + 8 * g + chA], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(ct * 16 + t) * K + 8 * g + chB], X[(tt * 16 + t) * K + 8 * g + chB], acc, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) D[(ct * 16 + 4 * kq + r) * 32 + tt * 16 + t] = acc[r];
        }
}
int main() {
    float *hX = (float*)malloc(32 * K * 4), *hW = (float*)malloc(32 * K * 4), h32[1024], h16[1024];
    srand(7);
    for (int i = 0; i < 32 * K; ++i) { hX[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; hW[i] = (rand() / (float)RAND_MAX - 0.5f); }
    float *X, *W, *D;
    hipMalloc(&X, 32 * K * 4); hipMalloc(&W, 32 * K * 4); hipMalloc(&D, 4096);
    hipMemcpy(X, hX, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW, 32 * K * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, X, W, D); hipMemcpy(h32, D, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, X, W, D); hipMemcpy(h16, D, 4096, hipMemcpyDeviceToHost);
    int diff = 0; double maxd = 0;
    for (int i = 0; i < 1024; ++i) { if (memcmp(&h32[i], &h16[i], 4)) ++diff; maxd = fmax(maxd, fabs((double)h32[i] - h16[i])); }
    // host fmaf chain in the same order, for element (co=5, t=9)
    float c = 0.25f * ((5 & 3) + 4 * (5 >> 3));
    for (int g = 0; g < K / 8; ++g) for (int e = 0; e < 4; ++e) for (int h = 0; h < 2; ++h) c = fmaf(hW[5 * K + 8 * g + 4 * h + e], hX[9 * K + 8 * g + 4 * h + e], c);
    printf("16x16x4 vs 32x32x2: %d of 1024 elements differ bitwise, max |diff| %.3e; host fmaf chain for (5,9): %.9g, gpu32 %.9g, gpu16 %.9g\n",
           diff, maxd, c, h32[5 * 32 + 9], h16[5 * 32 + 9]);
    return diff != 0;
}
