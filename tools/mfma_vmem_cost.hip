// Calibration microbenchmark (not product code): what does ONE vector-memory instruction between v_mfma_f32_32x32x2_f32s cost?
//
// Round 4's ablations of the wide fp32 MRF kernel put ~9 % of its time on the in-loop loads although they move few bytes, hit
// L1 / L2 and are requested groups ahead (profiles/r04_notes.md).  This loop is that kernel's inner structure reduced to its
// instruction mix: a group = 8 MFMAs on two accumulators (MT = 2) whose A operand comes from a register ring fed by ONE
// buffer_load_dwordx4 per group (always the same 1 KB: an L1 hit), optionally plus `EXTRA` more loads and `DS` ds_read_b128 per
// group, against the same loop with the loads removed.  1 or 2 blocks (4 waves each) per CU = 1 or 2 waves per SIMD.
// Prints cycles per group (from the wall time at the clock the chip holds; 512 = the matrix pipe's own time at one wave per SIMD).
// Second question: the same loop with a few HBM-miss loads per 64-group "phase" in front of the weight loads, at ring depths 2 / 4 / 8.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/mfma_vmem_cost tools/mfma_vmem_cost.hip && tools/mfma_vmem_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// VM buffer loads and DS LDS reads per group of 8 MFMAs; weight ring DB groups deep; SLOW: a "phase" is 16 trips of the ring
// (16 (DB + 1) groups); in its first trip every group issues one MORE load that streams 16 bytes per lane from a 1 GB buffer that is
// never re-read (an HBM miss), kept in a register of its own and consumed only at the END of the phase -- the kernel's window
// staging.  vmcnt retires in order, so the waits for the (L1-resident) weight fragments behind such a load wait for it too.
template <int VM, int DS, int DB, int SLOW>
__global__ void __launch_bounds__(256, 2) loop(const float* w, float* out, int groups, const float* big, unsigned big_bytes) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 68];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 68; i += 256) lds[i] = w[(i * 7 + blockIdx.x) & 0x3ffff];   // (operand data: whatever `w` holds)
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 1 << 20, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big), 0, (int)big_bytes, 0x00020000);
    unsigned slow_off = (unsigned)((blockIdx.x * 256u + threadIdx.x) * 16u) % big_bytes;
    f32x4 slowv[DB + 1], slowsum = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d <= DB; ++d) slowv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    int trip = 0;
    f32x16 acc[2];
    for (int m = 0; m < 2; ++m) for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
    f32x4 bw[DB + 1];
    for (int d = 0; d <= DB; ++d) bw[d] = f32x4{1.f + lane * 1e-3f, 0.5f, 0.25f, 0.125f};
    f32x4 extra[VM > 1 ? VM - 1 : 1];
    for (int i = 0; i < (VM > 1 ? VM - 1 : 1); ++i) extra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av[2][2];
    const float* ap = lds + (lane & 31) * 68 + 4 * (lane >> 5);
    for (int m = 0; m < 2; ++m) av[0][m] = *reinterpret_cast<const f32x4*>(ap + m * 32 * 68 / 2);
    for (int g0 = 0; g0 < groups; g0 += DB + 1) {
#pragma unroll
        for (int d = 0; d <= DB; ++d) {                       // one group; ring slot d is consumed, slot (d + DB) % (DB + 1) refilled
            if constexpr (VM >= 1)
                bw[(d + DB) % (DB + 1)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (d & 3) * 1024, 0));
#pragma unroll
            for (int i = 0; i + 1 < VM; ++i)
                extra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 8192 + i * 1024, 0));
            if constexpr (SLOW > 0) {
                if (trip == 0) {                               // (block-uniform branch)
                    slowv[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, slow_off, 0, 0));
                    slow_off += 64u * 1024u * 1024u + 4096u * 16u; if (slow_off >= big_bytes) slow_off -= big_bytes;
                }
            }
            if constexpr (DS > 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (m < DS) av[(d + 1) & 1][m] = *reinterpret_cast<const f32x4*>(ap + ((d + 1) & 7) * 8 + m * 32 * 68 / 2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[d][e], av[DS > 0 ? (d & 1) : 0][m][e], acc[m], 0, 0, 0);
            {   // one request slotted behind each of the first MFMAs, like the kernel
                int ds_left = DS, vm_left = VM;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ds_left > 0) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); --ds_left; }
                    else if (vm_left > 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); --vm_left; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SLOW > 0) {
            if (++trip == 16) {                                // end of the phase: the staged data is consumed here
                trip = 0;
#pragma unroll
                for (int d = 0; d <= DB; ++d) slowsum = slowsum + slowv[d];
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int q = 0; q < 16; ++q) s += acc[m][q];
    for (int i = 0; i < (VM > 1 ? VM - 1 : 1); ++i) s += extra[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s + slowsum[0];
}

template <int VM, int DS, int DB, int SLOW = 0>
void run(const char* what, int blocks, int groups, const float* w, float* out, double clock_ghz, const float* big = nullptr, unsigned big_bytes = 16) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<VM, DS, DB, SLOW>), dim3(blocks), dim3(256), 0, 0, w, out, groups, big ? big : w, big ? big_bytes : 16u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double waves_per_simd = blocks / 256.0;
    const double flops = (double)blocks * 4 * groups * 8 * 4096.0;
    // cycles of SIMD time per group and wave: wall * clock / (groups * waves per SIMD)
    printf("%-46s %d wave(s)/SIMD: %7.3f ms  %6.1f TFLOP/s  %6.1f SIMD-cycles per group (512 = pipe-bound at %.2f GHz)\n", what, (int)waves_per_simd, ms,
           flops / ms / 1e9, ms * 1e-3 * clock_ghz * 1e9 / (groups * waves_per_simd), clock_ghz);
}

// Third question: does the DATA matter?  argv[1] = "random": the weight buffer (and through it the LDS image) holds random values in
// [-1, 1) instead of zeros -- same instructions, same addresses; a difference is the clock the chip holds under the switching activity.
int main(int argc, char** argv) {
    float *w, *out;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    if (argc > 1 && argv[1][0] == 'r') {
        static float host[1 << 18];
        unsigned s = 12345u;
        for (int i = 0; i < (1 << 18); ++i) { s = s * 1664525u + 1013904223u; host[i] = (float)(int)(s >> 8) / 8388608.0f - 1.0f; }
        hipMemcpy(w, host, sizeof(host), hipMemcpyHostToDevice);
        printf("operands: random values in [-1, 1)\n");
    } else {
        printf("operands: zeros (weights) / zeros (activations)\n");
    }
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    float* big; const unsigned big_bytes = 1u << 30;
    hipMalloc(&big, big_bytes); hipMemset(big, 0, big_bytes);
    const double ghz = 2.4;
    const int G = 40000;    // groups per wave (x 5 ring slots per loop trip): ~10 ms launches
    for (int blocks : {256, 512}) {
        run<0, 0, 4>("MFMAs only", blocks, G, w, out, ghz);
        run<0, 2, 4>("+ 2 ds_read_b128 per group", blocks, G, w, out, ghz);
        run<1, 2, 4>("+ 2 ds_read_b128 + 1 buffer_load_dwordx4", blocks, G, w, out, ghz);
        run<2, 2, 4>("+ 2 ds_read_b128 + 2 buffer_load_dwordx4", blocks, G, w, out, ghz);
        run<3, 2, 4>("+ 2 ds_read_b128 + 3 buffer_load_dwordx4", blocks, G, w, out, ghz);
        run<1, 0, 4>("+ 1 buffer_load_dwordx4 only", blocks, G, w, out, ghz);
        run<1, 2, 2, 1>("ring 2: 3 HBM-miss loads per 48-group phase", blocks, G, w, out, ghz, big, big_bytes);
        run<1, 2, 4, 1>("ring 4: 5 HBM-miss loads per 80-group phase", blocks, G, w, out, ghz, big, big_bytes);
        run<1, 2, 8, 1>("ring 8: 9 HBM-miss loads per 144-group phase", blocks, G, w, out, ghz, big, big_bytes);
    }
    return 0;
}
