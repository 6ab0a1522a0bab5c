// generator_internal.h -- state shared by the translation units behind include/iris_hifigan.h
// (iris_hifigan.hip: C-ABI + fp32 path; iris_hifigan_bf16.hip: bf16-storage path).  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <utility>
#include <vector>

#include "../../include/iris_hifigan.h"
#include "diag_env.h"
#include "device_info.h"
#include <new>

namespace iris {

constexpr int kPassFrames = 65536;       // mel frames (batch x frames) one pass of a forward processes: bounds the workspace
int pass_items(int B, int T);             // batch items of one pass
constexpr int kTileCounterWords = 256;   // per-launch next-tile counters of one forward (MRF kernel, large batches)

// forward_until: stop after MRF step `step` of stage `stage` has been queued (stage < 0: run the whole forward)
struct ForwardStop { int stage; int step; };

// Runs the rest of the scope under the handle's device and restores the caller's (a process may hold generators on
// several GPUs; launches, events and the CU-count-based launch plans must all refer to the handle's device).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// records the message for iris_hifigan_last_error() and returns `code`
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::iris::fail(IRIS_HIFIGAN_HIP_ERROR, "%s failed: %s (%s:%d)", #expr,      \
                                hipGetErrorString(e__), __FILE__, __LINE__);                 \
    } while (0)

#define TRY(expr) do { int rc__ = (expr); if (rc__ != IRIS_HIFIGAN_OK) return rc__; } while (0)

// Every extern "C" body that can allocate runs between these: no C++ exception crosses the ABI (include/iris_hifigan.h).
#define IRIS_ABI_BEGIN try {
#define IRIS_ABI_END                                                                                              \
    } catch (const std::bad_alloc&) {                                                                             \
        return ::iris::fail(IRIS_HIFIGAN_OUT_OF_MEMORY, "host allocation failed (%s)", __func__);                 \
    } catch (...) {                                                                                               \
        return ::iris::fail(IRIS_HIFIGAN_HIP_ERROR, "unexpected C++ exception in %s", __func__);                  \
    }

struct ConvLayer {       // one Conv1d / ConvTranspose1d, weights resident on the device
    int kind = 0;                                     // 0 Conv1d, 1 ConvTranspose1d, 2 conv_post
    int C_in = 0, C_out = 0, k = 0, dil = 1, u = 1;  // u = stride of a ConvTranspose1d
    size_t w_off = 0, b_off = 0;                      // float offsets into the device blob
    size_t w_floats = 0;                              // packed size
    size_t ref_w_floats = 0;                          // size in the reference layout
    size_t w16_off = 0, w16_halfs = 0;                // bf16 path: offset/size (bf16 elements) in blob16
    size_t ws3_off = 0;                               // split path: offset (bf16 elements) of the hi plane in blob_s3
    size_t w16f_off = (size_t)-1;                     // small-problem kernel: float offset of the 16x16x4 packing in blob_w16, or -1
};

struct Stage {
    ConvLayer up;
    int C = 0;             // channels after the upsample
    int rate = 1;
    // convs[j][m][0|1] = resblocks[i*nk + j].convs{1,2}[m]
    std::vector<std::vector<ConvLayer>> c1, c2;
};

}  // namespace iris

struct iris_hifigan_handle {
    iris_hifigan_config cfg;
    iris::ConvLayer pre, post;
    std::vector<iris::Stage> stages;
    float* blob = nullptr;   // device: packed fp32 weights + biases
    size_t blob_floats = 0;
    unsigned* tile_counters = nullptr;  // device: one next-tile counter per MRF launch of a forward (zeroed per forward)
    uint16_t* blob16 = nullptr;  // device: packed bf16 weights (biases stay fp32 in `blob`)
    size_t blob16_halfs = 0;
    uint16_t* blob_s3 = nullptr; // device: hi/mid bf16 planes of the ResBlock conv weights (split-product mode), or null
    float* blob_w16 = nullptr;   // device: ResBlock conv weights in v_mfma_f32_16x16x4_f32 fragment order (small-problem kernel), or null
    int hop = 1;
    int device = 0;
    // create uploads both fp32 packings; those of the other dtypes are built on first use (iris_hifigan_prepare, or the
    // first forward of the dtype): the reference-layout weights are kept on the host until both exist or the caller
    // releases them (iris_hifigan_release_host_weights).  built_* = the build succeeded, or the config cannot have it.
    std::vector<float> ref_weights;
    bool built_bf16 = false, built_s3 = false;
    bool host_only = false;     // iris_hifigan_describe_plan: no device memory behind the pointers, nothing is launched
    // profiling
    int profiling = 0;          // 0 off, 1 one record per launch, 2 the MRF launches of a stage share one record
    int profiling_paused = 0;   // records are kept, forwards add none (hipGraph capture)
    std::vector<hipEvent_t> ev;          // event pool, n_ev in use
    size_t n_ev = 0;
    std::vector<iris_hifigan_launch_record> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rec_ev;  // (start, end) event of each record
    int n_rec = 0;
};

namespace iris {

// Walks the layers in blob order; fn(layer&) for each.
template <class Fn>
void for_each_layer(iris_hifigan_handle* h, Fn fn) {
    fn(h->pre);
    for (auto& st : h->stages) {
        fn(st.up);
        for (size_t j = 0; j < st.c1.size(); ++j) {
            for (auto& l : st.c1[j]) fn(l);
            for (auto& l : st.c2[j]) fn(l);
        }
    }
    fn(h->post);
}

// Launch timing with ONE event per launch boundary: event e_k sits between launch k-1 and launch k
// of a forward, so launch k lasted elapsed(e_k, e_{k+1}) (its start-up gap included).  A forward of
// n launches records n + 1 events.
struct Prof {
    iris_hifigan_handle* h;
    hipStream_t stream;
    int idx = 0;        // next record
    bool open = false;  // a start event for the next launch is already on the stream
    bool pending = false;  // grouped mode: record idx covers the MRF launches of a stage so far and has no end event yet
    int mark(hipEvent_t* out) {
        if (h->n_ev >= h->ev.size()) {
            const size_t old = h->ev.size();
            h->ev.resize(old + 64);
            for (size_t i = old; i < h->ev.size(); ++i) HIP_TRY(hipEventCreate(&h->ev[i]));
        }
        *out = h->ev[h->n_ev++];
        HIP_TRY(hipEventRecord(*out, stream));
        return IRIS_HIFIGAN_OK;
    }
    // grouped mode (set_profiling(h, 2)): the MRF launches of a stage share ONE record (flops and bytes summed,
    // `launches` counted), so a forward carries 11 events instead of 31 -- an event costs ~3 us of stream time
    int close_group() {
        if (!pending) return IRIS_HIFIGAN_OK;
        int rc = mark(&h->rec_ev[idx].second);
        if (rc != IRIS_HIFIGAN_OK) return rc;
        pending = false;
        open = true;
        ++idx;
        return IRIS_HIFIGAN_OK;
    }
    bool on() const { return h->profiling && !h->profiling_paused; }
    int begin(int kind, int stage, int step, double flops, double bytes) {
        if (!on()) return IRIS_HIFIGAN_OK;
        const bool grouped = h->profiling == 2 && kind == 2;
        if (pending) {
            iris_hifigan_launch_record& g = h->recs[idx];
            if (grouped && g.stage == stage) { g.flops += flops; g.bytes += bytes; g.launches += 1; return IRIS_HIFIGAN_OK; }
            int rc = close_group();
            if (rc != IRIS_HIFIGAN_OK) return rc;
        }
        if ((size_t)idx >= h->recs.size()) { h->recs.resize(idx + 64); h->rec_ev.resize(idx + 64); }
        iris_hifigan_launch_record& r = h->recs[idx];
        memset(&r, 0, sizeof(r));
        r.kind = kind; r.stage = stage; r.step = step; r.launches = 1; r.flops = flops; r.bytes = bytes;
        if (!open) {
            int rc = mark(&h->rec_ev[idx].first);
            if (rc != IRIS_HIFIGAN_OK) return rc;
        } else {
            h->rec_ev[idx].first = h->rec_ev[idx - 1].second;
        }
        pending = grouped;
        return IRIS_HIFIGAN_OK;
    }
    int end() {
        if (!on() || pending) return IRIS_HIFIGAN_OK;
        int rc = mark(&h->rec_ev[idx].second);
        if (rc != IRIS_HIFIGAN_OK) return rc;
        open = true;
        ++idx;
        return IRIS_HIFIGAN_OK;
    }
    // end of a forward (or of forward_until): closes an open group and publishes the record count
    int finish() {
        if (!on()) return IRIS_HIFIGAN_OK;
        int rc = close_group();
        if (rc != IRIS_HIFIGAN_OK) return rc;
        h->n_rec = idx;
        return IRIS_HIFIGAN_OK;
    }
};

// ---- bf16-storage path (iris_hifigan_bf16.hip) ----
int bf16_build_blob(iris_hifigan_handle* h, const float* weights_host);   // packs + uploads blob16 (host_only: packs only)
uint64_t bf16_workspace_bytes(const iris_hifigan_handle* h, int B, int T);
int bf16_workspace_map(const iris_hifigan_handle* h, int B, int T, iris_hifigan_workspace_map* out);
int bf16_forward(iris_hifigan_handle* h, const void* mel_dev, int B, int T, void* wav_dev,
                 void* workspace_dev, uint64_t workspace_bytes, hipStream_t stream, const ForwardStop& stop,
                 int32_t* until_flags);

// ---- fp32 storage with split-bf16 products for the ResBlock convs (conv_mfma_f32s.h; iris_hifigan_bf16.hip) ----
int f32s_build_blob(iris_hifigan_handle* h, const float* weights_host);   // packs + uploads blob_s3 (null if unsupported)
// one grouped MRF step (all branches): x/res/y pointers per branch as in the fp32 path
struct F32sStep { const float* x[IRIS_HIFIGAN_MAX_KERNELS]; const float* res[IRIS_HIFIGAN_MAX_KERNELS];
                  float* y[IRIS_HIFIGAN_MAX_KERNELS]; const ConvLayer* layer[IRIS_HIFIGAN_MAX_KERNELS]; };
bool f32s_step_applicable(const iris_hifigan_handle* h, int C, int L, int nk);
bool f32s_ups_applicable(const iris_hifigan_handle* h, const ConvLayer& l, int L_in);
int f32s_launch_ups(iris_hifigan_handle* h, const ConvLayer& l, const float* x, float* y, int B, int L_in, hipStream_t stream);
// sum_y: when set, the step stores only the mean of the branch outputs there (last conv step of a stage)
int f32s_launch_step(iris_hifigan_handle* h, const F32sStep& st, int nk, int B, int L, int C, float* sum_y, hipStream_t stream);

}  // namespace iris
