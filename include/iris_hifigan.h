/*
 * iris_hifigan.h -- C-ABI of the MI355X (gfx950) HiFiGAN generator path.
 *
 * The reference (ZECTBynmo/iris-tts) is pure Python and has no FFI for this path
 * (SURVEY.md section 8b): its boundary is two Python call surfaces,
 *   iris.hifigan_pretrained.HiFiGANGenerator.__call__   (src/iris/hifigan_pretrained.py:208-242)
 *   iris.vocoder.HiFiGANVocoder.infer                   (src/iris/vocoder.py:177-209)
 * both of which end in one call of the generator forward
 *   HiFiGANModel.forward                                (src/iris/hifigan_pretrained.py:123-143)
 *   HiFiGANGenerator.call                               (src/iris/vocoder.py:103-130).
 * The entry points below are what a binding for that forward has to provide; the
 * drop-in Python modules in iris-tts_amd/iris/ bind them with ctypes.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer named *_dev is a HIP device pointer
 *     owned by the caller (e.g. a PyTorch-ROCm tensor's data_ptr()), *_host is host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); forward is
 *     asynchronous on it and performs no allocation and no synchronisation -- with ONE exception: the first forward of
 *     IRIS_HIFIGAN_BF16 / IRIS_HIFIGAN_F32_SPLIT on a handle that was not prepared for that dtype
 *     (iris_hifigan_prepare) builds the dtype's weight packing first (host repack, hipMalloc, synchronous upload).
 *     Forwards of IRIS_HIFIGAN_F32 never do (create uploads everything fp32 needs).  Inside a stream capture that
 *     lazy build is refused with IRIS_HIFIGAN_NOT_PREPARED instead of invalidating the capture;
 *   - every function returns an iris_hifigan_status; on failure the calling thread's
 *     message is available from iris_hifigan_last_error(); no exception crosses the ABI;
 *   - a handle may be used by one thread at a time, with ONE forward in flight: a handle owns per-forward
 *     device state (the next-tile counters of the persistent MRF kernel, the profiling events) and every
 *     forward of a handle writes the same caller-provided workspace, so a second forward of the same handle
 *     (another stream, or a hipGraph replay that captured it) must be ordered after the first one -- put them
 *     on one stream or give each stream its own handle and workspace;
 *   - a handle belongs to the HIP device that was current in iris_hifigan_create; forward selects that device
 *     for its launches and restores the caller's, so `stream` and all *_dev pointers must belong to it;
 *   - the library reads no environment variable (diagnostic switches exist only in the `make diag` build).
 *
 * Activations inside the library are channels-last [B, L, C], fp32 (bf16 with dtype
 * IRIS_HIFIGAN_BF16); the mel comes in as the
 * reference hands it over, channels-first [B, n_mels, T] (hifigan_pretrained.py:228), and the
 * waveform goes out as [B, prod(upsample_rates) * T] (hifigan_pretrained.py:235-236).
 */
#ifndef IRIS_HIFIGAN_H
#define IRIS_HIFIGAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRIS_HIFIGAN_ABI_VERSION 4
#define IRIS_HIFIGAN_MAX_STAGES 8    /* upsample stages            */
#define IRIS_HIFIGAN_MAX_KERNELS 8   /* MRF branches per stage     */
#define IRIS_HIFIGAN_MAX_DILATIONS 8 /* conv pairs per ResBlock    */

typedef enum iris_hifigan_status {
    IRIS_HIFIGAN_OK = 0,
    IRIS_HIFIGAN_INVALID_ARGUMENT = 1,
    IRIS_HIFIGAN_HIP_ERROR = 2,
    IRIS_HIFIGAN_OUT_OF_MEMORY = 3,
    IRIS_HIFIGAN_UNSUPPORTED = 4,
    IRIS_HIFIGAN_WORKSPACE_TOO_SMALL = 5,
    IRIS_HIFIGAN_NOT_PREPARED = 6 /* a forward needs a weight packing that cannot be built now: the stream is being captured
                                    (call iris_hifigan_prepare first), or the host weights were released before the dtype's
                                    first use */
} iris_hifigan_status;

typedef enum iris_hifigan_dtype {
    IRIS_HIFIGAN_F32 = 0, /* fp32 storage, fp32 MFMA (exact fmaf chains): the parity path (<= 1e-4 vs reference) */
    IRIS_HIFIGAN_BF16 = 1, /* bf16 storage of activations and weights, fp32 accumulation (bf16 MFMA); mel in and
                             waveform out stay fp32.  BASELINE.json configs[2].  The reference has no bf16 path:
                             its error against the fp32 generator (~1e-2 max-abs) is documented, not pinned. */
    IRIS_HIFIGAN_F32_SPLIT = 2 /* fp32 storage everywhere, fp32 accumulation; the ResBlock convolutions form each product
                             from two bf16 terms per operand (hi*hi + hi*mid + mid*hi on the bf16 MFMA).  Within
                             north_star's 1e-4 of the fp32 generator (measured ~2e-5) but not the exact fp32 arithmetic
                             of IRIS_HIFIGAN_F32: opt-in, never the headline.  Same workspace as IRIS_HIFIGAN_F32. */
} iris_hifigan_dtype;

/* Generator hyper-parameters: the constructor arguments of HiFiGANModel
 * (hifigan_pretrained.py:77-85) == HiFiGANGenerator (vocoder.py:59-67). */
typedef struct iris_hifigan_config {
    int32_t in_channels;              /* 80  */
    int32_t upsample_initial_channel; /* 512 */
    int32_t num_upsamples;            /* 4   */
    int32_t upsample_rates[IRIS_HIFIGAN_MAX_STAGES];        /* 8,8,2,2   */
    int32_t upsample_kernel_sizes[IRIS_HIFIGAN_MAX_STAGES]; /* 16,16,4,4 */
    int32_t num_kernels;              /* 3   */
    int32_t resblock_kernel_sizes[IRIS_HIFIGAN_MAX_KERNELS]; /* 3,7,11 */
    int32_t num_dilations[IRIS_HIFIGAN_MAX_KERNELS];         /* 3,3,3  */
    int32_t resblock_dilations[IRIS_HIFIGAN_MAX_KERNELS][IRIS_HIFIGAN_MAX_DILATIONS]; /* 1,3,5 each */
    int32_t pre_kernel_size;          /* 7 (hifigan_pretrained.py:93)  */
    int32_t post_kernel_size;         /* 7 (hifigan_pretrained.py:120) */
    float lrelu_slope;                /* 0.1 everywhere (hifigan_pretrained.py:66,68,127,139) */
} iris_hifigan_config;

typedef struct iris_hifigan_handle iris_hifigan_handle;

/* Per-launch record filled when profiling is enabled (bench.py's roofline leg). */
typedef struct iris_hifigan_launch_record {
    int32_t kind;      /* 0 conv_pre, 1 upsample (ConvTranspose1d), 2 MRF ResBlock conv group, 3 conv_post */
    int32_t stage;     /* upsample stage index, -1 for conv_pre / conv_post */
    int32_t step;      /* 0..2*num_dilations-1 inside a stage's MRF, else 0 (grouped record: that of its first launch) */
    int32_t launches;  /* kernel launches this record covers: 1, or in grouped mode the MRF launches of a stage */
    double flops;      /* algorithmic FLOP of this launch (2*MAC, zero padding counted)   */
    double bytes;      /* algorithmic bytes of this launch, accounting L of SURVEY.md 8d  */
    float ms;          /* hipEventElapsedTime of this launch on the forward's stream      */
    float reserved2;
} iris_hifigan_launch_record;

int32_t iris_hifigan_abi_version(void);
const char* iris_hifigan_last_error(void);

/* Number of fp32 values iris_hifigan_create expects in `weights_host`: the folded
 * (weight-norm already applied) tensors of the reference state-dict, concatenated in
 * this order, each in the reference's own layout:
 *   conv_pre.weight [C0,in,kpre], conv_pre.bias [C0],
 *   for i in stages:  ups.i.weight [Cin,Cout,k] (ConvTranspose1d layout), ups.i.bias [Cout],
 *                     for j in kernels: for m in dilations: convs1.m.weight [C,C,k], convs1.m.bias [C]
 *                                       for m in dilations: convs2.m.weight [C,C,k], convs2.m.bias [C]
 *   conv_post.weight [1,Clast,kpost], conv_post.bias [1].                                         */
int32_t iris_hifigan_weight_count(const iris_hifigan_config* cfg, uint64_t* count);

/* Builds a generator: validates cfg, repacks the weights into MFMA fragment order and
 * uploads them to the current HIP device. Replaces HiFiGANModel() + load_state_dict
 * (hifigan_pretrained.py:186-190). */
int32_t iris_hifigan_create(const iris_hifigan_config* cfg, const float* weights_host,
                            uint64_t n_weights, iris_hifigan_handle** out);
int32_t iris_hifigan_destroy(iris_hifigan_handle* h);

/* create uploads what IRIS_HIFIGAN_F32 needs (the 32x32 MFMA fragments and the 16x16 fragments of the short-input kernel;
 * the repacking runs on up to 16 host threads).  The packings of the other arithmetic (bf16 fragments for IRIS_HIFIGAN_BF16,
 * hi/mid bf16 planes for IRIS_HIFIGAN_F32_SPLIT) are built on the first use of that dtype: by this call, or by the first
 * forward of the dtype (which is then synchronous and allocates once; refused with IRIS_HIFIGAN_NOT_PREPARED inside a stream
 * capture).  A failed build (out of memory) is reported and retried by the next call; it does not disable the dtype.
 * Call prepare before capturing forwards of a dtype into a hipGraph. */
int32_t iris_hifigan_prepare(iris_hifigan_handle* h, int32_t dtype);
/* Until both of those packings exist the handle keeps the reference-layout weights on the host (55.7 MB for V1, per handle)
 * to build them from.  A caller that has prepared every dtype it will use drops that copy here; a later first use of
 * another dtype then fails with IRIS_HIFIGAN_NOT_PREPARED. */
int32_t iris_hifigan_release_host_weights(iris_hifigan_handle* h);

/* Activation workspace needed by one forward of [B, in_channels, T].  Batch items are independent: a batch of more than
 * 65,536 mel frames in all runs as consecutive passes over sub-batches that share the workspace, so the figure is bounded
 * (fp32: 229 KB per frame of ONE pass, at most ~15 GB; bf16 half of that) instead of growing with the batch; one item
 * longer than that still needs its own length (split long utterances along time with iris.streaming). */
int32_t iris_hifigan_workspace_bytes(const iris_hifigan_handle* h, int32_t B, int32_t T,
                                     int32_t dtype, uint64_t* bytes);

/* mel_dev [B, in_channels, T] fp32 -> wav_dev [B, hop*T] fp32, hop = prod(upsample_rates).
 * Replaces `self.model(mel_tensor)` (hifigan_pretrained.py:231-232; vocoder.py:200). */
int32_t iris_hifigan_forward(iris_hifigan_handle* h, const void* mel_dev, int32_t B, int32_t T,
                             void* wav_dev, void* workspace_dev, uint64_t workspace_bytes,
                             int32_t dtype, void* stream);

/* ---- intermediates (parity tests of the layers inside a forward; not needed by a caller) ----
 * forward_until queues the same launches as forward up to and including MRF step `stop_step`
 * (0 .. 2*num_dilations-1: even = convs1[m], odd = convs2[m] + residual, hifigan_pretrained.py:64-71) of
 * upsample stage `stop_stage` and returns; no waveform is produced.  The intermediates are then in the
 * workspace at the offsets workspace_layout reports, channels-last [B, L_stage, C_stage]:
 *   pre  conv_pre output [B, T, C0];  up  the stage's ConvTranspose1d output;
 *   y[j] running x of ResBlock j (after an odd step);  xt[j] output of convs1 of ResBlock j (after an even step).
 * *flags (may be NULL) receives
 *   IRIS_HIFIGAN_UNTIL_MEAN_IN_Y0  stop_step is the last step of the stage and the kernel already stored
 *                                  (y[0]+y[1]+y[2])/num_kernels (hifigan_pretrained.py:131-137) in y[0] instead of the
 *                                  branch outputs;
 *   IRIS_HIFIGAN_UNTIL_X_IN_XT     the roles of the y and xt buffers are swapped at this point (the fused conv-pair
 *                                  kernel of the bf16 path cannot work in place and alternates between them). */
#define IRIS_HIFIGAN_UNTIL_MEAN_IN_Y0 1
#define IRIS_HIFIGAN_UNTIL_X_IN_XT 2
typedef struct iris_hifigan_workspace_map {
    uint64_t pre_offset, up_offset;                 /* byte offsets into the workspace */
    uint64_t y_offset[IRIS_HIFIGAN_MAX_KERNELS];
    uint64_t xt_offset[IRIS_HIFIGAN_MAX_KERNELS];
    uint64_t total_bytes;                           /* == iris_hifigan_workspace_bytes */
    int32_t element_bytes;                          /* 4 (fp32 storage) or 2 (bf16 storage) */
    int32_t reserved;
} iris_hifigan_workspace_map;
int32_t iris_hifigan_workspace_layout(const iris_hifigan_handle* h, int32_t B, int32_t T, int32_t dtype,
                                      iris_hifigan_workspace_map* out);
int32_t iris_hifigan_forward_until(iris_hifigan_handle* h, const void* mel_dev, int32_t B, int32_t T,
                                   void* workspace_dev, uint64_t workspace_bytes, int32_t dtype,
                                   int32_t stop_stage, int32_t stop_step, int32_t* flags, void* stream);

/* Samples of waveform per mel frame (256 for the V1 config). */
int32_t iris_hifigan_hop_length(const iris_hifigan_handle* h, int32_t* hop);

/* Profiling: enabled = 1 brackets every launch of forward by hipEvents on `stream`; enabled = 2 does the same but
 * gives the MRF launches of a stage ONE record (flops/bytes summed, `launches` counted): 11 events per forward
 * instead of 31 -- an event costs about 3 us of stream time, which is 10 % of a 100-frame forward.
 * Records accumulate over successive forwards until set_profiling is called again (which resets
 * them).  After the stream has been synchronised, read_profile copies up to `capacity` records and
 * returns how many launches were recorded. */
int32_t iris_hifigan_set_profiling(iris_hifigan_handle* h, int32_t enabled);
int32_t iris_hifigan_read_profile(iris_hifigan_handle* h, iris_hifigan_launch_record* out,
                                  int32_t capacity, int32_t* n_launches);
/* paused != 0: forwards record nothing until resumed; the records collected so far are kept (set_profiling would
 * reset them).  For forwards that are captured into a hipGraph (no events inside a capture). */
int32_t iris_hifigan_pause_profiling(iris_hifigan_handle* h, int32_t paused);

/* ---- launch plan of one forward, computed on the host alone (no device, nothing launched) ----
 * Runs the forward's own argument checks, workspace layout and launch planning for [B, in_channels, T] in `dtype` on a
 * chip of `cu_count` compute units (0 = 256, MI355X) and reports every launch it would issue, in order.  Used by the CPU
 * tests (address/undefined-behaviour sanitizer sweeps over the index arithmetic of the plans) and for inspection. */
#define IRIS_HIFIGAN_MAX_PLAN_LAUNCHES 96
typedef struct iris_hifigan_plan_launch {
    char kernel[80];        /* kernel (template) name as the launch site spells it */
    uint32_t grid[3];
    uint32_t block;
    uint64_t lds_bytes;     /* dynamic LDS per block */
} iris_hifigan_plan_launch;
typedef struct iris_hifigan_plan {
    uint64_t workspace_bytes;
    int32_t n_launches;     /* over all passes; may exceed IRIS_HIFIGAN_MAX_PLAN_LAUNCHES: only that many are described */
    int32_t cu_count;
    int32_t passes;         /* sub-batch passes the forward runs as (1 unless B * T exceeds 65,536 frames) */
    int32_t reserved;
    iris_hifigan_plan_launch launches[IRIS_HIFIGAN_MAX_PLAN_LAUNCHES];
} iris_hifigan_plan;
int32_t iris_hifigan_describe_plan(const iris_hifigan_config* cfg, int32_t B, int32_t T, int32_t dtype, int32_t cu_count,
                                   iris_hifigan_plan* out);

/* ---- single-layer entry points (bring-up and parity tests; synchronous, they allocate) ----
 * x_dev/y_dev/res_dev are channels-last [B, L, C]; weights/bias are HOST arrays in the
 * reference's layout. in_act: 0 none, 1 LeakyReLU(slope) applied to the input. */
int32_t iris_hifigan_op_conv1d(const float* x_dev, const float* w_host, const float* bias_host,
                               const float* res_dev, float* y_dev, int32_t B, int32_t L,
                               int32_t C_in, int32_t C_out, int32_t k, int32_t dilation,
                               int32_t in_act, float slope, int32_t x_channels_first, void* stream);
/* ConvTranspose1d(C_in->C_out, k, stride=u, padding=(k-u)/2), weight [C_in, C_out, k]; y is [B, u*L, C_out]. */
int32_t iris_hifigan_op_conv_transpose1d(const float* x_dev, const float* w_host,
                                         const float* bias_host, float* y_dev, int32_t B, int32_t L,
                                         int32_t C_in, int32_t C_out, int32_t k, int32_t u,
                                         int32_t in_act, float slope, void* stream);
/* tanh(Conv1d(C_in->1, k, pad=(k-1)/2)(LeakyReLU((x0+x1+x2)/n_in))): x1/x2 may be NULL (n_in = 1). */
int32_t iris_hifigan_op_conv_post(const float* x0_dev, const float* x1_dev, const float* x2_dev,
                                  const float* w_host, const float* bias_host, float* y_dev,
                                  int32_t B, int32_t L, int32_t C_in, int32_t k, float slope,
                                  void* stream);

/* One grouped MRF step of the fp32 path -- the hot kernel -- on its own: the three ResBlock branches
 * (kernel sizes k[j] = 3, 7, 11; hifigan_pretrained.py:64-71,130-136) each run
 *     y_j = Conv1d_{k[j], dil[j]}(LeakyReLU(x_j)) (+ res_j)
 * on fp32 channels-last tensors [B, L, C].  mean_dev != NULL makes it the last step of a stage: only
 * ((y_0 + y_1) + y_2) / 3 is stored, into mean_dev (y_dev is then unused).
 * plan: 0 = the library's own choice, 1 = persistent blocks with full-height tiles, 2 = with half-height tiles,
 * 3 = one branch per block, 4 = the small-problem kernel (16 x 16 jobs on v_mfma_f32_16x16x4_f32), 5 / 6 = snake-ordered
 * (tile, branch) jobs at half / full tile height (C >= 128); all of them produce identical bits.  mean_dev is
 * available in plans 0-2.  Returns IRIS_HIFIGAN_UNSUPPORTED when the shape cannot take
 * the requested kernel. */
int32_t iris_hifigan_op_mrf_step(const float* const* x_dev, const float* const* w_host, const float* const* bias_host,
                                 const float* const* res_dev, float* const* y_dev, float* mean_dev,
                                 int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                 float slope, int32_t plan, void* stream);

/* One ResBlock conv PAIR of the three branches in one launch, exact fp32 (csrc/mrf_pair_f32.h, mrf_pair_f32_pf.h; C = 32
 * or 64, k[j] in {3, 7, 11}): y_j = Conv1d_{k[j], 1}(LeakyReLU(Conv1d_{k[j], dil[j]}(LeakyReLU(x_j)))) + x_j
 * (hifigan_pretrained.py:64-71), bit for bit what two iris_hifigan_op_mrf_step calls produce.
 * mode: 0 = one block per (tile, branch) job; 1 = persistent blocks that prefetch the next job's window and draw their
 * jobs from a device counter; 2 = the same with a fixed job stride.
 * mean_dev != NULL (modes 1, 2) makes it the LAST pair of a stage: only ((y_0 + y_1) + y_2) / 3 is stored, into mean_dev
 * (hifigan_pretrained.py:131-137; y_dev is then unused).  No y_dev[i] / mean_dev may alias an x_dev[j].
 * The release library carries modes 1 / 2 in the summing form only (as plain pairs they measured slower than mode 0 at
 * every size and are compiled into the diagnostic build alone): modes 1 / 2 without mean_dev return IRIS_HIFIGAN_UNSUPPORTED
 * there, as do other shapes. */
int32_t iris_hifigan_op_mrf_pair(const float* const* x_dev, const float* const* w1_host, const float* const* b1_host,
                                 const float* const* w2_host, const float* const* b2_host, float* const* y_dev,
                                 float* mean_dev, int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                 float slope, int32_t mode, void* stream);

/* bf16 variants of the two layers above (dtype IRIS_HIFIGAN_BF16): x_dev / res_dev / y_dev are bf16
 * channels-last [B, L, C] (C_in % 8 == 0, C_out % 4 == 0); host weights are fp32 in the reference layout and are
 * rounded to bf16 (nearest even); bias stays fp32; accumulation is fp32, one rounding to bf16 at the store.
 * Same reference layers: hifigan_pretrained.py:50-57,64-71 (Conv1d) and :100-108,127-128 (ConvTranspose1d). */
int32_t iris_hifigan_op_conv1d_bf16(const void* x_dev, const float* w_host, const float* bias_host,
                                    const void* res_dev, void* y_dev, int32_t B, int32_t L, int32_t C_in,
                                    int32_t C_out, int32_t k, int32_t dilation, int32_t in_act, float slope,
                                    void* stream);
int32_t iris_hifigan_op_conv_transpose1d_bf16(const void* x_dev, const float* w_host, const float* bias_host,
                                              void* y_dev, int32_t B, int32_t L, int32_t C_in, int32_t C_out,
                                              int32_t k, int32_t u, int32_t in_act, float slope, void* stream);
/* One fused ResBlock conv pair of `n_branches` MRF branches in bf16 storage (hifigan_pretrained.py:64-71, one
 * iteration of the loop):  y_j = Conv1d_{k_j, 1}(LeakyReLU(Conv1d_{k_j, dil_j}(LeakyReLU(x_j)))) + x_j  on bf16
 * channels-last tensors [B, L, C], C = 32, 64 or 128 (the stages where the bf16 path is HBM-bound or at the ridge); the intermediate never
 * leaves the CU.  Same rounding points as the two separate bf16 layers (so: bit-identical to them).  NOT in place: no
 * y_dev[i] may be an x_dev[j] (a block's input window overlaps the rows its neighbours write).
 * Returns IRIS_HIFIGAN_UNSUPPORTED for other channel counts. */
int32_t iris_hifigan_op_mrf_pair_bf16(const void* const* x_dev, const float* const* w1_host, const float* const* b1_host,
                                      const float* const* w2_host, const float* const* b2_host, void* const* y_dev,
                                      int32_t n_branches, int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                      float slope, void* stream);
/* The LAST conv pair of a stage's three ResBlocks in bf16 storage, summed (hifigan_pretrained.py:64-71 and 131-137): one block
 * runs the pair of all three branches on its rows, rounds each y_j to bf16 as the kernel above stores it, and writes ONE tensor
 * [B, L, C]:  mean_f32 == 0: bf16(LeakyReLU(((y_0 + y_1) + y_2) * fp32(1/3))) -- bit for bit the operand the next
 * ConvTranspose1d builds from the three tensors, which then reads one tensor and applies no activation;  mean_f32 != 0: the
 * fp32 mean itself (conv_post's input after the last stage).  C = 32 or 64 (csrc/mrf_pair_bf16.h); mean_dev must not be an
 * input.  Returns IRIS_HIFIGAN_UNSUPPORTED for other shapes. */
int32_t iris_hifigan_op_mrf_pair_mean_bf16(const void* const* x_dev, const float* const* w1_host, const float* const* b1_host,
                                           const float* const* w2_host, const float* const* b2_host, void* mean_dev,
                                           int32_t mean_f32, int32_t B, int32_t L, int32_t C, const int32_t* k, const int32_t* dil,
                                           float slope, void* stream);
/* the ResBlock Conv1d (C -> C, LeakyReLU on the input, optional residual) with fp32 tensors and split-bf16 products
 * (dtype IRIS_HIFIGAN_F32_SPLIT; C % 32 == 0): hifigan_pretrained.py:50-57,64-71. */
int32_t iris_hifigan_op_conv1d_f32s(const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev,
                                    float* y_dev, int32_t B, int32_t L, int32_t C, int32_t k, int32_t dilation, float slope,
                                    void* stream);
/* LeakyReLU + ConvTranspose1d(C_in -> C_out, k, stride u, padding (k-u)/2) on fp32 tensors with split-bf16 products
 * (the upsamplers in dtype IRIS_HIFIGAN_F32_SPLIT): hifigan_pretrained.py:100-108,127-128. */
int32_t iris_hifigan_op_conv_transpose1d_f32s(const float* x_dev, const float* w_host, const float* bias_host, float* y_dev,
                                              int32_t B, int32_t L, int32_t C_in, int32_t C_out, int32_t k, int32_t u,
                                              float slope, void* stream);

/* ---- PostNet, the layer in front of the vocoder --------------------------
 * Replaces `postnet(mel_bt_f, training=False)` (scripts/synthesize.py:148-166; model src/iris/postnet.py:48-67):
 * num_layers Conv1D(kernel_size, 'same') layers over time, the first num_layers-1 with `channels`
 * filters + BatchNorm + tanh, the last back to n_mels + BatchNorm; output = mel + residual.
 * `weights_host`: per layer, weight [C_out, C_in, k] then bias [C_out], with the inference BatchNorm
 * already folded in (w' = w*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta).                     */
typedef struct iris_postnet_handle iris_postnet_handle;
int32_t iris_postnet_create(int32_t n_mels, int32_t num_layers, int32_t channels, int32_t kernel_size,
                            const float* weights_host, uint64_t n_weights, iris_postnet_handle** out);
int32_t iris_postnet_destroy(iris_postnet_handle* h);
int32_t iris_postnet_workspace_bytes(const iris_postnet_handle* h, int32_t B, int32_t T, uint64_t* bytes);
/* mel_dev, out_dev: [B, n_mels, T] fp32 (channels-first, like the reference); asynchronous on `stream`. */
int32_t iris_postnet_forward(iris_postnet_handle* h, const void* mel_dev, int32_t B, int32_t T,
                             void* out_dev, void* workspace_dev, uint64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IRIS_HIFIGAN_H */
