/* masic_hip.h -- C ABI of libmasic_hip.so: the MI355X (gfx950) implementation of the MASIC
 * stereo-codec hot path (reference: ywz978020607/MASIC, coremasic/mywork/MASIC.py:744-851
 * `HSIC.forward`, and the compressai layers/entropy models it calls).
 *
 * The reference has no FFI on this path -- the path sits behind a Python nn.Module API and
 * every op below is an ATen call there.  This header is therefore the boundary a maintainer
 * of the reference binds with ctypes (INTEGRATION.md shows the stub); each entry point cites
 * the reference interface it replaces.
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no C++/torch types.  All tensors are NCHW,
 *     contiguous, float32 unless stated; int32 for symbol streams.
 *   - the library never allocates, frees or retains device memory: inputs, outputs and
 *     workspaces are caller-owned (torch tensors in the Python host layer).
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*); no implicit
 *     synchronisation, safe to capture into a hipGraph.
 *   - return value: 0 = MASIC_OK, negative = error; masic_last_error() returns a thread-local
 *     message.  Nothing throws, nothing calls exit().
 */
#ifndef MASIC_HIP_H
#define MASIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MASIC_ABI_VERSION 1

enum { MASIC_OK = 0, MASIC_ERR_ARG = -1, MASIC_ERR_SHAPE = -2, MASIC_ERR_LAUNCH = -3, MASIC_ERR_UNSUPPORTED = -4 };

/* activation fused into conv epilogues (nn.ReLU / nn.LeakyReLU(0.01), MASIC.py:176-182,338-376,679-689) */
enum { MASIC_ACT_NONE = 0, MASIC_ACT_RELU = 1, MASIC_ACT_LEAKY = 2,
       MASIC_ACT_SOFTMAX_C = 3 /* softmax over the output channels (mask2weights, MASIC.py:497-502); Cout <= 8 only */ };
/* transform applied to conv inputs while they are staged (torch.abs MASIC.py:185; torch.round
 * of `_quantize(...,'dequantize')` entropy_models.py:116 feeding context_prediction MASIC.py:755-757) */
enum { MASIC_INOP_NONE = 0, MASIC_INOP_ABS = 1, MASIC_INOP_ROUND = 2 };
/* operand precision of the MFMA contraction; accumulation is always float32 */
enum { MASIC_PREC_F32 = 0, MASIC_PREC_BF16 = 1,
       MASIC_PREC_FP8 = 2 /* OCP e4m3fn operands on v_mfma_scale_f32_32x32x64_f8f6f4 (conv_f8k / gemm_f8k entry points only) */ };

int masic_version(void);
const char* masic_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Convolutions.  Replaces torch.nn.Conv2d / ConvTranspose2d as built by
 * compressai/models/utils.py:128-146 (`conv`: padding=k//2; `deconv`: padding=k//2,
 * output_padding=stride-1), compressai/layers/layers.py:52-78 (MaskedConv2d type 'A') and
 * layers.py:81-83 (conv3x3).
 *
 * Input  view: channels [in_coff, in_coff+Cin)  of a buffer [B, in_ctot,  Hi, Wi]
 * Output view: channels [out_coff,out_coff+Cout) of a buffer [B, out_ctot, Ho, Wo]
 * (views let producers write straight into torch.cat targets, MASIC.py:765,827, and consumers
 * read slices of them).  Optional gate: out *= gate[b, gate_c, oh, ow] from a buffer
 * [B, gate_ctot, Ho, Wo] (the mask2weights products of MASIC.py:827).
 */
typedef struct {
    int32_t B, Cin, Hi, Wi, in_ctot, in_coff;
    int32_t Cout, Ho, Wo, out_ctot, out_coff;
    int32_t KH, KW, stride, pad;
    int32_t transposed;   /* 0: Conv2d weight [Cout,Cin,KH,KW]; 1: ConvTranspose2d weight [Cin,Cout,KH,KW] */
    int32_t masked;       /* 1: MaskedConv2d type 'A' -- only the taps before the centre are live */
    int32_t in_op;        /* MASIC_INOP_* */
    int32_t act;          /* MASIC_ACT_*  */
    int32_t gate_ctot, gate_c;   /* used when the gate pointer is non-NULL */
    int32_t prec;         /* MASIC_PREC_* */
} masic_conv_desc_t;

/* bytes of the packed-weight buffer for this layer (weights are re-laid out once per weight
 * version into the K-major, Cout-contiguous order the implicit-GEMM kernel stages into LDS) */
size_t masic_conv_packed_bytes(const masic_conv_desc_t* d);
int masic_conv_pack_weight(const float* w, void* w_packed, const masic_conv_desc_t* d, void* stream);
/* which kernel (and how many launches of it) masic_conv2d_fwd issues for this layer -- used by
 * bench.py to attribute HIP-event timings to kernel symbols:
 *   variant 0/1: conv_direct_f32<3>/<8>; 2/3/4/5/7: conv_igemm_f32<2 waves x {1x1,1x2,1x4,2x4,2x2} tiles>; 8/9: conv_igemm_f32<1 wave x {1x2,3x2}>;
 *   6: deconv5s2_small_cout; launches = phases (1, or
 *   stride^2 for transposed convs). Returns variant, writes *launches if non-NULL; <0 on bad desc. */
int masic_conv_variant(const masic_conv_desc_t* d, int* launches);
/* the kernel symbol (as rocprofv3 prints it, without namespace/arguments) that this layer launches */
int masic_conv_kernel_name(const masic_conv_desc_t* d, char* buf, size_t n);
/* y = act(conv(in_op(x), w) + bias) [* gate].  bias may be NULL. */
int masic_conv2d_fwd(const float* x, const void* w_packed, const float* bias, const float* gate,
                     float* y, const masic_conv_desc_t* d, void* stream);
/* same, plus up to two residual tensors [B,Cout,Ho,Wo] added after the activation: the `out + identity` of
 * ResidualBlock (compressai/layers/layers.py:189), Enhancement_Block (MASIC.py:163) and Independent_EN (:1495-1496) */
int masic_conv2d_fwd_ex(const float* x, const void* w_packed, const float* bias, const float* gate,
                        const float* res1, const float* res2, float* y, const masic_conv_desc_t* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * The 1x1 stacks of the GMM parameter heads (MASIC.py:330-468) as bf16 GEMMs whose intermediate activations stay in
 * the matrix-core-friendly layout F16K = [B][C/16][H*W][16] bf16 (see masic_amd/csrc/gemm_bf16.hip):
 *   masic_nchw_to_f16k     float32 NCHW channel view -> F16K (zero padded to a multiple of 16 channels)
 *   masic_gemm1x1_*        y = act(W x + b) with x in F16K; output either F16K (next layer) or float32 NCHW view.
 * `transposed` = 1 packs a ConvTranspose2d(k=1) weight [Cin,Cout] (the first two layers of the y1 stacks, :339-342). */
size_t masic_f16k_bytes(int B, int C, int HW);
int masic_nchw_to_f16k(const float* x, void* y, int B, int C, int HW, int ctot, int coff, void* stream);
/* the same with |x| / round(x) applied on the way (MASIC_INOP_*): the inputs of the hyper-analysis transform (MASIC.py:184)
 * and of the context model in eval mode (:770, :812) */
int masic_nchw_to_f16k_op(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int in_op, void* stream);
size_t masic_gemm1x1_packed_bytes(int Cin, int Cout);
int masic_gemm1x1_pack_weight(const float* w, void* wp, int Cin, int Cout, int transposed, void* stream);
int masic_gemm1x1_bf16_fwd(const void* x_f16k, const void* w_packed, const float* bias, void* y_f16k, float* y_nchw,
                           int B, int Cin, int Cout, int HW, int out_ctot, int out_coff, int act, void* stream);

/* The same 1x1 layers with both operands staged by DMA (masic_amd/csrc/conv_f16k.hip: gemm_f16k), Cin % 16 == 0 and
 * Cout % 32 == 0; own weight pack (per 128-channel block, 64-channel chunks in LDS-image order). */
size_t masic_gemm_f16k_packed_bytes(int Cin, int Cout);
int masic_gemm_f16k_pack_weight(const float* w, void* wp, int Cin, int Cout, int transposed, void* stream);
/* n (<= 18) such packs in one launch: the nine layers of an entropy-parameter head, forward and transposed (training step) */
int masic_gemm_f16k_pack_weights(const float* const* w, void* const* wp, const int* Cin, const int* Cout, const int* transposed, int n, void* stream);
int masic_gemm_f16k_fwd(const void* x_f16k, const void* w_packed, const float* bias, void* y_f16k, float* y_nchw,
                        int B, int Cin, int Cout, int HW, int out_ctot, int out_coff, int act, void* stream);
/* Up to three such layers over the same B x HW pixels in ONE launch: layer i of the sigma / means / weights stacks of a GMM head
 * (reference MASIC.py:330-468: nine 1x1 layers per head, three per stack) -- 640 ... 864 workgroups instead of three launches of
 * 192 ... 288.  All groups bf16 operands (wscale NULL: x is F16K) or all fp8 (wscale set: x is F8K, weights from
 * masic_gemm_f8k_pack_weight, wscale = weight scales x the input tensor's scale); per group exactly one of y_f16k / y_f8k (quantised
 * with out_inv_scale) / y_nchw.  Results equal the single-layer calls bit for bit. */
typedef struct masic_gemm_group {
    const void* x;
    const void* w_packed;
    const float* wscale;
    const float* bias;
    void* y_f16k;
    void* y_f8k;
    float* y_nchw;
    float out_inv_scale;
    int Cin, Cout, out_ctot, out_coff, act;
} masic_gemm_group_t;
int masic_gemm_f16k_group_fwd(const masic_gemm_group_t* groups, int ngroups, int B, int HW, void* stream);

/* ------------------------------------------------------------------------------------------
 * The same convolutions (nn.Conv2d / nn.ConvTranspose2d / MaskedConv2d of MASIC.py:510-622, :170-187, :690-700) with
 * the input -- and optionally the output -- in F16K: the bf16-operand forward keeps the 128/192-channel activations of
 * the analysis / synthesis / hyper transforms in that layout between layers, so both MFMA operands reach LDS by
 * 16-byte DMA (masic_amd/csrc/conv_f16k.hip).  The descriptor is the one of masic_conv2d_fwd; in_ctot/in_coff (and
 * out_ctot/out_coff for an F16K output) count channels of the F16K buffers and must be multiples of 16; in_op must
 * be NONE.  Exactly one of y_nchw (float32 channel view, optional gate as in masic_conv2d_fwd) / y_f16k is non-null.
 *   masic_conv_f16k_supported    1 when the layer shape has a configuration (Cout >= 64, Cin >= 16, ...), else 0
 *   masic_conv_f16k_packed_bytes / _pack_weight   [phase-tap][ci/16][co][16] bf16 weights for this path */
int masic_conv_f16k_supported(const masic_conv_desc_t* d);
int masic_conv_f16k_kernel_name(const masic_conv_desc_t* d, int gdn, char* buf, size_t n);   /* symbol as rocprofv3 prints it */
size_t masic_conv_f16k_packed_bytes(const masic_conv_desc_t* d);
int masic_conv_f16k_pack_weight(const float* w, void* w_packed, const masic_conv_desc_t* d, void* stream);
/* Batched packing (a training step re-packs every weight it touches, each for its forward and for its input gradient):
 * masic_conv_f16k_pack_job writes, into host memory (masic_conv_f16k_pack_job_bytes() bytes), the job that one
 * masic_conv_f16k_pack_weight(w, w_packed, d) call stands for and returns its grid width (>= 1; negative: error code);
 * masic_conv_f16k_pack_jobs_run launches a table of such jobs, held in DEVICE memory, as ONE kernel (max_nb = the largest grid
 * width in the table).  Pointers inside the jobs must stay valid; the weights are read when the kernel runs. */
size_t masic_conv_f16k_pack_job_bytes(void);
int masic_conv_f16k_pack_job(const float* w, void* w_packed, const masic_conv_desc_t* d, void* job_host);
int masic_conv_f16k_pack_jobs_run(const void* jobs_dev, int njobs, int max_nb, void* stream);
int masic_conv_f16k_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                        float* y_nchw, void* y_f16k, const masic_conv_desc_t* d, void* stream);
/* The same with the (inverse) GDN that follows the 128-channel convolutions of the analysis / synthesis transforms
 * (MASIC.py:521-531, :544-554; compressai/layers/gdn.py:77-92) fused into the epilogue: y = GDN(conv(x) + bias).
 * gdn_packed: masic_gdn_pack_f16k of that GDN's stored beta/gamma (masic_gdn_f16k_packed_bytes bytes).
 * gdn_inverse (here and in masic_conv_f8k_fwd / masic_conv_a_gdn_fwd*): bit 0 = inverse GDN; bit 1 = contract gamma^ x x^2 as the
 * three-product bf16 hi/lo split (~2^-16 of the result) instead of ONE bf16 product (~2^-10: the order of the bf16 rounding the
 * result gets when it is stored for the next bf16-operand layer; a third of the epilogue's MFMAs). */
size_t masic_gdn_f16k_packed_bytes(void);
int masic_gdn_pack_f16k(const float* beta, const float* gamma, void* packed, int C, double beta_min, void* stream);
int masic_conv_f16k_gdn_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                            const void* gdn_packed, int gdn_inverse, float* y_nchw, void* y_f16k,
                            const masic_conv_desc_t* d, void* stream);

/* encoder2.pre_conv + pre_gdn (MASIC.py:559-560, :574-576) and decoder2.after_gdn + torch.cat + after_conv (:599-600,
 * :617-621) as one launch each: Conv2d / ConvTranspose2d(6 -> 3, k5, s1, p2) on channels [xa(3) | xb(3)] of two float32
 * tensors, optional 3-channel (I)GDN on xa while it is staged (gin_*: stored beta[3], gamma[3,3]) and / or on the result
 * (gout_*).  w_packed: masic_conv_pack_weight of that layer.  Same float32 operation order as the separate kernels. */
int masic_conv5s1_pair_fwd(const float* xa, const float* xb, const float* w_packed, const float* bias,
                           const float* gin_beta, const float* gin_gamma, int gin_inverse,
                           const float* gout_beta, const float* gout_gamma, int gout_inverse, double beta_min,
                           float* y, int B, int H, int W, void* stream);

/* The last synthesis layer g_s_conv4 = ConvTranspose2d(128 -> 3, k5, s2) (MASIC.py:550, :598) as its equivalent stride-1
 * 3x3 convolution to (2x2 phases) x C channels with a depth-to-space store.  `d`: that Conv2d(Cin -> 32 (zero padded), k3,
 * s1, p1) with weight row (4c + phase) = W_t[:, c, phase_h + 2(2-u), phase_w + 2(2-v)] (zero where the index exceeds 4;
 * phase = 2 phase_h + phase_w) and bias row (4c + phase) = bias[c]; y_nchw: [B][y_ctot][2 Hi][2 Wi], channels
 * y_coff .. y_coff+C-1 are written. */
int masic_conv_f16k_d2s_fwd(const void* x_f16k, const void* w_packed, const float* bias, float* y_nchw, int C,
                            int y_ctot, int y_coff, const masic_conv_desc_t* d, void* stream);

/* First analysis layer g_a_conv1 + g_a_gdn1 (MASIC.py:515-516, :563-564) in one kernel: Conv2d(3 -> 128, k5, s2, p2) on
 * channels in_coff..in_coff+2 of a float32 NCHW tensor, GDN, result in F16K [B][8][Ho*Wo][16]. */
/* ------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3fn) operand path -- BASELINE.json configs[4] ("fp8 MFMA conv path + int32 symbol quantise").  No reference
 * counterpart (the reference computes in float32): a declared-budget approximation of the layers above, gated against the
 * oracle in tests/test_gpu_fp8.py.  Activations: F8K = [B][C/32][H*W][32] fp8, stored as fp8(x / scale) with one scale per
 * tensor; weights: one scale per output channel, written by the pack calls.  Contraction:
 * v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales, float32 accumulate; dequantisation (wscale[co] = weight scale x
 * input scale) in the epilogue (masic_amd/csrc/conv_f16k.hip, fp8.hip).
 *   masic_nchw_to_f8k            float32 NCHW channel view -> F8K (in_op as masic_nchw_to_f16k_op), x * inv_scale, saturating
 *   masic_absmax                 *out = max(*out, max|x|) over n float32 (bf16 = 0) or bf16 (bf16 = 1) values: calibration
 *   masic_conv_f8k_pack_weight   d->prec = MASIC_PREC_FP8: fp8 slab stream + wscale[Cout] (max|W[co]| / 448)
 *   masic_conv_f8k_fwd           d->prec = MASIC_PREC_FP8: x is F8K, wscale required; d->prec = MASIC_PREC_BF16: x is F16K, wscale NULL
 *                                (= masic_conv_f16k_gdn_fwd).  Exactly one of y_nchw / y_f16k / y_f8k (fp8(y * out_inv_scale)).
 *   masic_gemm_f8k_*             the 1x1 layers (masic_gemm_f16k_*) with fp8 operands
 *   masic_conv_a_gdn_fwd_ex      masic_conv_a_gdn_fwd with the option of an F8K output */
size_t masic_f8k_bytes(int B, int C, int HW);
int masic_nchw_to_f8k(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int in_op, float inv_scale, void* stream);
int masic_absmax(const void* x, size_t n, int bf16, float* out, void* stream);
int masic_conv_f8k_pack_weight(const float* w, void* w_packed, float* wscale, const masic_conv_desc_t* d, void* stream);
int masic_conv_f8k_fwd(const void* x, const void* w_packed, const float* wscale, const float* bias, const float* gate,
                       const void* gdn_packed, int gdn_inverse, float* y_nchw, void* y_f16k, void* y_f8k, float out_inv_scale,
                       const masic_conv_desc_t* d, void* stream);
size_t masic_gemm_f8k_packed_bytes(int Cin, int Cout);
int masic_gemm_f8k_pack_weight(const float* w, void* wp, float* wscale, int Cin, int Cout, int transposed, void* stream);
int masic_gemm_f8k_fwd(const void* x_f8k, const void* w_packed, const float* wscale, const float* bias, void* y_f16k, void* y_f8k,
                       float* y_nchw, float out_inv_scale, int B, int Cin, int Cout, int HW, int out_ctot, int out_coff, int act, void* stream);
int masic_conv_a_gdn_fwd_ex(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                            void* y_f16k, void* y_f8k, float out_inv_scale, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream);

/* training-mode forms of the two fused conv + GDN entry points: F16K output plus the convolution's result BEFORE the GDN
 * (y_pre_f16k, same layout) -- the GDN backward needs its input, the next layer's weight gradient its output
 * (masic_amd/autograd.py: AnalysisFn / SynthesisFn run the inference kernels in the training step and keep bf16 activations) */
int masic_conv_f16k_gdn_dual_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                                 void* y_pre_f16k, void* y_f16k, const masic_conv_desc_t* d, void* stream);
int masic_conv_a_gdn_dual_fwd(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                              void* y_pre_f16k, void* y_f16k, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream);
/* Training step, input gradients of the picture-end layers (newtrain_codec_real.py:141 `loss.backward()` through MASIC.py:515, :550):
 * masic_conv_a_fwd -- Conv2d(3 -> 128, k5, s2, p2) alone (the first-layer kernel without its GDN) on a float32 NCHW channel view,
 *   F16K out: the input gradient of g_s_conv4 = ConvTranspose2d(128 -> 3) (its weight tensor [128][3][5][5] is that convolution's);
 * masic_deconv_s2_as_conv_weight -- ConvTranspose2d(Cin -> C <= 8, k5, s2, p2, op1) weight [Cin][C][5][5] (+ bias [C] or NULL) ->
 *   weight [32][Cin][3][3] and bias [32] of the stride-1 convolution masic_conv_f16k_d2s_fwd runs; with a Conv2d(C -> Cin) weight
 *   [Cin][C][5][5] passed as it is, that launch computes the convolution's input gradient (g_a_conv1). */
int masic_conv_a_fwd(const float* x, const void* w_packed, const float* bias, void* y_f16k, int B, int Hi, int Wi, int in_ctot, int in_coff,
                     void* stream);
int masic_deconv_s2_as_conv_weight(const float* w, const float* bias, float* w_out, float* bias_out, int Cin, int C, void* stream);
/* weight gradient of Conv2d(Cin -> Cout, k3, s1, p1) with both operands in F16K (Cin, Cout multiples of 32): the 3x3 layers of
 * Independent_EN in the CQE training step (newtrain_cqe_real.py:128-174).  Pixel-major records become MFMA operands through the
 * hardware transpose read ds_read_b64_tr_b16 (masic_amd/csrc/wgrad_f16k.hip).  dw: float32 [Cout][Cin][3][3]; workspace:
 * masic_conv3x3_wgrad_f16k_workspace_bytes bytes (zeroed by the call). */
size_t masic_conv3x3_wgrad_f16k_workspace_bytes(int Cin, int Cout);
int masic_conv3x3_wgrad_f16k(const void* x_f16k, const void* dy_f16k, float* dw, void* workspace,
                             int B, int Cin, int Cout, int H, int W, void* stream);
int masic_conv3x3_wgrad_f16k_ws(const void* x_f16k, const void* dy_f16k, float* dw, void* workspace,
                             int B, int Cin, int Cout, int H, int W, int workspace_clean, void* stream);
/* The same for Conv2d(Cin -> Cout, k5, s1, p2): encode_hyper's first layer and the context model at latent resolution
 * (reference MASIC.py:170-187, :627; a MaskedConv2d gets the dense gradient -- the reference masks weight.data, not the gradient).
 * dw: float32 [Cout][Cin][5][5]; workspace_clean as masic_conv2d_wgrad_ws. */
/* Weight gradient (float32, 128 x 3 x 5 x 5) of the two picture-end 5x5 stride-2 layers in the bf16 mode -- g_a_conv1 = Conv2d(3 -> 128)
 * (MASIC.py:515) and g_s_conv4 = ConvTranspose2d(128 -> 3) (:550): p_f16k = the 128-channel tensor at Hc x Wc in F16K (dy of the
 * former: dW = [Cout][Cin][5][5]; x of the latter: dW = [Cin][Cout][5][5]), q = the 3-channel tensor at 2 Hc x 2 Wc, float32 NCHW
 * (channels q_coff .. q_coff + 2 of q_ctot).  Per-workgroup partials + a finishing pass: deterministic.  workspace:
 * masic_pic_wgrad_f16k_workspace_bytes() bytes, contents irrelevant on entry and exit. */
size_t masic_pic_wgrad_f16k_workspace_bytes(void);
int masic_pic_wgrad_f16k(const void* p_f16k, const float* q, float* dw, void* workspace, int B, int Hc, int Wc,
                         int q_ctot, int q_coff, void* stream);
size_t masic_conv5x5_wgrad_f16k_workspace_bytes(int Cin, int Cout);
int masic_conv5x5_wgrad_f16k_ws(const void* x_f16k, const void* dy_f16k, float* dw, void* workspace,
                                int B, int Cin, int Cout, int H, int W, int workspace_clean, void* stream);
/* Weight gradient of the 1x1 layers (the entropy-parameter stacks, reference MASIC.py:330-468) from F16K operands:
 * dw[a][q] = sum over batch and pixels of rows[b][a][p] * cols[b][q][p]; rows / cols: F16K [B][C/16][HW][16] bf16 with CA / CQ
 * channels (multiples of 16); dw float32 [CA][CQ].  Conv2d weight [Cout][Cin]: rows = dy, cols = x; ConvTranspose2d(k1) weight
 * [Cin][Cout]: rows = x, cols = dy. */
int masic_gemm_wgrad_f16k(const void* rows_f16k, const void* cols_f16k, float* dw, int B, int CA, int CQ, int HW, void* stream);
/* ... with the layer's bias gradient in the same launch: bias_of 1: dw[CA*CQ .. CA*CQ + CA) = sums of the rows operand over batch and
 * pixels (dy of a Conv2d), 2: dw[CA*CQ .. CA*CQ + CQ) = sums of the columns operand (dy of a ConvTranspose2d); dw then holds that
 * many more floats.  0: masic_gemm_wgrad_f16k. */
int masic_gemm_wgrad_bias_f16k(const void* rows_f16k, const void* cols_f16k, float* dw, int bias_of, int B, int CA, int CQ, int HW, void* stream);
/* F16K in, F16K out with up to two F16K residual tensors added after the activation: out = act(conv(x) + bias) + res1 [+ res2]
 * (ResidualBlock: compressai/layers/layers.py:160-190; Enhancement_Block: MASIC.py:149-164) -- Independent_EN with bf16
 * operands keeps its 32 / 64 / 96-channel full-resolution activations in F16K.  y_f16k is a channel view (d->out_ctot / out_coff). */
int masic_conv_f16k_res_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                            void* y_f16k, const masic_conv_desc_t* d, void* stream);
/* the same with the pieces the CQE training step needs (masic_amd/autograd.py: EnhancementBlockFn), everything F16K:
 *   y = act(conv(x) + bias) * act'(mask) + res1 + res2;  y_pre (optional) = the value before the residual adds.
 * Forward: mask NULL, y_pre = the LeakyReLU output (its sign is the mask of the backward).  Input gradient (d->transposed on the same
 * weight): mask = the producer's activation output, mask_slope 0.01 (LeakyReLU) / 0 (ReLU), res1 / res2 = gradients of the identity paths. */
int masic_conv_f16k_res_ex_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                               const void* mask, float mask_slope, void* y_pre_f16k, void* y_f16k, const masic_conv_desc_t* d, void* stream);
/* F16K elementwise / reduction helpers of that backward: out = g * act'(y); out[c] = sum over (b, pixel) of x (a bias gradient) */
int masic_f16k_act_bwd(const void* g, const void* y, void* out, size_t n, float slope, void* stream);
size_t masic_f16k_channel_sum_workspace_bytes(int B, int C);
int masic_f16k_channel_sum(const void* x, float* out, void* workspace, int B, int C, int HW, void* stream);
/* out = g * act'(y) (as masic_f16k_act_bwd) and sums[C] = its per-channel sums (as masic_f16k_channel_sum of out) in one pass over the
 * tensors; workspace: masic_f16k_channel_sum_workspace_bytes(B, C) */
int masic_f16k_act_bwd_sum(const void* g, const void* y, void* out, float* sums, void* workspace, int B, int C, int HW, float slope, void* stream);
/* mask2weights_EN (reference MASIC.py:1411-1434, Kw = 2) in one launch: gates [B,2,H,W] = softmax over channels of four 3x3
 * stride-1 convolutions 1 -> 2 -> 4 -> 4 -> 2 with ReLUs on mask [B,1,H,W]; w_i [Cout][Cin][3][3], b_i [Cout], float32 on the
 * device.  Bit-identical to the four-launch form (masic_conv2d_fwd per layer). */
int masic_mask2weights_en_fwd(const float* mask, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                              const float* b3, const float* w4, const float* b4, float* gates, int B, int H, int W, void* stream);
/* Conv2d(Cin -> Cout, k3, s1, p1) on F16K with the whole weight tensor resident in LDS and persistent workgroups: Cout = 32 | 64,
 * Cin <= Cout (the input buffer holds round_up(Cin, 16) channels; a 3- or 6-channel picture is one zero-padded record per pixel) --
 * the 32- / 64-channel stages of Independent_EN and its input layers (reference MASIC.py:149-164, 1456-1482 / layers.py:99-121;
 * conv_f16k spends three quarters of such a launch in per-workgroup prologue / epilogue).  Same operand meaning as
 * masic_conv_f16k_res_ex_fwd; `transposed` in the pack selects the slabs of the INPUT gradient of a Conv2d(Cout -> Cin) layer
 * (weight [Cin][Cout][3][3]).  Needs W % 32 == 0 and H % 16 == 0 (Cout = 32) / H % 8 == 0 (Cout = 64). */
size_t masic_conv3x3_resident_packed_bytes(int Cin, int Cout);
int masic_conv3x3_resident_supported(int B, int Cin, int Cout, int H, int W);
int masic_conv3x3_resident_pack_weight(const float* w, void* w_packed, int Cin, int Cout, int transposed, void* stream);
int masic_conv3x3_resident_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                               const void* mask, float mask_slope, void* y_pre_f16k, void* y_f16k, int B, int Cin, int Cout, int H, int W,
                               int in_ctot, int in_coff, int out_ctot, int out_coff, int act, void* stream);
/* ... for the backward chain of a residual block (autograd of reference layers.py:160-190 / MASIC.py:149-164 through the node of
 * masic_amd/autograd.py: EnhancementBlockFn): additionally y2 = bf16(y) * act'(mask2) -- the next layer's dy, exactly what a separate
 * g * act'(u) pass over the stored y would give -- and sum_out[Cout] = per-channel sums (a bias gradient) of the value before the residual
 * adds (sum_of = 1) or of y2 (sum_of = 2), bf16-rounded as stored, from per-wave partials in sum_workspace
 * (masic_conv3x3_resident_sum_workspace_bytes()) by a fixed-order finishing launch.  mask2 / y2: res_ctot-channel tensors. */
size_t masic_conv3x3_resident_sum_workspace_bytes(void);
int masic_conv3x3_resident_ex_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                                  const void* mask, float mask_slope, void* y_pre_f16k, void* y_f16k, const void* mask2, float mask2_slope,
                                  void* y2_f16k, int sum_of, float* sum_out, void* sum_workspace, int B, int Cin, int Cout, int H, int W,
                                  int in_ctot, int in_coff, int out_ctot, int out_coff, int act, void* stream);
/* diagnostics: 16 uint64 on the device, filled by every following conv_f16k launch with {core-clock, 100 MHz} stamp pairs at kernel
 * entry, K-loop entry, K-loop exit and kernel exit of its first and of its last workgroup; NULL switches it off (tools/f16k_stamps.py) */
void masic_conv_f16k_set_stamps(void* device_buffer);
/* a layer with few output channels (Independent_EN.conv2: 96 -> 3 + the picture as residual, MASIC.py:1492-1496) on the same kernels:
 * `d` describes the convolution with its weight zero-padded to Cout = 32; y_nchw / res32 are float32 [B][cout_store][Ho][Wo]. */
int masic_conv_f16k_few_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* res32, float* y_nchw,
                            int cout_store, const masic_conv_desc_t* d, void* stream);
/* layout-side helpers of those chains (masic_amd/csrc/f16k_ops.hip):
 *   masic_f16k_gate           dst[:, dst_coff : dst_coff+C] = (minv ? warp_perspective(src, minv) : src) * (gate ? gate[:, gate_c] : 1)
 *                             -- the gated concats of MASIC.py:1470-1482 written straight into their slice of the F16K buffer
 *   masic_nchw_to_f16k_view   float32 NCHW channel view -> a channel slice of an F16K buffer
 *   masic_f16k_to_nchw        F16K channel slice -> float32 NCHW channel view */
int masic_f16k_gate(const void* src, const float* gate, const float* minv, void* dst, int B, int C, int H, int W,
                    int dst_ctot, int dst_coff, int gate_ctot, int gate_c, void* stream);
int masic_nchw_to_f16k_view(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int dst_ctot, int dst_coff, void* stream);
/* ... with |x| / round(x) (in_op) applied and an optional gate[b][gate_c][pixel] (float32 [B][gate_ctot][HW]) multiplied in before the
 * rounding to bf16: round(y) * gate into its slice of an F16K concat buffer (the gated concat of MASIC.py:827). */
int masic_nchw_to_f16k_view_op(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int dst_ctot, int dst_coff, int in_op,
                               const float* gate, int gate_ctot, int gate_c, void* stream);
int masic_f16k_to_nchw(const void* x, float* y, int B, int C, int HW, int src_ctot, int src_coff, int ctot, int coff, void* stream);
int masic_f16k_to_nchw_bf16(const void* x, void* y, int B, int C, int HW, int src_ctot, int src_coff, int ctot, int coff, void* stream);   /* y: bf16 NCHW */

size_t masic_conv_a_packed_bytes(void);
int masic_conv_a_pack_weight(const float* w, void* w_packed, void* stream);
int masic_conv_a_gdn_fwd(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                         void* y_f16k, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream);

/* ------------------------------------------------------------------------------------------
 * GDN / inverse GDN: compressai/layers/gdn.py:77-92 with the NonNegativeParametrizer of
 * compressai/ops/parametrizers.py:47-64 applied to the *stored* beta[C], gamma[C,C] inside the
 * kernel: y = x * rsqrt(beta^ + gamma^ . x^2)  (inverse: * sqrt).  beta_min as gdn.py:57.
 */
int masic_gdn_fwd(const float* x, const float* beta, const float* gamma, float* y,
                  int B, int C, int H, int W, int inverse, double beta_min, void* stream);
/* prec = MASIC_PREC_BF16: the C x C contraction runs as a bf16x3 split product (gamma^ and x^2 as bf16 hi+lo, float32
 * accumulate, ~2^-16 relative) at 16/3 of the f32 matrix rate -- HBM-bound; MASIC_PREC_F32: exact float32 MFMA */
int masic_gdn_fwd_ex(const float* x, const float* beta, const float* gamma, float* y,
                     int B, int C, int H, int W, int inverse, double beta_min, int prec, void* stream);
/* Simplified GDN ("GDN1", compressai/layers/gdn.py:95-121): y_i = x_i / (beta^_i + sum_j gamma^_ij |x_j|) (inverse: multiply); float32, any C */
int masic_gdn1_fwd(const float* x, const float* beta, const float* gamma, float* y,
                   int B, int C, int H, int W, int inverse, double beta_min, void* stream);
/* Same, C = 128 only, result written as F16K bf16 [B][8][H*W][16] for masic_conv_f16k_fwd (bf16x3 contraction). */
int masic_gdn_fwd_f16k(const float* x, const float* beta, const float* gamma, void* y_f16k,
                       int B, int C, int H, int W, int inverse, double beta_min, void* stream);

/* ------------------------------------------------------------------------------------------
 * Quantisation: compressai/entropy_models/entropy_models.py:98-125 with means=None.
 *   mode 3 'copy'      : y = x             (used with the gate / output view: x*w into a torch.cat slice, MASIC.py:1470-1482)
 *   mode 0 'dequantize': y = round(x)      (torch.round: half to even)
 *   mode 1 'noise'     : y = x + noise     (noise drawn by the caller, same shape)
 *   mode 2 'symbols'   : sym = (int32) round(x - median[c])   (median may be NULL -> 0)
 * Optional gate as above (y *= gate[b,gate_c,h,w]); output view as for convs.
 */
int masic_quantize_fwd(const float* x, const float* noise, const float* gate, float* y,
                       int B, int C, int H, int W, int out_ctot, int out_coff,
                       int gate_ctot, int gate_c, int mode, void* stream);
int masic_symbols_fwd(const float* x, const float* median, int32_t* sym,
                      int B, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * EntropyBottleneck.forward: entropy_models.py:384-411 (+ _likelihood :372-382,
 * _logits_cumulative :350-369, LowerBound 1e-9).  z [B,C,H,W]; params are the module's raw
 * parameters, concatenated per channel by the host into one table [C, 58]:
 *   matrices (3 + 9 + 9 + 9 + 3), biases (3 + 3 + 3 + 3 + 1), factors (3 + 3 + 3 + 3)
 * (filters (3,3,3,3), entropy_models.py:258); medians = quantiles[:,0,1].
 * training=1: z_hat = z + noise, noise given in the reference's draw layout (C, 1, B*H*W)
 * (entropy_models.py:386-394); training=0: z_hat = round(z - med) + med.
 */
#define MASIC_EB_PARAMS_PER_CHANNEL 58
int masic_entropy_bottleneck_fwd(const float* z, const float* params, const float* medians,
                                 const float* noise, float* z_hat, float* lik,
                                 int B, int C, int H, int W, int training, float lik_bound, void* stream);
/* EntropyBottleneck.loss(): entropy_models.py:345-348. out[0] = sum |logits(quantiles) - target| */
int masic_entropy_bottleneck_auxloss(const float* params, const float* quantiles, float* out,
                                     int C, double tail_mass, void* stream);

/* ------------------------------------------------------------------------------------------
 * GaussianMixtureConditional_gf.forward: entropy_models.py:808-858.
 *   y_hat = quantize(y) (mode as masic_quantize_fwd 0/1), K-component mixture likelihood with
 *   sigma lower-bounded at scale_bound (0.11) and the result at lik_bound (1e-9).
 * sigma, mu, wts: [B, K*M, H, W], component k of channel m at k*M+m (MASIC.py:389-393).
 * weights_are_logits=1 fuses the softmax over K of MASIC.py:389-393/459-464 into the kernel
 * (wts then holds the raw gmm_weights head output, and wts_out, if non-NULL, receives the
 * normalised weights).
 * training: 0 round(y), 1 y + noise, 2 y is the quantised latent already -- likelihood only, y_hat may be NULL
 * (the training forward draws y + noise where the reference does and evaluates the likelihood beside the synthesis transform).
 */
int masic_gmm_likelihood_fwd(const float* y, const float* noise, const float* sigma, const float* mu,
                             const float* wts, float* y_hat, float* lik, float* wts_out,
                             int B, int M, int K, int H, int W, int training, int weights_are_logits,
                             float scale_bound, float lik_bound, void* stream);
/* softmax over K on the (B,K,M,H,W) view: MASIC.py:389-393 */
int masic_softmax_k_fwd(const float* x, float* y, int B, int M, int K, int HW, void* stream);

/* ------------------------------------------------------------------------------------------
 * Perspective warp: kornia==0.5.0 warp_perspective(src, M, dsize) as called at
 * MASIC.py:638,644,781,821,833 (bilinear, zeros padding, align_corners=True).
 * minv_norm [B,3,3] = inverse(N_dst . M . N_src^-1) (normalised coordinates), computed by
 * masic_warp_matrix from the pixel-space M [B,3,3] in float64 and rounded to float32.
 * src NULL => warp an all-ones image (the mask of MASIC.py:636-638) with C=1.
 */
int masic_warp_matrix(const float* M, float* minv_norm, int B, int Hs, int Ws, int Hd, int Wd,
                      int invert_first, void* stream);
/* grid_sample convention of every warp of the library (forward, backward, masic_f16k_gate): 1 = align_corners=True -- kornia 0.5.0's
 * warp_perspective default, what the reference's pinned dependency runs (readme.md:12) and this library's default; 0 =
 * align_corners=False (kornia <= 0.4.1).  kornia is not vendored in the reference and no reference test pins a result at this
 * boundary, so the convention stays selectable.  Process-wide; set once before the first forward. */
void masic_set_warp_align_corners(int align_corners);
int masic_get_warp_align_corners(void);
int masic_warp_perspective_fwd(const float* src, const float* minv_norm, float* dst,
                               int B, int C, int Hs, int Ws, int Hd, int Wd,
                               int out_ctot, int out_coff, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small elementwise / reduction helpers.
 *   mul_inplace : MaskedConv2d's `weight.data *= mask` (layers.py:77)
 *   copy_view   : write a [B,C,H,W] tensor into a channel slice of a wider buffer (torch.cat)
 *   sum_log / sse : the reductions of RateDistortionLoss (newtrain_codec_real.py:79-83):
 *                   out[0] = sum(log(x)), out[0] = sum((a-b)^2); deterministic two-stage, fp64 final.
 *   workspace: at least masic_reduce_workspace_bytes() bytes.
 */
int masic_mul_inplace(float* x, const float* m, size_t n, void* stream);
/* LowerBound / LowerBoundFunction: compressai/ops/bound_ops.py:36-56 (standalone form; the hot path
 * fuses the bound into the GDN / EB / GMM kernels) */
int masic_lower_bound_fwd(const float* x, float* y, float bound, size_t n, void* stream);
int masic_lower_bound_bwd(const float* x, const float* g, float* gx, float bound, size_t n, void* stream);
int masic_copy_view(const float* x, float* y, int B, int C, int HW, int out_ctot, int out_coff, void* stream);
size_t masic_reduce_workspace_bytes(void);
int masic_sum_log(const float* x, size_t n, double* out, void* workspace, void* stream);
int masic_sse(const float* a, const float* b, size_t n, double* out, void* workspace, void* stream);
/* The rate-distortion criterion of the training step (newtrain_codec_real.py:73-87) in two launches: loss (float32) =
 * cm (mse1 + mse2) + bpp, mse_i = sum (x_i_hat - x_i)^2 / n_pic, bpp = cb sum_k sum log lik_k, per[k] = cb sum log lik_k (float64 device
 * scalars); nliks <= 4 (0: the distortion-only criterion of the CQE stage, newtrain_cqe_real.py:66-96).  Sums in the order of masic_sse /
 * masic_sum_log.  workspace: masic_rd_loss_workspace_bytes().  masic_rd_loss_bwd: its gradients times the float32 device scalar g in one
 * launch: g_x_i = (s_pic (x_i_hat - x_i)) g, g_liks[k] = (s_lik / lik_k) g. */
size_t masic_rd_loss_workspace_bytes(void);
int masic_rd_loss(const float* x1_hat, const float* x1, const float* x2_hat, const float* x2, size_t n_pic,
                  const float* const* liks, const size_t* lik_n, int nliks, double cb, double cm,
                  float* loss, double* mse1, double* mse2, double* bpp, double* const* per, void* workspace, void* stream);
int masic_rd_loss_bwd(const float* x1_hat, const float* x1, const float* x2_hat, const float* x2, size_t n_pic,
                      const float* const* liks, const size_t* lik_n, int nliks, float s_pic, float s_lik, const float* g,
                      float* g_x1, float* g_x2, float* const* g_liks, void* stream);

/* ==========================================================================================
 * Backward entry points (what torch autograd derives for the reference graph).
 * ========================================================================================== */

/* Weight gradient of Conv2d / ConvTranspose2d (same descriptor as the forward layer):
 *   Conv2d:          dW[co,ci,kh,kw] = sum_{b,oh,ow} dy[b,co,oh,ow] * x[b,ci,oh*s+kh-p,ow*s+kw-p]
 *   ConvTranspose2d: dW[ci,co,kh,kw] = sum_{b,ih,iw} x[b,ci,ih,iw] * dy[b,co,ih*s+kh-p,iw*s+kw-p]
 * x is the forward input view (in_ctot/in_coff of desc), dy a contiguous [B,Cout,Ho,Wo] tensor.
 * Pixel tiles are reduced with float atomics into `workspace` (>= masic_conv2d_wgrad_workspace_bytes(d) bytes,
 * zero-filled by the call itself with a stream-ordered memset) and then transposed into dw (weight layout).
 * Input gradients need no entry point of their own: dx of a Conv2d is masic_conv2d_fwd with transposed=1 and
 * Cin/Cout swapped on the same weight tensor (and vice versa), see masic_amd/autograd.py. */
size_t masic_conv2d_wgrad_workspace_bytes(const masic_conv_desc_t* d);
int masic_conv2d_wgrad(const float* x, const float* dy, float* dw, void* workspace, const masic_conv_desc_t* d, void* stream);
/* The same with a caller-owned PERSISTENT workspace: workspace_clean != 0 promises zeros on entry and gets zeros back on exit (the
 * last pass of the call re-zeroes what it reads), so a training step's ~80 weight gradients need no fill launch each. */
int masic_conv2d_wgrad_ws(const float* x, const float* dy, float* dw, void* workspace, const masic_conv_desc_t* d, int workspace_clean,
                          void* stream);
/* The 5x5 stride-2 layers of the bf16 mode with bf16 NCHW operands (x: [B][in_ctot][Hi][Wi], dy: [B][Cout][Ho][Wo], both bf16): the
 * GDN backward writes dx that way (masic_gdn_bwd_fused_ex2) and the saved F16K activations are converted that way
 * (masic_f16k_to_nchw_bf16) -- half the bytes of the float32 tensors on every side of the kernel. */
int masic_conv2d_wgrad_bf16in_supported(const masic_conv_desc_t* d);
int masic_conv2d_wgrad_bf16in(const void* x_bf16, const void* dy_bf16, float* dw, void* workspace, const masic_conv_desc_t* d,
                              int workspace_clean, void* stream);

/* y = op(a, b) elementwise over n floats; ops (s0,s1 scalars):
 *   0 act_bwd (a=grad, b=activation output, s0=MASIC_ACT_*)   1 abs_bwd (a=grad, b=x)   2 square   3 abs
 *   4 axpy s0*a+s1*b (b may be NULL)   5 s0/a   6 s0*(a-b)   7 a*b
 *   8 reparam max(a,s0)^2-s1 (parametrizers.py:61-64)   9 reparam_bwd (a=grad, b=stored param, s0=bound; incl. LowerBound rule)
 *   10 a+b */
int masic_elementwise(const float* a, const float* b, float* y, size_t n, int op, float s0, float s1, void* stream);
/* out[c] = sum over batch and pixels of channel coff+c of x[B,ctot,HW]  (bias / beta gradients; deterministic two-stage
 * reduction through `workspace` >= masic_channel_sum_workspace_bytes(C) bytes) */
size_t masic_channel_sum_workspace_bytes(int C);
int masic_channel_sum(const float* x, float* out, void* workspace, int B, int C, int HW, int ctot, int coff, void* stream);
/* y[B,C,HW] = x[B,ctot,HW][:, coff:coff+C]  (backward of copy_view / torch.cat) */
int masic_slice_copy(const float* x, float* y, int B, int C, int HW, int ctot, int coff, void* stream);
/* backward of y = x * gate[:,gate_c]: gx = g*gate, ggate[:,gate_c] = sum_c g*x (other gate channels untouched) */
int masic_gate_bwd(const float* g, const float* x, const float* gate, float* gx, float* ggate,
                   int B, int C, int HW, int gate_ctot, int gate_c, void* stream);
int masic_softmax_k_bwd(const float* g, const float* y, float* gx, int B, int M, int K, int HW, void* stream);
/* GDN backward, elementwise pieces (gdn.py:77-92; the three CxC contractions run on the conv / wgrad kernels):
 *   pre : s = g*n^(-1/2), t = dL/dn = -1/2 g x n^(-3/2)   (inverse: s = g*n^(1/2), t = +1/2 g x n^(-1/2))
 *   post: dx = s + 2 x u,  u = gamma^T t */
int masic_gdn_bwd_pre(const float* x, const float* nrm, const float* g, float* s, float* t, size_t n, int inverse, void* stream);
int masic_gdn_bwd_post(const float* x, const float* s, const float* u, float* dx, size_t n, void* stream);
/* The same backward for C <= 4 (MASIC.py:560 pre_gdn, :575 after_gdn: GDN(3) on full-resolution pictures) in one pass: float32
 * per-pixel algebra in registers, parameter sums reduced in float64 in a fixed order.  beta [C], gamma [C][C]: the STORED tensors;
 * g_beta, g_gamma: gradients with respect to them (reparametrisation rules included, as masic_gdn_bwd_fused). */
size_t masic_gdn_bwd_small_workspace_bytes(void);
int masic_gdn_bwd_small(const float* x, const float* g, const float* beta, const float* gamma, float* gx, float* g_beta,
                        float* g_gamma, void* workspace, int B, int C, int H, int W, int inverse, double beta_min, void* stream);
/* The whole GDN / inverse-GDN backward (gdn.py:77-92 under autograd, C = 128) in one pass over x and g = dL/dy, bf16
 * operands on the matrix cores with float32 accumulation -- the training step of the bf16-operand mode; the pieces above
 * remain the float32 parity path.  beta, gamma: the STORED tensors; g_beta [128], g_gamma [128][128]: gradients with
 * respect to them (NonNegativeParametrizer + LowerBound rules, parametrizers.py:61-64, bound_ops.py:40-42, included).
 * workspace: masic_gdn_bwd_fused_workspace_bytes() bytes of device memory. */
size_t masic_gdn_bwd_fused_workspace_bytes(void);
int masic_gdn_bwd_fused(const float* x, const float* g, const float* beta, const float* gamma, float* gx,
                        float* g_beta, float* g_gamma, void* workspace, int B, int C, int H, int W, int inverse,
                        double beta_min, void* stream);
/* The same with F16K operands: exactly one of x / x_f16k and of g / g_f16k (float32 NCHW or F16K bf16 [B][8][HW][16]); dx goes to
 * gx (float32 NCHW) and / or gx_f16k; g_sum [128] (or NULL) receives the per-channel sums of dx over batch and pixels -- the bias
 * gradient of the convolution in front of the GDN (autograd of `conv -> GDN`, reference MASIC.py:515-529, :538-552). */
int masic_gdn_bwd_fused_ex(const float* x, const void* x_f16k, const float* g, const void* g_f16k, const float* beta, const float* gamma,
                           float* gx, void* gx_f16k, float* g_sum, float* g_beta, float* g_gamma, void* workspace,
                           int B, int C, int H, int W, int inverse, double beta_min, void* stream);
/* ... with one more output: gx_bf16 (or NULL) = dx as bf16 NCHW [B][128][H][W], the operand of masic_conv2d_wgrad_bf16in. */
int masic_gdn_bwd_fused_ex2(const float* x, const void* x_f16k, const float* g, const void* g_f16k, const float* beta, const float* gamma,
                            float* gx, void* gx_f16k, void* gx_bf16, float* g_sum, float* g_beta, float* g_gamma, void* workspace,
                            int B, int C, int H, int W, int inverse, double beta_min, void* stream);
/* GaussianMixtureConditional_gf backward (entropy_models.py:808-858 + both LowerBound rules, bound_ops.py:40-42).
 * y_hat as returned by the forward; g_yhat may be NULL; weights_are_logits as in the forward. */
int masic_gmm_likelihood_bwd(const float* y_hat, const float* sigma, const float* mu, const float* wts,
                             const float* g_lik, const float* g_yhat, float* g_y, float* g_sigma, float* g_mu,
                             float* g_w, int B, int M, int K, int H, int W, int weights_are_logits,
                             float scale_bound, float lik_bound, void* stream);
/* EntropyBottleneck backward: g_z [B,C,H,W] and g_params [C,58] (same table layout as the forward) */
int masic_entropy_bottleneck_bwd(const float* z_hat, const float* params, const float* g_lik, const float* g_zhat,
                                 float* g_z, float* g_params, int B, int C, int H, int W, float lik_bound, void* stream);
/* Gradient of the [C][58] parameter table (torch.cat of the 14 per-channel tensors of an EntropyBottleneck, reference
 * entropy_models.py:289-300) split into the parameters' gradients in one launch: part t (width widths[t]) becomes the contiguous
 * block [C][widths[t]] at flat + C * (widths[0] + ... + widths[t-1]). */
int masic_eb_table_split(const float* g_table, float* flat, int C, const int* widths, int nparts, void* stream);
int masic_entropy_bottleneck_auxloss_bwd(const float* params, const float* quantiles, float* g_quantiles,
                                         int C, double tail_mass, float gout, void* stream);
/* The auxiliary step's loss and gradient (newtrain_codec_real.py:143-145: sum of EntropyBottleneck.loss() over the model's bottlenecks,
 * backward) for n <= 4 bottlenecks in two launches.  params[e]: [C[e]][58] tables, quantiles[e] / g_quantiles[e]: [C[e]][3];
 * loss[0] = total, loss[1 + e] = bottleneck e; workspace: >= n * 3 * max C floats. */
int masic_entropy_bottleneck_aux_step(const float* const* params, const float* const* quantiles, float* const* g_quantiles,
                                      const int* C, const double* tail_mass, int n, float* loss, float* workspace, int workspace_floats,
                                      void* stream);
/* warp backward w.r.t. the source image; g_src [B,C,Hs,Ws] must be ZERO-FILLED by the caller (float atomics) */
int masic_warp_perspective_bwd(const float* g_dst, const float* minv_norm, float* g_src,
                               int B, int C, int Hs, int Ws, int Hd, int Wd, void* stream);
/* The same adjoint (autograd's backward of kornia.warp_perspective -> F.grid_sample, reference MASIC.py:781, :1461-1480) as a gather over
 * source pixels: no atomics, a fixed summation order, g_src written once and needing NO zero fill.  flag: one int32 on the device, zero
 * on entry; set to 1 when some source pixel's footprint was outside the gather form's bounds (horizon inside the picture, magnification
 * beyond ~2 x) -- the call has then redone the whole tensor with the scatter form above (device-side test of the flag, no host round trip). */
int masic_warp_perspective_bwd_gather(const float* g_dst, const float* minv_norm, float* g_src, int* flag,
                                      int B, int C, int Hs, int Ws, int Hd, int Wd, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host-side entropy coding (no device work).  Bit-exact replacements of the reference's pybind11
 * extensions: compressai/cpp_exts/ops/ops.cpp:41-106 (pmf_to_quantized_cdf) and compressai/cpp_exts/rans/
 * rans_interface.cpp:108-283 (RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes: rANS with a 64-bit
 * state, 32-bit renormalisation, 16-bit probabilities, 4-bit bypass digits for symbols outside a table).
 *   cdfs: ncdfs rows of cdf_stride int32; row i holds cdf_sizes[i] entries 0 = c[0] < ... < c[size-1] = 65536; its last
 *   symbol (size-2) is the escape.  A coded value is symbols[k] - offsets[indexes[k]].
 *   masic_rans_encode_bound(n) bytes always suffice for n symbols; *out_len receives the stream length. */
int masic_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf /* n + 1 */);
size_t masic_rans_encode_bound(int nsymbols);
int masic_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int n, const int32_t* cdfs, int cdf_stride,
                                   const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, uint8_t* out, size_t out_cap,
                                   size_t* out_len);
int masic_rans_decode_with_indexes(const uint8_t* in, size_t in_len, const int32_t* indexes, int n, const int32_t* cdfs,
                                   int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, int32_t* symbols);

/* ---- upstream of the path, SURVEY.md 8(f)-3: h_matrix of a stereo pair from the UDH network's corner offsets --
 * udh/udh/model.py:100-111 (kornia.get_perspective_transform(corners, corners + delta), torch.inverse) + h_adjust
 * (newtrain_codec_real.py:49-59, scale_a = H / patch, scale_b = W / patch).  corners, delta: [B][4][2]; h_out: [B][3][3],
 * NaN for a degenerate quadrilateral. */
int masic_homography_from_corners(const float* corners, const float* delta, float* h_out, int B, float scale_a, float scale_b,
                                  void* stream);

/* ---- host data path, SURVEY.md 8(f)-4: one view of a dataset item (compressai/datasets/utils.py:207-285) from the decoded
 * uint8 RGB picture [H][W][3] on the device: pic_chw (nullable) = the crop [start_h : +ph, start_w : +pw] as float32
 * [3][ph][pw] / 255 (ToTensor); homo_patch (nullable) = the [homopatch][homopatch] window at (patch_x, patch_y) of
 * mean_c(Normalize(ToTensor(cv2.resize(crop, (homopic, homopic))))) -- uint8 INTER_LINEAR resize with OpenCV's fixed-point
 * arithmetic (2x decimation = rounded 2x2 box mean), scalar MEAN / STD of utils.py:26-27. */
int masic_pair_prep(const uint8_t* img_hwc, int H, int W, int start_h, int start_w, int ph, int pw, float* pic_chw,
                    int homopic, int patch_x, int patch_y, int homopatch, float* homo_patch, void* stream);

/* ---- HSIC.compress / decompress (MASIC.py:855-1408), SURVEY.md 8(f)-1: the y1 / y2 streams.
 * Per-symbol coding tables (MASIC.py:986-1044 in compress, :1262-1296 in decompress): the K-component Gaussian-mixture
 * PMF of latent element (pixel pix[i], channel chan[j]) over the alphabet 0 .. 2*minmax, clipped to [2^-16, 1],
 * renormalised to 2^16 and rounded (:1040-1043), the count total then forced to exactly 2^16 at the mode.  sigma / mu /
 * logits: [K*M][HW] head outputs of ONE image (the reference codes batch element 0), logits before the softmax over K.
 * Row r = i * nch + j; a negative pix[i] skips its rows (padding of a fixed-size list).  starts (nullable): [npix*nch][2*minmax+1] u16 interval starts -- the decoder's view;
 * y_hat + start_freq (nullable, together): the integer-valued latent [M][HW] and [npix*nch][2] (start, freq) of its
 * symbols -- the encoder's view.  err_flag (device int, caller-zeroed): bit 0 = a table could not be normalised,
 * bit 1 = a symbol outside the alphabet. */
int masic_gmm_cdf_rows(const float* sigma, const float* mu, const float* logits, int M, int K, int HW,
                       const int32_t* pix, int npix, const int32_t* chan, int nch, int minmax, float scale_bound,
                       const float* y_hat, uint16_t* starts, int32_t* start_freq, int32_t* err_flag, void* stream);
/* rANS with one table per symbol (host).  The reference drives the third-party `range_coder` package here
 * (MASIC.py:958, :1044, :1221, :1296), which is not part of its tree: the byte stream of this container is this
 * library's own (64-bit state, 32-bit renormalisation, 16-bit tables: the coder of the z streams above).
 * start_freq: [n][2] intervals in coding order.  The decoder is incremental -- the tables of a wavefront of latent pixels
 * exist only once the previous wavefront is decoded: open, decode_rows per wavefront (symbols[r] = index of the interval
 * of row r that holds the coder's value), close. */
int masic_rans_encode_freqs(const int32_t* start_freq, size_t n, uint8_t* out, size_t out_cap, size_t* out_len);
/* One rANS stream per channel -- the form the device decoder below reads (container "MSR2", masic_amd/codec.py).  start_freq:
 * [npix * nch][2], row i * nch + c = pixel i (coding order), channel c.  out: the nch streams back to back; lengths[c] bytes each
 * (multiples of 4, >= 8).  Replaces the reference's one `range_coder` stream per view (MASIC.py:1046-1140), whose bytes are parity-unpinned. */
int masic_rans_encode_channels(const int32_t* start_freq, int npix, int nch, uint8_t* out, size_t out_cap, uint32_t* lengths, size_t* out_len);
/* Device side of the decoder's loop over coding steps (reference MASIC.py:1262-1408 runs it per symbol on the host):
 * masic_gmm_cdf_rows_at = masic_gmm_cdf_rows (decoder's view) for the pixel list pix_all[*step][npix] (step: device int);
 * masic_rans_decode_step decodes that step's symbols -- one wavefront per channel stream -- writes them into the latent
 * y_hat [M][HW] (value = symbol - minmax at [chan[c]][pix]) and increments *step.  words / word_off / word_cnt: the channel streams
 * as 32-bit little-endian words, first word and word count per channel; state / pos: per-channel coder state and next word index,
 * caller-initialised (state = word0 | word1 << 32, pos = 2) and kept across steps; done: caller-zeroed device int; y_f16k (nullable):
 * the latent also as F16K bf16 [ceil16(M) / 16][HW][16].  err_flag bit 2: a stream ended early.  All pointers device memory; asynchronous on `stream`; capturable into a HIP graph. */
/* The entropy-parameter layers on a short pixel list (masic_amd/csrc/skinny.hip): up to three 1x1 layers (groups as in
 * masic_gemm_f16k_group_fwd, bf16 operands, w_packed from masic_gemm_f16k_pack_weight) or, with ctx != 0, ONE 5x5 type-A masked
 * convolution (w_packed from masic_skinny_ctx_pack_weight; latent h x w) evaluated at the pixels pix[(*step) * list_stride + i], i < npix
 * (step: device int or NULL), of one image.  x and the outputs are FULL-SIZE buffers (h * w pixels): only the listed pixels are
 * written.  gate (nullable): float32 [.][h * w], plane gate_c multiplies the result after the activation.  A pixel's result does not
 * depend on which other pixels are in the list: the encoder (all pixels, one launch) and the decoder (one coding wavefront per
 * launch) get bit-identical parameters.  Replaces the per-symbol crop + Conv2d calls of MASIC.py:986-1003 / :1262-1280. */
size_t masic_skinny_ctx_packed_bytes(int Cin, int Cout);
int masic_skinny_ctx_pack_weight(const float* w, void* w_packed, int Cin, int Cout, void* stream);
int masic_skinny_group_fwd(const masic_gemm_group_t* groups, int ngroups, int ctx, const int32_t* pix, const int32_t* step, int list_stride,
                           int npix, int h, int w, const float* gate, int gate_c, void* stream);
int masic_gmm_cdf_rows_at(const float* sigma, const float* mu, const float* logits, int M, int K, int HW,
                          const int32_t* pix_all, const int32_t* step, int npix, const int32_t* chan, int nch, int minmax, float scale_bound,
                          uint16_t* starts, int32_t* err_flag, void* stream);
int masic_rans_decode_step(const uint32_t* words, const uint32_t* word_off, const uint32_t* word_cnt, uint64_t* state, uint32_t* pos,
                           const uint16_t* starts, const int32_t* pix_all, int32_t* step, int npix, const int32_t* chan, int nch,
                           int L, int minmax, float* y_hat, void* y_f16k, int HW, int32_t* err_flag, int32_t* done, void* stream);
int masic_rans_decoder_open(const uint8_t* in, size_t in_len, void** handle);
int masic_rans_decoder_decode_rows(void* handle, const uint16_t* starts, int nrows, int L, int32_t* symbols);
/* the next n symbols with tables picked by indexes (arguments as masic_rans_decode_with_indexes), keeping the coder state between
 * calls: reference RansDecoder::set_stream / decode_stream (rans_interface.cpp:286-353) */
int masic_rans_decoder_decode_indexes(void* handle, const int32_t* indexes, int n, const int32_t* cdfs, int cdf_stride,
                                      const int32_t* cdf_sizes, const int32_t* offsets, int ncdfs, int32_t* symbols);
void masic_rans_decoder_close(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* MASIC_HIP_H */
