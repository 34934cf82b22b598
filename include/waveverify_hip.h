/*
 * waveverify_hip.h — C ABI of libwaveverify_hip.so: the MI355X (gfx950) embed/detect hot path
 * of WaveVerify.  Plain pointers and sizes only; no torch / HIP types in the signatures
 * (`stream` is a hipStream_t passed as void*, NULL = the default stream).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *   wv_generator_forward  Generator.forward            model/generator.py:360-423
 *                         (+ the `wm = delta + x` of AudioWatermarking._forward_audio_sample,
 *                          model/watermarking.py:423-441, when add_input != 0, and the message
 *                          batch broadcast of watermarking.py:320-329 through msg_rows)
 *   wv_detector_forward   Detector.forward             model/detector.py:366-391
 *                         (+ sigmoid/mean-over-time of waveverify/core.py:577-580 when
 *                          mean_prob != NULL, so the [B,nbits,T] logits need not be stored)
 *   wv_locator_forward    Locator.forward              model/locator.py:268-299
 *   wv_model_set_param*   nn.Module.load_state_dict on the stripped / parametrized key layouts
 *                         waveverify/core.py:324-426, scripts/train.py:1589-1676
 *   wv_op_*               the fused units of modules/seanet.py + modules/conv.py, exported one by
 *                         one so that parity tests can pin each kernel separately.
 *
 * Tensors are float32, contiguous, [B, C, T] with time innermost, in DEVICE memory unless the
 * parameter is documented as host memory.  The caller owns every buffer; the library owns only
 * the packed weights inside a wv_model.  Every function returns 0 on success or a negative
 * WV_E* code; wv_last_error() gives the message (thread-local).
 */
#ifndef WAVEVERIFY_HIP_H
#define WAVEVERIFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WV_OK 0
#define WV_EINVAL (-1)   /* bad argument / shape */
#define WV_ENOKEY (-2)   /* unknown or missing parameter name */
#define WV_EHIP (-3)     /* HIP runtime error */
#define WV_ESTATE (-4)   /* call order (e.g. forward before finalize) */
#define WV_ENOMEM (-5)   /* workspace too small */

#define WV_KIND_GENERATOR 0
#define WV_KIND_DETECTOR 1
#define WV_KIND_LOCATOR 2

#define WV_MAX_STRIDES 8

/* Hyper-parameters; field names follow the reference constructors
 * (model/generator.py:63-104, model/detector.py:82-114, model/locator.py:84-115). */
typedef struct wv_config {
    int32_t kind;
    int32_t dimension;
    int32_t msg_dimension;
    int32_t channels_enc;
    int32_t channels_dec;
    int32_t n_fft_base;
    int32_t n_residual_enc;
    int32_t n_residual_dec;
    int32_t n_strides;
    int32_t strides[WV_MAX_STRIDES];   /* as given to the constructor, e.g. {8,5,4,2} */
    int32_t kernel_size;
    int32_t last_kernel_size;
    int32_t residual_kernel_size;
    int32_t dilation_base;
    int32_t zero_init;                 /* 1: res_scale_param / scale_param tensors exist */
    int32_t nbits;
    int32_t output_dim;
    int32_t embedding_dim;
    int32_t embedding_layers;
    int32_t freq_bands;
    float res_scale_enc;
    float res_scale_dec;
    float wav_std;
    float spec_means[WV_MAX_STRIDES + 1];
    float spec_stds[WV_MAX_STRIDES + 1];
} wv_config;

typedef struct wv_model wv_model;

const char* wv_last_error(void);
const char* wv_version(void);

/* Fill cfg with the reference defaults for `kind`. */
int wv_config_default(int kind, wv_config* cfg);

/* ---- model lifetime --------------------------------------------------------------------- */
int wv_model_create(const wv_config* cfg, wv_model** out);
void wv_model_destroy(wv_model* m);

/* Parameter table: the state-dict keys (stripped layout) this model expects. */
int wv_model_num_params(const wv_model* m);
/* name_out: caller buffer of name_cap bytes; shape_out: 4 int64 (unused dims = 1). */
int wv_model_param_info(const wv_model* m, int index, char* name_out, int name_cap,
                        int64_t* shape_out, int* ndim_out, int* is_weight_normed);

/* Hand over one tensor in the reference's layout (HOST memory, float32, numel checked). */
int wv_model_set_param(wv_model* m, const char* name, const float* host_data, int64_t numel);
/* Same for a weight-normed tensor given as the (g, v) pair of
 * `...parametrizations.weight.original0/1`; folded as w = g * v / ||v|| (modules/conv.py:73-74). */
int wv_model_set_param_wn(wv_model* m, const char* name, const float* host_g, int64_t g_numel,
                          const float* host_v, int64_t v_numel);
/* Optional override of a CausalSTFT basis buffer `...spec.weight` [2F,1,n_fft] (checkpoints
 * carry it, modules/conv.py:1026); by default the basis is generated as conv.py:1003-1020 does. */
int wv_model_set_stft_basis(wv_model* m, const char* name, const float* host_data, int64_t numel);
/* Pack (transpose / interleave / compose heads) and upload. Fails if a parameter is missing. */
int wv_model_finalize(wv_model* m);

/* ---- forward passes --------------------------------------------------------------------- */
/* Device scratch needed for a [B,1,T] batch (bytes). */
size_t wv_workspace_bytes(const wv_model* m, int B, int T);

/* x [B,1,T]; msg [msg_rows, msg_dimension] float (0/1), msg_rows == B or 1 (broadcast);
 * out [B,1,T] = delta, or delta + x when add_input != 0. */
int wv_generator_forward(wv_model* m, const float* x, const float* msg, int msg_rows,
                         float* out, int add_input, int B, int T,
                         void* workspace, size_t workspace_bytes, void* stream);

/* x [B,1,T]; logits [B,nbits,T] or NULL; mean_prob [B,nbits] or NULL (mean_t sigmoid(logit)). */
int wv_detector_forward(wv_model* m, const float* x, float* logits, float* mean_prob,
                        int B, int T, void* workspace, size_t workspace_bytes, void* stream);

/* x [B,1,T]; logits [B,1,T]. */
int wv_locator_forward(wv_model* m, const float* x, float* logits, int B, int T,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Encoder only (SEANetEncoder.forward, modules/seanet.py:883-976): latent [B,dimension,ceil(T/hop)].
 * msg may be NULL (no FiLM), as for the detector / locator. */
int wv_encoder_forward(wv_model* m, const float* x, const float* msg, int msg_rows, float* latent,
                       int B, int T, void* workspace, size_t workspace_bytes, void* stream);

/* ---- single fused units ----------------------------------------------------------------
 * Activations (X, resid, film, Y, wav, P, H, x, Z, logits, mean_prob) are DEVICE pointers.
 * Weights / biases (w_*, *_bias, bias, basis) are HOST pointers in the reference's own layouts;
 * they are packed and uploaded per call and the call is synchronous: these entry points exist so
 * that every kernel can be parity-tested on its own, not for serving. */

/* Y = epilogue( DWconv_k,s,d( W1x1 @ act(pre_scale * X) ) + dw_bias )
 *   X [B,K,Tin], w_pw [M,K] (1x1, no bias), w_dw [M,ks], dw_bias [M] or NULL
 *   causal left pad (ks-1)*d-(s-1), zero right pad to complete the last frame
 *   (SConv1d, modules/conv.py:715-763); Tout = ceil(Tin/s).
 *   pre_elu: 1 -> act = ELU, 0 -> identity.
 *   film [B,bands,2] (gamma,beta) or NULL: y = y*gamma+beta per band (seanet.py:928-966);
 *   resid [B,M,Tout] or NULL: y = y*out_scale + resid (seanet.py:272-277).
 * This is the ResnetBlock half (seanet.py:39-116) and the Downsample+FiLM unit (seanet.py:733-772).
 * Yact [B,M,Tout] or NULL: second output ELU(act_scale * y) -- the NEXT unit's prologue hoisted into
 *   this epilogue, so that the consumer can stage its operand by LDS-DMA (a pure copy, pre_elu = 0).
 *   Y may be NULL when only Yact is wanted. */
int wv_op_pw_dw(const float* X, const float* w_pw, const float* w_dw, const float* dw_bias,
                const float* film, const float* resid, float* Y,
                int B, int K, int M, int Tin, int ks, int stride, int dilation,
                float pre_scale, int pre_elu, float out_scale, int bands,
                float* Yact, float act_scale, void* stream);

/* Whole SEANetResnetBlock in one launch, raw in / raw out (narrow layers, C in {64, 96, 128, 192}, k = 5,
 * dilation 1, T % 4 == 0; modules/seanet.py:245-281 with dws_conv_block :39-116):
 *   y = X + out_scale * ( DW5( W2 @ ELU( DW5( W1 @ ELU(pre_scale * X) ) + b1 ) ) + b2 )
 * X [B,C,T] is read once from HBM (activated inside, re-read from L2 as the residual operand); the intermediate
 * stays in LDS.  w_pw1/w_pw2 [C,C]; w_dw1/w_dw2 [C,5]; b1/b2 [C]; Y and/or Yact = ELU(act_scale*y) [B,C,T].
 * Bit-identical to the block run as two wv_op_pw_dw units.
 * Returns WV_EINVAL for shapes the fused kernel does not cover (the nets then run two wv_op_pw_dw units). */
int wv_op_resblock(const float* X, float pre_scale, const float* w_pw1, const float* w_dw1, const float* b1,
                   const float* w_pw2, const float* w_dw2, const float* b2, float* Y, float* Yact,
                   int B, int C, int T, float out_scale, float act_scale, void* stream);

/* Y = W1x1 @ producer(X) + bias, then optional L2-normalise over channels * sqrt(M).
 *   mode 0: producer = act(pre_scale*X)                                  (plain 1x1)
 *   mode 1: producer = causal DW conv k (no bias) of act(pre_scale*X)    (conv_post, seanet.py:797-823)
 *   mode 2: producer = causal DW ConvTranspose k=2r,s=r of act(pre_scale*X), right-trimmed by r
 *           (upsample, seanet.py:1112-1138; SConvTranspose1d conv.py:838-881); Tout = Tin*r
 *   accumulate != 0: Y += out_scale * (W @ producer(X))  (SpecBlock add, seanet.py:500-505).
 * Routing mirrors the model's: mode 2 and the accumulate form with M >= 128 run on the pw_dw kernel
 * (ConvTranspose producer in its operand loader / identity stencil with Y as the residual operand).
 * Yact / act_scale: as for wv_op_pw_dw (mode 2 and the accumulate form with M >= 128 only, else NULL). */
int wv_op_dw_pw(const float* X, const float* w_dw, const float* w_pw, const float* bias, float* Y,
                int B, int K, int M, int Tin, int mode, int ks_or_ratio,
                float pre_scale, int pre_elu, int l2norm, int accumulate, float out_scale,
                float* Yact, float act_scale, void* stream);

/* P[b,f,t] = (log(max(|STFT|,1e-5)) - mean)/std, CausalSTFT magnitude with eps 1e-12
 * (modules/conv.py:1036-1080, seanet.py:479-494). wav [B,1,T]; basis [2F,n_fft] host or NULL
 * (NULL: generated); P [B,F,ceil(T/hop)]. */
int wv_op_stft_logmag(const float* wav, const float* host_basis, float* P, int B, int T,
                      int n_fft, int hop, float mean, float std, void* stream);

/* Whole SpecBlock in one launch (modules/seanet.py:463-511 with CausalSTFT modules/conv.py:1036-1086):
 *   y = x + out_scale * ( W @ P ),  P = (log(max(|STFT(wav)|, 1e-5)) - mean) / std  -- the spectrogram stays in LDS, never in HBM.
 * For the scales whose whole spectrum is one tile and whose 1x1 has as many rows: n_fft = M in {64, 128}, more than 64 frames,
 * Tf = ceil(T/hop) a multiple of 4.  wav [B,1,T]; w_pw [M, n_fft/2+1] (HOST pointer, like every wv_op_* weight); x [B,M,Tf];
 * Y (may alias x) and/or Yact = ELU(act_scale * y).  Bit-identical to wv_op_stft_logmag followed by the accumulate form of wv_op_dw_pw.
 * Returns WV_EINVAL for shapes the fused kernel does not cover (the nets then run the two kernels). */
int wv_op_spec_block(const float* wav, const float* basis_or_null, const float* w_pw, const float* x, float* Y, float* Yact, int B, int T,
                     int n_fft, int hop, int M, float mean, float std, float out_scale, float act_scale, void* stream);

/* ---- f16-operand / f32-accumulate mode (BASELINE.json configs[4] "MFMA linears fp16"; csrc/wv_h16.hip): a throughput mode of the
 * detector next to the exact-f32 path.  Activations are f16 in the "c8" layout [B][roundup(C,16)/8][T][8] (channel groups of eight,
 * time-major inside a group = the B operand of v_mfma_f32_32x32x16_f16 as it lies in memory); accumulation, stencils, ELU, bias and
 * residual adds are f32.  Weights: HOST f32 pointers in the reference's layouts, as for every wv_op_*.
 *   wv_h16_from_f32 / wv_h16_to_f32   [B,C,T] f32 <-> c8 f16 (from: optionally ELU(scale * x) on the way; rows past C are zero)
 *   wv_h16_conv_pre   conv_pre (seanet.py:657-664) writing c8 f16
 *   wv_h16_resblock   whole SEANetResnetBlock (seanet.py:245-281), C in {32,64,96,128,192,256,384,512,768}, k = 5, dilation 1, any T
 *   wv_h16_conv       y = out_scale * (bias + Conv1d(x)) + resid with W[m][i][k] = w_dw[m][i] * w_pw[m][k] (w_dw NULL: 1), causal,
 *                     x = 0 outside [0,Tin), Tout = ceil(Tin/stride): the downsample unit (ks = 2r, stride r, pad r; seanet.py:739-760)
 *                     and the SpecBlock's 1x1 + add (ks = 1; seanet.py:500-502).  Outputs: Y16 / Yact16 = ELU(act_scale*y) in c8 f16,
 *                     Yf32 [B,M,Tout] f32 row-major (any of them NULL).
 *   wv_h16_spec_block whole SpecBlock in one launch (seanet.py:463-511): the STFT on the f16 pipe with the waveform split in two f16 terms,
 *                     log-magnitude, the 1x1 and the add; the spectrogram stays in LDS.  (n_fft = M, hop) in {(64,1),(128,2),(256,8),(512,40)}
 *                     (the default detector's scales), else WV_EINVAL; x16 / Y16 / Yact16 c8 f16 [B, M/8, ceil(T/hop), 8]
 *   wv_detector_forward_f16   Detector.forward (model/detector.py:366-391) in this mode: conv_pre, the ResnetBlocks, the SpecBlocks
 *                     (STFT as a split-f16 matrix product) incl. spec_post and the downsample units on the f16 pipe; with logits == NULL
 *                     (mean probabilities only) conv_post and the head as well, otherwise those two by the exact path's f32 kernels.
 *                     Same arguments and workspace as wv_detector_forward; WV_ESTATE for a model without an f16 plan.
 *   wv_locator_forward_f16    Locator.forward (model/locator.py:268-299) in this mode: the encoder stages on the f16 pipe (32- and
 *                     64-channel ResnetBlocks, composed downsample convs; its SpecBlocks, whose n_fft != C, by the exact path's STFT
 *                     kernel + the 1x1 on the f16 pipe), spec_post / conv_post / head by the exact f32 kernels.  Arguments of wv_locator_forward.
 *   wv_generator_forward_f16  Generator.forward (model/generator.py:360-423; modules/seanet.py:883-976, 1067-1226) in this mode: the
 *                     encoder as above with FiLM in the downsample convs' epilogues, conv_post as one composed conv, L2Norm, then
 *                     the decoder: first conv pair as one composed conv, every upsample unit (ELU -> depth-wise ConvTranspose1d ->
 *                     1x1, seanet.py:1147-1170) as ONE two-tap conv over (phase, channel) rows, ResnetBlocks of 768 / 384 / 192 / 96
 *                     channels in one launch each, the tail (f32 sums, tanh, + x) on the c8 stream.  Message MLP / FiLM scalars in f32.
 *                     Arguments and workspace of wv_generator_forward; WV_ESTATE for a model without an f16 plan.
 *   wv_h16_upsample   the decoder's upsample unit as that conv: x16 = the PRE-ACTIVATED input c8 [B, K/8, Tin, 8]; w_ct [K,1,2r], w_pw
 *                     [M,K,1], bias [M] HOST; Y16 / Yact16 c8 [B, M/8, Tin*r, 8] (either may be NULL)
 *   wv_h16_tail       decoder tail on the pre-activated c8 stream: out[B,1,T] = tanh(out_scale * (b + Conv1d(C -> 1, ks)(a16))) (+ x)
 *   wv_h16_l2norm     L2Norm over channels of lat [B,D,Fr] f32 (seanet.py:288-318) -> c8 f16
 *   wv_h16_conv_film  wv_h16_conv with FiLM behind it (film [B, bands, 2] DEVICE: gamma, beta per clip and band of M / bands rows) */
int wv_h16_round_host(const float* in, uint16_t* out, int64_t n);   /* HOST pointers: the weight packers' f32 -> f16 rounding (nearest even) */
int wv_h16_from_f32(const float* X, void* Y16, int B, int C, int T, float scale, int elu, void* stream);
int wv_h16_to_f32(const void* X16, float* Y, int B, int C, int T, void* stream);
int wv_h16_conv_pre(const float* x, const float* w, const float* bias, void* Y16, int B, int C, int T, int ks, float in_scale, void* stream);
int wv_h16_resblock(const void* X16, float pre_scale, const float* w_pw1, const float* w_dw1, const float* b1, const float* w_pw2, const float* w_dw2,
                    const float* b2, void* Y16, void* Yact16, int B, int C, int T, float out_scale, float act_scale, void* stream);
int wv_h16_conv(const void* X16, const float* w_pw, const float* w_dw, const float* bias, const void* resid16, void* Y16, void* Yact16, float* Yf32,
                int B, int K, int M, int Tin, int ks, int stride, int pad, float out_scale, float act_scale, void* stream);
int wv_h16_spec_block(const float* wav, const float* basis_or_null, const float* w_pw, const void* x16, void* Y16, void* Yact16, int B, int T,
                      int n_fft, int hop, int M, float mean, float std, float out_scale, float act_scale, void* stream);
int wv_detector_forward_f16(wv_model* m, const float* x, float* logits, float* mean_prob,
                            int B, int T, void* workspace, size_t workspace_bytes, void* stream);
int wv_locator_forward_f16(wv_model* m, const float* x, float* logits, int B, int T,
                           void* workspace, size_t workspace_bytes, void* stream);
int wv_generator_forward_f16(wv_model* m, const float* x, const float* msg, int msg_rows,
                             float* out, int add_input, int B, int T,
                             void* workspace, size_t workspace_bytes, void* stream);
int wv_h16_upsample(const void* X16, const float* w_ct, const float* w_pw, const float* bias, void* Y16, void* Yact16,
                    int B, int K, int M, int Tin, int ratio, float act_scale, void* stream);
int wv_h16_tail(const void* A16, const float* w, const float* bias, const float* x, float* out, int B, int C, int Tin, int T, int ks,
                float out_scale, void* stream);
int wv_h16_l2norm(const float* lat, void* Y16, int B, int D, int Fr, void* stream);
int wv_h16_conv_film(const void* X16, const float* w_pw, const float* w_dw, const float* bias, const float* film, int bands, void* Y16, void* Yact16,
                     int B, int K, int M, int Tin, int ks, int stride, int pad, float act_scale, void* stream);

/* conv_pre: Y = Conv1d(1->C,k)(x * in_scale) + bias  (seanet.py:657-664). x [B,1,T], w [C,1,k]. */

/* The same op with the basis packed and uploaded ONCE (a training step computes these features for every scale at every step):
 * basis_or_null as in wv_op_stft_logmag (NULL = the reference's windowed DFT basis). */
typedef struct wv_stft_plan wv_stft_plan;
int wv_stft_plan_create(int n_fft, const float* basis_or_null, wv_stft_plan** out);
void wv_stft_plan_destroy(wv_stft_plan* p);
int wv_stft_plan_logmag(const wv_stft_plan* p, const float* wav, float* P, int B, int T, int hop, float mean, float std, void* stream);
/* backward of the features towards the audio: dwav[B,1,T] (+)= d/dwav of <dP, P(wav)> (training the generator through the detector's and
 * locator's spectrogram branches).  The clamps of the reference (conv.py:1078, seanet.py:484) pass no gradient where |STFT|^2 <= 1e-10. */
size_t wv_stft_plan_backward_workspace_bytes(const wv_stft_plan* p, int B, int T, int hop);
int wv_stft_plan_backward(const wv_stft_plan* p, const float* wav, const float* dP, float* dwav, int accumulate, int B, int T, int hop, float std,
                          void* workspace, size_t workspace_bytes, void* stream);
int wv_op_conv_pre(const float* x, const float* w, const float* bias, float* Y, int B, int C,
                   int T, int ks, float in_scale, void* stream);

/* decoder tail: out = tanh(out_scale*(Conv1d(C->1,k)(ELU(pre_scale*H)) + bias)) (+ x)
 * (seanet.py:1177-1202, generator.py:410, watermarking.py:440). H [B,C,Tin>=T], w [1,C,k]. */
int wv_op_tail(const float* H, const float* w, const float* bias, const float* x_or_null,
               float* out, int B, int C, int Tin, int T, int ks, float pre_scale, float out_scale,
               void* stream);

/* detector / locator head: ConvTranspose1d(D->O,k=s=hop)+bias -> trim to T -> Conv1d(O->nb,1)+bias
 * (detector.py:300-310). Z [B,D,Fr]; w_rev [D,O,hop]; w_last [nb,O]; either output may be NULL. */
int wv_op_head(const float* Z, const float* w_rev, const float* b_rev,
               const float* w_last, const float* b_last, float* logits, float* mean_prob,
               int B, int D, int O, int nb, int hop, int Fr, int T, void* stream);

/* message MLP + all FiLM gammas/betas (seanet.py:831-846,905-912): msg [rows,msg_dim] ->
 * film [B,n_scales,bands,2]; uses the model's parameters. */
int wv_model_film(wv_model* m, const float* msg, int msg_rows, float* film, int B, void* stream);

/* ---- first training-step slice ------------------------------------------------------
 * Forward and backward of one SEANetResnetBlock half with LIVE weight normalisation
 * (modules/seanet.py:39-116 dws_conv_block; modules/conv.py:47-88 weight norm recomputed every step;
 *  scripts/train.py:1421-1480):
 *     y = DW5( (g_pw v_pw/||v_pw||) @ ELU(pre_scale * x) ; g_dw v_dw/||v_dw|| ) + bias
 * ALL pointers are DEVICE pointers (parameters live on the GPU while training): x, y, dy, dx [B,C,T];
 * g_pw [C], v_pw [C,C] (the 1x1), g_dw [C], v_dw [C,5] (the depth-wise conv), bias [C]; gradients have the
 * shapes of what they differentiate.  Any C and T: shapes off the LDS-DMA core's grid (C <= 32, T % 4 != 0) run on the round-1 core.
 * The weight-norm fold runs on the device in every call (wv_train_half_forward and _backward both fold). */
typedef struct wv_train_unit wv_train_unit;
typedef wv_train_unit wv_train_half;
int wv_train_half_create(int C, wv_train_half** out);
void wv_train_half_destroy(wv_train_half* h);
size_t wv_train_half_workspace_bytes(const wv_train_half* h, int B, int T);   /* backward only */
int wv_train_half_forward(wv_train_half* h, const float* x, const float* g_pw, const float* v_pw,
                          const float* g_dw, const float* v_dw, const float* bias, float pre_scale,
                          float* y, int B, int T, void* stream);
int wv_train_half_backward(wv_train_half* h, const float* x, const float* g_pw, const float* v_pw,
                           const float* g_dw, const float* v_dw, float pre_scale, const float* dy,
                           float* dx, float* dg_pw, float* dv_pw, float* dg_dw, float* dv_dw, float* db,
                           int B, int T, void* workspace, size_t workspace_bytes, void* stream);

/* The general fused unit of the encoder/decoder trunk, forward and backward:
 *     y[B,M,Tout] = DW_{ks,stride}( (g_pw v_pw/||v_pw||) @ act(pre_scale * x[B,K,Tin]) ; g_dw v_dw/||v_dw|| ) + bias
 * act = ELU (pre_elu = 1) or identity; causal SConv1d geometry (conv.py:715-763): left pad ks - stride, Tout = ceil(Tin/stride).
 * K = M, ks = 5, stride = 1 is the ResnetBlock half above; M = 2K, ks = 2r, stride = r is the encoder's Downsample unit
 * (seanet.py:733-772).  v_pw [M,K], v_dw [M,ks]; ks <= 16.  dx may be NULL (first layer: no input gradient). */
int wv_train_unit_create(int K, int M, int ks, int stride, wv_train_unit** out);
void wv_train_unit_destroy(wv_train_unit* u);
size_t wv_train_unit_workspace_bytes(const wv_train_unit* u, int B, int Tin);   /* backward only */
int wv_train_unit_forward(wv_train_unit* u, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                          const float* v_dw, const float* bias, float pre_scale, int pre_elu, float* y, int B, int Tin, void* stream);
int wv_train_unit_backward(wv_train_unit* u, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                           const float* v_dw, float pre_scale, int pre_elu, const float* dy, float* dx, float* dg_pw, float* dv_pw,
                           float* dg_dw, float* dv_dw, float* db, int B, int Tin, void* workspace, size_t workspace_bytes, void* stream);

/* Whole SEANetResnetBlock with live weight norm (modules/seanet.py:245-281, identity shortcut):
 *     y = x + s * half2(half1(pre_scale * x)),   s = res_scale * res_scale_param[0]  (res_scale_param may be NULL: s = res_scale)
 * forward keeps the two intermediate activations and the two 1x1 outputs in `saved` (wv_train_block_saved_bytes: four activation-sized
 * tensors; the forward kernels store the 1x1 outputs themselves, backward then has no GEMM to recompute) for backward, which returns
 * dx, both halves' parameter gradients and d(res_scale_param).  Same shape limits as the half.
 * CONTRACT: wv_train_block_backward reuses the weight folds (W, its packed copies, 1/||v||) that the forward left in the handle, so it
 * must follow the wv_train_block_forward that produced `saved` with the SAME parameter tensors, and nothing may change their values or
 * run another forward on this handle in between (optimizer steps come after backward).  Other parameter pointers fail with WV_ESTATE. */
typedef struct wv_train_block wv_train_block;
typedef struct { const float *g_pw, *v_pw, *g_dw, *v_dw, *bias; } wv_half_params;     /* device pointers */
typedef struct { float *dg_pw, *dv_pw, *dg_dw, *dv_dw, *db; } wv_half_grads;          /* device pointers */
int wv_train_block_create(int C, wv_train_block** out);
void wv_train_block_destroy(wv_train_block* b);
size_t wv_train_block_saved_bytes(const wv_train_block* b, int B, int T);
size_t wv_train_block_workspace_bytes(const wv_train_block* b, int B, int T);       /* backward only */
int wv_train_block_forward(wv_train_block* b, const float* x, const wv_half_params* p /*[2]*/, const float* res_scale_param,
                           float pre_scale, float res_scale, float* y, void* saved, size_t saved_bytes, int B, int T, void* stream);
int wv_train_block_backward(wv_train_block* b, const float* x, const wv_half_params* p /*[2]*/, const float* res_scale_param,
                            float pre_scale, float res_scale, const float* dy, const void* saved, float* dx,
                            const wv_half_grads* g /*[2]*/, float* d_res_scale_param, int B, int T,
                            void* workspace, size_t workspace_bytes, void* stream);

/* conv_pre with live weight norm (seanet.py:657-664): y[B,C,T] = causal conv1d(in_scale * x[B,1,T], g v/||v|| [C,1,ks]) + bias.
 * backward: dg [C], dv [C,ks], db [C] and, optionally, dx [B,1,T] (the gradient towards the audio; NULL to skip). */
typedef struct wv_train_convpre wv_train_convpre;
int wv_train_convpre_create(int C, int ks, wv_train_convpre** out);
void wv_train_convpre_destroy(wv_train_convpre* h);
size_t wv_train_convpre_workspace_bytes(const wv_train_convpre* h, int B, int T);
int wv_train_convpre_forward(wv_train_convpre* h, const float* x, const float* g, const float* v, const float* bias, float in_scale,
                             float* y, int B, int T, void* stream);
int wv_train_convpre_backward(wv_train_convpre* h, const float* x, const float* g, const float* v, float in_scale, const float* dy,
                              float* dx, float* dg, float* dv, float* db, int B, int T, void* workspace, size_t workspace_bytes, void* stream);

/* SpecBlock add with live weight norm (seanet.py:463-511): y[B,C,T] = x + s * ((g v/||v||)[C,F] @ P[B,F,T]),
 * s = res_scale * scale_param[0] (scale_param: device scalar of zero_init blocks, or NULL).  P is the normalised
 * log-magnitude STFT of the waveform (no parameters; wv_op_stft_logmag / the model's STFT kernel).  dx = dy (identity)
 * is the caller's; backward returns dg [C], dv [C,F], d(scale_param) and, when asked, dP = s W^T dy (the gradient towards the
 * features, continued to the audio by wv_stft_plan_backward).  y may alias x. */
typedef struct wv_train_spec wv_train_spec;
int wv_train_spec_create(int C, int F, wv_train_spec** out);
void wv_train_spec_destroy(wv_train_spec* h);
size_t wv_train_spec_workspace_bytes(const wv_train_spec* h, int B, int T);
int wv_train_spec_forward(wv_train_spec* h, const float* x, const float* P, const float* g, const float* v, const float* scale_param,
                          float res_scale, float* y, int B, int T, void* stream);
int wv_train_spec_backward(wv_train_spec* h, const float* P, const float* g, const float* v, const float* scale_param, float res_scale,
                           const float* dy, float* dg, float* dv, float* d_scale_param, float* dP /* [B,F,T] or NULL */, int B, int T,
                           void* workspace, size_t workspace_bytes, void* stream);

/* conv_post with live weight norm (seanet.py:795-822): y[B,D,T] = L2Norm( (g_pw v_pw/||v_pw||)[D,C] @ DW_ks( ELU(x[B,C,T]) ; g_dw v_dw/||v_dw|| ) + bias )
 * L2Norm = x / max(||x||_2 over channels, 1e-12) * sqrt(D) (seanet.py:288-318; l2norm = 0 skips it).  D <= 128.
 * v_dw [C,ks] (no bias on the depth-wise conv), v_pw [D,C], bias [D]. */
typedef struct wv_train_convpost wv_train_convpost;
int wv_train_convpost_create(int C, int D, int ks, wv_train_convpost** out);
void wv_train_convpost_destroy(wv_train_convpost* h);
size_t wv_train_convpost_workspace_bytes(const wv_train_convpost* h, int B, int T);
int wv_train_convpost_forward(wv_train_convpost* h, const float* x, const float* g_dw, const float* v_dw, const float* g_pw, const float* v_pw,
                              const float* bias, int l2norm, float* y, int B, int T, void* stream);
int wv_train_convpost_backward(wv_train_convpost* h, const float* x, const float* g_dw, const float* v_dw, const float* g_pw, const float* v_pw,
                               const float* bias, int l2norm, const float* dy, float* dx, float* dg_dw, float* dv_dw, float* dg_pw, float* dv_pw,
                               float* db, int B, int T, void* workspace, size_t workspace_bytes, void* stream);

/* Detector / locator head (model/detector.py:209-218,278-318; locator.py likewise): plain (not weight-normed) parameters
 *   logits[B,nb,T] = Conv1d(O, nb, 1)( ConvTranspose1d(D, O, k = s = hop)(z[B,D,N])[:, :, :T] ),  T <= N * hop
 * w_rev [D,O,hop], b_rev [O], w_last [nb,O], b_last [nb].  (Inference composes the two layers into one weight; training keeps
 * them apart so that each gets its gradient.) */
typedef struct wv_train_head wv_train_head;
int wv_train_head_create(int D, int O, int nb, int hop, wv_train_head** out);
void wv_train_head_destroy(wv_train_head* h);
size_t wv_train_head_workspace_bytes(const wv_train_head* h, int B, int N);
int wv_train_head_forward(wv_train_head* h, const float* z, const float* w_rev, const float* b_rev, const float* w_last, const float* b_last,
                          float* logits, int B, int N, int T, void* workspace, size_t workspace_bytes, void* stream);
int wv_train_head_backward(wv_train_head* h, const float* z, const float* w_rev, const float* b_rev, const float* w_last, const float* dlogits,
                           float* dz, float* dw_rev, float* db_rev, float* dw_last, float* db_last, int B, int N, int T,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The decoder's upsample unit with live weight norm (seanet.py:1110-1135; conv.py:838-881):
 *   y[B,M,r*Tin] = (g_pw v_pw/||v_pw||)[M,K] @ ConvTranspose1d_depthwise(act(pre_scale * x[B,K,Tin]); g_ct v_ct/||v_ct|| [K,2r], stride r) + bias
 * (causal: the last r samples of the transposed conv are trimmed).  ratio <= 8. */
typedef struct wv_train_up wv_train_up;
int wv_train_up_create(int K, int M, int ratio, wv_train_up** out);
void wv_train_up_destroy(wv_train_up* h);
size_t wv_train_up_workspace_bytes(const wv_train_up* h, int B, int Tin);
int wv_train_up_forward(wv_train_up* h, const float* x, const float* g_ct, const float* v_ct, const float* g_pw, const float* v_pw, const float* bias,
                        float pre_scale, int pre_elu, float* y, int B, int Tin, void* stream);
int wv_train_up_backward(wv_train_up* h, const float* x, const float* g_ct, const float* v_ct, const float* g_pw, const float* v_pw, float pre_scale,
                         int pre_elu, const float* dy, float* dx, float* dg_ct, float* dv_ct, float* dg_pw, float* dv_pw, float* db, int B, int Tin,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The decoder's tail with live weight norm (seanet.py:1166-1204; generator.py:396-413 trims to T):
 *   delta[B,1,T] = tanh( wav_std * ( causal conv1d(ELU(post * x[B,C,Tin]), g v/||v|| [1,C,ks]) + bias ) )[..., :T],  T <= Tin
 * (the watermarked audio is delta + the input clip: watermarking.py:423-441).  g [1], v [1,C,ks] (norm over the whole tensor). */
typedef struct wv_train_tail wv_train_tail;
int wv_train_tail_create(int C, int ks, wv_train_tail** out);
void wv_train_tail_destroy(wv_train_tail* h);
size_t wv_train_tail_workspace_bytes(const wv_train_tail* h, int B);
int wv_train_tail_forward(wv_train_tail* h, const float* x, const float* g, const float* v, const float* bias, float post, float wav_std,
                          float* delta, int B, int Tin, int T, void* stream);
int wv_train_tail_backward(wv_train_tail* h, const float* x, const float* g, const float* v, float post, float wav_std, const float* delta,
                           const float* d_delta, float* dx, float* dg, float* dv, float* db, int B, int Tin, int T,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Message MLP + FiLM (seanet.py:518-550,831-846,905-966).  `params` / `dparams` are one packed block:
 *   W0 [E][msg_dim], b0 [E], layers x (W [E][E], b [E]), FW [n_scales][bands][2][E] (gamma row, beta row), FB [n_scales][bands][2]
 * (wv_train_film_param_count floats).  film[B][n_scales][bands][2] = (gamma, beta) per clip, scale and frequency band;
 *   e0 = W0 msg + b0, e_l = relu(W_l e_{l-1} + b_l), gamma/beta = <FW, e_L> + FB.
 * wv_train_film_apply: y[b,c,t] = x * gamma[b][scale][band(c)] + beta (band = c / (C / bands));  its backward returns dx and fills
 * dfilm's entries of `scale`; wv_train_film_backward turns the complete dfilm into parameter gradients (ws = forward's buffer). */
size_t wv_train_film_param_count(int msg_dim, int E, int layers, int n_scales, int bands);
size_t wv_train_film_workspace_bytes(int B, int msg_dim, int E, int layers, int n_scales, int bands);
int wv_train_film_forward(const float* msg, const float* params, float* film, int B, int msg_dim, int E, int layers, int n_scales, int bands,
                          void* workspace, size_t workspace_bytes, void* stream);
int wv_train_film_backward(const float* msg, const float* params, const float* dfilm, float* dparams, int B, int msg_dim, int E, int layers,
                           int n_scales, int bands, void* workspace, size_t workspace_bytes, void* stream);
int wv_train_film_apply(const float* x, const float* film, float* y, int B, int C, int T, int bands, int n_scales, int scale, void* stream);
int wv_train_film_apply_backward(const float* x, const float* film, const float* dy, float* dx, float* dfilm, int B, int C, int T, int bands,
                                 int n_scales, int scale, void* workspace, size_t workspace_bytes, void* stream);

/* The two BCE-with-logits losses of the training step (scripts/loss.py:947-1099), forward + gradient in one pass:
 *   LocalizationLoss: msg = NULL, Cz = 1:  mean BCE(logits[B,1,T], mask[B,1,T])
 *   DecodingLoss:     mean BCE(logits[B,Cz,T], msg[B,Cz] * mask[B,1,T])        (mask NULL = all ones)
 * loss: one device float; dlogits (optional, [B,Cz,T]) = grad_scale * dLoss/dlogits.  Fixed-order two-stage sum. */
size_t wv_train_bce_workspace_bytes(void);
int wv_train_bce_logits(const float* logits, const float* mask, const float* msg, float* loss, float* dlogits, float grad_scale,
                        int B, int Cz, int T, void* workspace, size_t workspace_bytes, void* stream);
/* Waveform loss of the generator update (scripts/train.py:1322, audiotools L1Loss): loss = mean |a - b|, da (optional) =
 * grad_scale * sign(a - b) / n.  Workspace: wv_train_bce_workspace_bytes(). */
int wv_train_l1(const float* a, const float* b, float* loss, float* da, float grad_scale, size_t n, void* workspace, size_t workspace_bytes, void* stream);

/* The training units' live weight-norm fold on its own: w[m][k] = g[m] * v[m][k] / ||v[m]||, inv_norm[m] = 1 / ||v[m]|| (modules/conv.py:73-74:
 * the norm runs over all dimensions but the first; K = their product).  The same device code folds the weights inside every training
 * forward, so a trained net written out in the reference's STRIPPED checkpoint layout (scripts/train.py:1624-1629 removes the
 * parametrizations before saving) holds exactly the weights the training forward used.  g [M], v [M][K], w [M][K], inv_norm [M]: device pointers. */
int wv_train_fold_weight(const float* g, const float* v, float* w, float* inv_norm, int M, int K, void* stream);

/* Optimizer step over FLAT device arenas (a net's parameters / gradients / AdamW moments are contiguous; scripts/train.py:1346-1358,
 * conf/base.yml:128-130): wv_train_sumsq = sum of squares of the gradient arena (fixed-order two-stage sum; device scalar `out`);
 * wv_train_adamw = torch.nn.utils.clip_grad_norm_(max_norm) -- when grad_sumsq is given -- followed by torch.optim.AdamW's
 * update at 1-based step `step`; the ExponentialLR schedule is the caller's lr. */
int wv_train_sumsq(const float* g, size_t n, float* out, void* workspace, size_t workspace_bytes /* wv_train_bce_workspace_bytes() */, void* stream);
int wv_train_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* grad_sumsq, float max_norm, void* stream);
const char* wv_train_last_error(void);

/* ---- temporal augmentations of the training step ------------------------------------
 * One bandwidth-bound pass that replaces the reference's per-clip / per-segment Python loops and its GPU->CPU->GPU
 * hop (model/watermarking.py:487-519,540).  All pointers are DEVICE pointers, tensors [B,C,T] contiguous f32.
 *
 * wv_aug_localize_sequence = LocalizationAugmentation.forward (utils/localization_augmentation.py:212-321) followed
 * by SequenceAugmentation.forward (utils/seq_augmentation.py:100-273), i.e. AudioWatermarking._apply_augmentations:
 *   plan  int32 [B][nseg], nseg = ceil(T / seg_len): what happens to segment s of clip b --
 *         0 keep | 1 revert to the original | 2 replace by zeros | 3 + j substitute clip j's ORIGINAL (0 <= j < B);
 *         NULL = no localisation augmentation.  The host draws it (waveverify_amd/augment.py, same RNG call order
 *         as the reference) and validates j.
 *   seq_* the sequence map applied afterwards (WV_SEQ_*); out[t] = in[src(t)] for all three outputs:
 *         REVERSE: torch.flip;  ROLL: torch.roll(shifts = seq_a), 0 < seq_a < T;
 *         PERMUTE: segments of seq_a samples, output segment i = input segment perm[i] (device int32, T_out / seq_a
 *                  entries, the caller validates the range); T_out = n_segments * seq_a <= T (the reference drops the tail);
 *         CHUNK_SWAP: chunks [seq_a, seq_a + seq_c) and [seq_b, seq_b + seq_c) exchanged (non-overlapping).
 *   outputs: wm_out (augmented watermarked), orig_out (updated original), mask_out (1 = watermark present), [B,C,T_out].
 * wv_aug_sequence applies only the sequence map to up to three [rows,T] tensors (NULL inputs are skipped). */
#define WV_SEQ_IDENTITY 0
#define WV_SEQ_REVERSE 1
#define WV_SEQ_ROLL 2
#define WV_SEQ_PERMUTE 3
#define WV_SEQ_CHUNK_SWAP 4
int wv_aug_localize_sequence(const float* original, const float* watermarked, const int* plan, int nseg, int seg_len,
                             int seq_mode, int seq_a, int seq_b, int seq_c, const int* perm,
                             float* wm_out, float* orig_out, float* mask_out, int B, int C, int T, int T_out, void* stream);
/* gradient of the augmented audio towards the watermarked input: d_wm[b,c,ts] = d_out[b,c,t] where out[t] was copied from wm[ts]
 * (plan code 0), else 0.  inv_* = the INVERSE of the forward's sequence map (reverse and chunk swap are their own inverses; roll by a
 * -> roll by T - a; permutation -> its inverse permutation). */
int wv_aug_backward(const float* d_out, const int* plan, int nseg, int seg_len, int inv_mode, int inv_a, int inv_b, int inv_c, const int* inv_perm,
                    float* d_wm, int B, int C, int T, int T_out, void* stream);
int wv_aug_sequence(const float* in0, const float* in1, const float* in2, float* out0, float* out1, float* out2,
                    int seq_mode, int seq_a, int seq_b, int seq_c, const int* perm, int rows, int T, int T_out, void* stream);

/* ---- sinc-filter / resample effects -------------------------------------------------------------
 * A FIR filter bank over a padded signal (device pointers):  y[row][f][n] = sum_j taps[f][j] * xpad[n*stride + j], n < Tout =
 * (T + pad_l + pad_r - L) / stride + 1; replicate = 1 pads with the edge samples (julius' filters), 0 with zeros (the polyphase resampler);
 * interleave = 1 stores y[row][n*n_filters + f].  n_filters <= 8.  The taps are built on the host the way julius 0.2.7 / torchaudio
 * publish them (waveverify_amd/effects.py) -- third-party arithmetic that is not in this image: parity with the libraries is UNPINNED. */
int wv_fx_fir_bank(const float* x, const float* taps, float* y, int rows, int T, int n_filters, int L, int stride, int pad_l, int pad_r,
                   int replicate, int interleave, void* stream);
/* Adjoints of the two effects (the reference's julius filters and torchaudio resampler are plain differentiable torch ops --
 * utils/effect_augmentation.py:1451-1501,1684-1870 -- so a loss gradient crosses them through the TRANSPOSED operator):
 * the transpose of wv_fx_fir_bank with stride 1 is the same call on dy with time-reversed taps and L-1 zeros of padding on both sides
 * (giving the gradient towards the PADDED signal), followed by wv_fx_fold_replicate, the transpose of the replicate padding:
 * dx[t] = dxp[t + pad_l], with the pad samples' gradients added to dx[0] / dx[T-1].  dxp [rows][T + pad_l + pad_r], dx [rows][T]. */
int wv_fx_fold_replicate(const float* dxp, float* dx, int rows, int T, int pad_l, int pad_r, void* stream);
/* transpose of wv_fx_resample: dx[row][s] = sum_{m = n*new + f < Tout, j : n*orig + j - width = s} kernels[f][j] * dy[row][m];  dy [rows][Tout], dx [rows][T] */
int wv_fx_resample_adjoint(const float* dy, const float* kernels, float* dx, int rows, int T, int orig, int new_, int L, int width, int Tout,
                           void* stream);
/* polyphase sinc resampling by orig : nw (both already divided by their gcd), torchaudio's formulation: kernels [nw][L], L = 2*width + orig;
 * y[row][m] = sum_j kernels[m % nw][j] * xz[(m / nw)*orig + j - width] for m < Tout = ceil(nw * T / orig), xz zero outside [0,T). */
int wv_fx_resample(const float* x, const float* kernels, float* y, int rows, int T, int orig, int nw, int L, int width, int Tout, void* stream);

/* ---- measurement hook (bench.py's roofline figures) ---------------------------------------
 * When enabled, every kernel launch is bracketed by a hipEvent pair on the launch stream and
 * aggregated by "<kernel>|<role>" together with its ALGORITHMIC flops and bytes (the per-unit
 * figures of DESIGN.md).  wv_profile_collect(-1, ...) synchronises, snapshots and returns the
 * number of entries; wv_profile_collect(i, ...) reads entry i of that snapshot. */
int wv_profile_enable(int on);
int wv_profile_reset(void);
int wv_profile_collect(int index, char* name_out, int name_cap, int64_t* launches,
                       double* total_ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* WAVEVERIFY_HIP_H */
