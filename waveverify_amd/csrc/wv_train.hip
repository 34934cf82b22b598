// The training step on the GPU: forward AND backward of every unit of the Generator, Detector and Locator with
// LIVE weight normalisation (modules/conv.py:47-88: g v / ||v|| recomputed every forward while training), the BCE and waveform
// losses, gradient clipping + AdamW, behind the wv_train_* C ABI (include/waveverify_hip.h).  waveverify_amd/train.py composes the
// units into the nets and the reference's generator-update step (model/watermarking.py:340-421, scripts/train.py:1296-1358).
//
// Units (each: create / [workspace_bytes] / forward / backward; device pointers only; every reduction two-stage with a fixed order):
//   unit      y = DW_{ks,stride}(W(g,v) @ act(s x)) + b      ResnetBlock half, encoder Downsample unit, decoder input   seanet.py:39-116,733-772
//   block     y = x + s (half2 . half1)(pre x)               SEANetResnetBlock incl. res_scale_param                     seanet.py:245-281
//   convpre   causal conv 1 -> C on the scaled waveform (+ dL/dx)                                                        seanet.py:657-664
//   spec      y = x + s W @ P  (P = log-magnitude STFT features; dP for the gradient towards the audio)                  seanet.py:463-511
//   convpost  ELU -> DW conv -> 1x1 + bias -> L2Norm                                                                     seanet.py:795-822,288-318
//   head      ConvTranspose(k = s = hop) -> trim -> 1x1, in frame layout                                                 detector.py:209-218,304-310
//   up        ELU -> DW ConvTranspose(2r, r) -> 1x1 + bias   decoder upsample unit                                      seanet.py:1110-1135
//   tail      ELU -> conv C -> 1 -> wav_std -> tanh, trimmed                                                             seanet.py:1166-1204
//   film      message MLP + FiLM heads, and the modulation on the activations                                            seanet.py:518-550,831-846,928-966
//   bce / l1 / sumsq / adamw                                                                                            loss.py:947-1099, train.py:1322,1346-1358
// How they are built: the forward of a unit IS the inference kernel (K1 on the LDS-DMA core, the round-1 core at ragged lengths) fed
// by wn_fold_kernel, which folds g v / ||v|| on the device straight into the kernels' weight layouts (k-inner and K-major packs of W and
// of W^T) and keeps 1 / ||v||.  Backward runs W^T @ dy on the same K1 kernel with the activation's derivative (and a block's identity
// shortcut) in its epilogue, the weight gradient dW = sum_{b,t} dy a^T as a time-contracting MFMA GEMM (gemm_nt_kernel), the stencils'
// transposes as per-row kernels (one output frame or four samples per thread), and the weight-norm backward (wn_bwd_kernel /
// dw_param_grads_kernel).  A ResnetBlock keeps its two 1x1 outputs from the forward kernels' epilogues (K1 RES = 4 / 5, the latter also
// emitting y = x + s v), so its backward has no GEMM to recompute; a standalone unit recomputes its 1x1 output.  Layers the LDS-DMA
// core does not take (ragged / narrow) run the same steps as separate kernels (elu_bwd, axpy_res, scale_dot, add_inplace).
// Gradient parity against the reference's autograd: tests/test_gpu_train.py (units) and tests/test_gpu_trainer.py (whole nets).
#include <hip/hip_runtime.h>

#include <cmath>

#include <string>
#include <vector>

#include "../../include/waveverify_hip.h"
#include "wv_dev.h"

namespace wv {

// ---- weight-norm fold (conv.py:73-74: norm over all dims but 0) ------------------------------------------------
// v [M][K], g [M] -> w [M][K] (plain), inv_norm [M]; optional packs: wq[k/4][Mp][4] (A operand of W @ X) and
// wqT[m/4][Kp'][4] (A operand of W^T @ X, Kp' = padded K as the row count), and their K-major twins wt[k][Mp] /
// wtT[m][Kp'] for the round-1 core that takes over at ragged lengths.  Padding is zeroed once by the host.
struct WnFoldArgs {
    const float* g; const float* v; float* w; float* inv_norm; float* wq; float* wqT; int M, K, Mp, KpT;
    const float* pack_param; float pack_scale; float* wt; float* wtT;
};
__device__ __forceinline__ void wn_fold_row(const WnFoldArgs& a, int m, float* red) {
    const int tid = threadIdx.x, K = a.K;
    const float* vr = a.v + (size_t)m * K;
    float ss = 0.f;
    for (int k = tid; k < K; k += 256) ss = fmaf(vr[k], vr[k], ss);
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.f / sqrtf(ss);
    const float sc = a.g[m] * inv;
    if (tid == 0) a.inv_norm[m] = inv;
    for (int k = tid; k < K; k += 256) {
        const float x = vr[k] * sc;
        a.w[(size_t)m * K + k] = x;
        const float xs = x * (a.pack_param ? a.pack_scale * a.pack_param[0] : a.pack_scale);   // a scalar folded into the GEMM operand only
        if (a.wq) a.wq[((size_t)(k >> 2) * a.Mp + m) * 4 + (k & 3)] = xs;
        if (a.wqT) a.wqT[((size_t)(m >> 2) * a.KpT + k) * 4 + (m & 3)] = xs;
        if (a.wt) a.wt[(size_t)k * a.Mp + m] = xs;                 // K-major packs of the round-1 core (ragged T, M <= 32)
        if (a.wtT) a.wtT[(size_t)m * a.KpT + k] = xs;
    }
}
__global__ __launch_bounds__(256) void wn_fold_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                      float* __restrict__ w, float* __restrict__ inv_norm,
                                                      float* __restrict__ wq, float* __restrict__ wqT,
                                                      int M, int K, int Mp, int KpT,
                                                      const float* __restrict__ pack_param = nullptr, float pack_scale = 1.f,
                                                      float* __restrict__ wt = nullptr, float* __restrict__ wtT = nullptr) {
    __shared__ float red[4];
    wn_fold_row(WnFoldArgs{g, v, w, inv_norm, wq, wqT, M, K, Mp, KpT, pack_param, pack_scale, wt, wtT}, blockIdx.x, red);
}
// both weights of a unit in one launch: blocks [0, a.M) fold a's rows, the rest b's
__global__ __launch_bounds__(256) void wn_fold_pair_kernel(WnFoldArgs a, WnFoldArgs b) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    if (m < a.M) wn_fold_row(a, m, red);
    else wn_fold_row(b, m - a.M, red);
}

// the one-launch ResnetBlock kernel's stencil table (wv_kernels.h pack_rb_table: per channel 5 taps, bias, 1, 0) from the folded taps
__global__ __launch_bounds__(256) void rb_table_pair_kernel(const float* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ t1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ t2, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * C * 8) return;
    const int half = i / (C * 8), r = i - half * C * 8, m = r >> 3, j = r & 7;
    const float* w = half ? w2 : w1; const float* b = half ? b2 : b1; float* t = half ? t2 : t1;
    t[r] = j < 5 ? w[m * 5 + j] : (j == 5 ? (b ? b[m] : 0.f) : (j == 6 ? 1.f : 0.f));
}

// (dg, dv) of w = g * v / ||v||:  dot = <dw, v>;  dg = dot / ||v||;  dv = g / ||v|| * (dw - dot / ||v||^2 * v)
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                     const float* __restrict__ inv_norm, const float* __restrict__ dw,
                                                     float* __restrict__ dg, float* __restrict__ dv, int K) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* vr = v + (size_t)m * K;
    const float* dr = dw + (size_t)m * K;
    float dot = 0.f;
    for (int k = tid; k < K; k += 256) dot = fmaf(dr[k], vr[k], dot);
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    const float inv = inv_norm[m];
    if (tid == 0) dg[m] = dot * inv;
    const float a = g[m] * inv, c = dot * inv * inv;
    for (int k = tid; k < K; k += 256) dv[(size_t)m * K + k] = a * (dr[k] - c * vr[k]);
}

// ---- depth-wise stencil backward ---------------------------------------------------------------------------------
// y[n] = b + sum_i w[i] * h[n * stride + i - pad]  (h = 0 outside [0, Tin); SConv1d causal: pad = ks - stride,
// Tout = ceil(Tin / stride), conv.py:715-763).  One workgroup per (channel m, clip b) row:
//   dh[t] = sum_{i : (t + pad - i) = n * stride, 0 <= n < Tout} w[i] * dy[n]
//   partial[b][m][i] = sum_n dy[n] * h[n * stride + i - pad]  (i < ks),  partial[b][m][ks] = sum_n dy[n].
constexpr int TRAIN_MAX_KS = 16;
// CKS / CSTRIDE: compile-time kernel size and stride of the net's own stencils (0 = read them at run time): the tap loops
// unroll to exactly ks steps and the divisions by the stride become shifts / multiplies.
template <int CKS, int CSTRIDE>
__global__ __launch_bounds__(256) void dw_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                     const float* __restrict__ w, float* __restrict__ dh,
                                                     float* __restrict__ partial, int M, int Tin, int Tout, int ks_rt, int stride_rt, int pad,
                                                     int h_shared) {
    constexpr int NK = CKS ? CKS : TRAIN_MAX_KS;
    const int ks = CKS ? CKS : ks_rt, stride = CSTRIDE ? CSTRIDE : stride_rt;
    __shared__ float red[4][TRAIN_MAX_KS + 1];
    const int m = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    // h_shared: one input row per clip feeds every channel (conv_pre: h = the waveform); dh may be null
    const size_t row_h = h_shared ? (size_t)b * Tin : ((size_t)b * M + m) * Tin, row_y = ((size_t)b * M + m) * Tout;
    const float* dyr = dy + row_y;
    const float* hr = h + row_h;
    float wt[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) wt[i] = (dh && i < ks) ? w[m * ks + i] : 0.f;
    for (int t = tid; dh && t < Tin; t += 256) {
        float g = 0.f;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int u = t + pad - i;
            if (i < ks && u >= 0 && u % stride == 0 && u / stride < Tout) g = fmaf(wt[i], dyr[u / stride], g);
        }
        dh[((size_t)b * M + m) * Tin + t] = g;
    }
    float acc[NK + 1];
#pragma unroll
    for (int i = 0; i <= NK; ++i) acc[i] = 0.f;
    for (int n = tid; n < Tout; n += 256) {
        const float d = dyr[n];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int tb = n * stride + i - pad;
            if (i < ks && tb >= 0 && tb < Tin) acc[i] = fmaf(d, hr[tb], acc[i]);
        }
        acc[NK] += d;
    }
#pragma unroll
    for (int i = 0; i <= NK; ++i) {
        float v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid <= ks) {
        const int src = tid < ks ? tid : NK;
        partial[((size_t)b * M + m) * (ks + 1) + tid] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// The ResnetBlock stencil (ks = 5, stride 1, T % 4 == 0) four samples per thread: two 16-byte loads of dy give the 8 values dh[t..t+3]
// needs, two of h the 8 values the tap sums need.
// FUSE (the ResnetBlock's second half): dy arrives unscaled -- the kernel multiplies it by s = dy_scale * dy_scale_ptr[0] on the way in
// (what scale_dot_kernel used to write out as a tensor) and also leaves sum(dy * dot_v) of its row in dot_partial[b * M + m]
// (the gradient of res_scale_param, finished by finish_sum_kernel).
template <bool FUSE>
__global__ __launch_bounds__(256) void dw_bwd51_vec_kernel(const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ w,
                                                            float* __restrict__ dh, float* __restrict__ partial, int M, int T,
                                                            const float* __restrict__ dy_scale_ptr, float dy_scale, const float* __restrict__ dot_v,
                                                            float* __restrict__ dot_partial) {
    __shared__ float red[4][7];
    const float sc = FUSE ? (dy_scale_ptr ? dy_scale * dy_scale_ptr[0] : dy_scale) : 1.f;
    float adot = 0.f;
    const int m = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t row = ((size_t)b * M + m) * T;
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy + row);
    const f32x4* h4 = reinterpret_cast<const f32x4*>(h + row);
    const float w0 = w[m * 5], w1 = w[m * 5 + 1], w2 = w[m * 5 + 2], w3 = w[m * 5 + 3], w4 = w[m * 5 + 4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, ab = 0.f;
    const int n4 = T / 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int q = tid; q < n4; q += 256) {
        f32x4 d0 = dy4[q], d1 = q + 1 < n4 ? dy4[q + 1] : zero;             // dy[t .. t+7]
        if (FUSE) {
            const f32x4 vv = reinterpret_cast<const f32x4*>(dot_v + row)[q];
            adot += (d0.x * vv.x + d0.y * vv.y) + (d0.z * vv.z + d0.w * vv.w);
            d0 = f32x4{d0.x * sc, d0.y * sc, d0.z * sc, d0.w * sc};
            d1 = f32x4{d1.x * sc, d1.y * sc, d1.z * sc, d1.w * sc};
        }
        const f32x4 hm = q > 0 ? h4[q - 1] : zero, h0 = h4[q];                // h[t-4 .. t+3]
        const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float hv[8] = {hm.x, hm.y, hm.z, hm.w, h0.x, h0.y, h0.z, h0.w};
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e)                                          // dh[t+e] = sum_i w[i] dy[t+e+4-i]
            g[e] = fmaf(w0, d[e + 4], fmaf(w1, d[e + 3], fmaf(w2, d[e + 2], fmaf(w3, d[e + 1], w4 * d[e]))));
        reinterpret_cast<f32x4*>(dh + row)[q] = g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                        // dw[i] += dy[n] h[n-4+i], n = t+e
            const float dn = d[e];
            a0 = fmaf(dn, hv[e], a0); a1 = fmaf(dn, hv[e + 1], a1); a2 = fmaf(dn, hv[e + 2], a2); a3 = fmaf(dn, hv[e + 3], a3);
            a4 = fmaf(dn, hv[e + 4], a4); ab += dn;
        }
    }
    float acc[7] = {a0, a1, a2, a3, a4, ab, adot};
#pragma unroll
    for (int i = 0; i < (FUSE ? 7 : 6); ++i) {
        float v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 6) partial[((size_t)b * M + m) * 6 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (FUSE && tid == 6) dot_partial[(size_t)b * M + m] = red[0][6] + red[1][6] + red[2][6] + red[3][6];
}

// The downsample stencils (ks = 2 r, stride r, pad = r, Tin = r Tout: seanet.py:724-741) one output frame q per thread: the r samples
// h[q r .. q r + r - 1] meet exactly two frames, dy[q] (taps p + r) and dy[q + 1] (taps p):
//   dh[q r + p] = w[p] dy[q + 1] + w[p + r] dy[q],   dw[p + r] += dy[q] h[q r + p],   dw[p] += dy[q + 1] h[q r + p]
// -- r + 2 loads (16-byte ones when r % 4 == 0), r stores and 4 r FMAs per thread instead of 2 r tap tests per sample.
template <int R>
__global__ __launch_bounds__(256) void dw_bwd_down_kernel(const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ w,
                                                          float* __restrict__ dh, float* __restrict__ partial, int M, int Tout) {
    constexpr int KS = 2 * R;
    __shared__ float red[4][KS + 1];
    const int m = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t row_y = ((size_t)b * M + m) * Tout, row_h = row_y * R;
    const float* dyr = dy + row_y;
    const float* hr = h + row_h;
    float* dhr = dh + row_h;
    float wt[KS], acc[KS + 1];
#pragma unroll
    for (int i = 0; i < KS; ++i) { wt[i] = w[m * KS + i]; acc[i] = 0.f; }
    acc[KS] = 0.f;
    for (int q = tid; q < Tout; q += 256) {
        const float d0 = dyr[q], d1 = q + 1 < Tout ? dyr[q + 1] : 0.f;
        float hv[R], g[R];
        if constexpr (R % 4 == 0) {
#pragma unroll
            for (int c = 0; c < R / 4; ++c) {
                const f32x4 v = reinterpret_cast<const f32x4*>(hr)[q * (R / 4) + c];
                hv[4 * c] = v[0]; hv[4 * c + 1] = v[1]; hv[4 * c + 2] = v[2]; hv[4 * c + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int p = 0; p < R; ++p) hv[p] = hr[q * R + p];
        }
#pragma unroll
        for (int p = 0; p < R; ++p) {
            g[p] = fmaf(wt[p + R], d0, wt[p] * d1);
            acc[p + R] = fmaf(d0, hv[p], acc[p + R]);
            acc[p] = fmaf(d1, hv[p], acc[p]);
        }
        acc[KS] += d0;
        if constexpr (R % 4 == 0) {
#pragma unroll
            for (int c = 0; c < R / 4; ++c) {
                const f32x4 v = {g[4 * c], g[4 * c + 1], g[4 * c + 2], g[4 * c + 3]};
                reinterpret_cast<f32x4*>(dhr)[q * (R / 4) + c] = v;
            }
        } else {
#pragma unroll
            for (int p = 0; p < R; ++p) dhr[q * R + p] = g[p];
        }
    }
#pragma unroll
    for (int i = 0; i <= KS; ++i) {
        float v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid <= KS) partial[((size_t)b * M + m) * (KS + 1) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// can the ResnetBlock's second half take its upstream gradient unscaled (dw_bwd51_vec_kernel<true>)?
static bool dw_bwd_can_fuse_scale(const float* dy, const float* h, const float* dh, const float* v, int T) {
    return (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
}

static void launch_dw_bwd(hipStream_t s, const float* dy, const float* h, const float* w, float* dh, float* partial, int M, int B, int Tin,
                          int Tout, int ks, int stride, int pad, int h_shared, const float* dy_scale_ptr = nullptr, float dy_scale = 1.f,
                          const float* dot_v = nullptr, float* dot_partial = nullptr) {
#define WV_DWB(K, S) hipLaunchKernelGGL((dw_bwd_kernel<K, S>), dim3(M, B), dim3(256), 0, s, dy, h, w, dh, partial, M, Tin, Tout, ks, stride, pad, h_shared)
    const bool al16 = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dh)) & 15) == 0;
    // the stencil's transpose is bandwidth work: read dy [B, M, Tout] and h [B, M, Tin] (and v for the fused res_scale dot), write dh [B, M, Tin];
    // ks multiply-adds per sample for dh and ks for the tap sums.  Named and priced so that the training bench can put it against the HBM roof.
    std::string pname;
    if (prof::enabled()) pname = "dw_bwd<k" + std::to_string(ks) + ",s" + std::to_string(stride) + (dot_partial ? ",dot>" : ">");
    prof::Scope ps(s, pname.c_str(), 4.0 * ks * (double)B * M * Tout, 4.0 * (double)B * M * ((double)Tout * (dot_partial ? 2.0 : 1.0) + (double)Tin * (dh ? 2.0 : 1.0)));
    if (dot_partial)                                         // caller checked dw_bwd_can_fuse_scale
        hipLaunchKernelGGL((dw_bwd51_vec_kernel<true>), dim3(M, B), dim3(256), 0, s, dy, h, w, dh, partial, M, Tin, dy_scale_ptr, dy_scale, dot_v, dot_partial);
    else if (ks == 5 && stride == 1 && pad == 4 && dh && !h_shared && Tin == Tout && (Tin & 3) == 0 && al16)
        hipLaunchKernelGGL((dw_bwd51_vec_kernel<false>), dim3(M, B), dim3(256), 0, s, dy, h, w, dh, partial, M, Tin, (const float*)nullptr, 1.f,
                           (const float*)nullptr, (float*)nullptr);
    else if (dh && !h_shared && ks == 2 * stride && pad == stride && (long long)Tout * stride == Tin && (stride % 4 != 0 || al16) &&
             (stride == 2 || stride == 4 || stride == 5 || stride == 8)) {
#define WV_DWD(R) hipLaunchKernelGGL((dw_bwd_down_kernel<R>), dim3(M, B), dim3(256), 0, s, dy, h, w, dh, partial, M, Tout)
        if (stride == 2) WV_DWD(2); else if (stride == 4) WV_DWD(4); else if (stride == 5) WV_DWD(5); else WV_DWD(8);
#undef WV_DWD
    }
    else if (ks == 5 && stride == 1) WV_DWB(5, 1);
    else if (ks == 1 && stride == 1) WV_DWB(1, 1);
    else if (ks == 4 && stride == 2) WV_DWB(4, 2);
    else if (ks == 8 && stride == 4) WV_DWB(8, 4);
    else if (ks == 10 && stride == 5) WV_DWB(10, 5);
    else if (ks == 16 && stride == 8) WV_DWB(16, 8);
    else WV_DWB(0, 0);
#undef WV_DWB
}

// out[j] = sum_{s < S} part[s][j], fixed order (deterministic): workgroup = 64 outputs x 16 waves, wave w adds the splits
// w, w + 16, ... (eight loads in flight), then the 16 wave sums are added in wave order
constexpr int SP_WAVES = 16;
__global__ __launch_bounds__(64 * SP_WAVES) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int S, size_t n) {
    __shared__ float red[SP_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t j = (size_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (j < n) {
#pragma unroll 8
        for (int i = w; i < S; i += SP_WAVES) s += part[(size_t)i * n + j];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && j < n) {
        float t = red[0][lane];
#pragma unroll
        for (int k = 1; k < SP_WAVES; ++k) t += red[k][lane];
        out[j] = t;
    }
}
static void launch_sum_parts(hipStream_t st, const float* part, float* out, int S, size_t n) {
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * SP_WAVES), 0, st, part, out, S, n);
}

// The depth-wise stencil's parameter gradients of one channel row in one launch: the per-clip partial sums partial[b][m][0..ks] added
// over the clips (eight clip groups, fixed order), then db[m] = the bias column and the weight-norm backward of the ks taps
// (same formulas as wn_bwd_kernel).  Replaces sum_parts + split_dwdb + wn_bwd (three ~5 us launches per unit).
__global__ __launch_bounds__(256) void dw_param_grads_kernel(const float* __restrict__ partial, const float* __restrict__ g, const float* __restrict__ v,
                                                              const float* __restrict__ inv_norm, float* __restrict__ dg, float* __restrict__ dv,
                                                              float* __restrict__ db, int B, int M, int ks) {
    __shared__ float red[8][32];
    __shared__ float col[32];
    const int m = blockIdx.x, tid = threadIdx.x, c = tid & 31, grp = tid >> 5;
    float a = 0.f;
    if (c <= ks)
        for (int b = grp; b < B; b += 8) a += partial[((size_t)b * M + m) * (ks + 1) + c];
    red[grp][c] = a;
    __syncthreads();
    if (tid <= ks) col[tid] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
    __syncthreads();
    if (tid == 0) {
        const float* vr = v + (size_t)m * ks;
        float dot = 0.f;
        for (int i = 0; i < ks; ++i) dot = fmaf(col[i], vr[i], dot);
        const float inv = inv_norm[m];
        dg[m] = dot * inv;
        const float aa = g[m] * inv, cc = dot * inv * inv;
        for (int i = 0; i < ks; ++i) dv[(size_t)m * ks + i] = aa * (col[i] - cc * vr[i]);
        db[m] = col[ks];
    }
}

// tap / bias gradient rows: dwdb[m][0..ks] -> dw_dw[m][ks] and db[m]
__global__ void split_dwdb_kernel(const float* __restrict__ dwdb, float* __restrict__ dw, float* __restrict__ db, int M, int ks,
                                  float tap_scale = 1.f) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    for (int i = 0; i < ks; ++i) dw[m * ks + i] = dwdb[m * (ks + 1) + i] * tap_scale;
    db[m] = dwdb[m * (ks + 1) + ks];
}

// dx = da * ELU'(s x) * s,  ELU'(z) = z > 0 ? 1 : exp(z)
__global__ __launch_bounds__(256) void elu_bwd_kernel(const float* __restrict__ da, const float* __restrict__ x,
                                                      float* __restrict__ dx, float s, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 a = reinterpret_cast<const f32x4*>(da)[i], xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float z = s * xv[e]; o[e] = a[e] * (z > 0.f ? 1.f : __expf(z)) * s; }
    reinterpret_cast<f32x4*>(dx)[i] = o;
}
__global__ void elu_bwd_tail_kernel(const float* da, const float* x, float* dx, float s, size_t lo, size_t n) {
    const size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float z = s * x[i];
    dx[i] = da[i] * (z > 0.f ? 1.f : __expf(z)) * s;
}

// ---- dW = sum_{b,t} dh[b,m,t] * act(s x[b,k,t]) -----------------------------------------------------------------
// "NT" GEMM: both operands contract over their contiguous time axis.  Workgroup = 4 waves = a (64 RB) x (64 RB) tile of dW, each wave
// RB x RB blocks of 32 x 32 (RB = 2: 64 accumulator registers; every LDS operand read then feeds two MFMAs).  Per step 32 time samples
// of the tile's rows of both operands are staged in LDS ([row][t], row stride 33 floats: the 32 lanes of a fragment read -- same t,
// consecutive rows -- hit 32 banks) from 16-byte global loads that are issued one step ahead of the matrix work; the activation of
// the second operand is applied on the way in.  (clip, time chunk) items are dealt round-robin to the gridDim.z splits;
// part[split][M][K] partial sums, a fixed-order pass adds the splits.
// VEC (T % 4 == 0, 16-byte aligned operands): the fetch has no branches -- rows past M / K are clamped (they only feed accumulator
// rows that are never stored), a time overrun is clamped and the dh operand zeroed at the commit -- so the eight loads of a step
// issue back to back with one wait at the commit (with per-load bounds branches the compiler waited between the loads: 0.8x).
template <int RB, bool VEC>
__device__ __forceinline__ void gemm_nt_body(const float* __restrict__ dh, const float* __restrict__ x, float* __restrict__ part, float s, int elu,
                                             int B, int M, int K, int T, int TC) {
    constexpr int TS = 32, LD = TS + 1, ROWS = 64 * RB, NV = ROWS * TS / 4 / 256;      // float4 loads per thread and operand
    __shared__ float As[ROWS * LD], Bs[ROWS * LD];
    const int m0 = blockIdx.x * ROWS, k0 = blockIdx.y * ROWS, split = blockIdx.z, S = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wk = wave & 1;
    const int i31 = lane & 31, hh = lane >> 5;
    const int r0 = tid >> 3, tq = (tid & 7) * 4;               // this thread's row (+ 32 v) and first sample within a step
    f32x16 acc[RB][RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    unsigned offa[NV], offb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        offa[v] = (unsigned)min(m0 + r0 + 32 * v, M - 1) * (unsigned)T;
        offb[v] = (unsigned)min(k0 + r0 + 32 * v, K - 1) * (unsigned)T;
    }
    f32x4 ra[NV], rb[NV];
    auto fetch = [&](const float* dhb, const float* xb, int t0, int tb, int te) {
        if (VEC) {
            const int t = t0 + tq, tc = t < te ? t : tb;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                ra[v] = *reinterpret_cast<const f32x4*>(dhb + offa[v] + tc);
                rb[v] = *reinterpret_cast<const f32x4*>(xb + offb[v] + tc);
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t0 + tq + e < te) { a4[e] = dhb[offa[v] + t0 + tq + e]; b4[e] = xb[offb[v] + t0 + tq + e]; }
                ra[v] = a4; rb[v] = b4;
            }
        }
    };
    auto commit = [&](int t0, int te) {
        const float keep = (!VEC || t0 + tq < te) ? 1.f : 0.f;   // VEC: te - t0 is a multiple of 4, a float4 is all in or all out
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int row = r0 + 32 * v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[row * LD + tq + e] = VEC ? ra[v][e] * keep : ra[v][e];
                const float xv = rb[v][e] * s;
                Bs[row * LD + tq + e] = (!elu || xv > 0.f) ? xv : (__expf(xv) - 1.f);     // act(0) = 0 keeps the padding neutral
            }
        }
    };
    const int nch = (T + TC - 1) / TC;                         // work items = (clip, TC-sample time chunk), dealt round-robin to the splits
    for (int item = split; item < B * nch; item += S) {
        const int b = item / nch, tb = (item - b * nch) * TC, te = min(T, tb + TC);
        const float* dhb = dh + (size_t)b * M * T;
        const float* xb = x + (size_t)b * K * T;
        fetch(dhb, xb, tb, tb, te);
        for (int t0 = tb; t0 < te; t0 += TS) {
            __syncthreads();                                       // the previous step's fragments have been read
            commit(t0, te);
            __syncthreads();
            if (t0 + TS < te) fetch(dhb, xb, t0 + TS, tb, te);      // next step's loads fly under this step's MFMAs
#pragma unroll
            for (int kk = 0; kk < TS; kk += 2) {
                float av[RB], bv[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) av[i] = As[(32 * (RB * wm + i) + i31) * LD + kk + hh];
#pragma unroll
                for (int j = 0; j < RB; ++j) bv[j] = Bs[(32 * (RB * wk + j) + i31) * LD + kk + hh];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < RB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* P = part + (size_t)split * M * K;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * (RB * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * hh, k = k0 + 32 * (RB * wk + j) + i31;
                if (m < M && k < K) P[(size_t)m * K + k] = acc[i][j][r];
            }
}
// four waves per SIMD (128 registers): measured 0.85x the time of the compiler's own choice (184 registers, two waves per SIMD)
template <int RB, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                                                                float* __restrict__ part, float s, int elu, int B, int M,
                                                                                                int K, int T, int TC) {
    gemm_nt_body<RB, VEC>(dh, x, part, s, elu, B, M, K, T, TC);
}

// tile edge: 128 (two 32 x 32 blocks per wave each way) unless it pads the matrix much more than 64 does (192-row matrices: 1.8x)
static int nt_tile(int M, int K) {
    if (M <= 64 || K <= 64) return 64;
    const long long p128 = (long long)((M + 127) / 128) * ((K + 127) / 128) * 128 * 128, p64 = (long long)((M + 63) / 64) * ((K + 63) / 64) * 64 * 64;
    return p128 * 100 <= p64 * 115 ? 128 : 64;
}
static hipError_t launch_gemm_nt(hipStream_t st, const float* dh, const float* x, float* part, float s, int elu, int B, int M, int K, int T, int S, int TC) {
    const bool vec = (T & 3) == 0 && (TC & 3) == 0 && ((reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    if ((long long)std::max(M, K) * T >= (1LL << 32)) return hipErrorInvalidValue;     // 32-bit row offsets within one clip
    const int R = nt_tile(M, K);
    const dim3 g((M + R - 1) / R, (K + R - 1) / R, S);
    // event profiler (bench.py --workload train_step): dW = sum over B * T samples of an M x K outer product, both operands read once
    prof::Scope ps(st, R == 128 ? "gemm_nt<128>" : "gemm_nt<64>", 2.0 * B * (double)T * M * K, 4.0 * B * (double)T * (M + K) + 4.0 * S * (double)M * K);
    if (R == 128) {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<2, true>), g, dim3(256), 0, st, dh, x, part, s, elu, B, M, K, T, TC);
        else hipLaunchKernelGGL((gemm_nt_kernel<2, false>), g, dim3(256), 0, st, dh, x, part, s, elu, B, M, K, T, TC);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<1, true>), g, dim3(256), 0, st, dh, x, part, s, elu, B, M, K, T, TC);
        else hipLaunchKernelGGL((gemm_nt_kernel<1, false>), g, dim3(256), 0, st, dh, x, part, s, elu, B, M, K, T, TC);
    }
    return hipGetLastError();
}

// ---- residual block glue: y = x + s * v;  dv = s * dy and sum(dy * v);  dx += dy ---------------------------------
// s = res_scale * (*param) when the block carries the trainable res_scale_param (seanet.py:237-241,271-275), else res_scale
__device__ __forceinline__ float res_s(const float* param, float res_scale) { return param ? res_scale * param[0] : res_scale; }

__global__ __launch_bounds__(256) void axpy_res_kernel(const float4* __restrict__ x, const float4* __restrict__ v, float4* __restrict__ y,
                                                        const float* __restrict__ param, float res_scale, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float s = res_s(param, res_scale);
    const float4 a = x[i], b = v[i];
    y[i] = make_float4(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z), fmaf(b.w, s, a.w));   // y.mul_(scale).add_(shortcut)
}

constexpr int RED_BLOCKS = 1024;        // fixed grid of the two-stage reductions: the sum order never depends on the device

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return threadIdx.x == 0 ? (sh[0] + sh[1]) + (sh[2] + sh[3]) : 0.f;
}

__global__ __launch_bounds__(256) void scale_dot_kernel(const float4* __restrict__ dy, const float4* __restrict__ v, float4* __restrict__ dv,
                                                         const float* __restrict__ param, float res_scale, float* __restrict__ partial, size_t n4) {
    __shared__ float sh[4];
    const float s = res_s(param, res_scale);
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)RED_BLOCKS * 256) {
        const float4 g = dy[i], b = v[i];
        dv[i] = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
        acc += (g.x * b.x + g.y * b.y) + (g.z * b.z + g.w * b.w);
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// 256 threads: thread i adds its contiguous run of the partials in double, thread 0 adds the 256 thread sums in thread order (fixed order)
constexpr int FIN_T = 256;
__global__ __launch_bounds__(FIN_T) void finish_sum_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    __shared__ double red[FIN_T];
    const int tid = threadIdx.x, per = (n + FIN_T - 1) / FIN_T;
    double a = 0.0;
    for (int i = tid * per, e = min(n, (tid + 1) * per); i < e; ++i) a += (double)partial[i];
    red[tid] = a;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < FIN_T; ++k) t += red[k];
        out[0] = (float)(t * (double)scale);
    }
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float4* __restrict__ dx, const float4* __restrict__ dy, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = dx[i], b = dy[i];
    dx[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// ---- conv_pre backward towards the waveform: dx[b,t] = in_scale * sum_c sum_i w[c][i] * dy[b,c,t + (ks-1) - i] ----------
__global__ __launch_bounds__(256) void convpre_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                          int C, int T, int ks, float in_scale) {
    extern __shared__ float wl[];                                   // [C][ks]
    for (int i = threadIdx.x; i < C * ks; i += 256) wl[i] = w[i];
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    const float* dyb = dy + (size_t)b * C * T;
    float acc = 0.f;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < ks; ++i) {
            const int u = t + (ks - 1) - i;
            if (u < T) acc = fmaf(wl[c * ks + i], dyb[(size_t)c * T + u], acc);
        }
    dx[(size_t)b * T + t] = acc * in_scale;
}

// partial[block] = sum over the block's grid-stride elements of a[i] * b[i]  (RED_BLOCKS blocks, then finish_sum_kernel: fixed order)
__global__ __launch_bounds__(256) void dot_parts_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)RED_BLOCKS * 256) acc = fmaf(a[i], b[i], acc);
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// out[0] = scale * sum_i a[i] * b[i]  (one workgroup, fixed order);  buf *= res_s(param, res_scale)
__global__ __launch_bounds__(256) void dot_small_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float scale,
                                                         float* __restrict__ out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256) acc = fmaf(a[i], b[i], acc);
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = t * scale;
}
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ buf, size_t n, const float* __restrict__ param, float res_scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) buf[i] *= res_s(param, res_scale);
}

// ---- conv_post pieces (seanet.py:795-822): a = ELU(x), h = causal depth-wise conv(a);  L2Norm backward ----------------------
__global__ __launch_bounds__(256) void elu_dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ a,
                                                          float* __restrict__ h, int C, int T, int ks) {
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t row = ((size_t)b * C + c) * T;
    float wt[TRAIN_MAX_KS];
#pragma unroll
    for (int i = 0; i < TRAIN_MAX_KS; ++i) wt[i] = i < ks ? w[c * ks + i] : 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < TRAIN_MAX_KS; ++i) {
            const int u = t - (ks - 1) + i;
            if (i < ks && u >= 0) { const float v = x[row + u]; acc = fmaf(wt[i], v > 0.f ? v : (__expf(v) - 1.f), acc); }
        }
        const float v = x[row + t];
        a[row + t] = v > 0.f ? v : (__expf(v) - 1.f);
        h[row + t] = acc;
    }
}

// y = z / max(||z||_2 over channels, eps) * sqrt(D)  (seanet.py:288-318, F.normalize).  In place: z <- dL/dz.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(float* __restrict__ z, const float* __restrict__ dy, int D, int T, float eps) {
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    float* zb = z + (size_t)b * D * T + t;
    const float* db = dy + (size_t)b * D * T + t;
    float ss = 0.f, dot = 0.f;
    for (int d = 0; d < D; ++d) { const float v = zb[(size_t)d * T]; ss = fmaf(v, v, ss); dot = fmaf(v, db[(size_t)d * T], dot); }
    const float n = sqrtf(ss), sc = sqrtf((float)D);
    if (n > eps) {
        const float inv = 1.f / n, k = dot * inv * inv;
        for (int d = 0; d < D; ++d) zb[(size_t)d * T] = sc * inv * (db[(size_t)d * T] - zb[(size_t)d * T] * k);
    } else {
        for (int d = 0; d < D; ++d) zb[(size_t)d * T] = sc / eps * db[(size_t)d * T];
    }
}

// ---- detector / locator head (detector.py:209-218,278-318): ConvTranspose1d(D, O, k = s = hop) -> trim -> Conv1d(O, nb, 1) ------
// Everything runs in FRAME layout q[B][rows][hop][N] (row-major over (j, n)): the transposed conv is then a plain GEMM over the
// latent frames and the 1x1 a GEMM over the hop*N axis; two transposes move logits / their gradient between [B][nb][T] and frames.
__global__ __launch_bounds__(256) void time_to_frames_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int hop, int N) {
    const size_t row = blockIdx.y;                                  // (b, r)
    const int i = blockIdx.x * 256 + threadIdx.x;                   // index into [hop][N]
    if (i >= hop * N) return;
    const int j = i / N, n = i - j * N, t = n * hop + j;
    out[row * hop * N + i] = t < T ? in[row * T + t] : 0.f;         // positions past T were trimmed: no gradient
}
__global__ __launch_bounds__(256) void frames_to_time_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int hop, int N) {
    const size_t row = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int n = t / hop, j = t - n * hop;
    out[row * T + t] = in[row * hop * N + (size_t)j * N + n];
}
// dst wt[k][Mp] (K-major pack of an [M][K] matrix); src is [M][K] (transposed = 0) or [K][M] (transposed = 1); padding pre-zeroed
__global__ __launch_bounds__(256) void pack_wt_kernel(const float* __restrict__ src, float* __restrict__ dst, int M, int K, int Mp, int transposed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * K) return;
    const int m = (int)(i % M), k = (int)(i / M);
    dst[(size_t)k * Mp + m] = transposed ? src[(size_t)k * M + m] : src[(size_t)m * K + k];
}
__global__ void expand_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int O, int hop) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < O * hop) out[i] = b[i / hop];
}

// ---- upsample unit pieces (decoder, seanet.py:1110-1135): u = depth-wise ConvTranspose1d(k = 2r, stride r) of a = act(s x), right-trimmed
// to r * Tin (conv.py:838-881):  u[t] = a[l] w[j] + a[l-1] w[j + r],  l = t / r, j = t % r.
__global__ __launch_bounds__(256) void convtr_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ u,
                                                          int K, int Tin, int r, float s, int elu) {
    const int k = blockIdx.x, b = blockIdx.y, Tout = Tin * r;
    const float* xr = x + ((size_t)b * K + k) * Tin;
    float* ur = u + ((size_t)b * K + k) * Tout;
    const float* wk = w + (size_t)k * 2 * r;
    for (int t = threadIdx.x; t < Tout; t += 256) {
        const int l = t / r, j = t - l * r;
        float a0 = s * xr[l];
        if (elu) a0 = a0 > 0.f ? a0 : (__expf(a0) - 1.f);
        float v = a0 * wk[j];
        if (l > 0) {
            float a1 = s * xr[l - 1];
            if (elu) a1 = a1 > 0.f ? a1 : (__expf(a1) - 1.f);
            v = fmaf(a1, wk[j + r], v);
        }
        ur[t] = v;
    }
}

// The same, one INPUT frame l per thread (R = the ratio at compile time): a[l] is activated once (the per-output form above evaluates two
// exponentials per output sample), the R outputs u[l R .. l R + R - 1] = a[l] w[0..R) + a[l-1] w[R..2R) leave as 16-byte stores when
// R % 4 == 0.  Same arithmetic per output (product, then one fma).
template <int R>
__global__ __launch_bounds__(256) void convtr_fwd_frame_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ u,
                                                                int K, int Tin, float s, int elu) {
    const int k = blockIdx.x, b = blockIdx.y;
    const float* xr = x + ((size_t)b * K + k) * Tin;
    float* ur = u + ((size_t)b * K + k) * Tin * R;
    float wt[2 * R];
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) wt[j] = w[(size_t)k * 2 * R + j];
    for (int l = threadIdx.x; l < Tin; l += 256) {
        float a0 = s * xr[l], a1 = l > 0 ? s * xr[l - 1] : 0.f;
        if (elu) { a0 = a0 > 0.f ? a0 : (__expf(a0) - 1.f); a1 = a1 > 0.f ? a1 : (__expf(a1) - 1.f); }
        float v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = l > 0 ? fmaf(a1, wt[j + R], a0 * wt[j]) : a0 * wt[j];
        if constexpr (R % 4 == 0) {
#pragma unroll
            for (int c = 0; c < R / 4; ++c) reinterpret_cast<f32x4*>(ur)[l * (R / 4) + c] = f32x4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) ur[l * R + j] = v[j];
        }
    }
}

// ... and its backward per input frame: the 2R gradients du[l R .. l R + 2R - 1] that meet a[l] are this frame's R samples and the next
// frame's (zero past the end); one exponential per frame serves both the activation and its derivative.
template <int R>
__global__ __launch_bounds__(256) void convtr_bwd_frame_kernel(const float* __restrict__ du, const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ dx, float* __restrict__ partial, int K, int Tin, float s, int elu) {
    constexpr int KS = 2 * R;
    __shared__ float red[4][KS];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* dur = du + ((size_t)b * K + k) * Tin * R;
    const float* xr = x + ((size_t)b * K + k) * Tin;
    float wt[KS], acc[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) { wt[j] = w[(size_t)k * KS + j]; acc[j] = 0.f; }
    for (int l = tid; l < Tin; l += 256) {
        const float z = s * xr[l];
        const float e = (elu && !(z > 0.f)) ? __expf(z) : 1.f;
        const float a = (!elu || z > 0.f) ? z : (e - 1.f);
        float d[KS];
        if constexpr (R % 4 == 0) {
#pragma unroll
            for (int c = 0; c < KS / 4; ++c) {
                const bool in = c < R / 4 || l + 1 < Tin;
                const f32x4 v = in ? reinterpret_cast<const f32x4*>(dur)[l * (R / 4) + c] : f32x4{0.f, 0.f, 0.f, 0.f};
                d[4 * c] = v[0]; d[4 * c + 1] = v[1]; d[4 * c + 2] = v[2]; d[4 * c + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < KS; ++j) d[j] = (j < R || l + 1 < Tin) ? dur[l * R + j] : 0.f;
        }
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) { g = fmaf(wt[j], d[j], g); acc[j] = fmaf(a, d[j], acc[j]); }
        if (dx) dx[((size_t)b * K + k) * Tin + l] = g * ((!elu || z > 0.f) ? 1.f : e) * s;
    }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        float v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][j] = v;
    }
    __syncthreads();
    if (tid < KS) partial[((size_t)b * K + k) * KS + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// backward: da[l] = sum_{j < 2r, l r + j < Tout} w[j] du[l r + j];  dx[l] = da[l] act'(s x[l]) s;
// partial[b][k][j] = sum_l a[l] du[l r + j]   (2r <= TRAIN_MAX_KS taps)
__global__ __launch_bounds__(256) void convtr_bwd_kernel(const float* __restrict__ du, const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ dx, float* __restrict__ partial, int K, int Tin, int r, float s, int elu) {
    __shared__ float red[4][TRAIN_MAX_KS];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, Tout = Tin * r, ks = 2 * r;
    const float* dur = du + ((size_t)b * K + k) * Tout;
    const float* xr = x + ((size_t)b * K + k) * Tin;
    float wt[TRAIN_MAX_KS], acc[TRAIN_MAX_KS];
#pragma unroll
    for (int j = 0; j < TRAIN_MAX_KS; ++j) { wt[j] = j < ks ? w[(size_t)k * ks + j] : 0.f; acc[j] = 0.f; }
    for (int l = tid; l < Tin; l += 256) {
        const float z = s * xr[l];
        const float a = (!elu || z > 0.f) ? z : (__expf(z) - 1.f);
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < TRAIN_MAX_KS; ++j) {
            const int t = l * r + j;
            if (j < ks && t < Tout) { const float d = dur[t]; g = fmaf(wt[j], d, g); acc[j] = fmaf(a, d, acc[j]); }
        }
        if (dx) dx[((size_t)b * K + k) * Tin + l] = g * ((!elu || z > 0.f) ? 1.f : __expf(z)) * s;
    }
#pragma unroll
    for (int j = 0; j < TRAIN_MAX_KS; ++j) {
        float v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][j] = v;
    }
    __syncthreads();
    if (tid < ks) partial[((size_t)b * K + k) * ks + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// taps [K][2r] -> the loader's transposed, zero padded copy [2r][Kp]
__global__ void pack_ct_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int Kp, int ks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * ks) return;
    const int k = i / ks, j = i - k * ks;
    wt[(size_t)j * Kp + k] = w[i];
}

// ---- decoder tail (seanet.py:1166-1204): delta = tanh(wav_std * (conv1d(ELU(post * h), w[1,C,ks]) + b)) --------------------------
// One workgroup per (channel, clip) row.  dq[u] = d_delta[u] (1 - delta[u]^2) wav_std is formed on the fly (u < T; the decoder's
// extra samples past T were trimmed: no gradient);  dh[t] = post ELU'(post h[t]) sum_i w[c][i] dq[t + ks-1 - i];
// partial[b][c][i] = sum_n dq[n] ELU(post h[n - (ks-1) + i]),  partial[b][c][ks] = sum_n dq[n].
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ delta,
                                                        const float* __restrict__ dd, float* __restrict__ dh, float* __restrict__ partial,
                                                        int C, int Tin, int T, int ks, float post, float wav_std) {
    __shared__ float red[4][TRAIN_MAX_KS + 1];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* hr = h + ((size_t)b * C + c) * Tin;
    const float* ddr = dd + (size_t)b * T;
    const float* der = delta + (size_t)b * T;
    float wt[TRAIN_MAX_KS], acc[TRAIN_MAX_KS + 1];
#pragma unroll
    for (int i = 0; i < TRAIN_MAX_KS; ++i) { wt[i] = i < ks ? w[c * ks + i] : 0.f; acc[i] = 0.f; }
    acc[TRAIN_MAX_KS] = 0.f;
    for (int t = tid; t < Tin; t += 256) {
        float g = 0.f;
#pragma unroll
        for (int i = 0; i < TRAIN_MAX_KS; ++i) {
            const int u = t + (ks - 1) - i;
            if (i < ks && u < T) { const float dl = der[u]; g = fmaf(wt[i], ddr[u] * (1.f - dl * dl) * wav_std, g); }
        }
        const float z = post * hr[t];
        dh[((size_t)b * C + c) * Tin + t] = g * (z > 0.f ? 1.f : __expf(z)) * post;
    }
    for (int n = tid; n < T; n += 256) {
        const float dl = der[n], d = ddr[n] * (1.f - dl * dl) * wav_std;
#pragma unroll
        for (int i = 0; i < TRAIN_MAX_KS; ++i) {
            const int tb = n - (ks - 1) + i;
            if (i < ks && tb >= 0) { const float z = post * hr[tb]; acc[i] = fmaf(d, z > 0.f ? z : (__expf(z) - 1.f), acc[i]); }
        }
        acc[TRAIN_MAX_KS] += d;
    }
#pragma unroll
    for (int i = 0; i <= TRAIN_MAX_KS; ++i) {
        float v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid <= ks) {
        const int src = tid < ks ? tid : TRAIN_MAX_KS;
        partial[((size_t)b * C + c) * (ks + 1) + tid] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// The same for ks = 5 (the net's tail), four samples per thread: h[t..t+3] is activated ONCE per sample (the per-tap form above evaluates
// ks + 1 exponentials per sample) and serves the derivative, and every tap sum takes its term from the sample's side:
//     dh[t] = post ELU'(z_t) sum_i w[i] dq[t + 4 - i],   dw[i] += dq[t + 4 - i] ELU(z_t),   db += dq[t]   (dq = 0 past T)
// Needs T % 4 == 0, Tin % 4 == 0 and 16-byte aligned rows.
__global__ __launch_bounds__(256) void tail_bwd5_vec_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ delta,
                                                             const float* __restrict__ dd, float* __restrict__ dh, float* __restrict__ partial,
                                                             int C, int Tin, int T, float post, float wav_std) {
    __shared__ float red[4][6];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t row = ((size_t)b * C + c) * Tin;
    const f32x4* h4 = reinterpret_cast<const f32x4*>(h + row);
    const f32x4* dd4 = reinterpret_cast<const f32x4*>(dd + (size_t)b * T);
    const f32x4* de4 = reinterpret_cast<const f32x4*>(delta + (size_t)b * T);
    const float w0 = w[c * 5], w1 = w[c * 5 + 1], w2 = w[c * 5 + 2], w3 = w[c * 5 + 3], w4 = w[c * 5 + 4];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int n4 = Tin / 4, m4 = T / 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int q = tid; q < n4; q += 256) {
        const f32x4 hv = h4[q];
        const f32x4 da = q < m4 ? dd4[q] : zero, db = q + 1 < m4 ? dd4[q + 1] : zero;
        const f32x4 ea = q < m4 ? de4[q] : zero, eb = q + 1 < m4 ? de4[q + 1] : zero;
        float dq[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { dq[e] = da[e] * (1.f - ea[e] * ea[e]) * wav_std; dq[4 + e] = db[e] * (1.f - eb[e] * eb[e]) * wav_std; }
        f32x4 g4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = post * hv[e];
            const float ex = z > 0.f ? 1.f : __expf(z), a = z > 0.f ? z : (ex - 1.f);
            float g = w0 * dq[e + 4];
            g = fmaf(w1, dq[e + 3], g); g = fmaf(w2, dq[e + 2], g); g = fmaf(w3, dq[e + 1], g); g = fmaf(w4, dq[e], g);
            g4[e] = g * ex * post;
            acc[0] = fmaf(dq[e + 4], a, acc[0]); acc[1] = fmaf(dq[e + 3], a, acc[1]); acc[2] = fmaf(dq[e + 2], a, acc[2]);
            acc[3] = fmaf(dq[e + 1], a, acc[3]); acc[4] = fmaf(dq[e], a, acc[4]);
            acc[5] += dq[e];
        }
        reinterpret_cast<f32x4*>(dh + row)[q] = g4;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 6) partial[((size_t)b * C + c) * 6 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---- message MLP + FiLM (seanet.py:518-550,831-846,905-966) -------------------------------------------------------------------------
// Packed parameter block (and gradient block, same layout):  W0 [E][Dm], b0 [E], then L x (W [E][E], b [E]), then the FiLM heads
// FW [S][bands][2][E] (gamma row, beta row) and FB [S][bands][2].  One workgroup per clip; E <= 256.
//   e0 = W0 msg + b0;  e_l = relu(W_l e_{l-1} + b_l);  film[b][s][band][0|1] = <FW[s][band][0|1], e_L> + FB
constexpr int FILM_MAX_E = 256, FILM_MAX_L = 4;
__global__ __launch_bounds__(256) void msg_film_fwd_kernel(const float* __restrict__ msg, const float* __restrict__ P, float* __restrict__ film,
                                                            float* __restrict__ acts, int Dm, int E, int L, int NF) {
    __shared__ float e[2][FILM_MAX_E];
    const int b = blockIdx.x, j = threadIdx.x;
    const float* W0 = P; const float* b0 = P + (size_t)E * Dm;
    if (j < E) {
        float a = b0[j];
        for (int i = 0; i < Dm; ++i) a = fmaf(W0[(size_t)j * Dm + i], msg[(size_t)b * Dm + i], a);
        e[0][j] = a;
        if (acts) acts[((size_t)b * (L + 1)) * E + j] = a;
    }
    __syncthreads();
    const float* Wl = b0 + E;
    int cur = 0;
    for (int l = 0; l < L; ++l) {
        if (j < E) {
            float a = Wl[(size_t)E * E + j];
            for (int i = 0; i < E; ++i) a = fmaf(Wl[(size_t)j * E + i], e[cur][i], a);
            a = fmaxf(a, 0.f);
            e[cur ^ 1][j] = a;
            if (acts) acts[((size_t)b * (L + 1) + l + 1) * E + j] = a;
        }
        __syncthreads();
        cur ^= 1;
        Wl += (size_t)E * E + E;
    }
    const float* FW = Wl; const float* FB = FW + (size_t)NF * E;
    for (int f = j; f < NF; f += 256) {
        float a = FB[f];
        for (int i = 0; i < E; ++i) a = fmaf(FW[(size_t)f * E + i], e[cur][i], a);
        film[(size_t)b * NF + f] = a;
    }
}

// backward of the above for one clip: per-clip gradient block gpart[b][...] (summed over clips afterwards, fixed order)
__global__ __launch_bounds__(256) void msg_film_bwd_kernel(const float* __restrict__ msg, const float* __restrict__ P, const float* __restrict__ acts,
                                                            const float* __restrict__ dfilm, float* __restrict__ gpart, int Dm, int E, int L, int NF, size_t np) {
    __shared__ float de[FILM_MAX_E], dz[FILM_MAX_E];
    const int b = blockIdx.x, j = threadIdx.x;
    float* G = gpart + (size_t)b * np;
    const size_t off_l0 = (size_t)E * Dm + E, per = (size_t)E * E + E, off_f = off_l0 + (size_t)L * per;
    const float* FW = P + off_f;
    const float* eL = acts + ((size_t)b * (L + 1) + L) * E;
    const float* df = dfilm + (size_t)b * NF;
    // FiLM heads
    for (int f = j; f < NF; f += 256) G[off_f + (size_t)NF * E + f] = df[f];
    for (size_t i = j; i < (size_t)NF * E; i += 256) G[off_f + i] = df[i / E] * eL[i % E];
    if (j < E) {
        float a = 0.f;
        for (int f = 0; f < NF; ++f) a = fmaf(df[f], FW[(size_t)f * E + j], a);
        de[j] = a;
    }
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
        const float* Wl = P + off_l0 + (size_t)l * per;
        const float* eout = acts + ((size_t)b * (L + 1) + l + 1) * E;
        const float* ein = acts + ((size_t)b * (L + 1) + l) * E;
        if (j < E) dz[j] = eout[j] > 0.f ? de[j] : 0.f;
        __syncthreads();
        float* GW = G + off_l0 + (size_t)l * per;
        for (size_t i = j; i < (size_t)E * E; i += 256) GW[i] = dz[i / E] * ein[i % E];
        if (j < E) GW[(size_t)E * E + j] = dz[j];
        float a = 0.f;
        if (j < E) for (int o = 0; o < E; ++o) a = fmaf(Wl[(size_t)o * E + j], dz[o], a);
        __syncthreads();
        if (j < E) de[j] = a;
        __syncthreads();
    }
    for (size_t i = j; i < (size_t)E * Dm; i += 256) G[i] = de[i / Dm] * msg[(size_t)b * Dm + i % Dm];
    if (j < E) G[(size_t)E * Dm + j] = de[j];
}

// FiLM on the activations: y = x * gamma[b][band] + beta[b][band], band = c / (C / bands);  film points at this scale's [B][NF] block
__global__ __launch_bounds__(256) void film_apply_kernel(const float* __restrict__ x, const float* __restrict__ film, float* __restrict__ y,
                                                          int C, int T, int bands, int NF, int s_off) {
    const int c = blockIdx.x, b = blockIdx.y, band = c / (C / bands);
    const float g = film[(size_t)b * NF + s_off + band * 2], be = film[(size_t)b * NF + s_off + band * 2 + 1];
    const size_t row = ((size_t)b * C + c) * T;
    for (int t = threadIdx.x; t < T; t += 256) y[row + t] = fmaf(x[row + t], g, be);
}
// dx = dy * gamma;  rowsum[b][c] = (sum_t dy x, sum_t dy)
__global__ __launch_bounds__(256) void film_apply_bwd_kernel(const float* __restrict__ x, const float* __restrict__ film, const float* __restrict__ dy,
                                                              float* __restrict__ dx, float* __restrict__ rowsum, int C, int T, int bands, int NF, int s_off) {
    __shared__ float sh[4];
    const int c = blockIdx.x, b = blockIdx.y, band = c / (C / bands);
    const float g = film[(size_t)b * NF + s_off + band * 2];
    const size_t row = ((size_t)b * C + c) * T;
    float a0 = 0.f, a1 = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { const float d = dy[row + t]; a0 = fmaf(d, x[row + t], a0); a1 += d; dx[row + t] = d * g; }
    const float s0 = block_sum(a0, sh);
    __syncthreads();
    const float s1 = block_sum(a1, sh);
    if (threadIdx.x == 0) { rowsum[((size_t)b * C + c) * 2] = s0; rowsum[((size_t)b * C + c) * 2 + 1] = s1; }
}
// dfilm[b][s_off + band*2 + {0,1}] = sum over the band's channels of rowsum (fixed order)
__global__ void film_band_reduce_kernel(const float* __restrict__ rowsum, float* __restrict__ dfilm, int B, int C, int bands, int NF, int s_off) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * bands * 2) return;
    const int b = i / (bands * 2), r = i - b * bands * 2, band = r >> 1, which = r & 1, bw = C / bands;
    float a = 0.f;
    for (int c = band * bw; c < (band + 1) * bw; ++c) a += rowsum[((size_t)b * C + c) * 2 + which];
    dfilm[(size_t)b * NF + s_off + band * 2 + which] = a;
}

// ---- waveform loss: mean |a - b| and its gradient towards a (audiotools L1Loss in scripts/train.py:1322; sign(0) = 0 as torch) ----
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da,
                                                  float* __restrict__ partial, float gscale, size_t n) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)RED_BLOCKS * 256) {
        const float d = a[i] - b[i];
        acc += fabsf(d);
        if (da) da[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---- optimizer step over a FLAT parameter arena (scripts/train.py:1346-1358, conf/base.yml:128-130) ------------------
// Parameters, gradients and both AdamW moments of a net live in contiguous arenas, so gradient clipping is one
// two-stage sum of squares and the update one launch (and the DDP buckets are plain slices of the gradient arena).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)RED_BLOCKS * 256) acc = fmaf(g[i], g[i], acc);
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// torch.nn.utils.clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6))) followed by torch.optim.AdamW's update:
//   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                                     float bc1, float bc2_sqrt, const float* __restrict__ sumsq, float max_norm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float coef = 1.f;
    if (sumsq) coef = fminf(max_norm / (sqrtf(sumsq[0]) + 1e-6f), 1.f);
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);                         // lerp, as torch does it
    const float vi = v[i] * b2 + gi * gi * (1.f - b2);
    m[i] = mi; v[i] = vi;
    pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi;
}

// ---- BCE-with-logits losses of the training step (scripts/loss.py:947-1099) -----------------------------------------
//   LocalizationLoss: mean BCE(z[B,1,T], mask[B,1,T]);  DecodingLoss: mean BCE(z[B,nb,T], msg[B,nb] * mask[B,1,T]).
// One pass gives the loss partials and dL/dz = grad_scale * (sigmoid(z) - y) / N.  The loss term is torch's stable
// form max(z,0) - z*y + log1p(exp(-|z|)).
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, const float* __restrict__ mask, const float* __restrict__ msg,
                                                   float* __restrict__ dz, float* __restrict__ partial, float gscale, int Cz, int T, size_t n) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)RED_BLOCKS * 256) {
        const size_t bc = i / T;
        const int t = (int)(i - bc * T);
        const size_t b = bc / Cz;
        float y = mask ? mask[b * T + t] : 1.f;
        if (msg) y *= msg[bc];
        const float v = z[i];
        acc += fmaxf(v, 0.f) - v * y + log1pf(expf(-fabsf(v)));
        if (dz) dz[i] = (1.f / (1.f + expf(-v)) - y) * gscale;
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

}  // namespace wv

// ================================================================================================ C ABI
namespace {
thread_local std::string g_terr;
int tfail(int code, const std::string& msg) { g_terr = msg; return code; }
size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

struct wv_train_unit {
    int K = 0, M = 0, ks = 5, stride = 1, pad = 4, Mp = 0, KpT = 0;
    float *w_pw = nullptr, *inv_pw = nullptr, *wq = nullptr, *wqT = nullptr, *wt = nullptr, *wtT = nullptr;   // folded 1x1 weight + packs
    float *w_dw = nullptr, *inv_dw = nullptr, *id_taps = nullptr;                 // folded taps [M][ks]; identity stencil rows
    float *dW = nullptr, *dwdb = nullptr, *dw_taps = nullptr;                     // weight-gradient scratch
    const float* folded[4] = {nullptr, nullptr, nullptr, nullptr};               // the (g_pw, v_pw, g_dw, v_dw) the folded copies were made from
    std::vector<void*> owned;
    ~wv_train_unit() { for (void* p : owned) (void)hipFree(p); }
};

extern "C" {

const char* wv_train_last_error(void) { return g_terr.c_str(); }

int wv_train_unit_create(int K, int M, int ks, int stride, wv_train_unit** out) {
    if (!out || K < 1 || K > 4096 || M < 1 || M > 4096) return tfail(WV_EINVAL, "bad channel count");
    if (ks < 1 || ks > wv::TRAIN_MAX_KS || stride < 1 || ks - stride < 0) return tfail(WV_EINVAL, "bad kernel size / stride");
    auto* h = new wv_train_unit();
    h->K = K; h->M = M; h->ks = ks; h->stride = stride; h->pad = ks - stride;
    h->Mp = wv::round_up(M, wv::M_ALIGN); h->KpT = wv::round_up(K, wv::M_ALIGN);
    const size_t nq = (size_t)wv::round_up(K, 32) * h->Mp, nqT = (size_t)wv::round_up(M, 32) * h->KpT;
    auto alloc = [&](float** p, size_t n, bool zero) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return !zero || hipMemset(*p, 0, n * sizeof(float)) == hipSuccess;
    };
    const int R = std::max(K, M);
    std::vector<float> taps((size_t)R * 5, 0.f);
    for (int m = 0; m < R; ++m) taps[(size_t)m * 5 + 4] = 1.f;
    bool ok = alloc(&h->w_pw, (size_t)M * K, false) && alloc(&h->inv_pw, M, false) && alloc(&h->wq, nq, true) &&
              alloc(&h->wqT, nqT, true) && alloc(&h->wt, (size_t)wv::round_up(K, wv::BK) * h->Mp, true) &&
              alloc(&h->wtT, (size_t)wv::round_up(M, wv::BK) * h->KpT, true) && alloc(&h->w_dw, (size_t)M * ks, false) && alloc(&h->inv_dw, M, false) &&
              alloc(&h->id_taps, taps.size(), false) && alloc(&h->dW, (size_t)M * K, false) &&
              alloc(&h->dwdb, (size_t)M * (ks + 1), false) && alloc(&h->dw_taps, (size_t)M * ks, false) &&
              hipMemcpy(h->id_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { delete h; return tfail(WV_EHIP, "device allocation failed"); }
    *out = h;
    return WV_OK;
}

void wv_train_unit_destroy(wv_train_unit* h) { delete h; }
int wv_train_half_create(int C, wv_train_unit** out) { return wv_train_unit_create(C, C, 5, 1, out); }
void wv_train_half_destroy(wv_train_unit* h) { delete h; }

// splits of the dW GEMM: enough workgroups to fill the chip even when the weight matrix is one 64 x 64 tile (C = 64 layers), within
// a 128 MB scratch; a function of the shapes only, so the summation order -- and the result -- is reproducible
struct NtPlan { int S, TC; };
static NtPlan nt_plan(int B, int T, int M, int K) {
    const int te = wv::nt_tile(M, K);
    const long long tiles = (long long)((M + te - 1) / te) * ((K + te - 1) / te);
    const int TC = 512;
    const long long items = (long long)B * ((T + TC - 1) / TC);
    long long S = std::min<long long>(items, std::max<long long>(1, 1024 / tiles));
    while (S > 1 && S * M * K > (32LL << 20)) S /= 2;          // <= 128 MB of partial sums (a 16 MB cap left one workgroup per CU: 1.07x the step)
    return NtPlan{(int)S, TC};
}
static int t_out(const wv_train_unit* h, int Tin) { return (Tin + h->stride - 1) / h->stride; }

size_t wv_train_unit_workspace_bytes(const wv_train_unit* h, int B, int Tin) {
    if (!h || B < 1 || Tin < 1) return 0;
    const size_t am = al256((size_t)B * h->M * Tin * 4), ak = al256((size_t)B * h->K * Tin * 4);
    return 2 * am + ak + al256((size_t)B * h->M * (h->ks + 1) * 4) + al256((size_t)nt_plan(B, Tin, h->M, h->K).S * h->M * h->K * 4);
}
size_t wv_train_half_workspace_bytes(const wv_train_unit* h, int B, int T) { return wv_train_unit_workspace_bytes(h, B, T); }

#define T_LAUNCH(expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return tfail(WV_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// fold both weights of the unit for this step (live weight norm)
static int fold_step(wv_train_unit* h, const float* g_pw, const float* v_pw, const float* g_dw, const float* v_dw, hipStream_t s) {
    const wv::WnFoldArgs fa{g_pw, v_pw, h->w_pw, h->inv_pw, h->wq, h->wqT, h->M, h->K, h->Mp, h->KpT, nullptr, 1.f, h->wt, h->wtT};
    const wv::WnFoldArgs fb{g_dw, v_dw, h->w_dw, h->inv_dw, nullptr, nullptr, h->M, h->ks, 0, 0, nullptr, 1.f, nullptr, nullptr};
    hipLaunchKernelGGL(wv::wn_fold_pair_kernel, dim3(2 * h->M), dim3(256), 0, s, fa, fb);
    T_LAUNCH(hipGetLastError());
    h->folded[0] = g_pw; h->folded[1] = v_pw; h->folded[2] = g_dw; h->folded[3] = v_dw;
    return WV_OK;
}

static wv::PwWeight pack_of(const wv_train_unit* h, bool transposed) {
    wv::PwWeight p;
    if (!transposed) { p.M = h->M; p.K = h->K; p.Mp = h->Mp; p.Kp = wv::round_up(h->K, wv::BK); p.wq = h->wq; p.wt = h->wt; }
    else { p.M = h->K; p.K = h->M; p.Mp = h->KpT; p.Kp = wv::round_up(h->M, wv::BK); p.wq = h->wqT; p.wt = h->wtT; }
    return p;
}

// h_out (optional, stride-1 units): the 1x1 output H = W act(s x) [B, M, Tin], which the backward otherwise recomputes -- stored by the
// forward kernel itself next to y where the LDS-DMA core runs the layer, else by one more pass with the identity stencil
// sum_x / sum_scale_ptr / sum_scale / ysum (optional, with h_out): ysum = sum_x + sum_scale * sum_scale_ptr[0] * y from the same
// epilogue (the ResnetBlock's output); *summed reports whether the kernel took it (else the caller adds it).
static int unit_forward_impl(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw, const float* v_dw,
                             const float* bias, float pre_scale, int pre_elu, float* y, float* h_out, int B, int Tin, void* stream,
                             const float* sum_x = nullptr, const float* sum_scale_ptr = nullptr, float sum_scale = 1.f, float* ysum = nullptr,
                             bool* summed = nullptr) {
    if (summed) *summed = false;
    if (!h || !x || !g_pw || !v_pw || !g_dw || !v_dw || !y || B < 1 || Tin < 1) return tfail(WV_EINVAL, "null / bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = fold_step(h, g_pw, v_pw, g_dw, v_dw, s);
    if (rc) return rc;
    wv::PwDwArgs a{};
    a.X = x; a.pw = pack_of(h, false); a.dw_w = h->w_dw; a.dw_b = bias; a.Y = y;
    a.B = B; a.Tin = Tin; a.Tout = t_out(h, Tin); a.ks = h->ks; a.stride = h->stride; a.dil = 1; a.pad = h->pad;
    a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.out_scale = 1.f; a.bands = 1; a.film_stride = 2;
    if (h_out) {
        if (h->stride != 1) return tfail(WV_EINVAL, "saved 1x1 output: stride-1 units only");
        wv::PwDwArgs f = a;
        f.Yraw = h_out;
        if (ysum && sum_x) { f.resid = sum_x; f.Ysum = ysum; f.scale_ptr = sum_scale_ptr; f.out_scale = sum_scale; }
        const hipError_t e = wv::launch_pw_dw(f, s);
        if (e == hipSuccess) { if (summed && ysum && sum_x) *summed = true; return WV_OK; }
        if (e != hipErrorNotSupported) T_LAUNCH(e);
        wv::PwDwArgs r = a;                                   // ragged / narrow layers: H by the identity stencil, as the backward used to
        r.dw_w = h->id_taps; r.dw_b = nullptr; r.Y = h_out; r.Tout = Tin; r.ks = 5; r.stride = 1; r.pad = 4;
        T_LAUNCH(wv::launch_pw_dw(r, s));
    }
    T_LAUNCH(wv::launch_pw_dw(a, s));
    return WV_OK;
}

int wv_train_unit_forward(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                          const float* v_dw, const float* bias, float pre_scale, int pre_elu, float* y, int B, int Tin, void* stream) {
    return unit_forward_impl(h, x, g_pw, v_pw, g_dw, v_dw, bias, pre_scale, pre_elu, y, nullptr, B, Tin, stream);
}

int wv_train_half_forward(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                          const float* v_dw, const float* bias, float pre_scale, float* y, int B, int T, void* stream) {
    return wv_train_unit_forward(h, x, g_pw, v_pw, g_dw, v_dw, bias, pre_scale, 1, y, B, T, stream);
}

// h_saved (optional): the forward's 1x1 output (unit_forward_impl), else it is recomputed here.  dx_add (optional, with pre_elu): a
// tensor added to dx (the ResnetBlock's identity shortcut); *added reports whether the dx kernel's epilogue took it.
static int unit_backward_impl(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                              const float* v_dw, float pre_scale, int pre_elu, const float* dy, float* dx, float* dg_pw, float* dv_pw,
                              float* dg_dw, float* dv_dw, float* db, int B, int Tin, void* ws, size_t ws_bytes, void* stream,
                              const float* h_saved, const float* dx_add, bool* added, const float* dy_scale_ptr = nullptr, float dy_scale = 1.f,
                              const float* dot_v = nullptr, float* dot_partial = nullptr);

int wv_train_unit_backward(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                           const float* v_dw, float pre_scale, int pre_elu, const float* dy, float* dx, float* dg_pw, float* dv_pw,
                           float* dg_dw, float* dv_dw, float* db, int B, int Tin, void* ws, size_t ws_bytes, void* stream) {
    return unit_backward_impl(h, x, g_pw, v_pw, g_dw, v_dw, pre_scale, pre_elu, dy, dx, dg_pw, dv_pw, dg_dw, dv_dw, db, B, Tin, ws, ws_bytes, stream,
                              nullptr, nullptr, nullptr);
}

static int unit_backward_impl(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                              const float* v_dw, float pre_scale, int pre_elu, const float* dy, float* dx, float* dg_pw, float* dv_pw,
                              float* dg_dw, float* dv_dw, float* db, int B, int Tin, void* ws, size_t ws_bytes, void* stream,
                              const float* h_saved, const float* dx_add, bool* added, const float* dy_scale_ptr, float dy_scale,
                              const float* dot_v, float* dot_partial) {
    if (added) *added = false;
    if (!h || !x || !g_pw || !v_pw || !g_dw || !v_dw || !dy || !dg_pw || !dv_pw || !dg_dw || !dv_dw || !db)
        return tfail(WV_EINVAL, "null argument");
    if (B < 1 || Tin < 1 || !ws || ws_bytes < wv_train_unit_workspace_bytes(h, B, Tin)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int M = h->M, K = h->K, ks = h->ks, Tout = t_out(h, Tin);
    const size_t am = al256((size_t)B * M * Tin * 4), ak = al256((size_t)B * K * Tin * 4);
    char* w = (char*)ws;
    float* Hws = (float*)w; float* DH = (float*)(w + am); float* DA = (float*)(w + 2 * am);
    const float* H = h_saved ? h_saved : Hws;
    float* partial = (float*)(w + 2 * am + ak);
    float* parts = (float*)(w + 2 * am + ak + al256((size_t)B * M * (ks + 1) * 4));
    // the step's weights: folded here unless this call continues a forward that kept its 1x1 output (the block's forward folded the
    // same parameters into the same buffers moments ago)
    // A backward on saved activations reuses the folds (W, its packs, 1 / ||v||) its forward left in the handle: it must follow THAT forward,
    // with the same parameter tensors and no parameter update or other forward on this handle in between.  Other tensors are caught here.
    if (h_saved && (h->folded[0] != g_pw || h->folded[1] != v_pw || h->folded[2] != g_dw || h->folded[3] != v_dw))
        return tfail(WV_ESTATE, "backward on saved activations: the parameters are not the ones this handle's last forward folded");
    int rc = h_saved ? WV_OK : fold_step(h, g_pw, v_pw, g_dw, v_dw, s);
    if (rc) return rc;
    if (!h_saved) {
        // h = W @ act(s x), recomputed (this forward kept no activations): K1 with the identity stencil
        wv::PwDwArgs a{};
        a.X = x; a.pw = pack_of(h, false); a.dw_w = h->id_taps; a.dw_b = nullptr; a.Y = Hws;
        a.B = B; a.Tin = Tin; a.Tout = Tin; a.ks = 5; a.stride = 1; a.dil = 1; a.pad = 4;
        a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.out_scale = 1.f; a.bands = 1; a.film_stride = 2;
        T_LAUNCH(wv::launch_pw_dw(a, s));
    }
    // dh, and the per-clip partial sums of the tap / bias gradients
    wv::launch_dw_bwd(s, dy, H, h->w_dw, DH, partial, M, B, Tin, Tout, ks, h->stride, h->pad, 0, dy_scale_ptr, dy_scale, dot_v, dot_partial);
    hipLaunchKernelGGL(wv::dw_param_grads_kernel, dim3(M), dim3(256), 0, s, partial, g_dw, v_dw, h->inv_dw, dg_dw, dv_dw, db, B, M, ks);
    T_LAUNCH(hipGetLastError());
    if (dx) {
        // da = W^T @ dh on the forward's GEMM kernel, then through the activation
        wv::PwDwArgs t{};
        t.X = DH; t.pw = pack_of(h, true); t.dw_w = h->id_taps; t.dw_b = nullptr; t.Y = pre_elu ? DA : dx;
        t.B = B; t.Tin = Tin; t.Tout = Tin; t.ks = 5; t.stride = 1; t.dil = 1; t.pad = 4;
        t.pre_scale = pre_elu ? 1.f : pre_scale; t.pre_elu = 0; t.out_scale = 1.f; t.bands = 1; t.film_stride = 2;   // no ELU: dx = s W^T dh
        bool fused = false;
        if (pre_elu) {
            // the activation's derivative in the GEMM's epilogue (x rides in as the residual operand): no da round trip through HBM
            wv::PwDwArgs f = t;
            f.Y = dx; f.resid = x; f.res_mode = 2; f.out_scale = pre_scale; f.resid2 = dx_add;
            const hipError_t e = wv::launch_pw_dw(f, s);
            if (e == hipSuccess) { fused = true; if (added && dx_add) *added = true; }
            else if (e != hipErrorNotSupported) T_LAUNCH(e);
        }
        if (!fused) {
            T_LAUNCH(wv::launch_pw_dw(t, s));
            if (pre_elu) {
                const size_t n = (size_t)B * K * Tin, n4 = n / 4;
                if (n4) hipLaunchKernelGGL(wv::elu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, DA, x, dx, pre_scale, n4);
                if (n % 4) hipLaunchKernelGGL(wv::elu_bwd_tail_kernel, dim3(1), dim3(256), 0, s, DA, x, dx, pre_scale, n4 * 4, n);
            }
        }
    }
    // dW = sum dh a^T, then the weight-norm backward
    const NtPlan np_ = nt_plan(B, Tin, M, K);
    const int S = np_.S;
    T_LAUNCH(wv::launch_gemm_nt(s, DH, x, parts, pre_scale, pre_elu, B, M, K, Tin, S, np_.TC));
    wv::launch_sum_parts(s, parts, h->dW, S, (size_t)M * K);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(M), dim3(256), 0, s, g_pw, v_pw, h->inv_pw, h->dW, dg_pw, dv_pw, K);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_half_backward(wv_train_unit* h, const float* x, const float* g_pw, const float* v_pw, const float* g_dw,
                           const float* v_dw, float pre_scale, const float* dy, float* dx, float* dg_pw, float* dv_pw,
                           float* dg_dw, float* dv_dw, float* db, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!dx) return tfail(WV_EINVAL, "null argument");
    return wv_train_unit_backward(h, x, g_pw, v_pw, g_dw, v_dw, pre_scale, 1, dy, dx, dg_pw, dv_pw, dg_dw, dv_dw, db, B, T, ws, ws_bytes, stream);
}

// ---- whole SEANetResnetBlock: y = x + s * half2(half1(pre_scale * x)) (seanet.py:245-281) ------------------------------
struct wv_train_block {
    wv_train_unit* h[2] = {nullptr, nullptr};
    float* partial = nullptr;
    float* tab = nullptr;                                        // [2][C][8]: the one-launch forward kernel's stencil tables (rebuilt every step)
    ~wv_train_block() { delete h[0]; delete h[1]; if (partial) (void)hipFree(partial); if (tab) (void)hipFree(tab); }
};

int wv_train_block_create(int C, wv_train_block** out) {
    if (!out) return tfail(WV_EINVAL, "null argument");
    auto* b = new wv_train_block();
    int rc = wv_train_half_create(C, &b->h[0]);
    if (!rc) rc = wv_train_half_create(C, &b->h[1]);
    if (!rc && hipMalloc((void**)&b->partial, wv::RED_BLOCKS * sizeof(float)) != hipSuccess) rc = tfail(WV_EHIP, "device allocation failed");
    if (!rc && hipMalloc((void**)&b->tab, (size_t)2 * C * 8 * sizeof(float)) != hipSuccess) rc = tfail(WV_EHIP, "device allocation failed");
    if (rc) { delete b; return rc; }
    *out = b;
    return WV_OK;
}

void wv_train_block_destroy(wv_train_block* b) { delete b; }

size_t wv_train_block_saved_bytes(const wv_train_block* b, int B, int T) {
    return (b && B > 0 && T > 0) ? 4 * al256((size_t)B * b->h[0]->M * T * 4) : 0;       // u, v and the two 1x1 outputs
}

size_t wv_train_block_workspace_bytes(const wv_train_block* b, int B, int T) {
    if (!b || B < 1 || T < 1) return 0;
    return wv_train_half_workspace_bytes(b->h[0], B, T) + 2 * al256((size_t)B * b->h[0]->M * T * 4);
}

int wv_train_block_forward(wv_train_block* b, const float* x, const wv_half_params* p, const float* res_scale_param,
                           float pre_scale, float res_scale, float* y, void* saved, size_t saved_bytes, int B, int T, void* stream) {
    if (!b || !x || !p || !y || !saved) return tfail(WV_EINVAL, "null argument");
    if (B < 1 || T < 1 || (T & 3) || saved_bytes < wv_train_block_saved_bytes(b, B, T)) return tfail(WV_ENOMEM, "saved-activation buffer too small (or T % 4 != 0)");
    hipStream_t s = (hipStream_t)stream;
    const size_t act = al256((size_t)B * b->h[0]->M * T * 4);
    float* u = (float*)saved; float* v = (float*)((char*)saved + act);
    float* H0 = (float*)((char*)saved + 2 * act); float* H1 = (float*)((char*)saved + 3 * act);
    // The narrow layers (C = 64, 96, 128, 192, k = 5): the whole block in ONE launch of the inference kernel (wv_rb.hip) in its training
    // form -- x read once, u kept in LDS for the second GEMM, and the four saved tensors written from the registers they are computed
    // in (6 HBM passes instead of the two launches' 8); same accumulation order, stencil and ELU as the two K1 launches below.
    if (b->h[0]->ks == 5 && b->h[1]->ks == 5) {
        const int C = b->h[0]->M;
        wv::RbArgs f{};
        f.X = x; f.pre_scale = pre_scale; f.pw1 = pack_of(b->h[0], false); f.pw2 = pack_of(b->h[1], false);
        f.tab1 = b->tab; f.tab2 = b->tab + (size_t)C * 8;
        f.Y = y; f.Yact = nullptr; f.out_scale = res_scale; f.out_scale_ptr = res_scale_param; f.act_scale = 0.f; f.B = B; f.C = C; f.T = T;
        f.sv_h0 = H0; f.sv_u = u; f.sv_h1 = H1; f.sv_v = v;
        if (wv::rb_supported(f)) {
            int rc = fold_step(b->h[0], p[0].g_pw, p[0].v_pw, p[0].g_dw, p[0].v_dw, s);
            if (!rc) rc = fold_step(b->h[1], p[1].g_pw, p[1].v_pw, p[1].g_dw, p[1].v_dw, s);
            if (rc) return rc;
            hipLaunchKernelGGL(wv::rb_table_pair_kernel, dim3((2 * C * 8 + 255) / 256), dim3(256), 0, s, b->h[0]->w_dw, p[0].bias, b->tab,
                               b->h[1]->w_dw, p[1].bias, b->tab + (size_t)C * 8, C);
            T_LAUNCH(hipGetLastError());
            T_LAUNCH(wv::launch_resblock(f, s));
            return WV_OK;
        }
    }
    int rc = unit_forward_impl(b->h[0], x, p[0].g_pw, p[0].v_pw, p[0].g_dw, p[0].v_dw, p[0].bias, pre_scale, 1, u, H0, B, T, stream);
    bool summed = false;                                       // y = x + s v from the second half's own epilogue where the LDS-DMA core runs it
    if (!rc) rc = unit_forward_impl(b->h[1], u, p[1].g_pw, p[1].v_pw, p[1].g_dw, p[1].v_dw, p[1].bias, 1.f, 1, v, H1, B, T, stream,
                                    x, res_scale_param, res_scale, y, &summed);
    if (rc) return rc;
    const size_t n4 = (size_t)B * b->h[0]->M * T / 4;
    if (!summed)
        hipLaunchKernelGGL(wv::axpy_res_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float4*)x, (const float4*)v, (float4*)y,
                           res_scale_param, res_scale, n4);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_block_backward(wv_train_block* b, const float* x, const wv_half_params* p, const float* res_scale_param,
                            float pre_scale, float res_scale, const float* dy, const void* saved, float* dx, const wv_half_grads* g,
                            float* d_res_scale_param, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!b || !x || !p || !dy || !saved || !dx || !g) return tfail(WV_EINVAL, "null argument");
    if (B < 1 || T < 1 || (T & 3) || !ws || ws_bytes < wv_train_block_workspace_bytes(b, B, T)) return tfail(WV_ENOMEM, "workspace too small (or T % 4 != 0)");
    if (res_scale_param && !d_res_scale_param) return tfail(WV_EINVAL, "res_scale_param without a gradient slot");
    hipStream_t s = (hipStream_t)stream;
    const int C = b->h[0]->M;
    const size_t act = al256((size_t)B * C * T * 4), n4 = (size_t)B * C * T / 4;
    const float* u = (const float*)saved; const float* v = (const float*)((const char*)saved + act);
    const float* H0 = (const float*)((const char*)saved + 2 * act); const float* H1 = (const float*)((const char*)saved + 3 * act);
    float* DV = (float*)ws; float* DU = (float*)((char*)ws + act);
    void* hws = (char*)ws + 2 * act;
    const size_t hws_bytes = ws_bytes - 2 * act;
    // dv = s * dy, d(res_scale_param) = res_scale * sum(dy * v): inside the second half's stencil backward when the operands are 16-byte
    // aligned (dy is scaled on the way in, the dot leaves one partial per (clip, channel) row in the otherwise unused DV region), else
    // by a pass of their own
    const bool fuse = wv::dw_bwd_can_fuse_scale(dy, H1, (const float*)hws, v, T);
    int rc;
    if (fuse) {
        rc = unit_backward_impl(b->h[1], u, p[1].g_pw, p[1].v_pw, p[1].g_dw, p[1].v_dw, 1.f, 1, dy, DU, g[1].dg_pw, g[1].dv_pw,
                                g[1].dg_dw, g[1].dv_dw, g[1].db, B, T, hws, hws_bytes, stream, H1, nullptr, nullptr, res_scale_param, res_scale, v, DV);
        if (!rc && d_res_scale_param) {
            hipLaunchKernelGGL(wv::finish_sum_kernel, dim3(1), dim3(wv::FIN_T), 0, s, DV, B * C, res_scale, d_res_scale_param);
            T_LAUNCH(hipGetLastError());
        }
    } else {
        hipLaunchKernelGGL(wv::scale_dot_kernel, dim3(wv::RED_BLOCKS), dim3(256), 0, s, (const float4*)dy, (const float4*)v, (float4*)DV,
                           res_scale_param, res_scale, b->partial, n4);
        if (d_res_scale_param) hipLaunchKernelGGL(wv::finish_sum_kernel, dim3(1), dim3(wv::FIN_T), 0, s, b->partial, wv::RED_BLOCKS, res_scale, d_res_scale_param);
        T_LAUNCH(hipGetLastError());
        rc = unit_backward_impl(b->h[1], u, p[1].g_pw, p[1].v_pw, p[1].g_dw, p[1].v_dw, 1.f, 1, DV, DU, g[1].dg_pw, g[1].dv_pw,
                                g[1].dg_dw, g[1].dv_dw, g[1].db, B, T, hws, hws_bytes, stream, H1, nullptr, nullptr);
    }
    bool added = false;
    if (!rc) rc = unit_backward_impl(b->h[0], x, p[0].g_pw, p[0].v_pw, p[0].g_dw, p[0].v_dw, pre_scale, 1, DU, dx, g[0].dg_pw, g[0].dv_pw,
                                     g[0].dg_dw, g[0].dv_dw, g[0].db, B, T, hws, hws_bytes, stream, H0, dy, &added);      // + the identity shortcut
    if (rc) return rc;
    if (!added) hipLaunchKernelGGL(wv::add_inplace_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (float4*)dx, (const float4*)dy, n4);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- BCE losses -----------------------------------------------------------------------------------------------------------
size_t wv_train_bce_workspace_bytes(void) { return wv::RED_BLOCKS * sizeof(float); }

int wv_train_bce_logits(const float* logits, const float* mask, const float* msg, float* loss, float* dlogits, float grad_scale,
                        int B, int Cz, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!logits || !loss || B < 1 || Cz < 1 || T < 1) return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < wv_train_bce_workspace_bytes()) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * Cz * T;
    hipLaunchKernelGGL(wv::bce_kernel, dim3(wv::RED_BLOCKS), dim3(256), 0, s, logits, mask, msg, dlogits, (float*)ws,
                       grad_scale / (float)n, Cz, T, n);
    hipLaunchKernelGGL(wv::finish_sum_kernel, dim3(1), dim3(wv::FIN_T), 0, s, (const float*)ws, wv::RED_BLOCKS, 1.f / (float)n, loss);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- conv_pre: y = conv1d(in_scale * x[B,1,T], W(g,v)[C,1,ks]) + b, causal (seanet.py:657-664) ------------------------------
struct wv_train_convpre {
    int C = 0, ks = 0;
    float *w = nullptr, *inv = nullptr, *dwdb = nullptr, *taps = nullptr;
    std::vector<void*> owned;
    ~wv_train_convpre() { for (void* p : owned) (void)hipFree(p); }
};

int wv_train_convpre_create(int C, int ks, wv_train_convpre** out) {
    if (!out || C < 1 || C > 4096 || ks < 1 || ks > wv::TRAIN_MAX_KS) return tfail(WV_EINVAL, "bad channel count / kernel size");
    auto* h = new wv_train_convpre();
    h->C = C; h->ks = ks;
    auto alloc = [&](float** p, size_t n) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return true;
    };
    if (!(alloc(&h->w, (size_t)C * ks) && alloc(&h->inv, C) && alloc(&h->dwdb, (size_t)C * (ks + 1)) && alloc(&h->taps, (size_t)C * ks))) {
        delete h;
        return tfail(WV_EHIP, "device allocation failed");
    }
    *out = h;
    return WV_OK;
}
void wv_train_convpre_destroy(wv_train_convpre* h) { delete h; }
size_t wv_train_convpre_workspace_bytes(const wv_train_convpre* h, int B, int T) {
    return (h && B > 0 && T > 0) ? al256((size_t)B * h->C * (h->ks + 1) * 4) : 0;
}

int wv_train_convpre_forward(wv_train_convpre* h, const float* x, const float* g, const float* v, const float* bias, float in_scale,
                             float* y, int B, int T, void* stream) {
    if (!h || !x || !g || !v || !y || B < 1 || T < 1) return tfail(WV_EINVAL, "null / bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(h->C), dim3(256), 0, s, g, v, h->w, h->inv, (float*)nullptr, (float*)nullptr, h->C, h->ks, 0, 0,
                       (const float*)nullptr, 1.f);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(wv::launch_conv_pre(x, h->w, bias, y, nullptr, 0.f, B, h->C, T, h->ks, in_scale, s));
    return WV_OK;
}

int wv_train_convpre_backward(wv_train_convpre* h, const float* x, const float* g, const float* v, float in_scale, const float* dy,
                              float* dx, float* dg, float* dv, float* db, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !x || !g || !v || !dy || !dg || !dv || !db) return tfail(WV_EINVAL, "null argument");
    if (B < 1 || T < 1 || !ws || ws_bytes < wv_train_convpre_workspace_bytes(h, B, T)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int C = h->C, ks = h->ks;
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(C), dim3(256), 0, s, g, v, h->w, h->inv, (float*)nullptr, (float*)nullptr, C, ks, 0, 0,
                       (const float*)nullptr, 1.f);
    // per-clip partial sums of dW[c][i] = sum_t dy[c][t] x[t - (ks-1) + i] and of db, then the fixed-order sum over clips
    wv::launch_dw_bwd(s, dy, x, h->w, (float*)nullptr, (float*)ws, C, B, T, T, ks, 1, ks - 1, 1);
    wv::launch_sum_parts(s, (const float*)ws, h->dwdb, B, (size_t)C * (ks + 1));
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((C + 255) / 256), dim3(256), 0, s, h->dwdb, h->taps, db, C, ks, in_scale);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(C), dim3(256), 0, s, g, v, h->inv, h->taps, dg, dv, ks);
    if (dx) hipLaunchKernelGGL(wv::convpre_dx_kernel, dim3((T + 255) / 256, B), dim3(256), (size_t)C * ks * 4, s, dy, h->w, dx, C, T, ks, in_scale);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- SpecBlock add: y = x + s * (W(g,v)[C,F] @ P[B,F,T]),  s = res_scale * scale_param[0] (seanet.py:463-511) ----------------
struct wv_train_spec {
    int C = 0, F = 0, Mp = 0;
    float *w = nullptr, *inv = nullptr, *wq = nullptr, *wt = nullptr, *wqT = nullptr, *wtT = nullptr, *dW = nullptr, *id_taps = nullptr;
    int KpT = 0;
    std::vector<void*> owned;
    ~wv_train_spec() { for (void* p : owned) (void)hipFree(p); }
};

int wv_train_spec_create(int C, int F, wv_train_spec** out) {
    if (!out || C < 1 || C > 4096 || F < 1 || F > 4096) return tfail(WV_EINVAL, "bad channel count");
    auto* h = new wv_train_spec();
    h->C = C; h->F = F; h->Mp = wv::round_up(C, wv::M_ALIGN); h->KpT = wv::round_up(F, wv::M_ALIGN);
    auto alloc = [&](float** p, size_t n, bool zero) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return !zero || hipMemset(*p, 0, n * sizeof(float)) == hipSuccess;
    };
    const int R = std::max(C, F);
    std::vector<float> taps((size_t)R * 5, 0.f);
    for (int m = 0; m < R; ++m) taps[(size_t)m * 5 + 4] = 1.f;
    bool ok = alloc(&h->wqT, (size_t)wv::round_up(C, 32) * h->KpT, true) && alloc(&h->wtT, (size_t)wv::round_up(C, wv::BK) * h->KpT, true) &&
              alloc(&h->w, (size_t)C * F, false) && alloc(&h->inv, C, false) && alloc(&h->wq, (size_t)wv::round_up(F, 32) * h->Mp, true) &&
              alloc(&h->wt, (size_t)wv::round_up(F, wv::BK) * h->Mp, true) && alloc(&h->dW, (size_t)C * F, false) && alloc(&h->id_taps, taps.size(), false) &&
              hipMemcpy(h->id_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { delete h; return tfail(WV_EHIP, "device allocation failed"); }
    *out = h;
    return WV_OK;
}
void wv_train_spec_destroy(wv_train_spec* h) { delete h; }
size_t wv_train_spec_workspace_bytes(const wv_train_spec* h, int B, int T) {
    return (h && B > 0 && T > 0) ? al256((size_t)nt_plan(B, T, h->C, h->F).S * h->C * h->F * 4) : 0;
}

int wv_train_spec_forward(wv_train_spec* h, const float* x, const float* P, const float* g, const float* v, const float* scale_param,
                          float res_scale, float* y, int B, int T, void* stream) {
    if (!h || !x || !P || !g || !v || !y || B < 1 || T < 1) return tfail(WV_EINVAL, "null / bad argument");
    hipStream_t s = (hipStream_t)stream;
    // the scalar s rides in the GEMM operand: wq = s * W
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(h->C), dim3(256), 0, s, g, v, h->w, h->inv, h->wq, (float*)nullptr, h->C, h->F, h->Mp, 0,
                       scale_param, res_scale, h->wt, (float*)nullptr);
    T_LAUNCH(hipGetLastError());
    wv::PwDwArgs a{};
    a.X = P; a.pw.M = h->C; a.pw.K = h->F; a.pw.Mp = h->Mp; a.pw.Kp = wv::round_up(h->F, wv::BK); a.pw.wq = h->wq; a.pw.wt = h->wt;
    a.dw_w = h->id_taps; a.dw_b = nullptr; a.resid = x; a.Y = y;
    a.B = B; a.Tin = T; a.Tout = T; a.ks = 5; a.stride = 1; a.dil = 1; a.pad = 4;
    a.pre_scale = 1.f; a.pre_elu = 0; a.out_scale = 1.f; a.bands = 1; a.film_stride = 2;
    T_LAUNCH(wv::launch_pw_dw(a, s));
    return WV_OK;
}

int wv_train_spec_backward(wv_train_spec* h, const float* P, const float* g, const float* v, const float* scale_param, float res_scale,
                           const float* dy, float* dg, float* dv, float* d_scale_param, float* dP, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !P || !g || !v || !dy || !dg || !dv) return tfail(WV_EINVAL, "null argument");
    if (scale_param && !d_scale_param) return tfail(WV_EINVAL, "scale_param without a gradient slot");
    if (B < 1 || T < 1 || !ws || ws_bytes < wv_train_spec_workspace_bytes(h, B, T)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int C = h->C, F = h->F;
    const NtPlan np_ = nt_plan(B, T, C, F);
    const int S = np_.S;
    const size_t n = (size_t)C * F;
    // plain W for the parameter gradients; (s W)^T packs when the gradient towards the features is wanted
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(C), dim3(256), 0, s, g, v, h->w, h->inv, (float*)nullptr, dP ? h->wqT : (float*)nullptr, C, F, 0, h->KpT,
                       scale_param, res_scale, (float*)nullptr, dP ? h->wtT : (float*)nullptr);
    if (dP) {                                                   // dP = (s W)^T @ dy
        T_LAUNCH(hipGetLastError());
        wv::PwDwArgs t{};
        t.X = dy; t.pw.M = F; t.pw.K = C; t.pw.Mp = h->KpT; t.pw.Kp = wv::round_up(C, wv::BK); t.pw.wq = h->wqT; t.pw.wt = h->wtT;
        t.dw_w = h->id_taps; t.dw_b = nullptr; t.Y = dP;
        t.B = B; t.Tin = T; t.Tout = T; t.ks = 5; t.stride = 1; t.dil = 1; t.pad = 4;
        t.pre_scale = 1.f; t.pre_elu = 0; t.out_scale = 1.f; t.bands = 1; t.film_stride = 2;
        T_LAUNCH(wv::launch_pw_dw(t, s));
    }
    // G = sum_{b,t} dy P^T;  d scale_param = res_scale * <W, G>  (= res_scale * sum dy . (W @ P));  dW = s * G
    T_LAUNCH(wv::launch_gemm_nt(s, dy, P, (float*)ws, 1.f, 0, B, C, F, T, S, np_.TC));
    wv::launch_sum_parts(s, (const float*)ws, h->dW, S, n);
    if (d_scale_param) {
        if (ws_bytes >= wv::RED_BLOCKS * sizeof(float)) {        // the split partials have been summed: their buffer takes the dot's partials
            hipLaunchKernelGGL(wv::dot_parts_kernel, dim3(wv::RED_BLOCKS), dim3(256), 0, s, h->w, h->dW, n, (float*)ws);
            hipLaunchKernelGGL(wv::finish_sum_kernel, dim3(1), dim3(wv::FIN_T), 0, s, (const float*)ws, wv::RED_BLOCKS, res_scale, d_scale_param);
        } else hipLaunchKernelGGL(wv::dot_small_kernel, dim3(1), dim3(256), 0, s, h->w, h->dW, n, res_scale, d_scale_param);
    }
    hipLaunchKernelGGL(wv::scale_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dW, n, scale_param, res_scale);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(C), dim3(256), 0, s, g, v, h->inv, h->dW, dg, dv, F);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- conv_post: ELU -> causal depth-wise conv (ks, no bias) -> 1x1 (C -> D, bias) -> L2Norm * sqrt(D) (seanet.py:795-822) --------
struct wv_train_convpost {
    int C = 0, D = 0, ks = 0, Mp = 0, KpT = 0;
    float *w_dw = nullptr, *inv_dw = nullptr, *w_pw = nullptr, *inv_pw = nullptr, *wt = nullptr, *wtT = nullptr, *wq = nullptr, *wqT = nullptr;
    float *dW = nullptr, *dwdb = nullptr, *taps = nullptr, *dbsum = nullptr, *junk = nullptr, *id_taps = nullptr;
    std::vector<void*> owned;
    ~wv_train_convpost() { for (void* p : owned) (void)hipFree(p); }
};

int wv_train_convpost_create(int C, int D, int ks, wv_train_convpost** out) {
    if (!out || C < 1 || C > 4096 || D < 1 || D > 128 || ks < 1 || ks > wv::TRAIN_MAX_KS) return tfail(WV_EINVAL, "bad channel count / kernel size (D <= 128)");
    auto* h = new wv_train_convpost();
    h->C = C; h->D = D; h->ks = ks; h->Mp = wv::round_up(D, wv::M_ALIGN); h->KpT = wv::round_up(C, wv::M_ALIGN);
    auto alloc = [&](float** p, size_t n, bool zero) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return !zero || hipMemset(*p, 0, n * sizeof(float)) == hipSuccess;
    };
    std::vector<float> taps((size_t)C * 5, 0.f);
    for (int m = 0; m < C; ++m) taps[(size_t)m * 5 + 4] = 1.f;
    bool ok = alloc(&h->w_dw, (size_t)C * ks, false) && alloc(&h->inv_dw, C, false) && alloc(&h->w_pw, (size_t)D * C, false) && alloc(&h->inv_pw, D, false) &&
              alloc(&h->wt, (size_t)wv::round_up(C, wv::BK) * h->Mp, true) && alloc(&h->wtT, (size_t)wv::round_up(D, wv::BK) * h->KpT, true) &&
              alloc(&h->wq, (size_t)wv::round_up(C, 32) * h->Mp, true) && alloc(&h->wqT, (size_t)wv::round_up(D, 32) * h->KpT, true) &&
              alloc(&h->dW, (size_t)D * C, false) && alloc(&h->dwdb, (size_t)std::max(C * (ks + 1), D * 2), false) && alloc(&h->taps, (size_t)C * ks, false) &&
              alloc(&h->dbsum, (size_t)D, false) && alloc(&h->junk, (size_t)std::max(C, D), false) && alloc(&h->id_taps, taps.size(), false) &&
              hipMemcpy(h->id_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { delete h; return tfail(WV_EHIP, "device allocation failed"); }
    *out = h;
    return WV_OK;
}
void wv_train_convpost_destroy(wv_train_convpost* h) { delete h; }

size_t wv_train_convpost_workspace_bytes(const wv_train_convpost* h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    const size_t ac = al256((size_t)B * h->C * T * 4), ad = al256((size_t)B * h->D * T * 4);
    return 3 * ac + ad + al256((size_t)B * std::max(h->C * (h->ks + 1), h->D * 2) * 4) + al256((size_t)nt_plan(B, T, h->D, h->C).S * h->D * h->C * 4);
}

static int convpost_fold(wv_train_convpost* h, const float* g_dw, const float* v_dw, const float* g_pw, const float* v_pw, hipStream_t s) {
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(h->C), dim3(256), 0, s, g_dw, v_dw, h->w_dw, h->inv_dw, (float*)nullptr, (float*)nullptr, h->C, h->ks, 0, 0,
                       (const float*)nullptr, 1.f, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(h->D), dim3(256), 0, s, g_pw, v_pw, h->w_pw, h->inv_pw, h->wq, h->wqT, h->D, h->C, h->Mp, h->KpT,
                       (const float*)nullptr, 1.f, h->wt, h->wtT);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

static wv::PwWeight convpost_pack(const wv_train_convpost* h, bool transposed) {
    wv::PwWeight p;
    if (!transposed) { p.M = h->D; p.K = h->C; p.Mp = h->Mp; p.Kp = wv::round_up(h->C, wv::BK); p.wq = h->wq; p.wt = h->wt; }
    else { p.M = h->C; p.K = h->D; p.Mp = h->KpT; p.Kp = wv::round_up(h->D, wv::BK); p.wq = h->wqT; p.wt = h->wtT; }
    return p;
}

int wv_train_convpost_forward(wv_train_convpost* h, const float* x, const float* g_dw, const float* v_dw, const float* g_pw, const float* v_pw,
                              const float* bias, int l2norm, float* y, int B, int T, void* stream) {
    if (!h || !x || !g_dw || !v_dw || !g_pw || !v_pw || !y || B < 1 || T < 1) return tfail(WV_EINVAL, "null / bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = convpost_fold(h, g_dw, v_dw, g_pw, v_pw, s);
    if (rc) return rc;
    wv::DwPwArgs a{};
    a.X = x; a.dw_w = h->w_dw; a.pw = convpost_pack(h, false); a.bias = bias; a.Y = y;
    a.B = B; a.Tin = T; a.Tout = T; a.mode = 1; a.ks = h->ks; a.pre_scale = 1.f; a.pre_elu = 1; a.l2norm = l2norm; a.out_scale = 1.f;
    T_LAUNCH(wv::launch_dw_pw(a, s));
    return WV_OK;
}

int wv_train_convpost_backward(wv_train_convpost* h, const float* x, const float* g_dw, const float* v_dw, const float* g_pw, const float* v_pw,
                               const float* bias, int l2norm, const float* dy, float* dx, float* dg_dw, float* dv_dw, float* dg_pw, float* dv_pw,
                               float* db, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !x || !g_dw || !v_dw || !g_pw || !v_pw || !dy || !dx || !dg_dw || !dv_dw || !dg_pw || !dv_pw || !db) return tfail(WV_EINVAL, "null argument");
    if (B < 1 || T < 1 || !ws || ws_bytes < wv_train_convpost_workspace_bytes(h, B, T)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int C = h->C, D = h->D, ks = h->ks;
    const NtPlan np_ = nt_plan(B, T, D, C);
    const int S = np_.S;
    const size_t ac = al256((size_t)B * C * T * 4), ad = al256((size_t)B * D * T * 4);
    char* w = (char*)ws;
    float* A = (float*)w; float* H = (float*)(w + ac); float* DH = (float*)(w + 2 * ac); float* Z = (float*)(w + 3 * ac);
    float* partial = (float*)(w + 3 * ac + ad);
    float* parts = (float*)(w + 3 * ac + ad + al256((size_t)B * std::max(C * (ks + 1), D * 2) * 4));
    int rc = convpost_fold(h, g_dw, v_dw, g_pw, v_pw, s);
    if (rc) return rc;
    // recompute a = ELU(x), h = DW(a), z = W h + b
    hipLaunchKernelGGL(wv::elu_dw_fwd_kernel, dim3(C, B), dim3(256), 0, s, x, h->w_dw, A, H, C, T, ks);
    T_LAUNCH(hipGetLastError());
    const float* dZ = dy;
    if (l2norm) {
        wv::DwPwArgs a{};
        a.X = H; a.pw = convpost_pack(h, false); a.bias = bias; a.Y = Z;
        a.B = B; a.Tin = T; a.Tout = T; a.mode = 0; a.ks = 1; a.pre_scale = 1.f; a.pre_elu = 0; a.l2norm = 0; a.out_scale = 1.f;
        T_LAUNCH(wv::launch_dw_pw(a, s));
        hipLaunchKernelGGL(wv::l2norm_bwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, Z, dy, D, T, 1e-12f);
        dZ = Z;
    }
    // db = sum dz (the bias slot of the row-sum kernel with one tap), dW = sum dz h^T
    wv::launch_dw_bwd(s, dZ, dZ, h->junk, (float*)nullptr, partial, D, B, T, T, 1, 1, 0, 0);
    wv::launch_sum_parts(s, partial, h->dwdb, B, (size_t)D * 2);
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((D + 255) / 256), dim3(256), 0, s, h->dwdb, h->junk, db, D, 1, 1.f);
    T_LAUNCH(wv::launch_gemm_nt(s, dZ, H, parts, 1.f, 0, B, D, C, T, S, np_.TC));
    wv::launch_sum_parts(s, parts, h->dW, S, (size_t)D * C);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(D), dim3(256), 0, s, g_pw, v_pw, h->inv_pw, h->dW, dg_pw, dv_pw, C);
    T_LAUNCH(hipGetLastError());
    // dh = W^T dz
    wv::PwDwArgs t{};
    t.X = dZ; t.pw = convpost_pack(h, true); t.dw_w = h->id_taps; t.dw_b = nullptr; t.Y = DH;
    t.B = B; t.Tin = T; t.Tout = T; t.ks = 5; t.stride = 1; t.dil = 1; t.pad = 4;
    t.pre_scale = 1.f; t.pre_elu = 0; t.out_scale = 1.f; t.bands = 1; t.film_stride = 2;
    T_LAUNCH(wv::launch_pw_dw(t, s));
    // through the depth-wise conv (da into the H buffer) and the ELU
    wv::launch_dw_bwd(s, DH, A, h->w_dw, H, partial, C, B, T, T, ks, 1, ks - 1, 0);
    wv::launch_sum_parts(s, partial, h->dwdb, B, (size_t)C * (ks + 1));
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((C + 255) / 256), dim3(256), 0, s, h->dwdb, h->taps, h->junk, C, ks, 1.f);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(C), dim3(256), 0, s, g_dw, v_dw, h->inv_dw, h->taps, dg_dw, dv_dw, ks);
    const size_t n = (size_t)B * C * T, n4 = n / 4;
    if (n4) hipLaunchKernelGGL(wv::elu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, H, x, dx, 1.f, n4);
    if (n % 4) hipLaunchKernelGGL(wv::elu_bwd_tail_kernel, dim3(1), dim3(256), 0, s, H, x, dx, 1.f, n4 * 4, n);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- head ------------------------------------------------------------------------------------------------------------------------
struct wv_train_head {
    int D = 0, O = 0, nb = 0, hop = 0;
    float *wt_q = nullptr, *wt_dz = nullptr, *wt_l = nullptr, *wt_lT = nullptr, *bias_q = nullptr, *scr = nullptr, *junk = nullptr;
    std::vector<void*> owned;
    ~wv_train_head() { for (void* p : owned) (void)hipFree(p); }
};

int wv_train_head_create(int D, int O, int nb, int hop, wv_train_head** out) {
    if (!out || D < 1 || D > 4096 || O < 1 || O > 4096 || nb < 1 || nb > 4096 || hop < 1 || (long long)O * hop > (1 << 20)) return tfail(WV_EINVAL, "bad head shape");
    auto* h = new wv_train_head();
    h->D = D; h->O = O; h->nb = nb; h->hop = hop;
    const int OH = O * hop;
    auto alloc = [&](float** p, size_t n) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return hipMemset(*p, 0, n * sizeof(float)) == hipSuccess;
    };
    const size_t big = std::max<size_t>((size_t)std::max(nb, O) * 2, 64);
    bool ok = alloc(&h->wt_q, (size_t)wv::round_up(D, wv::BK) * wv::round_up(OH, wv::M_ALIGN)) &&
              alloc(&h->wt_dz, (size_t)wv::round_up(OH, wv::BK) * wv::round_up(D, wv::M_ALIGN)) &&
              alloc(&h->wt_l, (size_t)wv::round_up(O, wv::BK) * wv::round_up(nb, wv::M_ALIGN)) &&
              alloc(&h->wt_lT, (size_t)wv::round_up(nb, wv::BK) * wv::round_up(O, wv::M_ALIGN)) && alloc(&h->bias_q, (size_t)OH) &&
              alloc(&h->scr, big) && alloc(&h->junk, big);
    if (!ok) { delete h; return tfail(WV_EHIP, "device allocation failed"); }
    *out = h;
    return WV_OK;
}
void wv_train_head_destroy(wv_train_head* h) { delete h; }


size_t wv_train_head_workspace_bytes(const wv_train_head* h, int B, int N) {
    if (!h || B < 1 || N < 1) return 0;
    const size_t hn = (size_t)h->hop * N;
    const size_t aq = al256((size_t)B * h->O * hn * 4), al = al256((size_t)B * h->nb * hn * 4);
    const size_t p1 = (size_t)nt_plan(B, N, h->D, h->O * h->hop).S * h->D * h->O * h->hop, p2 = (size_t)nt_plan(B, (int)hn, h->nb, h->O).S * h->nb * h->O;
    return 2 * aq + al + al256(std::max(p1, p2) * 4) + al256((size_t)B * std::max(h->nb, h->O) * 2 * 4);
}

static wv::PwWeight head_pw(int M, int K, const float* wt) {
    wv::PwWeight p; p.M = M; p.K = K; p.Mp = wv::round_up(M, wv::M_ALIGN); p.Kp = wv::round_up(K, wv::BK); p.wt = wt; p.wq = nullptr;
    return p;
}
static hipError_t head_gemm(const float* X, const wv::PwWeight& pw, const float* bias, float* Y, int B, int T, hipStream_t s) {
    wv::DwPwArgs a{};
    a.X = X; a.pw = pw; a.bias = bias; a.Y = Y; a.B = B; a.Tin = T; a.Tout = T; a.mode = 0; a.ks = 1;
    a.pre_scale = 1.f; a.pre_elu = 0; a.l2norm = 0; a.out_scale = 1.f;
    return wv::launch_dw_pw(a, s);
}
// q[B][O*hop][N] = ConvTranspose frames of z
static int head_q(wv_train_head* h, const float* z, const float* w_rev, const float* b_rev, float* q, int B, int N, hipStream_t s) {
    const int OH = h->O * h->hop;
    hipLaunchKernelGGL(wv::pack_wt_kernel, dim3((unsigned)(((size_t)OH * h->D + 255) / 256)), dim3(256), 0, s, w_rev, h->wt_q, OH, h->D, wv::round_up(OH, wv::M_ALIGN), 1);
    hipLaunchKernelGGL(wv::expand_bias_kernel, dim3((OH + 255) / 256), dim3(256), 0, s, b_rev, h->bias_q, h->O, h->hop);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(head_gemm(z, head_pw(OH, h->D, h->wt_q), b_rev ? h->bias_q : nullptr, q, B, N, s));
    return WV_OK;
}

int wv_train_head_forward(wv_train_head* h, const float* z, const float* w_rev, const float* b_rev, const float* w_last, const float* b_last,
                          float* logits, int B, int N, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !z || !w_rev || !w_last || !logits || B < 1 || N < 1 || T < 1 || T > N * h->hop) return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < wv_train_head_workspace_bytes(h, B, N)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t hn = (size_t)h->hop * N, aq = al256((size_t)B * h->O * hn * 4);
    float* q = (float*)ws; float* lq = (float*)((char*)ws + 2 * aq);
    int rc = head_q(h, z, w_rev, b_rev, q, B, N, s);
    if (rc) return rc;
    hipLaunchKernelGGL(wv::pack_wt_kernel, dim3((unsigned)(((size_t)h->nb * h->O + 255) / 256)), dim3(256), 0, s, w_last, h->wt_l, h->nb, h->O, wv::round_up(h->nb, wv::M_ALIGN), 0);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(head_gemm(q, head_pw(h->nb, h->O, h->wt_l), b_last, lq, B, (int)hn, s));
    hipLaunchKernelGGL(wv::frames_to_time_kernel, dim3((T + 255) / 256, B * h->nb), dim3(256), 0, s, lq, logits, T, h->hop, N);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_head_backward(wv_train_head* h, const float* z, const float* w_rev, const float* b_rev, const float* w_last, const float* dlogits,
                           float* dz, float* dw_rev, float* db_rev, float* dw_last, float* db_last, int B, int N, int T,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!h || !z || !w_rev || !w_last || !dlogits || !dz || !dw_rev || !db_rev || !dw_last || !db_last || B < 1 || N < 1 || T < 1 || T > N * h->hop)
        return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < wv_train_head_workspace_bytes(h, B, N)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->D, O = h->O, nb = h->nb, hop = h->hop, OH = O * hop;
    const size_t hn = (size_t)hop * N, aq = al256((size_t)B * O * hn * 4), al = al256((size_t)B * nb * hn * 4);
    const NtPlan np1 = nt_plan(B, N, D, OH), np2 = nt_plan(B, (int)hn, nb, O);
    const size_t p1 = (size_t)np1.S * D * OH, p2 = (size_t)np2.S * nb * O;
    char* w = (char*)ws;
    float* q = (float*)w; float* dq = (float*)(w + aq); float* dlq = (float*)(w + 2 * aq);
    float* parts = (float*)(w + 2 * aq + al);
    float* partial = (float*)(w + 2 * aq + al + al256(std::max(p1, p2) * 4));
    int rc = head_q(h, z, w_rev, b_rev, q, B, N, s);                               // recomputed (forward keeps nothing)
    if (rc) return rc;
    hipLaunchKernelGGL(wv::time_to_frames_kernel, dim3((unsigned)((hn + 255) / 256), B * nb), dim3(256), 0, s, dlogits, dlq, T, hop, N);
    // last layer: dw_last[k][o] = sum dlq[k] . q[o] over (b, j, n);  db_last[k] = sum dlogits
    const int S2 = np2.S;
    T_LAUNCH(wv::launch_gemm_nt(s, dlq, q, parts, 1.f, 0, B, nb, O, (int)hn, S2, np2.TC));
    wv::launch_sum_parts(s, parts, dw_last, S2, (size_t)nb * O);
    wv::launch_dw_bwd(s, dlq, dlq, h->junk, (float*)nullptr, partial, nb, B, (int)hn, (int)hn, 1, 1, 0, 0);
    wv::launch_sum_parts(s, partial, h->scr, B, (size_t)nb * 2);
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, h->scr, h->junk, db_last, nb, 1, 1.f);
    // dq = w_last^T @ dlq
    hipLaunchKernelGGL(wv::pack_wt_kernel, dim3((unsigned)(((size_t)O * nb + 255) / 256)), dim3(256), 0, s, w_last, h->wt_lT, O, nb, wv::round_up(O, wv::M_ALIGN), 1);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(head_gemm(dlq, head_pw(O, nb, h->wt_lT), nullptr, dq, B, (int)hn, s));
    // db_rev[o] = sum dq[o];  dw_rev[d][(o,j)] = sum_{b,n} z[d][n] dq[(o,j)][n];  dz = w_rev @ dq
    wv::launch_dw_bwd(s, dq, dq, h->junk, (float*)nullptr, partial, O, B, (int)hn, (int)hn, 1, 1, 0, 0);
    wv::launch_sum_parts(s, partial, h->scr, B, (size_t)O * 2);
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((O + 255) / 256), dim3(256), 0, s, h->scr, h->junk, db_rev, O, 1, 1.f);
    const int S1 = np1.S;
    T_LAUNCH(wv::launch_gemm_nt(s, z, dq, parts, 1.f, 0, B, D, OH, N, S1, np1.TC));
    wv::launch_sum_parts(s, parts, dw_rev, S1, (size_t)D * OH);
    hipLaunchKernelGGL(wv::pack_wt_kernel, dim3((unsigned)(((size_t)D * OH + 255) / 256)), dim3(256), 0, s, w_rev, h->wt_dz, D, OH, wv::round_up(D, wv::M_ALIGN), 0);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(head_gemm(dq, head_pw(D, OH, h->wt_dz), nullptr, dz, B, N, s));
    return WV_OK;
}

// ---- upsample unit: y[B,M,r*Tin] = W(g,v)[M,K] @ ConvT_dw(act(s x[B,K,Tin]); taps(g,v)[K,2r]) + bias (seanet.py:1110-1135) -------
struct wv_train_up {
    int K = 0, M = 0, r = 0, Mp = 0, KpT = 0, Kp = 0;
    float *w_ct = nullptr, *inv_ct = nullptr, *ct_wt = nullptr, *w_pw = nullptr, *inv_pw = nullptr, *wq = nullptr, *wqT = nullptr, *wt = nullptr, *wtT = nullptr;
    float *id_taps = nullptr, *dW = nullptr, *taps = nullptr, *scr = nullptr, *junk = nullptr;
    std::vector<void*> owned;
    ~wv_train_up() { for (void* p : owned) (void)hipFree(p); }
};

int wv_train_up_create(int K, int M, int ratio, wv_train_up** out) {
    if (!out || K < 1 || K > 4096 || M < 1 || M > 4096 || ratio < 1 || 2 * ratio > wv::TRAIN_MAX_KS) return tfail(WV_EINVAL, "bad upsample shape (ratio <= 8)");
    auto* h = new wv_train_up();
    h->K = K; h->M = M; h->r = ratio; h->Mp = wv::round_up(M, wv::M_ALIGN); h->KpT = wv::round_up(K, wv::M_ALIGN); h->Kp = wv::round_up(K, wv::BK);
    auto alloc = [&](float** p, size_t n) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return hipMemset(*p, 0, n * sizeof(float)) == hipSuccess;
    };
    const int R = std::max(K, M), ks = 2 * ratio;
    std::vector<float> taps((size_t)R * 5, 0.f);
    for (int m = 0; m < R; ++m) taps[(size_t)m * 5 + 4] = 1.f;
    bool ok = alloc(&h->w_ct, (size_t)K * ks) && alloc(&h->inv_ct, K) && alloc(&h->ct_wt, (size_t)ks * h->Kp) && alloc(&h->w_pw, (size_t)M * K) &&
              alloc(&h->inv_pw, M) && alloc(&h->wq, (size_t)wv::round_up(K, 32) * h->Mp) && alloc(&h->wqT, (size_t)wv::round_up(M, 32) * h->KpT) &&
              alloc(&h->wt, (size_t)h->Kp * h->Mp) && alloc(&h->wtT, (size_t)wv::round_up(M, wv::BK) * h->KpT) && alloc(&h->id_taps, taps.size()) &&
              alloc(&h->dW, (size_t)M * K) && alloc(&h->taps, (size_t)K * ks) && alloc(&h->scr, (size_t)R * 2) && alloc(&h->junk, (size_t)R) &&
              hipMemcpy(h->id_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { delete h; return tfail(WV_EHIP, "device allocation failed"); }
    *out = h;
    return WV_OK;
}
void wv_train_up_destroy(wv_train_up* h) { delete h; }

size_t wv_train_up_workspace_bytes(const wv_train_up* h, int B, int Tin) {
    if (!h || B < 1 || Tin < 1) return 0;
    const int Tout = Tin * h->r;
    const size_t au = al256((size_t)B * h->K * Tout * 4);
    return 2 * au + al256((size_t)B * std::max(h->K * 2 * h->r, h->M * 2) * 4) + al256((size_t)nt_plan(B, Tout, h->M, h->K).S * h->M * h->K * 4);
}

static int up_fold(wv_train_up* h, const float* g_ct, const float* v_ct, const float* g_pw, const float* v_pw, hipStream_t s) {
    const int ks = 2 * h->r;
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(h->K), dim3(256), 0, s, g_ct, v_ct, h->w_ct, h->inv_ct, (float*)nullptr, (float*)nullptr, h->K, ks, 0, 0,
                       (const float*)nullptr, 1.f, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(wv::pack_ct_wt_kernel, dim3((h->K * ks + 255) / 256), dim3(256), 0, s, h->w_ct, h->ct_wt, h->K, h->Kp, ks);
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(h->M), dim3(256), 0, s, g_pw, v_pw, h->w_pw, h->inv_pw, h->wq, h->wqT, h->M, h->K, h->Mp, h->KpT,
                       (const float*)nullptr, 1.f, h->wt, h->wtT);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_up_forward(wv_train_up* h, const float* x, const float* g_ct, const float* v_ct, const float* g_pw, const float* v_pw, const float* bias,
                        float pre_scale, int pre_elu, float* y, int B, int Tin, void* stream) {
    if (!h || !x || !g_ct || !v_ct || !g_pw || !v_pw || !y || B < 1 || Tin < 1) return tfail(WV_EINVAL, "null / bad argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = up_fold(h, g_ct, v_ct, g_pw, v_pw, s);
    if (rc) return rc;
    wv::PwDwArgs a{};
    a.X = x; a.pw.M = h->M; a.pw.K = h->K; a.pw.Mp = h->Mp; a.pw.Kp = h->Kp; a.pw.wq = h->wq; a.pw.wt = h->wt;
    a.ct_w = h->w_ct; a.ct_wt = h->ct_wt; a.ratio = h->r; a.dw_w = h->id_taps; a.dw_b = bias; a.Y = y;
    a.B = B; a.Tin = Tin; a.Tout = Tin * h->r; a.ks = 5; a.stride = 1; a.dil = 1; a.pad = 4;
    a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.out_scale = 1.f; a.bands = 1; a.film_stride = 2;
    T_LAUNCH(wv::launch_pw_dw(a, s));
    return WV_OK;
}

int wv_train_up_backward(wv_train_up* h, const float* x, const float* g_ct, const float* v_ct, const float* g_pw, const float* v_pw, float pre_scale,
                         int pre_elu, const float* dy, float* dx, float* dg_ct, float* dv_ct, float* dg_pw, float* dv_pw, float* db, int B, int Tin,
                         void* ws, size_t ws_bytes, void* stream) {
    if (!h || !x || !g_ct || !v_ct || !g_pw || !v_pw || !dy || !dx || !dg_ct || !dv_ct || !dg_pw || !dv_pw || !db) return tfail(WV_EINVAL, "null argument");
    if (B < 1 || Tin < 1 || !ws || ws_bytes < wv_train_up_workspace_bytes(h, B, Tin)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int K = h->K, M = h->M, r = h->r, ks = 2 * r, Tout = Tin * r;
    const size_t au = al256((size_t)B * K * Tout * 4);
    char* w = (char*)ws;
    float* U = (float*)w; float* DU = (float*)(w + au);
    float* partial = (float*)(w + 2 * au);
    float* parts = (float*)(w + 2 * au + al256((size_t)B * std::max(K * ks, M * 2) * 4));
    int rc = up_fold(h, g_ct, v_ct, g_pw, v_pw, s);
    if (rc) return rc;
    {
        const bool al = (reinterpret_cast<uintptr_t>(U) & 15) == 0;
#define WV_CTF(R) hipLaunchKernelGGL((wv::convtr_fwd_frame_kernel<R>), dim3(K, B), dim3(256), 0, s, x, h->w_ct, U, K, Tin, pre_scale, pre_elu)
        if (r == 2) WV_CTF(2); else if (r == 4 && al) WV_CTF(4); else if (r == 5) WV_CTF(5); else if (r == 8 && al) WV_CTF(8);
        else hipLaunchKernelGGL(wv::convtr_fwd_kernel, dim3(K, B), dim3(256), 0, s, x, h->w_ct, U, K, Tin, r, pre_scale, pre_elu);
#undef WV_CTF
    }
    // db = sum dy;  dW = sum dy u^T
    wv::launch_dw_bwd(s, dy, dy, h->junk, (float*)nullptr, partial, M, B, Tout, Tout, 1, 1, 0, 0);
    wv::launch_sum_parts(s, partial, h->scr, B, (size_t)M * 2);
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((M + 255) / 256), dim3(256), 0, s, h->scr, h->junk, db, M, 1, 1.f);
    const NtPlan np_ = nt_plan(B, Tout, M, K);
    T_LAUNCH(wv::launch_gemm_nt(s, dy, U, parts, 1.f, 0, B, M, K, Tout, np_.S, np_.TC));
    wv::launch_sum_parts(s, parts, h->dW, np_.S, (size_t)M * K);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(M), dim3(256), 0, s, g_pw, v_pw, h->inv_pw, h->dW, dg_pw, dv_pw, K);
    T_LAUNCH(hipGetLastError());
    // du = W^T dy
    wv::PwDwArgs t{};
    t.X = dy; t.pw.M = K; t.pw.K = M; t.pw.Mp = h->KpT; t.pw.Kp = wv::round_up(M, wv::BK); t.pw.wq = h->wqT; t.pw.wt = h->wtT;
    t.dw_w = h->id_taps; t.dw_b = nullptr; t.Y = DU;
    t.B = B; t.Tin = Tout; t.Tout = Tout; t.ks = 5; t.stride = 1; t.dil = 1; t.pad = 4;
    t.pre_scale = 1.f; t.pre_elu = 0; t.out_scale = 1.f; t.bands = 1; t.film_stride = 2;
    T_LAUNCH(wv::launch_pw_dw(t, s));
    // through the transposed conv and the activation
    {
        const bool al = (reinterpret_cast<uintptr_t>(DU) & 15) == 0;
#define WV_CTB(R) hipLaunchKernelGGL((wv::convtr_bwd_frame_kernel<R>), dim3(K, B), dim3(256), 0, s, DU, x, h->w_ct, dx, partial, K, Tin, pre_scale, pre_elu)
        if (r == 2) WV_CTB(2); else if (r == 4 && al) WV_CTB(4); else if (r == 5) WV_CTB(5); else if (r == 8 && al) WV_CTB(8);
        else hipLaunchKernelGGL(wv::convtr_bwd_kernel, dim3(K, B), dim3(256), 0, s, DU, x, h->w_ct, dx, partial, K, Tin, r, pre_scale, pre_elu);
#undef WV_CTB
    }
    wv::launch_sum_parts(s, partial, h->taps, B, (size_t)K * ks);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(K), dim3(256), 0, s, g_ct, v_ct, h->inv_ct, h->taps, dg_ct, dv_ct, ks);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- decoder tail ------------------------------------------------------------------------------------------------------------------
struct wv_train_tail {
    int C = 0, ks = 0;
    float *w = nullptr, *inv = nullptr, *dwdb = nullptr, *taps = nullptr, *dbv = nullptr;
    std::vector<void*> owned;
    ~wv_train_tail() { for (void* p : owned) (void)hipFree(p); }
};

int wv_train_tail_create(int C, int ks, wv_train_tail** out) {
    if (!out || C < 1 || C > 4096 || ks < 1 || ks > wv::TRAIN_MAX_KS) return tfail(WV_EINVAL, "bad channel count / kernel size");
    auto* h = new wv_train_tail();
    h->C = C; h->ks = ks;
    auto alloc = [&](float** p, size_t n) {
        if (hipMalloc((void**)p, n * sizeof(float)) != hipSuccess) return false;
        h->owned.push_back(*p);
        return true;
    };
    if (!(alloc(&h->w, (size_t)C * ks) && alloc(&h->inv, 1) && alloc(&h->dwdb, (size_t)C * (ks + 1)) && alloc(&h->taps, (size_t)C * ks) && alloc(&h->dbv, C))) {
        delete h;
        return tfail(WV_EHIP, "device allocation failed");
    }
    *out = h;
    return WV_OK;
}
void wv_train_tail_destroy(wv_train_tail* h) { delete h; }
size_t wv_train_tail_workspace_bytes(const wv_train_tail* h, int B) { return (h && B > 0) ? al256((size_t)B * h->C * (h->ks + 1) * 4) : 0; }

int wv_train_tail_forward(wv_train_tail* h, const float* x, const float* g, const float* v, const float* bias, float post, float wav_std,
                          float* delta, int B, int Tin, int T, void* stream) {
    if (!h || !x || !g || !v || !delta || B < 1 || T < 1 || T > Tin) return tfail(WV_EINVAL, "null / bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(1), dim3(256), 0, s, g, v, h->w, h->inv, (float*)nullptr, (float*)nullptr, 1, h->C * h->ks, 0, 0,
                       (const float*)nullptr, 1.f, (float*)nullptr, (float*)nullptr);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(wv::launch_tail(x, h->w, bias, nullptr, delta, B, h->C, Tin, T, h->ks, post, wav_std, s));
    return WV_OK;
}

int wv_train_tail_backward(wv_train_tail* h, const float* x, const float* g, const float* v, float post, float wav_std, const float* delta,
                           const float* d_delta, float* dx, float* dg, float* dv, float* db, int B, int Tin, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !x || !g || !v || !delta || !d_delta || !dx || !dg || !dv || !db) return tfail(WV_EINVAL, "null argument");
    if (B < 1 || T < 1 || T > Tin || !ws || ws_bytes < wv_train_tail_workspace_bytes(h, B)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int C = h->C, ks = h->ks;
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3(1), dim3(256), 0, s, g, v, h->w, h->inv, (float*)nullptr, (float*)nullptr, 1, C * ks, 0, 0,
                       (const float*)nullptr, 1.f, (float*)nullptr, (float*)nullptr);
    if (ks == 5 && (T & 3) == 0 && (Tin & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(delta) | reinterpret_cast<uintptr_t>(d_delta) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
        hipLaunchKernelGGL(wv::tail_bwd5_vec_kernel, dim3(C, B), dim3(256), 0, s, x, h->w, delta, d_delta, dx, (float*)ws, C, Tin, T, post, wav_std);
    else
        hipLaunchKernelGGL(wv::tail_bwd_kernel, dim3(C, B), dim3(256), 0, s, x, h->w, delta, d_delta, dx, (float*)ws, C, Tin, T, ks, post, wav_std);
    wv::launch_sum_parts(s, (const float*)ws, h->dwdb, B, (size_t)C * (ks + 1));
    hipLaunchKernelGGL(wv::split_dwdb_kernel, dim3((C + 255) / 256), dim3(256), 0, s, h->dwdb, h->taps, h->dbv, C, ks, 1.f);
    hipLaunchKernelGGL(wv::wn_bwd_kernel, dim3(1), dim3(256), 0, s, g, v, h->inv, h->taps, dg, dv, C * ks);
    T_LAUNCH(hipGetLastError());
    T_LAUNCH(hipMemcpyAsync(db, h->dbv, sizeof(float), hipMemcpyDeviceToDevice, s));      // every channel row holds the same sum of dq
    return WV_OK;
}

// ---- message MLP + FiLM ------------------------------------------------------------------------------------------------------------
size_t wv_train_film_param_count(int msg_dim, int E, int layers, int n_scales, int bands) {
    if (msg_dim < 1 || E < 1 || layers < 0 || n_scales < 1 || bands < 1) return 0;
    return (size_t)E * msg_dim + E + (size_t)layers * ((size_t)E * E + E) + (size_t)n_scales * bands * 2 * (E + 1);
}
size_t wv_train_film_workspace_bytes(int B, int msg_dim, int E, int layers, int n_scales, int bands) {
    return al256((size_t)B * (layers + 1) * E * 4) + al256((size_t)B * wv_train_film_param_count(msg_dim, E, layers, n_scales, bands) * 4);
}

static bool film_shape_ok(int B, int Dm, int E, int L, int S, int bands) {
    return B >= 1 && Dm >= 1 && E >= 1 && E <= wv::FILM_MAX_E && L >= 0 && L <= wv::FILM_MAX_L && S >= 1 && bands >= 1;
}

int wv_train_film_forward(const float* msg, const float* params, float* film, int B, int msg_dim, int E, int layers, int n_scales, int bands,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!msg || !params || !film || !film_shape_ok(B, msg_dim, E, layers, n_scales, bands)) return tfail(WV_EINVAL, "null / bad argument (E <= 256, layers <= 4)");
    if (!ws || ws_bytes < wv_train_film_workspace_bytes(B, msg_dim, E, layers, n_scales, bands)) return tfail(WV_ENOMEM, "workspace too small");
    hipLaunchKernelGGL(wv::msg_film_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, msg, params, film, (float*)ws, msg_dim, E, layers, n_scales * bands * 2);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

/* `ws` must be the buffer forward used (it holds the MLP activations). */
int wv_train_film_backward(const float* msg, const float* params, const float* dfilm, float* dparams, int B, int msg_dim, int E, int layers,
                           int n_scales, int bands, void* ws, size_t ws_bytes, void* stream) {
    if (!msg || !params || !dfilm || !dparams || !film_shape_ok(B, msg_dim, E, layers, n_scales, bands)) return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < wv_train_film_workspace_bytes(B, msg_dim, E, layers, n_scales, bands)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t np = wv_train_film_param_count(msg_dim, E, layers, n_scales, bands);
    float* acts = (float*)ws;
    float* gpart = (float*)((char*)ws + al256((size_t)B * (layers + 1) * E * 4));
    hipLaunchKernelGGL(wv::msg_film_bwd_kernel, dim3(B), dim3(256), 0, s, msg, params, acts, dfilm, gpart, msg_dim, E, layers, n_scales * bands * 2, np);
    wv::launch_sum_parts(s, gpart, dparams, B, np);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_film_apply(const float* x, const float* film, float* y, int B, int C, int T, int bands, int n_scales, int scale, void* stream) {
    if (!x || !film || !y || B < 1 || C < 1 || T < 1 || bands < 1 || C % bands || scale < 0 || scale >= n_scales) return tfail(WV_EINVAL, "null / bad argument");
    hipLaunchKernelGGL(wv::film_apply_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, film, y, C, T, bands, n_scales * bands * 2, scale * bands * 2);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

/* dfilm: the [B][n_scales*bands*2] gradient block; this call fills the entries of `scale`.  ws: B*C*2 floats. */
int wv_train_film_apply_backward(const float* x, const float* film, const float* dy, float* dx, float* dfilm, int B, int C, int T, int bands,
                                 int n_scales, int scale, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !film || !dy || !dx || !dfilm || B < 1 || C < 1 || T < 1 || bands < 1 || C % bands || scale < 0 || scale >= n_scales)
        return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < (size_t)B * C * 2 * sizeof(float)) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int NF = n_scales * bands * 2, so = scale * bands * 2;
    hipLaunchKernelGGL(wv::film_apply_bwd_kernel, dim3(C, B), dim3(256), 0, s, x, film, dy, dx, (float*)ws, C, T, bands, NF, so);
    hipLaunchKernelGGL(wv::film_band_reduce_kernel, dim3((B * bands * 2 + 255) / 256), dim3(256), 0, s, (const float*)ws, dfilm, B, C, bands, NF, so);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_l1(const float* a, const float* b, float* loss, float* da, float grad_scale, size_t n, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !b || !loss || !n) return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < wv_train_bce_workspace_bytes()) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wv::l1_kernel, dim3(wv::RED_BLOCKS), dim3(256), 0, s, a, b, da, (float*)ws, grad_scale / (float)n, n);
    hipLaunchKernelGGL(wv::finish_sum_kernel, dim3(1), dim3(wv::FIN_T), 0, s, (const float*)ws, wv::RED_BLOCKS, 1.f / (float)n, loss);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

// ---- gradient norm + AdamW over flat arenas ---------------------------------------------------------------------------------
int wv_train_fold_weight(const float* g, const float* v, float* w, float* inv_norm, int M, int K, void* stream) {
    if (!g || !v || !w || !inv_norm || M < 1 || K < 1) return tfail(WV_EINVAL, "null / bad argument");
    hipLaunchKernelGGL(wv::wn_fold_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, g, v, w, inv_norm, (float*)nullptr,
                       (float*)nullptr, M, K, 0, 0, (const float*)nullptr, 1.f, (float*)nullptr, (float*)nullptr);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_sumsq(const float* g, size_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!g || !out || !n) return tfail(WV_EINVAL, "null / bad argument");
    if (!ws || ws_bytes < wv_train_bce_workspace_bytes()) return tfail(WV_ENOMEM, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wv::sumsq_kernel, dim3(wv::RED_BLOCKS), dim3(256), 0, s, g, n, (float*)ws);
    hipLaunchKernelGGL(wv::finish_sum_kernel, dim3(1), dim3(wv::FIN_T), 0, s, (const float*)ws, wv::RED_BLOCKS, 1.f, out);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

int wv_train_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* grad_sumsq, float max_norm, void* stream) {
    if (!p || !g || !m || !v || !n || step < 1 || !(lr >= 0.f) || beta1 < 0.f || beta1 >= 1.f || beta2 < 0.f || beta2 >= 1.f)
        return tfail(WV_EINVAL, "null / bad argument");
    const float bc1 = (float)(1.0 - std::pow((double)beta1, step));
    const float bc2s = (float)std::sqrt(1.0 - std::pow((double)beta2, step));
    hipLaunchKernelGGL(wv::adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2s, grad_sumsq, max_norm);
    T_LAUNCH(hipGetLastError());
    return WV_OK;
}

}  // extern "C"
