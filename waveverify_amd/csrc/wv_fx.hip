// Sinc-filter and resample effects on the device (the arithmetic is third-party in the reference:
// julius low / high / band-pass filters, torchaudio.transforms.Resample -- utils/effect_augmentation.py:1451-1501,1684-1870,
// waveverify/utils.py:213).  Both are FIR filter banks over a padded signal:
//     y[row][f][n] = sum_j taps[f][j] * xpad[n * stride + j],   xpad = x with pad_l / pad_r samples in front / behind
// (interleave = 1 stores y[row][n * n_filters + f] instead: the polyphase resampler's output order)
// (replicate padding for julius' filters, zero padding and stride = orig_freq / gcd with one filter per output phase for the polyphase
// resampler).  The host builds the taps the way the libraries publish them (waveverify_amd/effects.py); this file only convolves.
// One workgroup = one row x 256 outputs: the input window ((255 stride + L) samples, in 4096-sample pieces when long) is staged
// through LDS together with the matching piece of every filter.
#include <hip/hip_runtime.h>

#include "../../include/waveverify_hip.h"

namespace wv {

constexpr int FX_TILE = 256, FX_JC = 1024, FX_MAX_F = 8;

__global__ __launch_bounds__(256) void fir_bank_kernel(const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
                                                        int T, int Tout, int L, int nf, int stride, int pad_l, int replicate, int interleave) {
    extern __shared__ float sm[];                      // window piece [(FX_TILE - 1) * stride + FX_JC], then taps piece [nf][FX_JC]
    const int row = blockIdx.y, n0 = blockIdx.x * FX_TILE, tid = threadIdx.x;
    const int wlen = (FX_TILE - 1) * stride + FX_JC;
    float* win = sm;
    float* tp = sm + wlen;
    const float* xr = x + (size_t)row * T;
    float acc[FX_MAX_F];
#pragma unroll
    for (int f = 0; f < FX_MAX_F; ++f) acc[f] = 0.f;
    for (int j0 = 0; j0 < L; j0 += FX_JC) {
        const int jc = min(FX_JC, L - j0);
        const int need = (FX_TILE - 1) * stride + jc;
        __syncthreads();
        for (int i = tid; i < need; i += 256) {
            int s = n0 * stride + j0 + i - pad_l;              // index into the unpadded signal
            float v = 0.f;
            if (replicate) v = xr[min(max(s, 0), T - 1)];
            else if (s >= 0 && s < T) v = xr[s];
            win[i] = v;
        }
        for (int i = tid; i < nf * jc; i += 256) tp[(i / jc) * FX_JC + i % jc] = taps[(size_t)(i / jc) * L + j0 + i % jc];
        __syncthreads();
        const float* w = win + tid * stride;
        for (int j = 0; j < jc; ++j) {
            const float v = w[j];
#pragma unroll
            for (int f = 0; f < FX_MAX_F; ++f)
                if (f < nf) acc[f] = fmaf(tp[f * FX_JC + j], v, acc[f]);
        }
    }
    const int n = n0 + tid;
    if (n < Tout)
        for (int f = 0; f < nf; ++f) y[interleave ? ((size_t)row * Tout + n) * nf + f : ((size_t)row * nf + f) * Tout + n] = acc[f];
}

// Polyphase sinc resampler (torchaudio's formulation): output m = n * new + f reads input samples n * orig - width .. with phase f's filter:
//     y[row][m] = sum_j K[f][j] * xz[n * orig + j - width],  xz = x with zeros outside [0, T)
// One thread per output; the filter bank [new][L] stays in L2.  Any ratio (file loading: 44.1 kHz -> 16 kHz = 441 : 160).
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, const float* __restrict__ K, float* __restrict__ y,
                                                        int T, int Tout, int orig, int nw, int L, int width) {
    const int row = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Tout) return;
    const int n = m / nw, f = m - n * nw;
    const float* xr = x + (size_t)row * T;
    const float* kf = K + (size_t)f * L;
    const int s0 = n * orig - width;
    float acc = 0.f;
    for (int j = 0; j < L; ++j) {
        const int s = s0 + j;
        if (s >= 0 && s < T) acc = fmaf(kf[j], xr[s], acc);
    }
    y[(size_t)row * Tout + m] = acc;
}

// ---- adjoints (the gradient of a loss through these effects: in the reference they are plain differentiable torch ops, so the
// generator's gradient passes through the TRANSPOSED filter, not through an identity) --------------------------------------------
// Transpose of the replicate padding in front of a 'same' FIR: dxp [rows][T + pad_l + pad_r] is the gradient towards the padded signal;
// dx[t] = dxp[t + pad_l], and the pad samples (copies of x[0] / x[T-1]) send theirs to the two end samples.  One workgroup per row.
__global__ __launch_bounds__(256) void fold_replicate_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int T, int pad_l, int pad_r) {
    __shared__ float red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x, Tp = T + pad_l + pad_r;
    const float* src = dxp + (size_t)row * Tp;
    float* dst = dx + (size_t)row * T;
    for (int t = tid; t < T; t += 256) dst[t] = src[t + pad_l];
    float a = 0.f, b = 0.f;
    for (int i = tid; i < pad_l; i += 256) a += src[i];
    for (int i = tid; i < pad_r; i += 256) b += src[pad_l + T + i];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
        const float sa = red[0][0] + red[0][1] + red[0][2] + red[0][3], sb = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (T == 1) dst[0] = dst[0] + sa + sb;
        else { dst[0] += sa; dst[T - 1] += sb; }
    }
}

// Transpose of resample_kernel: dx[row][s] = sum over outputs m = n * nw + f (m < Tout) and taps j with n * orig + j - width = s of
// K[f][j] * dy[row][m].  One thread per input sample.
__global__ __launch_bounds__(256) void resample_adjoint_kernel(const float* __restrict__ dy, const float* __restrict__ K, float* __restrict__ dx,
                                                                int T, int Tout, int orig, int nw, int L, int width) {
    const int row = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= T) return;
    const float* dr = dy + (size_t)row * Tout;
    const int hi = (s + width) / orig;                              // j = s + width - n * orig >= 0
    int lo = s + width - (L - 1);                                   // j <= L - 1
    lo = lo <= 0 ? 0 : (lo + orig - 1) / orig;
    float acc = 0.f;
    for (int n = lo; n <= hi; ++n) {
        const int j = s + width - n * orig;
        for (int f = 0; f < nw; ++f) {
            const int m = n * nw + f;
            if (m < Tout) acc = fmaf(K[(size_t)f * L + j], dr[m], acc);
        }
    }
    dx[(size_t)row * T + s] = acc;
}

}  // namespace wv

extern "C" int wv_fx_fold_replicate(const float* dxp, float* dx, int rows, int T, int pad_l, int pad_r, void* stream) {
    if (!dxp || !dx || rows < 1 || T < 1 || pad_l < 0 || pad_r < 0) return WV_EINVAL;
    hipLaunchKernelGGL(wv::fold_replicate_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dxp, dx, T, pad_l, pad_r);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

extern "C" int wv_fx_resample_adjoint(const float* dy, const float* kernels, float* dx, int rows, int T, int orig, int nw, int L, int width, int Tout,
                                      void* stream) {
    if (!dy || !kernels || !dx || rows < 1 || rows > 65535 || T < 1 || orig < 1 || nw < 1 || L < 1 || width < 0 || Tout < 1) return WV_EINVAL;
    hipLaunchKernelGGL(wv::resample_adjoint_kernel, dim3((T + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, dy, kernels, dx, T, Tout, orig, nw, L,
                       width);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

extern "C" int wv_fx_resample(const float* x, const float* kernels, float* y, int rows, int T, int orig, int nw, int L, int width, int Tout, void* stream) {
    if (!x || !kernels || !y || rows < 1 || rows > 65535 || T < 1 || orig < 1 || nw < 1 || L < 1 || width < 0 || Tout < 1) return WV_EINVAL;
    if ((long long)Tout > ((long long)T + orig - 1) / orig * nw + nw) return WV_EINVAL;
    hipLaunchKernelGGL(wv::resample_kernel, dim3((Tout + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, x, kernels, y, T, Tout, orig, nw, L, width);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

extern "C" int wv_fx_fir_bank(const float* x, const float* taps, float* y, int rows, int T, int n_filters, int L, int stride, int pad_l, int pad_r,
                              int replicate, int interleave, void* stream) {
    if (!x || !taps || !y || rows < 1 || T < 1 || n_filters < 1 || n_filters > wv::FX_MAX_F || L < 1 || stride < 1 || pad_l < 0 || pad_r < 0 ||
        rows > 65535)
        return WV_EINVAL;
    const long long padded = (long long)T + pad_l + pad_r;
    if (padded < L) return WV_EINVAL;
    const int Tout = (int)((padded - L) / stride + 1);
    const size_t smem = ((size_t)(wv::FX_TILE - 1) * stride + wv::FX_JC + (size_t)n_filters * wv::FX_JC) * sizeof(float);
    if (smem > 64 * 1024) return WV_EINVAL;                       // stride <= ~50 with the default LDS budget
    hipLaunchKernelGGL(wv::fir_bank_kernel, dim3((Tout + wv::FX_TILE - 1) / wv::FX_TILE, rows), dim3(256), smem, (hipStream_t)stream, x, taps, y, T, Tout, L,
                       n_filters, stride, pad_l, replicate, interleave);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}
