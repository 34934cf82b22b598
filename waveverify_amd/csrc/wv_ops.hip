// wv_op_* entry points: each fused unit on its own, with weights handed over in the
// reference's layouts (HOST pointers; packed and uploaded per call, synchronously).  These are
// the handles the kernel-level parity tests pull; the model forward passes (wv_model.hip) use
// the same launchers with weights packed once at finalize.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/waveverify_hip.h"
#include "wv_kernels.h"

namespace {

extern "C" const char* wv_last_error(void);

struct Tmp {                       // scoped device uploads
    std::vector<void*> d;
    bool ok = true;
    const float* up(const float* h, size_t n) {
        if (!h || !ok) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)) != hipSuccess) { ok = false; return nullptr; }
        d.push_back(p);
        if (n && hipMemcpy(p, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { ok = false; return nullptr; }
        return (const float*)p;
    }
    const float* upv(const std::vector<float>& v) { return up(v.data(), v.size()); }
    const void* upb(const void* h, size_t bytes) {
        if (!h || !ok) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) { ok = false; return nullptr; }
        d.push_back(p);
        if (bytes && hipMemcpy(p, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { ok = false; return nullptr; }
        return p;
    }
    wv::H16Weight h16(const float* pw, const float* dw, int M, int K, int ks) {
        wv::H16Weight w;
        const std::vector<uint16_t> q = wv::pack_h16(pw, dw, M, K, ks, &w);
        w.wq = upb(q.data(), q.size() * sizeof(uint16_t));
        return w;
    }
    wv::PwWeight pw(const float* w, int M, int K) {
        wv::PwWeight p; p.M = M; p.K = K; p.Kp = wv::round_up(K, wv::BK); p.Mp = wv::round_up(M, wv::M_ALIGN);
        std::vector<float> t((size_t)p.Kp * p.Mp, 0.f);
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < K; ++k) t[(size_t)k * p.Mp + m] = w[(size_t)m * K + k];
        p.wt = upv(t);
        std::vector<float> q((size_t)wv::round_up(K, 32) * p.Mp, 0.f);   // wq[roundup(K,32)/4][Mp][4]
        for (int mm = 0; mm < M; ++mm)
            for (int k = 0; k < K; ++k) q[((size_t)(k / 4) * p.Mp + mm) * 4 + (k & 3)] = w[(size_t)mm * K + k];
        p.wq = upv(q);
        return p;
    }
    ~Tmp() { (void)hipDeviceSynchronize(); for (void* p : d) (void)hipFree(p); }
};

int done(Tmp& t, hipError_t e, hipStream_t s) {
    if (!t.ok) return WV_EHIP;
    if (e != hipSuccess) return e == hipErrorInvalidValue ? WV_EINVAL : WV_EHIP;
    return hipStreamSynchronize(s) == hipSuccess ? WV_OK : WV_EHIP;
}

}  // namespace

extern "C" {

int wv_op_pw_dw(const float* X, const float* w_pw, const float* w_dw, const float* dw_bias,
                const float* film, const float* resid, float* Y, int B, int K, int M, int Tin,
                int ks, int stride, int dilation, float pre_scale, int pre_elu, float out_scale,
                int bands, float* Yact, float act_scale, void* stream) {
    if (!X || !w_pw || !w_dw || (!Y && !Yact) || B < 1 || K < 1 || M < 1 || Tin < 1) return WV_EINVAL;
    Tmp t;
    wv::PwDwArgs a{};
    a.X = X; a.pw = t.pw(w_pw, M, K); a.dw_w = t.up(w_dw, (size_t)M * ks);
    a.dw_b = t.up(dw_bias, M); a.film = film; a.resid = resid; a.Y = Y; a.Yact = Yact; a.act_scale = act_scale;
    a.B = B; a.Tin = Tin; a.Tout = (Tin + stride - 1) / stride; a.ks = ks; a.stride = stride;
    a.dil = dilation; a.pad = (ks - 1) * dilation - (stride - 1);
    a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.out_scale = out_scale;
    a.bands = bands > 0 ? bands : 1; a.film_stride = 2 * a.bands;
    if (a.pad < 0) return WV_EINVAL;
    return done(t, wv::launch_pw_dw(a, (hipStream_t)stream), (hipStream_t)stream);
}

int wv_op_resblock(const float* X, float pre_scale, const float* w_pw1, const float* w_dw1, const float* b1,
                   const float* w_pw2, const float* w_dw2, const float* b2, float* Y, float* Yact,
                   int B, int C, int T, float out_scale, float act_scale, void* stream) {
    if (!X || !w_pw1 || !w_dw1 || !w_pw2 || !w_dw2 || (!Y && !Yact) || B < 1 || C < 1 || T < 1) return WV_EINVAL;
    Tmp t;
    wv::RbArgs a{};
    a.X = X; a.pre_scale = pre_scale; a.pw1 = t.pw(w_pw1, C, C); a.pw2 = t.pw(w_pw2, C, C);
    a.tab1 = t.upv(wv::pack_rb_table(w_dw1, b1, C)); a.tab2 = t.upv(wv::pack_rb_table(w_dw2, b2, C));
    a.Y = Y; a.Yact = Yact; a.out_scale = out_scale; a.act_scale = act_scale; a.B = B; a.C = C; a.T = T;
    const hipError_t e = wv::launch_resblock(a, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return WV_EINVAL;
    return done(t, e, (hipStream_t)stream);
}

// ---- the f16 mode's units (wv_h16.hip): activations in the c8 f16 layout, weights as HOST f32 pointers in the reference's layouts
// host-side f32 -> f16 rounding the weight packers use (round to nearest even; no device needed)
int wv_h16_round_host(const float* in, uint16_t* out, int64_t n) {
    if (!in || !out || n < 0) return WV_EINVAL;
    for (int64_t i = 0; i < n; ++i) out[i] = wv::f32_to_f16_bits(in[i]);
    return WV_OK;
}
int wv_h16_from_f32(const float* X, void* Y16, int B, int C, int T, float scale, int elu, void* stream) {
    Tmp t;
    return done(t, wv::launch_f32_to_c8(X, Y16, B, C, T, scale, elu, (hipStream_t)stream), (hipStream_t)stream);
}
int wv_h16_to_f32(const void* X16, float* Y, int B, int C, int T, void* stream) {
    Tmp t;
    return done(t, wv::launch_c8_to_f32(X16, Y, B, C, T, (hipStream_t)stream), (hipStream_t)stream);
}
int wv_h16_conv_pre(const float* x, const float* w, const float* bias, void* Y16, int B, int C, int T, int ks, float in_scale, void* stream) {
    if (!x || !w || !Y16 || C < 1 || ks < 1) return WV_EINVAL;
    Tmp t;
    return done(t, wv::launch_conv_pre16(x, t.up(w, (size_t)C * ks), t.up(bias, C), Y16, B, C, T, ks, in_scale, (hipStream_t)stream), (hipStream_t)stream);
}
int wv_h16_resblock(const void* X16, float pre_scale, const float* w_pw1, const float* w_dw1, const float* b1, const float* w_pw2, const float* w_dw2,
                    const float* b2, void* Y16, void* Yact16, int B, int C, int T, float out_scale, float act_scale, void* stream) {
    if (!X16 || !w_pw1 || !w_dw1 || !w_pw2 || !w_dw2 || (!Y16 && !Yact16) || B < 1 || C < 1 || T < 1) return WV_EINVAL;
    Tmp t;
    wv::RhArgs a{};
    a.X = X16; a.pre_scale = pre_scale;
    { const std::vector<uint16_t> q = wv::pack_rh_pw(w_pw1, C, &a.w1); a.w1.wq = t.upb(q.data(), q.size() * sizeof(uint16_t)); }
    { const std::vector<uint16_t> q = wv::pack_rh_pw(w_pw2, C, &a.w2); a.w2.wq = t.upb(q.data(), q.size() * sizeof(uint16_t)); }
    a.tab1 = t.upv(wv::pack_rh_table(w_dw1, b1, C, wv::RH_LOG2E)); a.tab2 = t.upv(wv::pack_rh_table(w_dw2, b2, C, 1.0));
    a.Y = Y16; a.Yact = Yact16; a.out_scale = out_scale; a.act_scale = act_scale; a.B = B; a.C = C; a.T = T;
    const hipError_t e = wv::launch_resblock16(a, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return WV_EINVAL;
    return done(t, e, (hipStream_t)stream);
}
int wv_h16_conv(const void* X16, const float* w_pw, const float* w_dw, const float* bias, const void* resid16, void* Y16, void* Yact16, float* Yf32,
                int B, int K, int M, int Tin, int ks, int stride, int pad, float out_scale, float act_scale, void* stream) {
    if (!X16 || !w_pw || B < 1 || K < 1 || M < 1 || Tin < 1 || ks < 1 || stride < 1 || pad < 0) return WV_EINVAL;
    Tmp t;
    wv::Conv16Args a{};
    a.X = X16; a.w = t.h16(w_pw, w_dw, M, K, ks); a.bias = t.up(bias, M); a.resid = resid16; a.Y = Y16; a.Yact = Yact16; a.Yf32 = Yf32;
    a.out_scale = out_scale; a.act_scale = act_scale; a.B = B; a.M = M; a.Tin = Tin; a.Tout = (Tin + stride - 1) / stride;
    a.ks = ks; a.stride = stride; a.pad = pad;
    return done(t, wv::launch_conv16(a, (hipStream_t)stream), (hipStream_t)stream);
}

int wv_op_dw_pw(const float* X, const float* w_dw, const float* w_pw, const float* bias, float* Y,
                int B, int K, int M, int Tin, int mode, int ks_or_ratio, float pre_scale, int pre_elu,
                int l2norm, int accumulate, float out_scale, float* Yact, float act_scale, void* stream) {
    if (!X || !w_pw || !Y || B < 1 || K < 1 || M < 1 || Tin < 1) return WV_EINVAL;
    if (mode == 2 && !l2norm && !accumulate) {
        // upsample unit = K1 kernel with the ConvTranspose producer in its loader (as the model runs it)
        if (!w_dw || ks_or_ratio < 1) return WV_EINVAL;
        Tmp t;
        std::vector<float> taps((size_t)M * 5, 0.f);
        for (int m = 0; m < M; ++m) taps[(size_t)m * 5 + 4] = 1.f;
        wv::PwDwArgs a{};
        a.X = X; a.pw = t.pw(w_pw, M, K); a.ct_w = t.up(w_dw, (size_t)K * 2 * ks_or_ratio); a.ratio = ks_or_ratio;
        a.ct_wt = t.upv(wv::pack_ct_wt(w_dw, K, a.pw.Kp, ks_or_ratio));
        a.dw_w = t.upv(taps); a.dw_b = t.up(bias, M); a.Y = Y; a.Yact = Yact; a.act_scale = act_scale;
        a.B = B; a.Tin = Tin; a.Tout = Tin * ks_or_ratio; a.ks = 5; a.stride = 1; a.dil = 1; a.pad = 4;
        a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.out_scale = 1.f; a.bands = 1;
        a.film_stride = 2;
        return done(t, wv::launch_pw_dw(a, (hipStream_t)stream), (hipStream_t)stream);
    }
    Tmp t;
    wv::DwPwArgs a{};
    a.X = X; a.pw = t.pw(w_pw, M, K); a.bias = t.up(bias, M); a.Y = Y;
    a.B = B; a.Tin = Tin; a.mode = mode; a.Tout = Tin;
    if (mode == 2) return WV_EINVAL;                  // the upsample unit has neither L2-norm nor accumulate
    const bool k1_form = mode == 0 && accumulate && !l2norm && !bias && M >= 128;
    if (Yact && !k1_form) return WV_EINVAL;           // the second output exists on the pw_dw kernel only
    if (mode == 0 && accumulate && !l2norm && !bias && M >= 128) {
        // SpecBlock add as the model runs it for M >= 128: K1 kernel, identity stencil, Y as residual
        Tmp t;
        std::vector<float> taps((size_t)M * 5, 0.f);
        for (int m = 0; m < M; ++m) taps[(size_t)m * 5 + 4] = 1.f;
        wv::PwDwArgs a{};
        a.X = X; a.pw = t.pw(w_pw, M, K); a.dw_w = t.upv(taps); a.resid = Y; a.Y = Y; a.Yact = Yact; a.act_scale = act_scale;
        a.B = B; a.Tin = Tin; a.Tout = Tin; a.ks = 5; a.stride = 1; a.dil = 1; a.pad = 4;
        a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.out_scale = out_scale; a.bands = 1; a.film_stride = 2;
        a.spec_add = 1;
        return done(t, wv::launch_pw_dw(a, (hipStream_t)stream), (hipStream_t)stream);
    }
    if (mode == 1) { a.ks = ks_or_ratio; a.dw_w = t.up(w_dw, (size_t)K * a.ks); }
    a.pre_scale = pre_scale; a.pre_elu = pre_elu; a.l2norm = l2norm; a.accumulate = accumulate;
    a.out_scale = out_scale;
    return done(t, wv::launch_dw_pw(a, (hipStream_t)stream), (hipStream_t)stream);
}

// host: the reference's windowed DFT basis [2F][n_fft] (conv.py:1003-1026), or the caller's
static std::vector<float> stft_basis_host(const float* basis_or_null, int n_fft) {
    const int F = n_fft / 2 + 1;
    std::vector<float> basis((size_t)2 * F * n_fft);
    if (basis_or_null) {
        std::memcpy(basis.data(), basis_or_null, basis.size() * sizeof(float));
    } else {
        // float32 arithmetic in the reference's order (bit-identical to its buffer: tests/golden/dft_basis.npz)
        const float c = (float)(-2.0 * M_PI / n_fft), wc = (float)(2.0 * M_PI / n_fft);
        for (int k = 0; k < F; ++k) {
            volatile float ck = c * (float)k;
            for (int n = 0; n < n_fft; ++n) {
                volatile float ang = ck * (float)n;
                volatile float wa = (float)n * wc;
                const float win = 0.5f - 0.5f * (float)std::cos((double)wa);
                basis[(size_t)k * n_fft + n] = (float)std::cos((double)ang) * win;
                basis[(size_t)(F + k) * n_fft + n] = (float)std::sin((double)ang) * win;
            }
        }
    }
    return basis;
}

int wv_op_stft_logmag(const float* wav, const float* basis_or_null, float* P, int B, int T, int n_fft,
                      int hop, float mean, float std, void* stream) {
    if (!wav || !P || B < 1 || T < 1 || hop < 1) return WV_EINVAL;
    if (n_fft < 4 || (n_fft & 1)) return WV_EINVAL;
    const int F = n_fft / 2 + 1;
    const std::vector<float> basis = stft_basis_host(basis_or_null, n_fft);
    std::vector<float> bt, side;
    int Mp = 0;
    wv::pack_stft_basis(basis.data(), n_fft, bt, side, &Mp);
    Tmp t;
    wv::StftArgs a{};
    a.wav = wav; a.basis_t = t.upv(bt); a.basis_q = t.upv(wv::pack_stft_q(bt, n_fft, Mp)); a.side = t.upv(side); a.P = P; a.B = B; a.T = T; a.Tf = (T + hop - 1) / hop;
    a.n_fft = n_fft; a.hop = hop; a.F = F; a.Mp = Mp; a.mean = mean; a.inv_std = 1.f / std;
    return done(t, wv::launch_stft_logmag(a, (hipStream_t)stream), (hipStream_t)stream);
}

int wv_op_spec_block(const float* wav, const float* basis_or_null, const float* w_pw, const float* x, float* Y, float* Yact, int B, int T,
                     int n_fft, int hop, int M, float mean, float std, float out_scale, float act_scale, void* stream) {
    if (!wav || !w_pw || !x || (!Y && !Yact) || B < 1 || T < 1 || hop < 1 || M < 1) return WV_EINVAL;
    if (n_fft < 4 || (n_fft & 1)) return WV_EINVAL;
    const int F = n_fft / 2 + 1;
    const std::vector<float> basis = stft_basis_host(basis_or_null, n_fft);
    std::vector<float> bt, side;
    int Mp = 0;
    wv::pack_stft_basis(basis.data(), n_fft, bt, side, &Mp);
    Tmp t;
    wv::StftArgs a{};
    a.wav = wav; a.basis_t = t.upv(bt); a.basis_q = t.upv(wv::pack_stft_q(bt, n_fft, Mp)); a.side = t.upv(side); a.P = nullptr; a.B = B; a.T = T;
    a.Tf = (T + hop - 1) / hop; a.n_fft = n_fft; a.hop = hop; a.F = F; a.Mp = Mp; a.mean = mean; a.inv_std = 1.f / std;
    wv::SpecAddArgs q{};
    q.pw = t.pw(w_pw, M, F); q.resid = x; q.Y = Y; q.Yact = Yact; q.out_scale = out_scale; q.act_scale = act_scale;
    const hipError_t e = wv::launch_stft_spec(a, q, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return WV_EINVAL;
    return done(t, e, (hipStream_t)stream);
}

int wv_h16_upsample(const void* X16, const float* w_ct, const float* w_pw, const float* bias, void* Y16, void* Yact16,
                    int B, int K, int M, int Tin, int ratio, float act_scale, void* stream) {
    if (!X16 || !w_ct || !w_pw || (!Y16 && !Yact16) || B < 1 || K < 1 || M < 1 || Tin < 1 || ratio < 1) return WV_EINVAL;
    Tmp t;
    wv::Conv16Args a{};
    const int mb = wv::up16_block(M, ratio) ? wv::up16_block(M, ratio) : M;
    { const std::vector<uint16_t> q = wv::pack_up16(w_pw, w_ct, M, K, ratio, mb, &a.w); a.w.wq = t.upb(q.data(), q.size() * sizeof(uint16_t)); }
    a.up_mb = mb;
    a.X = X16; a.bias = t.up(bias, M); a.Y = Y16; a.Yact = Yact16; a.out_scale = 1.f; a.act_scale = act_scale;
    a.B = B; a.M = M * ratio; a.Tin = Tin; a.Tout = Tin; a.ks = 2; a.stride = 1; a.pad = 1; a.up = ratio;
    return done(t, wv::launch_conv16(a, (hipStream_t)stream), (hipStream_t)stream);
}
int wv_h16_tail(const void* A16, const float* w, const float* bias, const float* x, float* out, int B, int C, int Tin, int T, int ks,
                float out_scale, void* stream) {
    if (!A16 || !w || !out || C < 1 || ks < 1) return WV_EINVAL;
    Tmp t;
    const hipError_t e = wv::launch_tail16(A16, t.up(w, (size_t)C * ks), t.up(bias, 1), x, out, B, C, Tin, T, ks, out_scale, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return WV_EINVAL;
    return done(t, e, (hipStream_t)stream);
}
int wv_h16_l2norm(const float* lat, void* Y16, int B, int D, int Fr, void* stream) {
    Tmp t;
    return done(t, wv::launch_l2norm_c8(lat, Y16, B, D, Fr, (hipStream_t)stream), (hipStream_t)stream);
}
int wv_h16_conv_film(const void* X16, const float* w_pw, const float* w_dw, const float* bias, const float* film, int bands, void* Y16, void* Yact16,
                     int B, int K, int M, int Tin, int ks, int stride, int pad, float act_scale, void* stream) {
    if (!X16 || !w_pw || !film || bands < 1 || B < 1 || K < 1 || M < 1 || Tin < 1 || ks < 1 || stride < 1 || pad < 0) return WV_EINVAL;
    Tmp t;
    wv::Conv16Args a{};
    a.X = X16; a.w = t.h16(w_pw, w_dw, M, K, ks); a.bias = t.up(bias, M); a.Y = Y16; a.Yact = Yact16;
    a.out_scale = 1.f; a.act_scale = act_scale; a.B = B; a.M = M; a.Tin = Tin; a.Tout = (Tin + stride - 1) / stride; a.ks = ks; a.stride = stride; a.pad = pad;
    a.film = film; a.bands = bands; a.film_stride = 2 * bands;
    return done(t, wv::launch_conv16(a, (hipStream_t)stream), (hipStream_t)stream);
}
int wv_h16_spec_block(const float* wav, const float* basis_or_null, const float* w_pw, const void* x16, void* Y16, void* Yact16, int B, int T,
                      int n_fft, int hop, int M, float mean, float std_, float out_scale, float act_scale, void* stream) {
    if (!wav || !w_pw || !x16 || (!Y16 && !Yact16) || B < 1 || T < 1 || n_fft < 2 || (n_fft & 1) || hop < 1 || (M != n_fft && 2 * M != n_fft) || !(std_ > 0.f)) return WV_EINVAL;
    Tmp t;
    const std::vector<float> basis = stft_basis_host(basis_or_null, n_fft);
    std::vector<uint16_t> q4[4];
    wv::H16Weight w4[4];
    wv::Spec16Args a{};
    wv::pack_stft16(basis.data(), n_fft, q4, w4);
    for (int k = 0; k < 4; ++k) w4[k].wq = t.upb(q4[k].data(), q4[k].size() * sizeof(uint16_t));
    a.cosw = w4[0]; a.sinw = w4[1]; a.cosl = w4[2]; a.sinl = w4[3];
    a.pw = t.h16(w_pw, nullptr, M, n_fft / 2 + 1, 1);
    a.wav = wav; a.resid = x16; a.Y = Y16; a.Yact = Yact16; a.out_scale = out_scale; a.act_scale = act_scale;
    a.c1 = 0.5f * 0.69314718055994531f / std_; a.c0 = -mean / std_;
    a.B = B; a.T = T; a.Tf = (T + hop - 1) / hop; a.n_fft = n_fft; a.hop = hop;
    const hipError_t e = wv::launch_spec16(a, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return WV_EINVAL;
    return done(t, e, (hipStream_t)stream);
}

}  // extern "C"

// A resident plan of the same op (the basis packed and uploaded once): what a training step calls once per scale and step.
// It also carries the BACKWARD of the features towards the audio (training the generator through the detector / locator):
//   C = Basis[2F][n] @ frames(wav)  (re rows, then im rows);  p = re^2 + im^2;  P = (0.5 ln max(p, 1e-10) - mean) / std
//   dC = dP * {re, im} / (std * p)  where p > 1e-10 (the clamps of conv.py:1078 / seanet.py:484 pass no gradient below), else 0
//   dwav = overlap-add( Basis^T[n][2F] @ dC )
// on the generic GEMM core: frames are materialised ([B][n_fft][Tf], a few MB per clip), two GEMMs, three small kernels.
struct wv_stft_plan {
    int n_fft = 0, Mp = 0;
    float *basis_t = nullptr, *basis_q = nullptr, *side = nullptr;
    float *wt_fwd = nullptr, *wt_bwd = nullptr;      // K-major packs of Basis [2F][n] and of Basis^T [n][2F]
    ~wv_stft_plan() { (void)hipFree(basis_t); (void)hipFree(basis_q); (void)hipFree(side); (void)hipFree(wt_fwd); (void)hipFree(wt_bwd); }
};

namespace {
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ wav, float* __restrict__ X, int T, int Tf, int n_fft, int hop) {
    const int n = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Tf) return;
    const int sidx = t * hop + n - (n_fft - 1);
    X[((size_t)b * n_fft + n) * Tf + t] = (sidx >= 0 && sidx < T) ? wav[(size_t)b * T + sidx] : 0.f;
}
// in place: C[b][f] (re), C[b][F + f] (im) <- their gradients
__global__ __launch_bounds__(256) void stft_dc_kernel(float* __restrict__ Cm, const float* __restrict__ dP, int F, int Tf, float inv_std) {
    const int f = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Tf) return;
    float* re = Cm + ((size_t)b * 2 * F + f) * Tf + t;
    float* im = Cm + ((size_t)b * 2 * F + F + f) * Tf + t;
    const float r = *re, i = *im, p = fmaf(r, r, i * i);
    const float g = p > 1e-10f ? dP[((size_t)b * F + f) * Tf + t] * inv_std / p : 0.f;
    *re = g * r; *im = g * i;
}
// dwav[b][s] (+)= sum over frames t and taps n with t*hop + n - (n_fft-1) = s of Q[b][n][t]   (fixed order: t ascending)
__global__ __launch_bounds__(256) void stft_overlap_add_kernel(const float* __restrict__ Q, float* __restrict__ dwav, int T, int Tf, int n_fft, int hop, int accumulate) {
    const int b = blockIdx.y;
    const int sidx = blockIdx.x * 256 + threadIdx.x;
    if (sidx >= T) return;
    const int base = sidx + (n_fft - 1);                     // n = base - t*hop in [0, n_fft)
    int t_lo = (base - (n_fft - 1) + hop - 1) / hop;
    if (t_lo < 0) t_lo = 0;
    int t_hi = base / hop;
    if (t_hi > Tf - 1) t_hi = Tf - 1;
    float a = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) a += Q[((size_t)b * n_fft + (base - t * hop)) * Tf + t];
    dwav[(size_t)b * T + sidx] = accumulate ? dwav[(size_t)b * T + sidx] + a : a;
}
size_t al256o(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

extern "C" {

int wv_stft_plan_create(int n_fft, const float* basis_or_null, wv_stft_plan** out) {
    if (!out || n_fft < 4 || (n_fft & 1)) return WV_EINVAL;
    const std::vector<float> basis = stft_basis_host(basis_or_null, n_fft);
    std::vector<float> bt, side;
    auto* p = new wv_stft_plan();
    p->n_fft = n_fft;
    wv::pack_stft_basis(basis.data(), n_fft, bt, side, &p->Mp);
    const std::vector<float> bq = wv::pack_stft_q(bt, n_fft, p->Mp);
    const int F = n_fft / 2 + 1, M2 = 2 * F;
    const int Mp_f = wv::round_up(M2, wv::M_ALIGN), Kp_f = wv::round_up(n_fft, wv::BK);
    const int Mp_b = wv::round_up(n_fft, wv::M_ALIGN), Kp_b = wv::round_up(M2, wv::BK);
    std::vector<float> wf((size_t)Kp_f * Mp_f, 0.f), wb((size_t)Kp_b * Mp_b, 0.f);
    for (int m = 0; m < M2; ++m)
        for (int n = 0; n < n_fft; ++n) {
            wf[(size_t)n * Mp_f + m] = basis[(size_t)m * n_fft + n];
            wb[(size_t)m * Mp_b + n] = basis[(size_t)m * n_fft + n];
        }
    auto up = [](float** d, const std::vector<float>& v) {
        return hipMalloc((void**)d, v.size() * sizeof(float)) == hipSuccess &&
               hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!(up(&p->basis_t, bt) && up(&p->basis_q, bq) && up(&p->side, side) && up(&p->wt_fwd, wf) && up(&p->wt_bwd, wb))) { delete p; return WV_EHIP; }
    *out = p;
    return WV_OK;
}

void wv_stft_plan_destroy(wv_stft_plan* p) { delete p; }

int wv_stft_plan_logmag(const wv_stft_plan* p, const float* wav, float* P, int B, int T, int hop, float mean, float std, void* stream) {
    if (!p || !wav || !P || B < 1 || T < 1 || hop < 1 || !(std > 0.f)) return WV_EINVAL;
    wv::StftArgs a{};
    a.wav = wav; a.basis_t = p->basis_t; a.basis_q = p->basis_q; a.side = p->side; a.P = P; a.B = B; a.T = T; a.Tf = (T + hop - 1) / hop;
    a.n_fft = p->n_fft; a.hop = hop; a.F = p->n_fft / 2 + 1; a.Mp = p->Mp; a.mean = mean; a.inv_std = 1.f / std;
    const hipError_t e = wv::launch_stft_logmag(a, (hipStream_t)stream);
    return e == hipSuccess ? WV_OK : (e == hipErrorInvalidValue ? WV_EINVAL : WV_EHIP);
}

size_t wv_stft_plan_backward_workspace_bytes(const wv_stft_plan* p, int B, int T, int hop) {
    if (!p || B < 1 || T < 1 || hop < 1) return 0;
    const int Tf = (T + hop - 1) / hop, F = p->n_fft / 2 + 1;
    return al256o((size_t)B * p->n_fft * Tf * 4) + al256o((size_t)B * 2 * F * Tf * 4);
}

int wv_stft_plan_backward(const wv_stft_plan* p, const float* wav, const float* dP, float* dwav, int accumulate, int B, int T, int hop, float std,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!p || !wav || !dP || !dwav || B < 1 || T < 1 || hop < 1 || !(std > 0.f) || B > 65535) return WV_EINVAL;
    if (!ws || ws_bytes < wv_stft_plan_backward_workspace_bytes(p, B, T, hop)) return WV_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int n = p->n_fft, F = n / 2 + 1, Tf = (T + hop - 1) / hop;
    float* X = (float*)ws;
    float* Cm = (float*)((char*)ws + al256o((size_t)B * n * Tf * 4));
    hipLaunchKernelGGL(stft_frames_kernel, dim3((Tf + 255) / 256, n, B), dim3(256), 0, s, wav, X, T, Tf, n, hop);
    auto gemm = [&](const float* Xin, int M, int K, const float* wt, float* Y) {
        wv::DwPwArgs a{};
        a.X = Xin; a.pw.M = M; a.pw.K = K; a.pw.Mp = wv::round_up(M, wv::M_ALIGN); a.pw.Kp = wv::round_up(K, wv::BK); a.pw.wt = wt; a.pw.wq = nullptr;
        a.bias = nullptr; a.Y = Y; a.B = B; a.Tin = Tf; a.Tout = Tf; a.mode = 0; a.ks = 1; a.pre_scale = 1.f; a.pre_elu = 0; a.l2norm = 0; a.out_scale = 1.f;
        return wv::launch_dw_pw(a, s);
    };
    hipError_t e = gemm(X, 2 * F, n, p->wt_fwd, Cm);
    if (e != hipSuccess) return WV_EHIP;
    hipLaunchKernelGGL(stft_dc_kernel, dim3((Tf + 255) / 256, F, B), dim3(256), 0, s, Cm, dP, F, Tf, 1.f / std);
    e = gemm(Cm, n, 2 * F, p->wt_bwd, X);                          // Q into the frames buffer
    if (e != hipSuccess) return WV_EHIP;
    hipLaunchKernelGGL(stft_overlap_add_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, X, dwav, T, Tf, n, hop, accumulate);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

int wv_op_conv_pre(const float* x, const float* w, const float* bias, float* Y, int B, int C, int T,
                   int ks, float in_scale, void* stream) {
    if (!x || !w || !Y) return WV_EINVAL;
    Tmp t;
    return done(t, wv::launch_conv_pre(x, t.up(w, (size_t)C * ks), t.up(bias, C), Y, nullptr, 0.f, B, C, T, ks,
                                       in_scale, (hipStream_t)stream), (hipStream_t)stream);
}

int wv_op_tail(const float* H, const float* w, const float* bias, const float* x_or_null, float* out,
               int B, int C, int Tin, int T, int ks, float pre_scale, float out_scale, void* stream) {
    if (!H || !w || !out) return WV_EINVAL;
    Tmp t;
    return done(t, wv::launch_tail(H, t.up(w, (size_t)C * ks), t.up(bias, 1), x_or_null, out, B, C, Tin,
                                   T, ks, pre_scale, out_scale, (hipStream_t)stream), (hipStream_t)stream);
}

int wv_op_head(const float* Z, const float* w_rev, const float* b_rev, const float* w_last,
               const float* b_last, float* logits, float* mean_prob, int B, int D, int O, int nb,
               int hop, int Fr, int T, void* stream) {
    if (!Z || !w_rev || !b_rev || !w_last || !b_last || (!logits && !mean_prob)) return WV_EINVAL;
    if (T > Fr * hop) return WV_EINVAL;
    std::vector<float> wc((size_t)D * nb * hop), bc(nb);
    for (int d = 0; d < D; ++d)
        for (int n = 0; n < nb; ++n)
            for (int j = 0; j < hop; ++j) {
                double acc = 0;
                for (int o = 0; o < O; ++o) acc += (double)w_last[(size_t)n * O + o] * w_rev[((size_t)d * O + o) * hop + j];
                wc[((size_t)d * nb + n) * hop + j] = (float)acc;
            }
    for (int n = 0; n < nb; ++n) {
        double acc = b_last[n];
        for (int o = 0; o < O; ++o) acc += (double)w_last[(size_t)n * O + o] * b_rev[o];
        bc[n] = (float)acc;
    }
    Tmp t;
    wv::HeadArgs a{};
    a.Z = Z; a.wc = t.upv(wc); a.bc = t.upv(bc); a.logits = logits; a.mean_prob = mean_prob;
    a.B = B; a.D = D; a.nb = nb; a.hop = hop; a.Fr = Fr; a.T = T;
    return done(t, wv::launch_head(a, (hipStream_t)stream), (hipStream_t)stream);
}

}  // extern "C"
