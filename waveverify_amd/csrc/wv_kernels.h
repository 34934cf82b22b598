// Internal launcher interface between the host-side model plan (wv_model.hip) and the
// gfx950 kernels (wv_kernels.hip).  All pointers are device pointers; weights are in the
// PACKED layouts described next to each struct.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <vector>

namespace wv {

constexpr int BK = 16;          // K-chunk staged per pipeline step
constexpr int M_ALIGN = 128;    // packed 1x1 weights: Mp = roundup(M, M_ALIGN)

inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

// Packed pointwise weight: Wt[Kp][Mp] (transposed, zero padded), Kp = roundup(K, BK).
struct PwWeight {
    const float* wt = nullptr;
    int K = 0, M = 0, Kp = 0, Mp = 0;
    // k-inner f32 layout for the row-strip cores: wq[roundup(K,32)/4][Mp][4] (W[m][4kq .. 4kq+3], zero padded;
    // the LDS-DMA core reads whole 32-deep chunks); may be null
    const float* wq = nullptr;
};
// ---- K1: 1x1 GEMM -> depth-wise stencil epilogue ------------------------------------------
struct PwDwArgs {
    const float* X;       // [B, K, Tin]
    PwWeight pw;          // 1x1 (no bias)
    const float* dw_w;    // [M, ks]
    const float* dw_b;    // [M] or null
    const float* film;    // [B, film_stride] (gamma,beta interleaved per band) or null
    const float* resid;   // [B, M, Tout] or null
    const float* resid2;  // res_mode 2 only: [B, M, Tout] added after the derivative, or null
    float* Yraw;          // training forward, LDS-DMA core with the ks = 5 stencil only: the 1x1 output H [B, M, Tout] stored next to Y, or null
    float* Ysum;          // with Yraw and resid: [B, M, Tout] = resid + out_scale * scale_ptr[0] * y (the ResnetBlock's output), Y keeps y
    const float* scale_ptr; // device scalar for Ysum, or null (= 1)
    int res_mode;         // 0: y = resid + out_scale * y;  2 (training, LDS-DMA core only): y = y * ELU'(out_scale * resid) * out_scale
    float* Y;             // [B, M, Tout]
    int B, Tin, Tout, ks, stride, dil, pad;
    float pre_scale;
    int pre_elu;
    float out_scale;
    int bands, film_stride;
    float* Yact;          // optional SECOND output [B, M, Tout] = ELU(act_scale * y): the consumer's prologue
    float act_scale;      //   hoisted into the producer, so that the consumer stages its operand by LDS-DMA
                          //   (a pure copy).  Y may be null when only the activated copy is consumed.
    int num_m, num_t;         // tile counts (filled by launch_pw_dw; XCD-aware 1-D grid)
    int stagger, first_gen;   // de-phasing of the first workgroup generation (see kernel)
    int tto, off;         // filled by launch_pw_dw: outputs per time tile; stencil offset inside
                          // the 4-aligned H window
    int spec_add;         // 1: the SpecBlock add (identity stencil, resid == Y): own kernel instantiation
    int flat, Tv;         // LDS-DMA core, filled by its launcher: flattened (clip, time) tiling with period Tv = T + pad
    const float* ct_w;    // upsample unit only: DW ConvTranspose taps [K, 2*ratio] and ratio; X is then
    int ratio;            //   [B, K, Tin], Tout = Tin*ratio
    const float* ct_wt;   // the same taps transposed and zero padded, [2*ratio][pw.Kp] (pack_ct_wt)
};
hipError_t launch_pw_dw(const PwDwArgs& a, hipStream_t s);
// the LDS-DMA core (wv_k1.hip); launch_pw_dw routes to it when k1_supported()
bool k1_supported(const PwDwArgs& a);
hipError_t launch_k1(const PwDwArgs& a, hipStream_t s);   // hipErrorNotSupported: use the round-1 kernel
bool pw_dw_geometry(PwDwArgs& a, int BN);                 // time-tile geometry shared by both cores
// ---- whole ResnetBlock in one launch, raw in / raw out (C in {64, 96, 128, 192}; wv_rb.hip) -------------
struct RbArgs {
    const float* X;       // [B, C, T] x: the block's input (activated inside) and its residual operand
    float pre_scale;      // the block's Scale in front of its first ELU (seanet.py:183)
    PwWeight pw1, pw2;    // the two 1x1 convs (no bias); only the k-inner layout wq is read
    const float* tab1;    // [C][8] per channel: 5 depth-wise taps, bias, 1, 0 (pack_rb_table)
    const float* tab2;
    float* Y;             // [B, C, T] y = x + out_scale * block(x), or null
    float* Yact;          // [B, C, T] ELU(act_scale * y), or null
    float out_scale, act_scale;
    int B, C, T;
    int num_t, ntiles;    // filled by the launcher: tiles per clip, tiles in all
    // training forward (wv_train_block_forward): out_scale is multiplied by out_scale_ptr[0] (a device scalar: res_scale_param) when
    // given; sv_* (all four or none, with Y): the tensors the block's backward wants, [B, C, T] each -- h0 = W1 @ ELU(c x), u = DW5(h0) + b1
    // (BEFORE the second half's ELU), h1 = W2 @ ELU(u), v = DW5(h1) + b2 (y = x + s v) -- written from the same registers the kernel computes them in
    const float* out_scale_ptr = nullptr;
    float *sv_h0 = nullptr, *sv_u = nullptr, *sv_h1 = nullptr, *sv_v = nullptr;
};
bool rb_supported(const RbArgs& a);
hipError_t launch_resblock(const RbArgs& a, hipStream_t s);   // hipErrorNotSupported: run it as two K1 launches
inline std::vector<float> pack_rb_table(const float* dw_w, const float* dw_b, int C) {   // dw_w [C][5]
    std::vector<float> t((size_t)C * 8, 0.f);
    for (int m = 0; m < C; ++m) {
        for (int i = 0; i < 5; ++i) t[(size_t)m * 8 + i] = dw_w[(size_t)m * 5 + i];
        t[(size_t)m * 8 + 5] = dw_b ? dw_b[m] : 0.f;
        t[(size_t)m * 8 + 6] = 1.f;
    }
    return t;
}
// host: ConvTranspose taps [K][2r] -> [2r][Kp], zero padded
inline std::vector<float> pack_ct_wt(const float* w, int K, int Kp, int ratio) {
    std::vector<float> t((size_t)2 * ratio * Kp, 0.f);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < 2 * ratio; ++i) t[(size_t)i * Kp + k] = w[(size_t)k * 2 * ratio + i];
    return t;
}

// ---- f16-operand / f32-accumulate detector mode (wv_h16.hip).  Activations: c8 layout f16 [B][roundup(C,16)/8][T][8] ----------------
// Weights as A fragments of v_mfma_f32_32x32x16_f16: wq[chunk][Mp][2][8] f16, chunk = (k/16) * taps + tap, element = W[m][tap][16*kc + 8*h + j];
// Kp = roundup(K, 16), Mp = roundup(M, 32), the chunk count padded to a multiple of 8 with zero chunks.
struct H16Weight { const void* wq = nullptr; int K = 0, M = 0, Kp = 0, Mp = 0, nchunks = 0; };
inline uint16_t f32_to_f16_bits(float f) {                      // round to nearest even, overflow -> inf
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);    // rounds to >= 65520 -> inf
    if (x < 0x38800000u) {                                      // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                              // 14 .. 24
        const uint32_t half = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
        return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1))) ? 1u : 0u)));
    }
    const uint32_t mant = x & 0x7fffffu, e = (x >> 23) - 112;
    uint32_t hv = (e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (hv & 1))) ++hv;
    return (uint16_t)(sign | hv);
}
// host: 1x1 weight pw [M][K] (and, for the composed downsample conv, depth-wise taps dw [M][ks]: W[m][i][k] = dw[m][i] * pw[m][k]) -> A fragments
// dw_by_k: the taps belong to the INPUT channels (a depth-wise conv in front of the 1x1: dw [K][ks], W[m][i][k] = pw[m][k] * dw[k][i])
inline std::vector<uint16_t> pack_h16(const float* pw, const float* dw, int M, int K, int ks, H16Weight* out, bool dw_by_k = false) {
    H16Weight w; w.K = K; w.M = M; w.Kp = round_up(K, 16); w.Mp = round_up(M, 32);
    const int nkc = w.Kp / 16;
    w.nchunks = round_up(ks * nkc, 8);
    std::vector<uint16_t> q((size_t)w.nchunks * w.Mp * 16, 0);
    for (int i = 0; i < ks; ++i)
        for (int m = 0; m < M; ++m) {
            const float t = (dw && !dw_by_k) ? dw[(size_t)m * ks + i] : 1.f;
            for (int k = 0; k < K; ++k) {
                const float tk = (dw && dw_by_k) ? dw[(size_t)k * ks + i] : t;
                q[(((size_t)((k / 16) * ks + i) * w.Mp + m) * 2 + ((k >> 3) & 1)) * 8 + (k & 7)] = f32_to_f16_bits(pw[(size_t)m * K + k] * tk);
            }
        }
    *out = w;
    return q;
}
// whole ResnetBlock, c8 f16 in / out (C in {32, 64, 96, 128, 192, 256, 384, 512, 768}, k = 5, dilation 1).  The kernel keeps both
// activations times log2(e) (one instruction less per ELU, wv_h16.hip elu_l2), so the operands arrive pre-scaled -- pack_rh_pw /
// pack_rh_table: w1, w2 = the 1x1 weights DIVIDED by log2(e); tab1 = the first stencil's taps and bias TIMES log2(e)
// (pack_rh_table(.., RH_LOG2E)); tab2 plain (pack_rh_table(.., 1.0)); both tables in the kernel's channel-pair layout.
struct RhArgs {
    const void* X; float pre_scale; H16Weight w1, w2; const float* tab1; const float* tab2;
    void* Y; void* Yact; float out_scale, act_scale; int B, C, T; int num_t, ntiles;
};
constexpr double RH_LOG2E = 1.4426950408889634;
inline std::vector<uint16_t> pack_rh_pw(const float* pw, int C, H16Weight* out) {
    std::vector<float> t((size_t)C * C);
    for (size_t i = 0; i < t.size(); ++i) t[i] = (float)((double)pw[i] / RH_LOG2E);
    return pack_h16(t.data(), nullptr, C, C, 1, out);
}
// stencil table of the f16 ResnetBlock kernel: per channel PAIR (2p, 2p + 1) twelve floats {w0a, w0b, w1a, w1b, w2a, w2b, w3a, w3b, w4a, w4b,
// bias_a, bias_b} (the kernel multiplies both rows of a pair in one packed instruction), times `scale` (log2(e) for the first stencil)
inline std::vector<float> pack_rh_table(const float* dw_w, const float* dw_b, int C, double scale) {
    std::vector<float> t((size_t)(C / 2) * 12, 0.f);
    for (int m = 0; m < C; ++m) {
        float* row = t.data() + (size_t)(m / 2) * 12 + (m & 1);
        for (int i = 0; i < 5; ++i) row[2 * i] = (float)((double)dw_w[(size_t)m * 5 + i] * scale);
        row[10] = dw_b ? (float)((double)dw_b[m] * scale) : 0.f;
    }
    return t;
}
bool rh_supported(const RhArgs& a);
hipError_t launch_resblock16(const RhArgs& a, hipStream_t s);
struct Conv16Args {           // y = out_scale * (bias + conv(x)) + resid; x, resid, Y, Yact c8 f16; Yf32 [B][M][Tout] f32 row-major
    const void* X; H16Weight w; const float* bias; const void* resid; void* Y; void* Yact; float* Yf32;
    float out_scale, act_scale; int B, M, Tin, Tout, ks, stride, pad;
    // FiLM (seanet.py:518-550, 928-966) behind the conv: y = gamma[b][band] * y + beta[b][band], band = m / (M / bands);
    // film[b * film_stride + 2 * band + {0, 1}], or null
    const float* film = nullptr; int bands = 1, film_stride = 0;
    // up = r > 0: the rows are r phases of Mo = M / r output channels (row p * Mo + m: pack_up16) and row (p, m) at input time l is
    // output channel m at time l * r + p -- the decoder's upsample unit as ONE conv over the input frames; Y / Yact are
    // [B][Mo / 8][Tout * r][8], bias has Mo entries.  No resid / Yf32 in this form.
    // The rows come in blocks of up * up_mb: row = (block, phase, channel in block), channel = block * up_mb + c -- a row block holds ALL phases
    // of its up_mb channels, so that a workgroup owning one produces whole runs of the output (wv_h16.hip conv16u_kernel).
    int up = 0, up_mb = 0;
};
inline int up16_block(int Mo, int r) {                          // channels per row block: the largest multiple of 8 with r * MB <= 256 that divides Mo
    for (int mb = (256 / r) / 8 * 8; mb >= 8; mb -= 8)
        if (Mo % mb == 0) return mb;
    return 0;
}
// host: the decoder's upsample unit ELU -> depth-wise ConvTranspose1d(2r, stride r), right-trimmed by r -> 1x1 (seanet.py:1147-1170,
// conv.py:838-881) composed into a 2-tap conv over the INPUT frames: out[m][r l + p] = b[m] + sum_k pw[m][k] * (ct[k][p] * a[k][l] +
// ct[k][p + r] * a[k][l - 1]), i.e. rows (block, p, m in block) (Conv16Args::up_mb = mb channels per block; mb = Mo: plain (p, m) order),
// taps i = 0 (frame l - 1: ct[k][p + r]) and i = 1 (frame l: ct[k][p]), causal pad 1.
inline std::vector<uint16_t> pack_up16(const float* pw, const float* ct, int Mo, int K, int r, int mb, H16Weight* out) {
    H16Weight w; w.K = K; w.M = Mo * r; w.Kp = round_up(K, 16); w.Mp = round_up(w.M, 32);
    const int nkc = w.Kp / 16;
    w.nchunks = round_up(2 * nkc, 8);
    std::vector<uint16_t> q((size_t)w.nchunks * w.Mp * 16, 0);
    for (int i = 0; i < 2; ++i)
        for (int p = 0; p < r; ++p)
            for (int m = 0; m < Mo; ++m) {
                const size_t row = (size_t)(m / mb) * r * mb + (size_t)p * mb + m % mb;
                for (int k = 0; k < K; ++k)
                    q[(((size_t)((k / 16) * 2 + i) * w.Mp + row) * 2 + ((k >> 3) & 1)) * 8 + (k & 7)] =
                        f32_to_f16_bits(pw[(size_t)m * K + k] * ct[(size_t)k * 2 * r + (i == 0 ? p + r : p)]);
            }
    *out = w;
    return q;
}
hipError_t launch_conv16(const Conv16Args& a, hipStream_t s);
// whole SpecBlock (STFT on the f16 pipe with a two-term split of the waveform -> log-magnitude -> 1x1 -> + x), the spectrogram stays in LDS.
// cosw / sinw: the basis' cos rows f = 0 .. n_fft/2 - 1 and sin rows (row 0 = the Nyquist bin's cos row) as A fragments (pack_stft16);
// pw: the 1x1 [n_fft][F] with K padded to n_fft/2 + 16.  (n_fft, hop) in {(64,1), (128,2), (256,8), (512,40), (1024,320)}, else hipErrorNotSupported.
struct Spec16Args {
    const float* wav; H16Weight cosw, sinw, cosl, sinl, pw; const void* resid; void* Y; void* Yact;   // cosl / sinl: (basis - f16(basis)) * 2^11
    float* Yf32;          // [B][n_fft][Tf] f32 row-major copy of y, or null
    float out_scale, act_scale, c1, c0; int B, T, Tf, n_fft, hop;
};
hipError_t launch_spec16(const Spec16Args& a, hipStream_t s);
// host: basis [2F][n_fft] (cos rows, then sin rows; modules/conv.py:1003-1026) -> the two A-fragment matrices of launch_spec16
inline float f16_bits_to_f32(uint16_t hb) {
    const uint32_t sign = (uint32_t)(hb & 0x8000u) << 16, e = (hb >> 10) & 31u, m = hb & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++sh; } x = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &x, 4);
    return f;
}
// q[0..3] = cos hi, sin hi, cos lo, sin lo; w[0..3] their descriptors (wq unset)
inline void pack_stft16(const float* basis, int n_fft, std::vector<uint16_t> (&q)[4], H16Weight (&w)[4]) {
    const int F = n_fft / 2 + 1, R = n_fft / 2;
    std::vector<float> m[4];
    for (auto& v : m) v.resize((size_t)R * n_fft);
    for (int f = 0; f < R; ++f)
        for (int n = 0; n < n_fft; ++n) {
            const float c = basis[(size_t)f * n_fft + n];
            const float s = f == 0 ? basis[(size_t)(F - 1) * n_fft + n] : basis[(size_t)(F + f) * n_fft + n];
            const size_t i = (size_t)f * n_fft + n;
            m[0][i] = c; m[1][i] = s;
            m[2][i] = (c - f16_bits_to_f32(f32_to_f16_bits(c))) * 2048.f;
            m[3][i] = (s - f16_bits_to_f32(f32_to_f16_bits(s))) * 2048.f;
        }
    for (int k = 0; k < 4; ++k) q[k] = pack_h16(m[k].data(), nullptr, R, n_fft, 1, &w[k]);
}
// detector head, mean-probability output only: L2Norm over channels of Y [B][D][Fr] (f32), composed head GEMM (w: [nb * hop][D] as A
// fragments), sigmoid, mean over time.  hipErrorNotSupported outside D % 16 == 0, D <= 128, nb % 4 == 0, hop % 32 == 0.
hipError_t launch_head16(const float* Y, const H16Weight& w, const float* bc, float* mean_prob, int B, int D, int nb, int hop, int Fr, int T, hipStream_t s);
hipError_t launch_conv_pre16(const float* x, const float* w, const float* bias, void* Y, int B, int C, int T, int ks, float in_scale, hipStream_t s);
hipError_t launch_f32_to_c8(const float* X, void* Y, int B, int C, int T, float scale, int elu, hipStream_t s);
// L2Norm over channels (seanet.py:288-318: y / max(||y||, 1e-12) * sqrt(D)) of a latent [B][D][Fr] f32 -> c8 f16 (the f16 decoder's input)
hipError_t launch_l2norm_c8(const float* X, void* Y, int B, int D, int Fr, hipStream_t s);
// decoder tail on a PRE-ACTIVATED c8 f16 input: out = tanh(out_scale * (b + conv_{C -> 1, ks}(a))) (+ x), first T samples (seanet.py:1177-1202)
hipError_t launch_tail16(const void* A16, const float* w, const float* bias, const float* x, float* out, int B, int C, int Tin, int T, int ks, float out_scale, hipStream_t s);
hipError_t launch_c8_to_f32(const void* X, float* Y, int B, int C, int T, hipStream_t s);

// ---- K2: (identity | DW conv) producer -> 1x1 GEMM -> epilogue --------------------------------
struct DwPwArgs {
    const float* X;       // [B, K, Tin]
    const float* dw_w;    // mode 1: [K, ks]; mode 0: unused
    PwWeight pw;
    const float* bias;    // [M] or null
    float* Y;             // [B, M, Tout]
    int B, Tin, Tout, mode, ks;
    float pre_scale;
    int pre_elu;
    int l2norm;           // normalise over channels * sqrt(M) (needs M <= 128)
    int accumulate;       // Y += out_scale * result
    float out_scale;
    float* Yact;          // optional second output ELU(act_scale * y) (see PwDwArgs::Yact); full aligned tiles and the
    float act_scale;      //   ragged path both write it
};
hipError_t launch_dw_pw(const DwPwArgs& a, hipStream_t s);

// ---- K3a: causal STFT log-magnitude ---------------------------------------------------------
// The 2F = n_fft + 2 basis rows are laid out so that the matrix part has exactly n_fft rows (no
// padding for n_fft = 64 * 2^s): row 0 = cos_0 (DC), row 1 = cos_{F-1} (Nyquist), rows 2f / 2f+1
// = cos_f / sin_f for f = 1 .. F-2.  The two remaining rows, sin_0 (identically zero for a
// reference-built basis) and sin_{F-1} (rounding-level values), are applied as two plain dot
// products per frame next to the matrix loop (`side`), so nothing of the reference's arithmetic
// is dropped.  basis_t[Kp][Mp] is the transposed, zero padded matrix part.
struct StftArgs {
    const float* wav;     // [B, 1, T]
    const float* basis_t; // [Kp][Mp]
    const float* side;    // [2][n_fft]: sin_0 row, sin_{F-1} row
    float* P;             // [B, F, Tf]
    int B, T, Tf, n_fft, hop, F, Mp;
    float mean, inv_std;
    const float* basis_q; // the same matrix in the k-inner layout of the LDS-DMA core, [roundup(n_fft,32)/4][Mp][4] (pack_stft_q); may be null
    int num_m, num_t;     // filled by the launcher
    float c1, c0;         // filled by the launcher: 0.5 ln2 / std, -mean / std
};
// host: basis_t[Kp][Mp] (pack_stft_basis) -> basis_q[roundup(n_fft,32)/4][Mp][4]
inline std::vector<float> pack_stft_q(const std::vector<float>& bt, int n_fft, int Mp) {
    std::vector<float> q((size_t)round_up(n_fft, 32) * Mp, 0.f);
    for (int n = 0; n < n_fft; ++n)
        for (int m = 0; m < Mp; ++m) q[((size_t)(n >> 2) * Mp + m) * 4 + (n & 3)] = bt[(size_t)n * Mp + m];
    return q;
}
hipError_t launch_stft_k1(const StftArgs& a, hipStream_t s);   // wv_k1.hip; hipErrorNotSupported -> round-1 kernel
// ---- whole SpecBlock in one launch (modules/seanet.py:463-511): y = resid + out_scale * (W @ logmag(STFT(wav))), the spectrogram
// stays in LDS.  For the scales whose whole spectrum is one m-tile and whose 1x1 has as many rows (n_fft = M = 64 or 128; Tf % 4 == 0,
// more than 64 frames).  StftArgs::P is not used.  hipErrorNotSupported: run launch_stft_logmag + the SpecBlock add.
struct SpecAddArgs {
    PwWeight pw;          // [M, F] 1x1 (no bias); only the k-inner layout wq is read
    const float* resid;   // [B, M, Tf] x
    float* Y;             // [B, M, Tf] x + out_scale * (W @ P), or null (may alias resid)
    float* Yact;          // [B, M, Tf] ELU(act_scale * y), or null
    float out_scale, act_scale;
};
hipError_t launch_stft_spec(const StftArgs& a, const SpecAddArgs& q, hipStream_t s);   // wv_k1.hip
// host: basis [2F][n_fft] (cos rows, then sin rows; modules/conv.py:1003-1026) -> basis_t, side
inline void pack_stft_basis(const float* basis, int n_fft, std::vector<float>& bt, std::vector<float>& side, int* Mp_out) {
    const int F = n_fft / 2 + 1, Mp = round_up(n_fft, M_ALIGN), Kp = round_up(n_fft, BK);
    bt.assign((size_t)Kp * Mp, 0.f);
    side.assign((size_t)2 * n_fft, 0.f);
    for (int n = 0; n < n_fft; ++n) {
        bt[(size_t)n * Mp + 0] = basis[(size_t)0 * n_fft + n];
        bt[(size_t)n * Mp + 1] = basis[(size_t)(F - 1) * n_fft + n];
        for (int f = 1; f + 1 < F; ++f) {
            bt[(size_t)n * Mp + 2 * f] = basis[(size_t)f * n_fft + n];
            bt[(size_t)n * Mp + 2 * f + 1] = basis[(size_t)(F + f) * n_fft + n];
        }
        side[n] = basis[(size_t)F * n_fft + n];
        side[(size_t)n_fft + n] = basis[(size_t)(2 * F - 1) * n_fft + n];
    }
    if (Mp_out) *Mp_out = Mp;
}
hipError_t launch_stft_logmag(const StftArgs& a, hipStream_t s);

// ---- K4: conv_pre ---------------------------------------------------------------------------
// Yact (optional): second output ELU(act_scale * y), the first ResnetBlock's hoisted prologue
hipError_t launch_conv_pre(const float* x, const float* w, const float* bias, float* Y, float* Yact, float act_scale,
                           int B, int C, int T, int ks, float in_scale, hipStream_t s);

// ---- K5: decoder tail -----------------------------------------------------------------------
hipError_t launch_tail(const float* H, const float* w, const float* bias, const float* x,
                       float* out, int B, int C, int Tin, int T, int ks, float pre_scale,
                       float out_scale, hipStream_t s);

// ---- K6: detector / locator head ------------------------------------------------------------
// wc[D][nb*hop] = sum_o w_last[n][o] * w_rev[d][o][j]; bc[nb] = w_last @ b_rev + b_last.
struct HeadArgs {
    const float* Z;       // [B, D, Fr]
    const float* wc;
    const float* bc;
    float* logits;        // [B, nb, T] or null
    float* mean_prob;     // [B, nb] or null
    int B, D, nb, hop, Fr, T;
};
hipError_t launch_head(const HeadArgs& a, hipStream_t s);

// ---- K7: message MLP + FiLM parameters ------------------------------------------------------
struct FilmArgs {
    const float* msg;     // [rows, msg_dim]
    int msg_rows, msg_dim, E, n_layers, n_out;
    const float* w0; const float* b0;          // [E, msg_dim], [E]
    const float* wl; const float* bl;          // [n_layers, E, E], [n_layers, E]
    const float* wf; const float* bf;          // [n_out, E], [n_out]
    float* film;          // [B, n_out]
    int B;
};
hipError_t launch_film(const FilmArgs& a, hipStream_t s);

// ---- optional per-launch profiling with HIP events on the launch stream ---------------------
// When enabled every launcher brackets its kernel with an event pair; entries aggregate by
// "<kernel symbol>|<role>".  Roles are set by the model plan (e.g. "enc.down_film").
namespace prof {
void enable(bool on);
bool enabled();
void reset();
void set_role(const char* role);
struct Entry { const char* name; long long launches; double ms, flops, bytes; };
int collect(Entry* out, int cap);   // synchronises the recorded events; returns entry count
// RAII bracket used by every launcher: records an event pair around the launch when profiling is on
// (thread-safe; skipped on a capturing stream).
struct Scope {
    hipStream_t s; void* ev_a = nullptr; void* ev_b = nullptr; int key_ = -1; bool armed = false;
    Scope(hipStream_t st, const char* kernel, double flops, double bytes);
    ~Scope();
    Scope(const Scope&) = delete;
    Scope& operator=(const Scope&) = delete;
};
}  // namespace prof

}  // namespace wv
