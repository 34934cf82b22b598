// Device-side pieces shared by the kernel translation units (wv_kernels.hip, wv_k1.hip).
#pragma once
#include "wv_kernels.h"

namespace wv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));   // 4-byte aligned pair (global loads may be unaligned)
constexpr int NT_ = 256;   // threads per workgroup

// ELU(x) = x > 0 ? x : exp(x) - 1, as the median of (x, exp(x) - 1, 0): exp(x) - 1 lies above x on both sides of zero, so for
// x > 0 the middle value is x and for x < 0 it is exp(x) - 1 -- the same values as the select form, one v_med3_f32 instead of a
// compare + select through VCC (f32 matrix work shares the SIMD's lanes with VALU work: every instruction here is matrix time)
__device__ __forceinline__ float elu1(float x) { return __builtin_amdgcn_fmed3f(x, __expf(x) - 1.f, 0.f); }
__device__ __forceinline__ float act(float x, float scale, int elu) {
    x *= scale;
    return elu ? elu1(x) : x;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// B operand for the k-inner core: a thread owns a 2(k) x 4(time) micro-tile -- two 16-byte row
// loads, exactly the coalescing of a plain row copy -- and scatters it as four 8-byte halves of
// the [kq][col] fragments.  Column slots are XOR-swizzled inside each group of 4 so the 16 lanes
// of a ds_write_b64 group land on 8 distinct bank pairs (2-way, free) instead of 2 (8-way).
__device__ __forceinline__ int q_slot(int n) { return (n & ~3) | ((n & 3) ^ ((n >> 3) & 3)); }

struct RowPairLoader {
    const float* base; int K, ld, ncols, c0; float scale; int elu;
    const float* p; int c; bool full, vec;
    __device__ __forceinline__ void init(int cg) {
        c = c0 + 4 * cg;
        full = c >= 0 && c + 3 < ncols;
        vec = full && ((ld & 3) == 0) && ((c & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
        p = base + c;
    }
    __device__ __forceinline__ void fetch2(int k0, float (&raw)[8]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + i;
            if (k < K && vec) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)k * ld);
                raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    raw[4 * i + j] = (k < K && c + j >= 0 && c + j < ncols) ? p[(size_t)k * ld + j] : 0.f;
            }
        }
    }
    __device__ __forceinline__ float xform(float v) const { return act(v, scale, elu); }
    static constexpr int NRAW = 8;
    // raw 2(k) x 4(t) micro-tile -> staged values o[4*i + j] = B[k0+i][t+j]
    __device__ __forceinline__ void finish2(int, const float (&raw)[8], float (&o)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = xform(raw[i]);
    }
};

// The same B operand with the INPUT window in LDS (wv_k1.hip, LDR 6 / 7; per-clip tiles, K a whole number of chunks, Tin % 4 == 0): the
// rows a[k][ls .. ls + LWP) of a chunk arrive by LDS-DMA (coalesced, each piece once per workgroup; frames outside [0, Tin) read zeros =
// the causal padding and the trimmed tail), and a thread builds its 2 (k) x 4 (t) micro-tile from two LDS reads per channel instead of
// narrow global gathers with clamps and selects.  Only the taps still come from global memory (two aligned float4 per channel, L2-resident).
// The values are ConvTrPair's: B[k][t] = fmaf(a[k][l - 1], w[k][ph + r], a[k][l] * w[k][ph]).
template <int RM>                                               // RM = 4: r % 4 == 0 (the four times share l); RM = 2: r == 2
struct ConvTrLds {
    static constexpr int NRAW = RM == 4 ? 16 : 8;
    const float* ct_w; int K, ratio, c0, LWP, ls, km;           // ls: first frame of the window (a multiple of 4, <= the tile's first l - 1); km: chunk depth - 1
    const float* Rcur;                                          // this chunk's rows in LDS: [BKC][LWP]
    int lrel, ph0;
    __device__ __forceinline__ void init(int cg) {
        const int t = max(c0 + 4 * cg, 0);                      // (columns at t < 0 -- the halo of a clip's first tile -- meet zero taps downstream)
        const int l = t / ratio;
        ph0 = t - l * ratio;
        lrel = l - ls;
    }
    template <bool FAST = false>
    __device__ __forceinline__ void fetch2(int k0, float (&raw)[NRAW]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* w = ct_w + (size_t)(k0 + i) * 2 * ratio;
            float* r = raw + i * (NRAW / 2);
            if (RM == 4) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + ph0), w1 = *reinterpret_cast<const f32x4*>(w + ph0 + ratio);
                r[0] = w0.x; r[1] = w0.y; r[2] = w0.z; r[3] = w0.w; r[4] = w1.x; r[5] = w1.y; r[6] = w1.z; r[7] = w1.w;
            } else {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w);               // (w0[0], w0[1], w1[0], w1[1])
                r[0] = wv.x; r[1] = wv.y; r[2] = wv.z; r[3] = wv.w;
            }
        }
    }
    template <bool FAST = false>
    __device__ __forceinline__ void finish2(int k0, const float (&raw)[NRAW], float (&o)[8]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* r = raw + i * (NRAW / 2);
            const float* a = Rcur + ((k0 + i) & km) * LWP + lrel;
            if (RM == 4) {
                const float xa = a[0], xb = a[-1];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * i + e] = fmaf(xb, r[4 + e], xa * r[e]);
            } else {
                const float am = a[-1], a0 = a[0], a1 = a[1];
                o[4 * i + 0] = fmaf(am, r[2], a0 * r[0]); o[4 * i + 1] = fmaf(am, r[3], a0 * r[1]);
                o[4 * i + 2] = fmaf(a0, r[2], a1 * r[0]); o[4 * i + 3] = fmaf(a0, r[3], a1 * r[1]);
            }
        }
    }
};

// log-magnitude feature of one STFT bin (modules/conv.py:1078, seanet.py:487-494):
//   (log(max(sqrt(max(re^2 + im^2, 1e-12)), 1e-5)) - mean) / std.
// sqrt(max(p, 1e-12)) >= 1e-6 and the outer clamp at 1e-5 = sqrt(1e-10) supersedes the inner one, so this is
// 0.5 * log(max(p, 1e-10)) -- one v_log_f32 and one fma instead of a correctly rounded sqrt, a log and two
// more operations per value (the matrix pipe waits for every VALU instruction of the epilogue).  The two forms
// differ by rounding of the log only (<= 2e-7 relative); the parity bar on the feature is 2e-5.
//   c1 = 0.5 * ln(2) / std,  c0 = -mean / std
__device__ __forceinline__ float stft_logmag(float re, float im, float c1, float c0) {
    const float pw = fmaxf(fmaf(re, re, im * im), 1e-10f);
    return fmaf(__builtin_amdgcn_logf(pw), c1, c0);
}

// B operand of the upsample unit for the k-inner core: act(s*x) -> depth-wise ConvTranspose1d(2r, r),
// right-trimmed (modules/conv.py SConvTranspose1d causal trim), produced on the fly:
//   B[k][t] = a(x[k][l]) * w[k][ph] + a(x[k][l-1]) * w[k][ph + r],   l = t / r, ph = t % r.
// A thread owns 2 channels x 4 consecutive output times t (t0 a multiple of 4).  RM picks the
// addressing: RM = 4 (r % 4 == 0): the four times share l and their taps are one aligned float4
// pair -> 2 scalar + 2 vector loads and 2 activations per channel;  RM = 2 (r == 2): three inputs
// and one float4 of taps;  RM = 0: any ratio, per-time scalar gathers.  Loads go to clamped
// (always valid) addresses and the zero-selects happen in finish2, at commit time, so that no
// s_waitcnt sits between issuing the loads and the matrix work.
template <int RM>
struct ConvTrPair {
    static constexpr int NRAW = RM == 4 ? 20 : (RM == 2 ? 14 : (RM == 1 ? 14 : 22));
    const float* Xb; const float* ct_w; const float* ct_wt; int K, Kt, Tin, Tout, c0, ratio; float scale; int elu;
    int t, l0, ph[4], dl[4];
    __device__ __forceinline__ void init(int cg) {
        t = c0 + 4 * cg;                                          // first output time of the micro-tile
        l0 = min(max(t, 0) / ratio, Tin - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int te = max(t + e, 0), le = te / ratio;
            ph[e] = te - le * ratio;
            dl[e] = min(le, Tin - 1) - l0;                         // 0 or 1 for ratio >= 2
        }
    }
    __device__ __forceinline__ void fetch2(int k0, float (&raw)[NRAW]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = min(k0 + i, K - 1);
            const float* xr = Xb + (size_t)k * Tin;
            const float* w = ct_w + (size_t)k * 2 * ratio;
            float* r = raw + i * (NRAW / 2);
            if (RM == 4) {                                         // one input pair, aligned tap vectors
                r[0] = xr[l0]; r[1] = xr[max(l0 - 1, 0)];
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + ph[0]);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + ph[0] + ratio);
                r[2] = w0.x; r[3] = w0.y; r[4] = w0.z; r[5] = w0.w;
                r[6] = w1.x; r[7] = w1.y; r[8] = w1.z; r[9] = w1.w;
            } else if (RM == 2) {                                  // inputs l0-1, l0, l0+1; w = (w0[0], w0[1], w1[0], w1[1])
                r[0] = xr[max(l0 - 1, 0)]; r[1] = xr[l0]; r[2] = xr[min(l0 + 1, Tin - 1)];
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w);
                r[3] = wv.x; r[4] = wv.y; r[5] = wv.z; r[6] = wv.w;
            } else if (RM == 1) {                                  // ratio 1: five inputs, two taps
#pragma unroll
                for (int e = 0; e < 5; ++e) r[e] = xr[min(max(l0 - 1 + e, 0), Tin - 1)];
                r[5] = w[0]; r[6] = w[1];
            } else {                                               // any ratio >= 2: three inputs, per-time taps
                r[0] = xr[max(l0 - 1, 0)]; r[1] = xr[l0]; r[2] = xr[min(l0 + 1, Tin - 1)];
            }
        }
        if (RM == 0) {
            // taps from the transposed copy ct_wt[2r][Kt] (channels contiguous): the two channel rows
            // of a (tap, time) pair are one 8-byte load (k0 is even, Kt is even)
            const int k = min(k0, Kt - 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 a = *reinterpret_cast<const f32x2*>(ct_wt + (size_t)ph[e] * Kt + k);
                const f32x2 b = *reinterpret_cast<const f32x2*>(ct_wt + (size_t)(ph[e] + ratio) * Kt + k);
                raw[3 + e] = a.x; raw[NRAW / 2 + 3 + e] = a.y;
                raw[7 + e] = b.x; raw[NRAW / 2 + 7 + e] = b.y;
            }
        }
    }
    __device__ __forceinline__ void finish2(int k0, const float (&raw)[NRAW], float (&o)[8]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool kv = k0 + i < K;
            const float* r = raw + i * (NRAW / 2);
            float a[5];                                            // activated inputs (each computed once)
            constexpr int NX = RM == 4 ? 2 : (RM == 1 ? 5 : 3);
#pragma unroll
            for (int e = 0; e < NX; ++e) a[e] = act(r[e], scale, elu);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int te = t + e;
                const bool ok = kv && te >= 0 && te < Tout;
                float xa, xb, w0, w1;
                if (RM == 4) { xa = a[0]; xb = a[1]; w0 = r[2 + e]; w1 = r[6 + e]; }
                else if (RM == 2) { xa = a[1 + (e >> 1)]; xb = a[e >> 1]; w0 = r[3 + (e & 1)]; w1 = r[5 + (e & 1)]; }
                else if (RM == 1) { xa = a[1 + e]; xb = a[e]; w0 = r[5]; w1 = r[6]; }
                else { xa = dl[e] ? a[2] : a[1]; xb = dl[e] ? a[1] : a[0]; w0 = r[3 + e]; w1 = r[7 + e]; }
                const float v = fmaf(te >= ratio ? xb : 0.f, w1, xa * w0);
                o[4 * i + e] = ok ? v : 0.f;
            }
        }
    }
};


// Buffer descriptor whose words are forced into scalar registers.  The base pointers below are the same for every lane of a workgroup
// (clip base or tensor base), but the compiler cannot always prove it and then wraps EVERY buffer load / store in a "waterfall" loop
// (4 v_readfirstlane + 2 v_cmp + exec juggling + a branch; 32-64 of them per K1 epilogue).  Only for wave-uniform arguments.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// XCD-aware tile mapping for K1.  Workgroup ids are dealt round-robin over the 8 XCDs (private
// L2 each), so ids L, L+8, L+16, ... share an L2.  We enumerate, per XCD, the m-tiles of ONE
// activation tile back to back: the X window is fetched into that L2 once and reused by all
// M/BM m-tiles instead of crossing the fabric M/BM times.  (Speed only: any placement is correct.)
struct TileId { int m_tile, t_tile, b; bool valid; };
__device__ __forceinline__ TileId decode_tile(const PwDwArgs& p) {
    const unsigned L = blockIdx.x;
    const unsigned xcd = L & 7, j = L >> 3;
    const unsigned m_tile = j % p.num_m, n_idx = (j / p.num_m) * 8 + xcd;
    TileId t;
    t.valid = n_idx < (unsigned)p.num_t * p.B;
    t.m_tile = m_tile; t.t_tile = n_idx % p.num_t; t.b = n_idx / p.num_t;
    return t;
}

}  // namespace wv
