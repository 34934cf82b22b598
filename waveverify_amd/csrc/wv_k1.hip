// K1 on the LDS-DMA core (round 2): Y = epi( DWconv( W1x1 @ B ) + b ), B = activation rows.
//
// What changed against the round-1 core (wv_kernels.hip, kept for ragged shapes and M < 128), and why
// (measured in tools/mfma_peak.hip, profiles/r02_mfma_ceiling_*.txt): the bare v_mfma_f32_32x32x2_f32
// loop runs 155 TFLOP/s and a loop that re-reads both operands from LDS still 150, but staging the B
// operand through registers with a scale -> ELU -> transposing ds_write commit cost 15 % of that at any
// occupancy, and the accumulator -> LDS strip -> stencil epilogue another 5-25 % depending on K.
//
//  * Both operands arrive by LDS-DMA (global_load_lds_dwordx4): no staging registers, no VALU, no
//    s_waitcnt in front of matrix work.  A = packed weights wq[k/4][m][4]; B = the activation rows as
//    they lie in HBM, [k][t] with time innermost.  Out-of-range columns (causal left pad, right edge)
//    are DMA'd from a 16-byte zero constant and rows past K are clamped (their weights are zero), so
//    the loop has no branch.
//  * B keeps its NATURAL layout in LDS.  One ds_read_b128 of row k gives a lane 4 CONSECUTIVE columns;
//    the wave's NT = 4 column tiles are therefore interleaved (tile e holds columns 4j + e) instead of
//    blocked.  Same instruction count as the round-1 k-inner layout (one read feeds 4 MFMAs), no
//    transposition on the write side -- and after the GEMM a lane ALREADY holds 4 consecutive columns
//    of each of its rows (acc[0..3][r]), which is exactly what the depth-wise stencil wants: its right
//    neighbours come from the next lane by DPP (wave_shl:1).  No accumulator round trip through LDS.
//  * The consumer-side prologue (scale -> ELU) is hoisted into the PRODUCER's epilogue, which writes
//    ELU(s*y) as a second output (PwDwArgs::Yact): done once per element instead of once per m-tile
//    workgroup, and the consumer's operand becomes a pure copy.  Units whose input is not available
//    pre-activated (or whose B operand is computed: the upsample unit's ConvTranspose) stage B through
//    registers (LDR >= 1) into the same natural layout with two ds_write_b128.
//
// Accumulation order over k is the round-1 order (lane half h owns k in [4h,4h+4) U [8+4h,12+4h) of
// every 16), so results are bit-identical to the round-1 kernel.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "wv_dev.h"

namespace wv {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float dpp_next(float v) {          // lane i <- lane i+1 (wave_shl:1), lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// Tile: BM = 32 * WM output channels (one 32-row strip per wave; WM = 2, 3, 4 waves) x BN = 32 * NT columns.
template <int NT_, int BKC_, int BM_ = 128>
struct K1 {
    static constexpr int NT = NT_, BN = 32 * NT_, BKC = BKC_, KQ = BKC_ / 4, BM = BM_, WM = BM_ / 32, NTHREADS = 64 * WM;
    static constexpr int A4 = KQ * BM;                      // f32x4 per A stage: [kq][m]
    static constexpr int B4 = BKC * BN / 4;                 // f32x4 per B stage: [k][t]
    static constexpr int STAGE4 = A4 + B4;
    static constexpr int A_PIECES = A4 / 64, B_PIECES = B4 / 64;   // 64-lane DMA pieces per chunk, dealt round-robin to the waves
    static constexpr int A_PER = (A_PIECES + WM - 1) / WM, B_PER_DMA = (B_PIECES + WM - 1) / WM;
    static constexpr int CG = BN / 4;                       // 4-column groups per row
    static constexpr int HLD = BN + 4;                      // generic epilogue: strip row
    static constexpr int NBT = (BKC / 2) * CG;              // register path: 2(k) x 4(t) micro-tiles per chunk
    static constexpr int B_PER = (NBT + NTHREADS - 1) / NTHREADS;
    static_assert(A4 % 64 == 0 && B4 % 64 == 0 && NTHREADS % CG == 0 && 64 % CG == 0, "staging maps");
};

template <int NT> struct NVec;
template <> struct NVec<4> { typedef f32x4 type; };
template <> struct NVec<2> { typedef f32x2 type; };

// ---- loaders ----------------------------------------------------------------------------------
// LDS-DMA.  The WEIGHT operand goes through a buffer descriptor (buffer_load_dwordx4 ... offen lds): the per-lane
// part of its address is a 32-bit voffset computed once per tile, the per-chunk part a scalar soffset -- no vector
// instruction is spent on it inside the loop (f32 MFMAs share the SIMD's lanes with VALU work, so every one counts;
// measured -2 % on the whole step).  The ACTIVATION rows stay on global_load_lds with 64-bit per-lane sources: the
// same descriptor form was measured 12-14 % SLOWER on the bandwidth-bound layers (same box, interleaved runs:
// 9.39 vs 8.21 ms per step on the 96-row tiles), so only their few address instructions remain.  A lane whose
// 4-column group lies outside [0, Tin) reads a 16-byte zero constant (the causal zero padding); rows past K are
// clamped to K - 1 (they meet zero weights).
constexpr int OOB_VOFF = 0x40000000;                            // > any num_records here, and + soffset cannot wrap

template <class C>
struct DmaRows {                        // LDR 0: B = X rows, pure copy (global_load_lds, per-lane 64-bit source)
    const float* src; size_t ld; int K; int rbase;
    __device__ __forceinline__ void init(const float* Xb, int K_, int Tin, int ti0, int wave, int lane) {
        const int col = ti0 + 4 * (lane % C::CG);
        rbase = lane / C::CG;
        const bool inr = col >= 0 && col + 3 < Tin;
        src = inr ? Xb + col : g_zero16;
        ld = inr ? (size_t)Tin : 0;
        K = K_;
        (void)wave;
    }
    // flattened (clip, time) axis: global column g = b * Tv + u, u = t + pad (see k1_kernel)
    __device__ __forceinline__ void init_flat(const float* X, int K_, int Tin, int B, int Tv, int pad, long long g0, int lane) {
        const long long g = g0 + 4 * (lane % C::CG);
        const int b = (int)(g / Tv), t = (int)(g - (long long)b * Tv) - pad;
        rbase = lane / C::CG;
        const bool inr = b < B && t >= 0;                      // t + 3 < Tin holds: Tv, pad and g are multiples of 4
        src = inr ? X + (size_t)b * K_ * Tin + t : g_zero16;
        ld = inr ? (size_t)Tin : 0;
        K = K_;
    }
    __device__ __forceinline__ void issue(int c, f32x4* Bst, int wave) const {
#pragma unroll
        for (int i = 0; i < C::B_PER_DMA; ++i) {
            const int pi = wave + i * C::WM;
            if (C::B_PIECES % C::WM == 0 || pi < C::B_PIECES) {
                const int k = min(c * C::BKC + pi * (64 / C::CG) + rbase, K - 1);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)k * ld), (lptr_t)(Bst + pi * 64), 16, 0, 0);
            }
        }
    }
};

template <class C>
struct DmaA {                           // A = packed weights wq[kq][Mp] (f32x4), rows m0 .. m0 + BM
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[C::A_PER];
    int chunk_bytes;                    // KQ * Mp * 16
    __device__ __forceinline__ void init(const f32x4* wq, int Mp, int m0, int nchunks, int wave, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(wq + m0), 0, (nchunks * C::KQ * Mp - m0) * 16, 0x00020000);
        chunk_bytes = C::KQ * Mp * 16;
#pragma unroll
        for (int i = 0; i < C::A_PER; ++i) {
            const int idx = (wave + i * C::WM) * 64 + lane;    // piece = 64 consecutive fragments of the [kq][m] stage
            voff[i] = ((idx / C::BM) * Mp + idx % C::BM) * 16;
        }
    }
    __device__ __forceinline__ void issue(int c, f32x4* Ast, int wave) const {
        const int soff = c * chunk_bytes;
#pragma unroll
        for (int i = 0; i < C::A_PER; ++i) {
            const int pi = wave + i * C::WM;
            if (C::A_PIECES % C::WM == 0 || pi < C::A_PIECES) {
                const int vo = voff[i];        // local copy: passing the dependent-size member array element directly makes
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(Ast + pi * 64), 16, vo, soff, 0, 0);   // hipcc's host pass drop the kernel stub
            }
        }
    }
};

// ---- epilogue -----------------------------------------------------------------------------------
// EPI 0: the ResnetBlock stencil (k5, stride 1, dilation 1, pad 4) straight from the accumulators.
// EPI 1: any (ks <= 16, stride, dilation): rows go through a wave-private LDS strip (one 16/8-byte
//        write per lane and row) and the taps are gathered from there.
// begin() runs before the GEMM (row table, first residual rows), finish() after it.
template <class C, int EPI, int RES>
struct K1Epi {
    static constexpr int NT = C::NT, HLD = C::HLD;
    // row table: taps (padded to a multiple of 4), then bias, FiLM gamma, beta.  The r = 2 / r = 4 stencils keep it at 8 / 12 floats per
    // row instead of 20: with it the K <= 128 downsample stages fit a fourth workgroup per CU (36-38 KB instead of 42)
    static constexpr int NTAP = EPI == 0 ? 5 : (EPI == 2 ? 4 : (EPI == 4 ? 8 : 16));
    static constexpr int WLD = EPI == 0 ? 8 : NTAP + 4;
    static constexpr int TABLE_FLOATS = C::BM * WLD;
    // RES: 0 none, 1 y = resid + out_scale y, 2 (training) y = y ELU'(out_scale resid) out_scale [+ resid2], 4 (training) no residual,
    // the raw 1x1 output H is stored next to y (p.Yraw), 5 (training) as 4 plus p.Ysum = resid + s y, s = out_scale * scale_ptr[0]
    static constexpr bool HASR = RES == 1 || RES == 2 || RES == 5;
    static constexpr bool RAWH = RES == 4 || RES == 5;
    static constexpr int RP = HASR ? 4 : 0;                      // residual rows in flight per lane
    typedef typename NVec<NT>::type ovec;
    int M, m0, b, to0, lane, wave, half, q, o, to;
    bool act_lane, vec;
    const float* Rb; float* Yb; float* Ab; float* Wl;
    const float* Rb2; float* Hb; float* Sb;                      // RES 2: addend; RES 4 / 5: raw 1x1 output; RES 5: residual sum
    int voff0, nrec;                                             // EPI 0: lane's first byte offset (or an out-of-range marker), buffer size
    float fgam, fbet;                                            // EPI 8, flat tiling: this lane's FiLM scalars

    __device__ __forceinline__ int row_of(int r) const { return 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half; }
    __device__ __forceinline__ ovec load_res(const PwDwArgs& p, int r) const {
        ovec v;
#pragma unroll
        for (int e = 0; e < NT; ++e) v[e] = 0.f;
        const int gm = m0 + row_of(r);
        if (act_lane && gm < M) {
            const float* rp = Rb + (size_t)gm * p.Tout + to;
            if (vec) v = *reinterpret_cast<const ovec*>(rp);
            else {
#pragma unroll
                for (int e = 0; e < NT; ++e)
                    if (o + e < p.tto && to + e < p.Tout) v[e] = rp[e];
            }
        }
        return v;
    }
    __device__ __forceinline__ void store(const PwDwArgs& p, int gm, const float (&y)[NT]) const {
        const size_t yo = (size_t)gm * p.Tout + to;
        if (Yb) {
            if (vec) { ovec v; _Pragma("unroll") for (int e = 0; e < NT; ++e) v[e] = y[e]; *reinterpret_cast<ovec*>(Yb + yo) = v; }
            else {
#pragma unroll
                for (int e = 0; e < NT; ++e)
                    if (o + e < p.tto && to + e < p.Tout) Yb[yo + e] = y[e];
            }
        }
        if (Ab) {
            float a[NT];
#pragma unroll
            for (int e = 0; e < NT; ++e) a[e] = elu1(y[e] * p.act_scale);
            if (vec) { ovec v; _Pragma("unroll") for (int e = 0; e < NT; ++e) v[e] = a[e]; *reinterpret_cast<ovec*>(Ab + yo) = v; }
            else {
#pragma unroll
                for (int e = 0; e < NT; ++e)
                    if (o + e < p.tto && to + e < p.Tout) Ab[yo + e] = a[e];
            }
        }
    }
    __device__ __forceinline__ void begin(const PwDwArgs& p, float* table, int m0_, int b_, int to0_, long long gflat = 0) {
        M = p.pw.M; m0 = m0_; b = b_; to0 = to0_;
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        half = lane >> 5; q = lane & 31;
        Wl = table;
        const size_t bo = (size_t)b * M * p.Tout;
        Yb = p.Y ? p.Y + bo : nullptr;
        Ab = p.Yact ? p.Yact + bo : nullptr;
        Rb = (HASR && p.resid) ? p.resid + bo : nullptr;
        Rb2 = (RES == 2 && p.resid2) ? p.resid2 + bo : nullptr;
        Hb = (RAWH && p.Yraw) ? p.Yraw + bo : nullptr;
        Sb = (RES == 5 && p.Ysum) ? p.Ysum + bo : nullptr;
        o = NT * q; to = to0 + o;
        act_lane = o < p.tto && to < p.Tout;
        vec = act_lane && to + NT - 1 < p.Tout && o + NT - 1 < p.tto && (p.Tout % NT) == 0;
        nrec = M * p.Tout * 4;
        voff0 = act_lane ? ((m0 + 32 * wave + 4 * half) * p.Tout + to) * 4 : 0x7f000000;
        if (EPI == 0 && p.flat) {
            // flattened tiling: this lane's outputs sit at global columns gflat + pad + NT*q .. of the padded
            // (clip, time) axis; whole-tensor buffers (rows are whole: M % BM == 0, checked by the launcher)
            const long long gc = gflat + p.pad + o;
            const int bo2 = (int)(gc / p.Tv), t = (int)(gc - (long long)bo2 * p.Tv) - p.pad;
            const bool ok = o < p.tto && bo2 < p.B && t >= 0;
            Yb = p.Y; Ab = p.Yact; Rb = HASR ? p.resid : nullptr; Rb2 = RES == 2 ? p.resid2 : nullptr; Hb = RAWH ? p.Yraw : nullptr; Sb = RES == 5 ? p.Ysum : nullptr;
            nrec = (int)((long long)p.B * M * p.Tout * 4);
            voff0 = ok ? (int)((((long long)bo2 * M + m0 + 32 * wave + 4 * half) * p.Tout + t) * 4) : (int)0x80000000u;
        }
        if (EPI == 8 && p.flat) {
            // this lane's output: flat index gflat / 8 + q / 2 -> (clip, frame); FiLM scalars of that clip (the tile lies in one band)
            const int TvO = p.Tv >> 3;
            const long long go = gflat / 8 + (q >> 1);
            const int bo2 = (int)(go / TvO), n = (int)(go - (long long)bo2 * TvO);
            const bool ok = (q & 1) == 0 && (q >> 1) < p.tto && bo2 < p.B && n < p.Tout;
            Yb = p.Y; Ab = p.Yact;
            nrec = (int)((long long)p.B * M * p.Tout * 4);
            voff0 = ok ? (int)((((long long)bo2 * M + m0 + 32 * wave + 4 * half) * p.Tout + n) * 4) : (int)0x80000000u;
            fgam = 1.f; fbet = 0.f;
            if (p.film) {
                const float* fl = p.film + (size_t)min(bo2, p.B - 1) * p.film_stride + 2 * (m0 / (M / p.bands));
                fgam = fl[0]; fbet = fl[1];
            }
        }
        const bool lane_film = EPI == 8 && p.flat;                   // flat tiles span clips: FiLM comes per lane, the table holds (1, 0)
        const int bw = p.film ? (M / p.bands) : 1;
        const float* filmb = (p.film && !lane_film) ? p.film + (size_t)b * p.film_stride : nullptr;
        for (int m = tid; m < C::BM; m += C::NTHREADS) {
            const int gm = m0 + m;
            float* row = Wl + m * WLD;
#pragma unroll
            for (int i = 0; i < NTAP; ++i) row[i] = (gm < M && i < p.ks) ? p.dw_w[(size_t)gm * p.ks + i] : 0.f;
            float gam = 1.f, bet = 0.f;
            if (filmb && gm < M) { const int band = gm / bw; gam = filmb[2 * band]; bet = filmb[2 * band + 1]; }
            row[NTAP] = (gm < M && p.dw_b) ? p.dw_b[gm] : 0.f;
            row[NTAP + 1] = gam; row[NTAP + 2] = bet;
            if (EPI != 0) row[WLD - 1] = 0.f;
        }
    }

    // EPI 0 addressing: buffer descriptors over ONE clip's [M][Tout] block and a per-lane byte offset that is
    // out of range for lanes without outputs.  Rows past M land beyond num_records as well, so loads return 0
    // and stores are dropped by the hardware range check: the loop below has no mask, no branch per element
    // and no 64-bit address arithmetic (every VALU cycle here is taken from the matrix pipe, which shares
    // the SIMD's lanes with f32 VALU work -- profiles/r02_f32mfma_valu_coissue.txt).
    typedef unsigned uvec __attribute__((ext_vector_type(NT)));
    __device__ __forceinline__ ovec buf_load(__amdgpu_buffer_rsrc_t r, int off) const {
        if constexpr (NT == 4) return __builtin_bit_cast(ovec, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        else return __builtin_bit_cast(ovec, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    }
    __device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, int off, ovec v) const {
        if constexpr (NT == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, v), r, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, v), r, off, 0, 0);
    }

    __device__ __forceinline__ void finish(f32x16 (&acc)[NT], const PwDwArgs& p, float* strips) {
        if constexpr (EPI == 0) {
            constexpr int NSH = 4 / NT;                          // lane shifts that bring 4 more columns
            const int clip_bytes = nrec;
            const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(Yb ? Yb : p.Yact, Yb ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(Ab ? Ab : p.Yact, Ab ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rR = uniform_rsrc(Rb ? Rb : p.X, Rb ? clip_bytes : 0);
            const int row_bytes = p.Tout * 4;
            const int voff = voff0;
            const float* Wrow = Wl + (32 * wave + 4 * half) * 8;
            ovec res4[HASR ? 4 : 1];                             // RES 1 / 2 instantiations always have a residual operand
            ovec add4[RES == 2 ? 4 : 1];                         // RES 2: the optional addend (a null one reads zeros: zero-record buffer)
            const __amdgpu_buffer_rsrc_t rR2 = uniform_rsrc(Rb2 ? Rb2 : p.X, Rb2 ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rH = uniform_rsrc(Hb ? Hb : p.Y, Hb ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rS = uniform_rsrc(Sb ? Sb : p.Y, Sb ? clip_bytes : 0);
            const float sres = (RES == 5 && p.scale_ptr) ? p.out_scale * p.scale_ptr[0] : p.out_scale;
            if constexpr (HASR) {
#pragma unroll
                for (int r = 0; r < 4; ++r) res4[r] = buf_load(rR, voff + ((r & 3) + 8 * (r >> 2)) * row_bytes);
            }
            if constexpr (RES == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) add4[r] = buf_load(rR2, voff + ((r & 3) + 8 * (r >> 2)) * row_bytes);
            }
            // The row table of step r+1 is requested during step r; a scheduling barrier per step keeps the
            // compiler from hoisting all 16 steps' loads to the top (128 live registers, spills).
            f32x4 w0n = *reinterpret_cast<const f32x4*>(Wrow), w1n = *reinterpret_cast<const f32x4*>(Wrow + 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2);           // row of this step inside the lane half's 32-row strip
                const f32x4 w0 = w0n, w1 = w1n;
                if (r + 1 < 16) {
                    const int cn = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
                    w0n = *reinterpret_cast<const f32x4*>(Wrow + cn * 8);
                    w1n = *reinterpret_cast<const f32x4*>(Wrow + cn * 8 + 4);
                }
                // columns NT*q .. NT*q + NT + 3 of this row: own registers + the next lane(s) by DPP
                // (all 64 lanes execute the DPP moves: there is no divergence anywhere in this loop)
                float hh[NT + 4], cur[NT];
#pragma unroll
                for (int e = 0; e < NT; ++e) { cur[e] = acc[e][r]; hh[e] = cur[e]; }
#pragma unroll
                for (int s2 = 1; s2 <= NSH; ++s2) {
#pragma unroll
                    for (int e = 0; e < NT; ++e) { cur[e] = dpp_next(cur[e]); hh[s2 * NT + e] = cur[e]; }
                }
                const int off = voff + cr * row_bytes;
                ovec y;
#pragma unroll
                for (int e = 0; e < NT; ++e) {
                    float v = fmaf(w0.x, hh[e], w1.y);                           // bias + 5 taps
                    v = fmaf(w0.y, hh[e + 1], v); v = fmaf(w0.z, hh[e + 2], v);
                    v = fmaf(w0.w, hh[e + 3], v); v = fmaf(w1.x, hh[e + 4], v);
                    y[e] = v;
                }
                if constexpr (HASR) {
                    const ovec rr = res4[r & 3];
                    ovec rsum;
                    (void)rsum;
                    if (r + 4 < 16) res4[r & 3] = buf_load(rR, voff + (((r + 4) & 3) + 8 * ((r + 4) >> 2)) * row_bytes);
#pragma unroll
                    for (int e = 0; e < NT; ++e) {
                        if constexpr (RES == 2) {                // training: dx = da * ELU'(s x) * s, the residual operand is x, out_scale = s
                            const float z = p.out_scale * rr[e];
                            y[e] = y[e] * (z > 0.f ? 1.f : __expf(z)) * p.out_scale;
                        } else if constexpr (RES == 5) {         // training forward: the branch output y stays, the block output goes to Ysum
                            rsum[e] = fmaf(y[e], sres, rr[e]);
                        } else y[e] = fmaf(y[e], p.out_scale, rr[e]);
                    }
                    if constexpr (RES == 5) buf_store(rS, off, rsum);
                }
                if constexpr (RES == 2) {                        // ... + the identity shortcut's gradient
                    const ovec ad = add4[r & 3];
                    if (r + 4 < 16) add4[r & 3] = buf_load(rR2, voff + (((r + 4) & 3) + 8 * ((r + 4) >> 2)) * row_bytes);
#pragma unroll
                    for (int e = 0; e < NT; ++e) y[e] += ad[e];
                }
                if constexpr (RAWH) {                            // the stencil's input at the output's own time: hh[e + 4] = H[t]
                    ovec hraw;
#pragma unroll
                    for (int e = 0; e < NT; ++e) hraw[e] = hh[e + 4];
                    buf_store(rH, off, hraw);
                }
                if (Yb) buf_store(rY, off, y);
                if (Ab) {
                    ovec a;
#pragma unroll
                    for (int e = 0; e < NT; ++e) a[e] = elu1(y[e] * p.act_scale);
                    buf_store(rA, off, a);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (EPI == 5) {
            // ks = 10, stride = 5, dil = 1 (the net's r = 5 downsample): the strip form with compile-time taps -- three 16-byte reads
            // of the row's taps and one of (bias, gamma, beta) instead of a run-time tap loop with a table read per tap
            float* Hw = strips + wave * (4 * HLD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* strip = Hw + (r & 1) * 2 * HLD;
                ovec hv;
#pragma unroll
                for (int e = 0; e < NT; ++e) hv[e] = acc[e][r];
                *reinterpret_cast<ovec*>(strip + half * HLD + NT * q) = hv;
                const int row = row_of(r), gm = m0 + row;
                const float* wt = Wl + row * WLD;
                const f32x4 ta = *reinterpret_cast<const f32x4*>(wt), tb = *reinterpret_cast<const f32x4*>(wt + 4);
                const f32x4 tc = *reinterpret_cast<const f32x4*>(wt + 8), td = *reinterpret_cast<const f32x4*>(wt + 16);
                if (gm < M && q < p.tto && to0 + q < p.Tout) {
                    const float* hp = strip + half * HLD + p.off + q * 5;
                    float y = td.x;
                    y = fmaf(ta.x, hp[0], y); y = fmaf(ta.y, hp[1], y); y = fmaf(ta.z, hp[2], y); y = fmaf(ta.w, hp[3], y);
                    y = fmaf(tb.x, hp[4], y); y = fmaf(tb.y, hp[5], y); y = fmaf(tb.z, hp[6], y); y = fmaf(tb.w, hp[7], y);
                    y = fmaf(tc.x, hp[8], y); y = fmaf(tc.y, hp[9], y);
                    y = fmaf(y, td.y, td.z);
                    const size_t ro = (size_t)gm * p.Tout + to0 + q;
                    if (Yb) Yb[ro] = y;
                    if (Ab) Ab[ro] = elu1(y * p.act_scale);
                }
            }
        } else if constexpr (EPI >= 2) {
            // Downsample stencil (ks = 2R, stride R, pad R; R = EPI in {2, 4, 8}; seanet.py:733-772) + FiLM, from
            // the accumulators: a lane holds H columns 4q..4q+3, the rest of an output's 2R taps sits in the
            // next 1 (R = 2, 4) or 3 (R = 8) lanes and comes by DPP.  Outputs per lane: R = 2 -> 2q, 2q+1 (one
            // 8-byte store), R = 4 -> q, R = 8 -> q/2 on the even lanes.  Taps are summed in ascending order like
            // the generic path (bit-identical).  Addressing and masking as in EPI 0 (range-checked buffers).
            static_assert(NT == 4 && RES == 0, "strided DPP epilogue: 128-column windows, no residual");
            constexpr int R = EPI;
            const bool flat = R == 8 && p.flat;
            const int clip_bytes = flat ? nrec : M * p.Tout * 4;
            const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(Yb ? Yb : p.Yact, Yb ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(Ab ? Ab : p.Yact, Ab ? clip_bytes : 0);
            const int row_bytes = p.Tout * 4;
            const int o0 = R == 2 ? 2 * q : (R == 4 ? q : q >> 1);       // first output of this lane inside the tile
            const bool lane_ok = (R != 8 || (q & 1) == 0) && o0 + (R == 2 ? 1 : 0) < p.tto && to0 + o0 + (R == 2 ? 1 : 0) < p.Tout;
            const int voff = flat ? voff0 : (lane_ok ? ((m0 + 32 * wave + 4 * half) * p.Tout + to0 + o0) * 4 : 0x7f000000);
            const float* Wrow = Wl + (32 * wave + 4 * half) * WLD;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2);
                constexpr int NL = R == 8 ? 4 : 2;                   // lanes an output's taps span
                float hh[4 * NL], cur[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { cur[e] = acc[e][r]; hh[e] = cur[e]; }
#pragma unroll
                for (int l = 1; l < NL; ++l) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cur[e] = dpp_next(cur[e]); hh[4 * l + e] = cur[e]; }
                }
                const float* wt = Wrow + cr * WLD;
                float w[2 * R];
#pragma unroll
                for (int i = 0; i < 2 * R; i += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wt + i);
                    w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
                }
                const f32x4 bgb = *reinterpret_cast<const f32x4*>(wt + NTAP);   // bias, gamma, beta
                const int off = voff + cr * row_bytes;
                constexpr int NO = R == 2 ? 2 : 1;
                float y[NO];
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    const int base = R == 2 ? 2 + 2 * j : 0;            // first H column of output j relative to 4q (off = 2 for R = 2)
                    float v = bgb.x;
#pragma unroll
                    for (int i = 0; i < 2 * R; ++i) v = fmaf(w[i], hh[base + i], v);
                    y[j] = fmaf(v, bgb.y, bgb.z);
                    if (R == 8 && flat) y[j] = fmaf(y[j], fgam, fbet);       // table holds (1, 0) then: same rounding as the one fma
                }
                if constexpr (R == 2) {
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    if (Yb) __builtin_amdgcn_raw_buffer_store_b64(u2{__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1])}, rY, off, 0, 0);
                    if (Ab) __builtin_amdgcn_raw_buffer_store_b64(u2{__builtin_bit_cast(unsigned, elu1(y[0] * p.act_scale)),
                                                                        __builtin_bit_cast(unsigned, elu1(y[1] * p.act_scale))}, rA, off, 0, 0);
                } else {
                    if (Yb) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[0]), rY, off, 0, 0);
                    if (Ab) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, elu1(y[0] * p.act_scale)), rA, off, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float* Hw = strips + wave * (4 * HLD);
            const int ks = p.ks;
            const int no = (p.tto + 31) / 32;                     // consecutive outputs per lane
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* strip = Hw + (r & 1) * 2 * HLD;
                ovec hv;
#pragma unroll
                for (int e = 0; e < NT; ++e) hv[e] = acc[e][r];
                *reinterpret_cast<ovec*>(strip + half * HLD + NT * q) = hv;
                const int row = row_of(r), gm = m0 + row;
                if (gm >= M) continue;
                const float* wt = Wl + row * WLD;
                const float bias = wt[16], gam = wt[17], bet = wt[18];
                const float* hrow = strip + half * HLD + p.off;
                const size_t ro = (size_t)gm * p.Tout + to0;
                for (int e = 0; e < no; ++e) {
                    const int oo = q * no + e;
                    if (oo >= p.tto || to0 + oo >= p.Tout) break;
                    const float* hp = hrow + oo * p.stride;
                    float y = bias;
                    for (int i = 0; i < ks; ++i) y = fmaf(wt[i], hp[i * p.dil], y);
                    y = fmaf(y, gam, bet);
                    if (RES == 1 && Rb) y = fmaf(y, p.out_scale, Rb[ro + oo]);
                    if (Yb) Yb[ro + oo] = y;
                    if (Ab) Ab[ro + oo] = elu1(y * p.act_scale);
                }
            }
        }
    }
};

// LDR: 0 DMA copy | 1 registers, scale -> ELU | 2,3,4,5 registers, DW ConvTranspose producer (RM 4,2,1,0)
template <int LDR> struct LdrSel { typedef RowPairLoader type; };
template <> struct LdrSel<2> { typedef ConvTrPair<4> type; };
template <> struct LdrSel<3> { typedef ConvTrPair<2> type; };
template <> struct LdrSel<4> { typedef ConvTrPair<1> type; };
template <> struct LdrSel<5> { typedef ConvTrPair<0> type; };
template <> struct LdrSel<6> { typedef ConvTrLds<4> type; };        // upsample unit, input window through LDS (per-clip tiles)
template <> struct LdrSel<7> { typedef ConvTrLds<2> type; };
template <class C> constexpr int k1_lwp(int ldr) { return ldr == 7 ? (C::BN / 2 + 6 + 3) / 4 * 4 : (C::BN / 4 + 6 + 3) / 4 * 4; }   // LDS row length of the input window (frames)

// One workgroup per (m-tile, time-tile, clip); the 1-D grid is mapped XCD-aware (decode_tile, wv_dev.h): the
// m-tiles of one activation window are adjacent on one XCD, whose L2 then fetches the window once.
// (A persistent form with the next tile's head issued before the epilogue was measured and dropped: the
// workgroup relaunch is not what limits this kernel -- at K = 384 the matrix pipe is busy 80 % of all SIMD
// cycles at the 2.05 GHz the chip holds under this load, profiles/r02_k1_sq_counters.txt -- and the second
// set of per-tile state cost the fourth resident wave per SIMD, which short-K layers need to hide the DMA
// latency.)
template <class C, int EPI, int LDR, int RES, int NS>
__global__ __launch_bounds__(C::NTHREADS, LDR >= 2 ? (C::B_PER > 1 ? 2 : 3) : 4) void k1_kernel(PwDwArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool REG = LDR != 0;
    typedef K1Epi<C, EPI, RES> Epi;
    typedef typename LdrSel<LDR>::type LB;
    typedef typename NVec<C::NT>::type bvec;
    // Tile decode.  Per-clip tiling: (m-tile, time tile, clip).  Flattened tiling (p.flat; k5 stencil + DMA operand
    // only): the clips' time axes, each with its `pad` causal zero columns in front, are laid end to end (period
    // Tv = T + pad, a multiple of 4) and tiles of tto = BN - 4 outputs run straight across the clip boundaries, so
    // the only columns computed twice are the 4 halo columns per tile and the 4 pad columns per clip: 4 % instead
    // of 9 % (T = 2000) or 12 % (T = 400, which also moves from 64- to 128-column windows).
    // The r = 8 downsample (EPI 8; 50 outputs per clip = 3.3 tiles of 15) is flattened the same way over its INPUT axis: period
    // Tv = Tin + 8, a multiple of 8, so a clip contributes Tv / 8 flat outputs, the last of which straddles the next clip's pad and is dropped.
    constexpr bool FLAT_OK = EPI == 0 || EPI == 8;
    TileId tile = decode_tile(p);
    if (FLAT_OK && p.flat) {
        const unsigned L = blockIdx.x, j = L >> 3;
        tile.m_tile = j % p.num_m; tile.t_tile = (j / p.num_m) * 8 + (L & 7); tile.b = 0;
        tile.valid = tile.t_tile < p.num_t;                    // num_t = flat tile count
    }
    if (!tile.valid) return;
    const long long gflat = (long long)tile.t_tile * p.tto * p.stride;      // flat tiling: first global (input) column of this tile
    f32x4* S4 = reinterpret_cast<f32x4*>(smem);
    float* table = smem + NS * C::STAGE4 * 4;
    static_assert(NS == 2 || (NS == 3 && LDR == 0), "three stages: DMA path only");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i31 = lane & 31;
    const int cg = tid % C::CG;                                // register path: columns 4cg..; rows 2kp, 2kp+1 per micro-tile
    const int K = p.pw.K, Mp = p.pw.Mp;
    const int nchunks = (K + C::BKC - 1) / C::BKC;
    const f32x4* wq = reinterpret_cast<const f32x4*>(p.pw.wq);
    const int m0 = tile.m_tile * C::BM, b = tile.b;
    const int to0 = tile.t_tile * p.tto;
    const int ti0 = to0 * p.stride - p.pad - p.off;
    const float* Xb = p.X + (size_t)b * K * p.Tin;

    constexpr int LWP = LDR >= 6 ? k1_lwp<C>(LDR) : 4;           // LDS row length of the upsample unit's input window (LDR 6 / 7)
    Epi epi;
    epi.begin(p, table, m0, b, to0, gflat);
    DmaRows<C> db{};
    DmaA<C> da;
    da.init(wq, Mp, m0, nchunks, wave, lane);
    LB lb{};
    float raw[REG ? C::B_PER : 1][REG ? LB::NRAW : 1];
    if constexpr (LDR == 0) {
        if (FLAT_OK && p.flat) db.init_flat(p.X, K, p.Tin, p.B, p.Tv, p.pad, gflat, lane);
        else db.init(Xb, K, p.Tin, ti0, wave, lane);
    }
    else {
        // register path: per-clip tiles share one clip base and window start; with flat tiling every thread derives
        // the clip and the local time of its own 4-column group (c0 is chosen so that the loader's c0 + 4*cg is it)
        const float* Xt = Xb;
        int c0 = ti0, ncols = p.Tin, tout = p.Tout;
        if (EPI == 0 && p.flat) {
            const long long g = gflat + 4 * cg;
            const int bb = (int)(g / p.Tv), t = (int)(g - (long long)bb * p.Tv) - p.pad;
            const bool inb = bb < p.B;
            Xt = p.X + (size_t)(inb ? bb : 0) * K * p.Tin;
            c0 = t - 4 * cg; ncols = inb ? p.Tin : 0; tout = inb ? p.Tout : 0;
        }
        if constexpr (LDR == 1) lb = LB{Xt, K, p.Tin, ncols, c0, p.pre_scale, p.pre_elu, nullptr, 0, false, false};
        else if constexpr (LDR >= 6) {
            const int lfirst = max(c0, 0) / p.ratio;
            lb = LB{p.ct_w, K, p.ratio, c0, LWP, ((lfirst - 1) & ~3), C::BKC - 1, nullptr, 0, 0};
        }
        else lb = LB{Xt, p.ct_w, p.ct_wt, K, p.pw.Kp, p.Tin, tout, c0, p.ratio, p.pre_scale, p.pre_elu, 0, 0, {}, {}};
    }
    // LDR 6 / 7: the input window of a chunk ([BKC rows][LWP frames] from frame lb.ls) by LDS-DMA into one of two buffers behind the row table
    float* Rbuf = table + Epi::TABLE_FLOATS;
    auto issue_rows = [&](int c, int buf) {
        if constexpr (LDR >= 6) {
            constexpr int PPR = LWP / 4, NPIECE = C::BKC * PPR, NINST = (NPIECE + 63) / 64, PER = (NINST + C::WM - 1) / C::WM;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int inst = wave + i * C::WM;
                const int q = inst * 64 + lane;
                const int row = q / PPR, f = lb.ls + 4 * (q - row * PPR);
                const bool inr = f >= 0 && f < p.Tin;            // Tin % 4 == 0 and ls % 4 == 0: a piece is all inside or all outside
                const float* src = inr ? Xb + (size_t)(c * C::BKC + min(row, C::BKC - 1)) * p.Tin + f : g_zero16;
                if (inst < NINST && q < NPIECE)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Rbuf + (size_t)buf * C::BKC * LWP + inst * 256), 16, 0, 0);
            }
        }
    };

    auto issue = [&](int c, int st) {
        f32x4* S = S4 + st * C::STAGE4;
        da.issue(c, S, wave);
        if constexpr (!REG) db.issue(c, S + C::A4, wave);
    };
    auto fetch = [&](int c) {
        if constexpr (LDR >= 6) { if (c + 1 < nchunks) issue_rows(c + 1, (c + 1) & 1); }   // the window of the chunk after this one (its buffer was read by the last commit)
        if constexpr (REG) {
#pragma unroll
            for (int r = 0; r < C::B_PER; ++r) {
                const int idx = tid + r * C::NTHREADS;
                if (C::NBT % C::NTHREADS == 0 || idx < C::NBT) lb.fetch2(c * C::BKC + 2 * (idx / C::CG), raw[r]);
            }
        }
    };
    auto commit = [&](int c, int st) {
        if constexpr (LDR >= 6) lb.Rcur = Rbuf + (size_t)(c & 1) * C::BKC * LWP;
        if constexpr (REG) {
            f32x4* Bq = S4 + st * C::STAGE4 + C::A4;
#pragma unroll
            for (int r = 0; r < C::B_PER; ++r) {
                const int idx = tid + r * C::NTHREADS;
                if (!(C::NBT % C::NTHREADS == 0 || idx < C::NBT)) continue;
                const int kp = idx / C::CG;
                float o[8];
                lb.finish2(c * C::BKC + 2 * kp, raw[r], o);
                Bq[(2 * kp) * C::CG + cg] = f32x4{o[0], o[1], o[2], o[3]};
                Bq[(2 * kp + 1) * C::CG + cg] = f32x4{o[4], o[5], o[6], o[7]};
            }
        }
    };
    // Pipeline depth NS.  2 stages: chunk c+1 is fetched while chunk c computes, one __syncthreads() per chunk
    // (which drains the DMA).  3 stages (DMA path): chunk c+2 is in flight while chunk c computes and survives
    // the barrier -- raw s_barrier behind a COUNTED vmcnt.  Measured (kbench, B = 256): +5..8 % on the
    // matrix-bound K >= 256 layers with 128-column windows (a chunk is ~1 us of matrix work, about one HBM round
    // trip), -3..19 % where the layer is bandwidth-bound or the third stage costs a resident workgroup (K = 128,
    // 64-column windows with BK = 32, 64/96-row tiles) -- the launcher picks per layer.
    constexpr int PIECES_MIN = C::A_PIECES / C::WM + C::B_PIECES / C::WM;   // DMA instructions a wave issues per chunk, at least
    f32x16 acc[C::NT];
#pragma unroll
    for (int e = 0; e < C::NT; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    issue(0, 0);
    if constexpr (NS == 2) {
        if constexpr (LDR >= 6) {
            // Window pipeline: fetch(c) requests the rows of chunk c + 1 while chunk c - 1 computes, commit(c + 1) reads them one barrier
            // later.  Here: chunk 0's rows, waited for and made visible, before the first commit.
            issue_rows(0, 0);
            __syncthreads();
        }
        if constexpr (REG) { lb.init(cg); fetch(0); commit(0, 0); }
        __syncthreads();                                       // first chunk landed, table visible
    } else {
        if (nchunks > 1) { issue(1, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES_MIN) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the row table (ds_write in begin())
        __builtin_amdgcn_s_barrier();                            // first chunk landed everywhere, table visible
    }
    int st = 0;
    for (int c = 0; c < nchunks; ++c) {
        const f32x4* S = S4 + st * C::STAGE4;
        const int stn = st + 1 == NS ? 0 : st + 1;
        if constexpr (NS == 2) {
            if (c + 1 < nchunks) { issue(c + 1, stn); fetch(c + 1); }
        } else {
            if (c + 2 < nchunks) issue(c + 2, stn + 1 == NS ? 0 : stn + 1);
        }
        const float* Bf = reinterpret_cast<const float*>(S + C::A4) + C::NT * i31;
#pragma unroll
        for (int g = 0; g < C::BKC / 16; ++g) {
            const f32x4 a0 = S[(4 * g + h) * C::BM + 32 * wave + i31];
            const f32x4 a1 = S[(4 * g + h + 2) * C::BM + 32 * wave + i31];
#define WV_K1_STEP(AV, ROW)                                                                        \
    { const bvec bv = *reinterpret_cast<const bvec*>(Bf + (ROW) * C::BN);                          \
      _Pragma("unroll") for (int e = 0; e < C::NT; ++e)                                            \
          acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, bv[e], acc[e], 0, 0, 0); }
            WV_K1_STEP(a0.x, 16 * g + 4 * h + 0) WV_K1_STEP(a0.y, 16 * g + 4 * h + 1)
            WV_K1_STEP(a0.z, 16 * g + 4 * h + 2) WV_K1_STEP(a0.w, 16 * g + 4 * h + 3)
            WV_K1_STEP(a1.x, 16 * g + 8 + 4 * h + 0) WV_K1_STEP(a1.y, 16 * g + 8 + 4 * h + 1)
            WV_K1_STEP(a1.z, 16 * g + 8 + 4 * h + 2) WV_K1_STEP(a1.w, 16 * g + 8 + 4 * h + 3)
#undef WV_K1_STEP
        }
        if constexpr (NS == 2) {
            if (c + 1 < nchunks) commit(c + 1, stn);
            __syncthreads();
        } else {
            // chunk c+1 must have landed (all waves' pieces) before anyone reads it; chunk c+2 may stay in flight
            if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES_MIN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        st = stn;
    }
    epi.finish(acc, p, smem);              // strips (EPI 1) alias the stages: the main loop ended with a barrier
}

// ---- launcher -----------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Shapes this core covers; everything else stays on the round-1 kernel (launch_pw_dw falls through, also
// when launch_k1 answers hipErrorNotSupported).
bool k1_supported(const PwDwArgs& a) {
    if (!a.pw.wq || a.pw.Mp % 128) return false;
    if (a.pw.M < 33) return false;                           // tiny layers: the round-1 core's 32-row tile
    // DMA'd activation rows need 16-byte aligned rows; the ConvTranspose producer gathers scalars, only its
    // output rows (Tout = Tin * ratio) must be aligned
    if (((a.ct_w ? a.Tout : a.Tin) & 3) || !aligned16(a.X) || a.pw.K < 1) return false;
    // 32-bit buffer offsets: one clip's operand block and output block stay below the out-of-range marker
    if ((long long)a.pw.K * a.Tin * 4 >= OOB_VOFF || (long long)a.pw.M * a.Tout * 4 >= OOB_VOFF) return false;
    if (a.Y && !aligned16(a.Y)) return false;
    if (a.Yact && !aligned16(a.Yact)) return false;
    if (a.resid && !aligned16(a.resid)) return false;
    if ((a.resid2 && !aligned16(a.resid2)) || (a.Yraw && !aligned16(a.Yraw)) || (a.Ysum && !aligned16(a.Ysum))) return false;
    if (a.ks < 1 || a.ks > 16 || (a.ks - 1) * a.dil + 1 + 3 > 64) return false;
    if (a.ct_w && a.ratio == 1) return false;                // degenerate ratio: rare, round-1 path
    return true;
}

template <class C, int EPI, int LDR, int RES, int NS = 2>
static hipError_t k1_run(PwDwArgs a, hipStream_t s, const char* base) {
    const size_t smem = NS * (size_t)C::STAGE4 * 16 + (size_t)K1Epi<C, EPI, RES>::TABLE_FLOATS * sizeof(float) +
                        (LDR >= 6 ? (size_t)2 * C::BKC * k1_lwp<C>(LDR) * sizeof(float) : 0);
    static_assert(C::WM * 4 * C::HLD <= 2 * C::STAGE4 * 4, "strips alias the stages");
    a.num_m = (a.pw.M + C::BM - 1) / C::BM;
    a.num_t = (a.Tout + a.tto - 1) / a.tto;
    a.stagger = 0; a.first_gen = 0;
    long long n_act = (long long)a.num_t * a.B;
    if (a.flat) {                                            // flat tiles over the padded (clip, time) axis
        if (!(C::NT == 4 && (EPI == 0 || (EPI == 8 && LDR == 0)))) return hipErrorInvalidValue;
        n_act = EPI == 8 ? ((long long)a.B * (a.Tv / 8) + a.tto - 1) / a.tto : ((long long)a.B * a.Tv - a.pad + a.tto - 1) / a.tto;
        if (n_act > 0x7fffffffLL) return hipErrorInvalidValue;
        a.num_t = (int)n_act;
    }
    const long long nblk = ((n_act + 7) / 8) * 8 * a.num_m;
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    std::string name;
    if (prof::enabled())
        name = std::string(base) + "<" + std::to_string(C::BM) + "," + std::to_string(C::BN) + (LDR == 0 ? (NS == 3 ? ",dma3" : ",dma") : (LDR >= 6 ? ",win" : ",reg")) + (a.flat ? ",flat>" : ">");
    const double M = a.pw.M, K = a.pw.K, Bd = a.B;
    const double outs = (a.Y ? 1.0 : 0.0) + (a.Yact ? 1.0 : 0.0) + (a.resid ? 1.0 : 0.0);
    const double flops = a.ct_w ? 2.0 * Bd * a.Tout * K * (M + 2.0) : 2.0 * Bd * M * (K * a.Tin + (double)a.ks * a.Tout);
    prof::Scope ps(s, name.c_str(), flops, 4.0 * Bd * (K * a.Tin + M * a.Tout * outs));
    hipLaunchKernelGGL((k1_kernel<C, EPI, LDR, RES, NS>), dim3((unsigned)nblk), dim3(C::NTHREADS), smem, s, a);
    return hipGetLastError();
}

bool pw_dw_geometry(PwDwArgs& a, int BN);                     // wv_kernels.hip

template <class C, int EPI, int RES>
static hipError_t k1_pick_ldr(const PwDwArgs& a, hipStream_t s) {
    if (a.ct_w) {
        if constexpr (EPI == 0 && RES == 0) {
#ifndef K1_NO_LDS_UPSAMPLE
            // per-clip tiles of an already activated input in whole chunks: the input window through LDS (ConvTrLds)
            if (!a.flat && a.pre_scale == 1.f && !a.pre_elu && a.pw.K % C::BKC == 0 && a.Tin % 4 == 0 && a.Tout == a.Tin * a.ratio && C::BM <= 128) {
                if (a.ratio % 4 == 0) return k1_run<C, 0, 6, false>(a, s, "convtr_pw_lds");
                if (a.ratio == 2) return k1_run<C, 0, 7, false>(a, s, "convtr_pw_lds");
            }
#endif
            if (a.ratio % 4 == 0) return k1_run<C, 0, 2, false>(a, s, "convtr_pw");
            if (a.ratio == 2) return k1_run<C, 0, 3, false>(a, s, "convtr_pw");
            return k1_run<C, 0, 5, false>(a, s, "convtr_pw");
        }
        return hipErrorInvalidValue;
    }
    const char* base = a.spec_add ? "spec_add" : (EPI == 0 ? (RES == 2 ? "pw_dw_k5_dact" : RES == 4 ? "pw_dw_k5_h" : RES == 5 ? "pw_dw_k5_hres" : RES ? "pw_dw_k5" : "pw_dw_k5_nr") : (EPI == 1 ? "pw_dw" : "pw_dw_s"));
    if (a.pre_elu || a.pre_scale != 1.f) return k1_run<C, EPI, 1, RES>(a, s, base);
    if constexpr (C::NT == 4 && C::BM == 128) {
        // matrix-bound k5 units: deeper DMA pipeline.  Not the strided units: their 8-10 KB row table makes the third stage cost a resident
        // workgroup (58 KB -> 2 per CU instead of 3 at 42 KB); measured r = 5: 1289 vs 1425 us, r = 8 (flat): 945 vs 1010 us with two stages
        if (EPI == 0 && a.pw.K >= 256) return k1_run<C, EPI, 0, RES, 3>(a, s, base);
    }
    return k1_run<C, EPI, 0, RES>(a, s, base);
}

template <class C>
static hipError_t k1_pick_epi(const PwDwArgs& a, hipStream_t s, bool k5) {
    const bool res = a.resid != nullptr;
    if (a.res_mode == 2) {                                   // ELU-derivative epilogue (training): the k5 DPP epilogue only
        if (!k5 || !res) return hipErrorNotSupported;
        return k1_pick_ldr<C, 0, 2>(a, s);
    }
    if (a.Yraw) {                                            // raw 1x1 output next to y (training forward): the k5 DPP epilogue only
        if (!k5 || a.ct_w || (res != (a.Ysum != nullptr))) return hipErrorNotSupported;
        return res ? k1_pick_ldr<C, 0, 5>(a, s) : k1_pick_ldr<C, 0, 4>(a, s);
    }
    if (k5) return res ? k1_pick_ldr<C, 0, true>(a, s) : k1_pick_ldr<C, 0, false>(a, s);
    if constexpr (C::NT == 4) {                              // the net's downsample stencils: ks = 2r, stride r, pad r
        if (!res && !a.ct_w && a.dil == 1 && a.ks == 2 * a.stride && a.pad == a.stride &&
            a.off == (a.stride == 2 ? 2 : 0)) {
            if (a.stride == 2) return k1_pick_ldr<C, 2, false>(a, s);
            if (a.stride == 4) return k1_pick_ldr<C, 4, false>(a, s);
            if (a.stride == 8) return k1_pick_ldr<C, 8, false>(a, s);
        }
        if (!res && !a.ct_w && a.dil == 1 && a.ks == 10 && a.stride == 5 && a.tto <= 32) return k1_pick_ldr<C, 5, false>(a, s);
    }
    return res ? k1_pick_ldr<C, 1, true>(a, s) : k1_pick_ldr<C, 1, false>(a, s);
}

hipError_t launch_k1(const PwDwArgs& a0, hipStream_t s) {
    PwDwArgs a = a0;
    const bool k5 = a.ks == 5 && a.stride == 1 && a.dil == 1 && a.pad == 4 && !a.film;   // FiLM: generic epilogue
    // window width: 64 columns when that computes fewer columns (tile quantisation) or the layer is short
    const int need = (a.ks - 1) * a.dil + 1;
    bool narrow = (a.ct_w ? a.Tout : a.Tin) + a.pad + 3 <= 64;        // the H window runs over output times for the upsample unit
    if (!narrow) {
        PwDwArgs g128 = a, g64 = a;
        if (pw_dw_geometry(g128, 128) && pw_dw_geometry(g64, 64)) {
            const long long c128 = (long long)((a.Tout + g128.tto - 1) / g128.tto) * 128;
            const long long c64 = (long long)((a.Tout + g64.tto - 1) / g64.tto) * 64;
            if (c64 * 100 < c128 * 95) narrow = true;
        }
    }
    (void)need;
    // Flattened (clip, time) tiling for the k5 stencil with a DMA'd operand (see k1_kernel): taken when it computes
    // at least 2 % fewer columns than per-clip tiles, rows are whole tiles and one tensor stays below 2 GB (32-bit
    // buffer offsets).
    a.flat = 0;
    if (k5 && a.B > 1 && (a.Tout & 3) == 0) {
        int bm = 128, best = (a.pw.M + 127) / 128 * 128;
        for (int cand : {96, 64}) { const int pd = (a.pw.M + cand - 1) / cand * cand; if (pd < best) { best = pd; bm = cand; } }
        const long long Tv = a.Tout + 4;                    // the H window runs over OUTPUT times (= input times for stride 1)
        const long long flat_cols = ((long long)a.B * Tv - 4 + 123) / 124 * 128;
        PwDwArgs g = a;
        const int bn = narrow ? 64 : 128;
        long long clip_cols = -1;
        if (pw_dw_geometry(g, bn)) clip_cols = (long long)a.B * ((a.Tout + g.tto - 1) / g.tto) * bn;
        if (a.pw.M % bm == 0 && (long long)a.B * a.pw.M * a.Tout * 4 < 0x7fffffffLL && clip_cols > 0 &&
            flat_cols * 100 < clip_cols * 98) {
            a.flat = 1; a.Tv = (int)Tv; narrow = false;
        }
    }
    if (!pw_dw_geometry(a, narrow ? 64 : 128)) return hipErrorNotSupported;
    // the r = 8 downsample on the DMA path: flat tiles over the input axis (see k1_kernel) when that computes >= 2 % fewer columns
    if (!narrow && !a.flat && !a.resid && !a.ct_w && a.ks == 16 && a.stride == 8 && a.dil == 1 && a.pad == 8 && a.off == 0 && a.B > 1 &&
        a.Tin % 8 == 0 && a.Tout * 8 == a.Tin && a.pw.M % 128 == 0 && (!a.film || (a.pw.M / a.bands) % 128 == 0) && !a.pre_elu && a.pre_scale == 1.f &&
        (long long)a.B * a.pw.M * a.Tout * 4 < 0x7fffffffLL && (long long)a.B * a.pw.K * a.Tin * 4 < 0x7fffffffLL) {
        const long long flat_tiles = ((long long)a.B * ((a.Tin + 8) / 8) + a.tto - 1) / a.tto;
        const long long clip_tiles = (long long)a.B * ((a.Tout + a.tto - 1) / a.tto);
        if (flat_tiles * 100 < clip_tiles * 98) { a.flat = 1; a.Tv = a.Tin + 8; }
    }
    // every tile's window must start on a multiple of 4 samples (16-byte DMA source addresses)
    if ((a.tto * a.stride) % 4 != 0 || (a.pad + a.off) % 4 != 0) return hipErrorNotSupported;
    // tile height: the one that pads the channel count least (128 on a tie): 192 -> 2 x 96, 64 -> 64, 130 -> 2 x 96
    int bm = 128, best = (a.pw.M + 127) / 128 * 128;
    for (int cand : {96, 64}) {
        const int padded = (a.pw.M + cand - 1) / cand * cand;
        if (padded < best) { best = padded; bm = cand; }
    }
    if (narrow) {
        if (bm == 128) return k1_pick_epi<K1<2, 32, 128>>(a, s, k5);
        if (bm == 96) return k1_pick_epi<K1<2, 32, 96>>(a, s, k5);
        return k1_pick_epi<K1<2, 32, 64>>(a, s, k5);
    }
    // The upsample unit computes its B operand (depth-wise ConvTranspose) in the loader, once per m-tile workgroup: 256-row tiles (8 waves)
    // halve that redundant vector work per matrix instruction where the channel count allows (M = 768: 2.27 -> 2.07 ms, 104 -> 117 TFLOP/s).
    // 192-row tiles (6 waves on 4 SIMDs) were measured and lose: M = 384 3.04 -> 3.93 ms, M = 192 3.17 -> 3.46 ms; one 384-row tile
    // (12 waves, one workgroup per CU) does not win either: 3.04 -> 3.13 ms.
    if (a.ct_w && k5 && !narrow && !a.resid && !a.Yraw && a.res_mode != 2 && a.pw.M % 256 == 0) return k1_pick_ldr<K1<4, 16, 256>, 0, false>(a, s);
    if (bm == 128) return k1_pick_epi<K1<4, 16, 128>>(a, s, k5);
    if (bm == 96) return k1_pick_epi<K1<4, 16, 96>>(a, s, k5);
    return k1_pick_epi<K1<4, 16, 64>>(a, s, k5);
}


// =================================================================================================
// CausalSTFT -> log-magnitude on the LDS-DMA core (modules/conv.py:1036-1086, seanet.py:479-494).
// The STFT is a GEMM against the windowed DFT basis: A = basis rows packed like a 1x1 weight (wq layout, rows
// arranged so that an accumulator register pair (2f, 2f+1) of one lane is (re, im) of bin f; row pair 0 =
// (cos_0, cos_Nyquist)), by descriptor DMA; B[k][t] = wav[t*hop + k - (n_fft-1)] (left zero history) gathered
// through registers into the natural [k][t] stage -- loads go to clamped addresses, the zero-selects happen at
// commit time.  The two left-over basis rows sin_0 / sin_Nyquist (see StftArgs) are two plain dot products per
// frame, accumulated per thread next to the commit (first m-tile only) and reduced through LDS after the loop.
// Epilogue: a lane holds 4 (2) consecutive frames of both halves of a bin -> sqrt(max(re^2+im^2,1e-12)) ->
// log(max(.,1e-5)) -> (y - mean)/std -> one 16-byte (8-byte) store.  Round 1's kernel (generic core, 4-byte
// scattered stores, 74 TFLOP/s) stays for frame counts that are not a multiple of the vector width.
// =================================================================================================
// FUSE: the SpecBlock's 1x1 and its add in the same launch (n_fft = M = BM: the workgroup holds every bin of its frames).  The
// log-magnitudes go to LDS as P[F][BN] (rows up to the next multiple of 16 zeroed) instead of HBM, a second GEMM W @ P runs on the same
// 32-row strips with its A fragments loaded straight into registers (3 or 5 chunks of the k-inner layout), and the epilogue forms
// y = resid + out_scale * (W @ P) [-> ELU(act_scale * y)] -- the values of the two-kernel path, bit for bit.
// FUSE at 64 rows: the residual operand of the add is requested BEFORE the second GEMM (16 rows in registers, two waves per SIMD): 814 ->
// 712 us per launch at n_fft = 64; at 128 rows the third wave per SIMD is worth more than the prefetch (1071 vs 1083 us).
template <class C, bool FUSE>
__global__ __launch_bounds__(C::NTHREADS, (FUSE && C::BM == 64) ? 2 : ((FUSE || C::B_PER > 1) ? 3 : 4)) void stft_k1_kernel(StftArgs p, SpecAddArgs q) {
    constexpr bool PRE_RES = FUSE && C::BM == 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef typename NVec<C::NT>::type bvec;
    typedef unsigned uvec __attribute__((ext_vector_type(C::NT)));
    const unsigned L = blockIdx.x, j = L >> 3;
    const int m_tile = j % p.num_m;
    const unsigned n_idx = (j / p.num_m) * 8 + (L & 7);
    if (n_idx >= (unsigned)p.num_t * p.B) return;
    const int t_tile = n_idx % p.num_t, b = n_idx / p.num_t;
    f32x4* S4 = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i31 = lane & 31, cg = tid % C::CG;
    const int K = p.n_fft, m0 = m_tile * C::BM, t0 = t_tile * C::BN;
    const int nchunks = (K + C::BKC - 1) / C::BKC;
    const float* wb = p.wav + (size_t)b * p.T;
    DmaA<C> da;
    da.init(reinterpret_cast<const f32x4*>(p.basis_q), p.Mp, m0, nchunks, wave, lane);

    // frame gather: this thread's 4 frames t0 + 4cg + e; rows 2kp, 2kp + 1 of a chunk are adjacent samples
    int sbase[4]; bool fv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = t0 + 4 * cg + e;
        fv[e] = t < p.Tf;
        sbase[e] = t * p.hop - (p.n_fft - 1);
    }
    // interior tiles (all but the first few and the last of a clip): every sample of the window exists -> one
    // unaligned 8-byte load per frame and row pair, no clamps, no selects
    const bool interior = (K % C::BKC) == 0 && t0 + C::BN <= p.Tf && (long long)t0 * p.hop - (p.n_fft - 1) >= 0 &&
                          (long long)(t0 + C::BN - 1) * p.hop < p.T;
    float raw[C::B_PER][8];
    const bool side_on = m_tile == 0;                          // workgroup-uniform
    float d0[4] = {0.f, 0.f, 0.f, 0.f}, d1[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int c, auto FAST) {
#pragma unroll
        for (int r = 0; r < C::B_PER; ++r) {
            const int idx = tid + r * C::NTHREADS;
            if (!(C::NBT % C::NTHREADS == 0 || idx < C::NBT)) continue;
            const int k0 = c * C::BKC + 2 * (idx / C::CG);
            if constexpr (decltype(FAST)::value) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2u v = *reinterpret_cast<const f32x2u*>(wb + (sbase[e] + k0));
                    raw[r][e] = v.x; raw[r][4 + e] = v.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) raw[r][4 * i + e] = wb[min(max(sbase[e] + k0 + i, 0), p.T - 1)];
            }
        }
    };
    auto commit = [&](int c, int st, auto FAST) {
        f32x4* Bq = S4 + st * C::STAGE4 + C::A4;
#pragma unroll
        for (int r = 0; r < C::B_PER; ++r) {
            const int idx = tid + r * C::NTHREADS;
            if (!(C::NBT % C::NTHREADS == 0 || idx < C::NBT)) continue;
            const int kp = idx / C::CG, k0 = c * C::BKC + 2 * kp;
            float o[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (decltype(FAST)::value) o[4 * i + e] = raw[r][4 * i + e];
                    else {
                        const int sidx = sbase[e] + k0 + i;
                        o[4 * i + e] = (fv[e] && k0 + i < K && sidx >= 0 && sidx < p.T) ? raw[r][4 * i + e] : 0.f;
                    }
                }
            Bq[(2 * kp) * C::CG + cg] = f32x4{o[0], o[1], o[2], o[3]};
            Bq[(2 * kp + 1) * C::CG + cg] = f32x4{o[4], o[5], o[6], o[7]};
            if (side_on) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = min(k0 + i, K - 1);           // rows past K carry zeros
                    const float s0 = p.side[k], s1 = p.side[K + k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { d0[e] = fmaf(s0, o[4 * i + e], d0[e]); d1[e] = fmaf(s1, o[4 * i + e], d1[e]); }
                }
            }
        }
    };
    f32x16 acc[C::NT];
#pragma unroll
    for (int e = 0; e < C::NT; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    auto gemm = [&](auto FAST) {
        da.issue(0, S4, wave);
        fetch(0, FAST); commit(0, 0, FAST);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int st = c & 1;
            const f32x4* S = S4 + st * C::STAGE4;
            if (c + 1 < nchunks) { da.issue(c + 1, S4 + (st ^ 1) * C::STAGE4, wave); fetch(c + 1, FAST); }
            const float* Bf = reinterpret_cast<const float*>(S + C::A4) + C::NT * i31;
#pragma unroll
            for (int g = 0; g < C::BKC / 16; ++g) {
                const f32x4 a0 = S[(4 * g + h) * C::BM + 32 * wave + i31];
                const f32x4 a1 = S[(4 * g + h + 2) * C::BM + 32 * wave + i31];
#define WV_ST_STEP(AV, ROW)                                                                        \
    { const bvec bv = *reinterpret_cast<const bvec*>(Bf + (ROW) * C::BN);                          \
      _Pragma("unroll") for (int e = 0; e < C::NT; ++e)                                            \
          acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, bv[e], acc[e], 0, 0, 0); }
                WV_ST_STEP(a0.x, 16 * g + 4 * h + 0) WV_ST_STEP(a0.y, 16 * g + 4 * h + 1)
                WV_ST_STEP(a0.z, 16 * g + 4 * h + 2) WV_ST_STEP(a0.w, 16 * g + 4 * h + 3)
                WV_ST_STEP(a1.x, 16 * g + 8 + 4 * h + 0) WV_ST_STEP(a1.y, 16 * g + 8 + 4 * h + 1)
                WV_ST_STEP(a1.z, 16 * g + 8 + 4 * h + 2) WV_ST_STEP(a1.w, 16 * g + 8 + 4 * h + 3)
#undef WV_ST_STEP
            }
            if (c + 1 < nchunks) commit(c + 1, st ^ 1, FAST);
            __syncthreads();
        }
    };
    if (interior) gemm(std::true_type{});
    else gemm(std::false_type{});
    // the side rows: per-thread partial dot products -> LDS [slot][column][2] (aliases the stages) -> summed per column
    float* sd = smem;
    if (side_on) {                                               // one slot per row group of threads (tid / CG)
        const int slot = tid / C::CG;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sd[(slot * C::BN + 4 * cg + e) * 2] = d0[e];
            sd[(slot * C::BN + 4 * cg + e) * 2 + 1] = d1[e];
        }
        __syncthreads();
    }
    constexpr int NSLOT = C::NTHREADS / C::CG;
    const int clip_bytes = p.F * p.Tf * 4;
    const __amdgpu_buffer_rsrc_t rP = uniform_rsrc(FUSE ? p.wav : p.P + (size_t)b * p.F * p.Tf, FUSE ? 0 : clip_bytes);
    const int tq = t0 + C::NT * i31;                            // Tf % NT == 0: the lane's frames are all in or all out
    const bool lane_ok = tq < p.Tf;
    auto logmag = [&](float re, float im) { return stft_logmag(re, im, p.c1, p.c0); };
    // FUSE: P[F16][BN] in LDS behind the side-row slots (the stages are free now)
    constexpr int F16 = FUSE ? (C::BM / 2 + 1 + 15) / 16 * 16 : 0, NC2 = F16 / 16;
    float* Pl = smem + NSLOT * C::BN * 2;
    f32x4 a2[FUSE ? NC2 : 1][2];
    if constexpr (FUSE) {
        // the 1x1's A fragments: chunk c, lane half h: wq[(4c + h)][m], wq[(4c + h + 2)][m], m = 32 wave + i31 -- in flight under the epilogue
        const __amdgpu_buffer_rsrc_t rW = uniform_rsrc(q.pw.wq, NC2 * 4 * q.pw.Mp * 16);
        const int avoff = (h * q.pw.Mp + 32 * wave + i31) * 16;
#pragma unroll
        for (int c = 0; c < NC2; ++c) {
            a2[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, avoff, c * 4 * q.pw.Mp * 16, 0));
            a2[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, avoff, (c * 4 + 2) * q.pw.Mp * 16, 0));
        }
        for (int i = tid; i < (F16 - (C::BM / 2 + 1)) * C::BN; i += C::NTHREADS) Pl[(C::BM / 2 + 1) * C::BN + i] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const int row = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;          // even: (re, im) rows 2f, 2f + 1
        const bool row_ok = lane_ok && row < p.n_fft;
        if (row == 0) {                                          // (cos_0, cos_Nyquist) + the two side rows
            bvec y0, y1;
#pragma unroll
            for (int e = 0; e < C::NT; ++e) {
                float s0 = 0.f, s1 = 0.f;
                const int col = C::NT * i31 + e;
                for (int sl = 0; sl < NSLOT; ++sl) { s0 += sd[(sl * C::BN + col) * 2]; s1 += sd[(sl * C::BN + col) * 2 + 1]; }
                y0[e] = logmag(acc[e][r], s0);
                y1[e] = logmag(acc[e][r + 1], s1);
            }
            if constexpr (FUSE) {
                *reinterpret_cast<bvec*>(Pl + C::NT * i31) = y0;
                *reinterpret_cast<bvec*>(Pl + (p.F - 1) * C::BN + C::NT * i31) = y1;
            } else {
                const int v0 = lane_ok ? tq * 4 : 0x7f000000, v1 = lane_ok ? ((p.F - 1) * p.Tf + tq) * 4 : 0x7f000000;
                if constexpr (C::NT == 4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, y0), rP, v0, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, y1), rP, v1, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, y0), rP, v0, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, y1), rP, v1, 0, 0);
                }
            }
        } else {
            bvec y;
#pragma unroll
            for (int e = 0; e < C::NT; ++e) y[e] = logmag(acc[e][r], acc[e][r + 1]);
            if constexpr (FUSE) {
                *reinterpret_cast<bvec*>(Pl + (row >> 1) * C::BN + C::NT * i31) = y;      // n_fft = BM: every row is a bin row
            } else {
                const int v = row_ok ? ((row >> 1) * p.Tf + tq) * 4 : 0x7f000000;
                if constexpr (C::NT == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, y), rP, v, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, y), rP, v, 0, 0);
            }
        }
    }
    if constexpr (FUSE) {
        bvec resA[PRE_RES ? 16 : 1];
        if constexpr (PRE_RES) {
            const int M_ = C::BM, blk_ = M_ * p.Tf * 4;
            const __amdgpu_buffer_rsrc_t rR_ = uniform_rsrc(q.resid + (size_t)b * M_ * p.Tf, blk_);
            const int voff_ = lane_ok ? ((32 * wave + 4 * h) * p.Tf + tq) * 4 : 0x7f000000;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o_ = voff_ + ((r & 3) + 8 * (r >> 2)) * p.Tf * 4;
                if constexpr (C::NT == 4) resA[r] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b128(rR_, o_, 0, 0));
                else resA[r] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b64(rR_, o_, 0, 0));
            }
        }
        __syncthreads();                                         // P complete
        // ---- second GEMM: W[M][F] @ P[F][BN] on this wave's 32-row strip (the k order of the K1 core)
#pragma unroll
        for (int e = 0; e < C::NT; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
        const float* Bf = Pl + C::NT * i31 + 4 * h * C::BN;
#pragma unroll
        for (int c = 0; c < NC2; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bvec bv = *reinterpret_cast<const bvec*>(Bf + (16 * c + (j < 4 ? j : 4 + j)) * C::BN);
                const float a = a2[c][j >> 2][j & 3];
#pragma unroll
                for (int e = 0; e < C::NT; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[e], acc[e], 0, 0, 0);
            }
        }
        // ---- y = resid + out_scale * (W @ P): range-checked buffers over the clip's [M][Tf] block (M = BM: every row exists)
        const int M = C::BM, blk = M * p.Tf * 4;
        const size_t bo = (size_t)b * M * p.Tf;
        const __amdgpu_buffer_rsrc_t rR = uniform_rsrc(q.resid + bo, blk);
        const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(q.Y ? q.Y + bo : q.resid, q.Y ? blk : 0);
        const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(q.Yact ? q.Yact + bo : q.resid, q.Yact ? blk : 0);
        const int voff0 = lane_ok ? ((32 * wave + 4 * h) * p.Tf + tq) * 4 : 0x7f000000;
        bvec res4[4];
        if constexpr (!PRE_RES) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (C::NT == 4) res4[r] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b128(rR, voff0 + ((r & 3) + 8 * (r >> 2)) * p.Tf * 4, 0, 0));
                else res4[r] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b64(rR, voff0 + ((r & 3) + 8 * (r >> 2)) * p.Tf * 4, 0, 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = voff0 + ((r & 3) + 8 * (r >> 2)) * p.Tf * 4;
            bvec rr;
            if constexpr (PRE_RES) rr = resA[r];
            else {
                rr = res4[r & 3];
                if (r + 4 < 16) {
                    const int o4 = voff0 + (((r + 4) & 3) + 8 * ((r + 4) >> 2)) * p.Tf * 4;
                    if constexpr (C::NT == 4) res4[r & 3] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b128(rR, o4, 0, 0));
                    else res4[r & 3] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b64(rR, o4, 0, 0));
                }
            }
            bvec y;
#pragma unroll
            for (int e = 0; e < C::NT; ++e) y[e] = fmaf(acc[e][r], q.out_scale, rr[e]);
            if (q.Y) {
                if constexpr (C::NT == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, y), rY, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, y), rY, off, 0, 0);
            }
            if (q.Yact) {
                bvec a;
#pragma unroll
                for (int e = 0; e < C::NT; ++e) a[e] = elu1(y[e] * q.act_scale);
                if constexpr (C::NT == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, a), rA, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, a), rA, off, 0, 0);
            }
        }
    }
}

template <class C, bool FUSE>
static hipError_t stft_k1_run(StftArgs a, const SpecAddArgs& q, hipStream_t s) {
    a.num_m = (a.n_fft + C::BM - 1) / C::BM;
    a.num_t = (a.Tf + C::BN - 1) / C::BN;
    const long long n_act = (long long)a.num_t * a.B;
    const long long nblk = ((n_act + 7) / 8) * 8 * a.num_m;
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t side_bytes = (size_t)(C::NTHREADS / C::CG) * C::BN * 2 * sizeof(float);
    constexpr int F16 = (C::BM / 2 + 1 + 15) / 16 * 16;
    const size_t smem = std::max<size_t>(2 * (size_t)C::STAGE4 * 16, side_bytes + (FUSE ? (size_t)F16 * C::BN * sizeof(float) : 0));
    std::string name;
    if (prof::enabled()) name = std::string(FUSE ? "stft_spec<" : "stft_logmag<") + std::to_string(C::BM) + "," + std::to_string(C::BN) + ",k1>";
    const double Bd = a.B, T = a.Tf, M = C::BM;
    const double outs = (q.Y ? 1.0 : 0.0) + (q.Yact ? 1.0 : 0.0);
    const double flops = 2.0 * Bd * (2.0 * a.F) * a.n_fft * T + (FUSE ? 2.0 * Bd * M * a.F * T : 0.0);
    const double bytes = FUSE ? 4.0 * Bd * ((double)a.T + M * T * (1.0 + outs)) : 4.0 * Bd * ((double)a.T + (double)a.F * T);
    prof::Scope ps(s, name.c_str(), flops, bytes);
    if (FUSE && smem > 64 * 1024) return hipErrorNotSupported;
    hipLaunchKernelGGL((stft_k1_kernel<C, FUSE>), dim3((unsigned)nblk), dim3(C::NTHREADS), smem, s, a, q);
    return hipGetLastError();
}

// hipErrorNotSupported: use the round-1 kernel (no k-inner basis, or a frame count that is not a multiple of the vector width)
hipError_t launch_stft_k1(const StftArgs& a, hipStream_t s) {
    if (!a.basis_q || a.n_fft < 4 || (a.n_fft & 1) || (long long)a.F * a.Tf * 4 >= OOB_VOFF) return hipErrorNotSupported;
    // <= 64 frames per clip (one 64-column tile, never interior: measured 416 vs 376 us on the 1024-point scale):
    // the round-1 kernel keeps those
    if (a.Tf <= 64 || a.Tf % 4) return hipErrorNotSupported;
    int bm = 128, best = (a.n_fft + 127) / 128 * 128;
    for (int cand : {96, 64}) { const int pd = (a.n_fft + cand - 1) / cand * cand; if (pd < best) { best = pd; bm = cand; } }
    const SpecAddArgs none{};
    if (bm == 128) return stft_k1_run<K1<4, 16, 128>, false>(a, none, s);
    if (bm == 96) return stft_k1_run<K1<4, 16, 96>, false>(a, none, s);
    return stft_k1_run<K1<4, 16, 64>, false>(a, none, s);
}

hipError_t launch_stft_spec(const StftArgs& a_in, const SpecAddArgs& q, hipStream_t s) {
    StftArgs a = a_in;
    a.c1 = 0.5f * 0.69314718055994531f * a.inv_std;             // see stft_logmag (wv_dev.h)
    a.c0 = -a.mean * a.inv_std;
    if (!a.basis_q || !a.side || !q.pw.wq || !q.resid || (!q.Y && !q.Yact)) return hipErrorNotSupported;
    if (a.n_fft != q.pw.M || (a.n_fft != 64 && a.n_fft != 128) || q.pw.K != a.F || a.F != a.n_fft / 2 + 1) return hipErrorNotSupported;
    if (a.Mp % M_ALIGN || a.Mp < a.n_fft || q.pw.Mp % M_ALIGN || a.Tf <= 64 || (a.Tf & 3)) return hipErrorNotSupported;
    if ((long long)q.pw.M * a.Tf * 4 >= OOB_VOFF) return hipErrorNotSupported;
    if (!aligned16(q.resid) || (q.Y && !aligned16(q.Y)) || (q.Yact && !aligned16(q.Yact))) return hipErrorNotSupported;
    if (a.n_fft == 128) return stft_k1_run<K1<4, 16, 128>, true>(a, q, s);
    return stft_k1_run<K1<4, 16, 64>, true>(a, q, s);
}

}  // namespace wv
