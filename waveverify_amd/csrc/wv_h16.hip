// The detector's f16-operand / f32-accumulate mode (BASELINE.json configs[4]: "MFMA linears fp16"): a throughput mode next to the
// exact-f32 path, never in place of it.  Activations live in HBM as f16 in the layout the f16 matrix instruction reads its B operand
// in -- channel groups of eight, time-major inside a group:
//
//     X16[b][g = c / 8][t][c % 8]        ("c8": one (group, time) pair = 16 bytes = the 8 consecutive k of a B fragment)
//
// so that a lane's B fragment of v_mfma_f32_32x32x16_f16 is ONE 16-byte read (LDS or global), a whole time run of a group is
// contiguous (LDS-DMA copies it as it lies), and an accumulator lane (4 consecutive rows of a group at one time) stores 8 bytes.
// Weights are packed as A fragments, wq16[chunk][m][k-half][8] (one 16-byte load per lane and 16-deep chunk).  Accumulation, the
// depth-wise stencils, ELU, bias, residual add are f32; only what crosses HBM (and the window in LDS) is f16.
//
//   rh_kernel      whole SEANetResnetBlock (modules/seanet.py:245-281) in one launch, the structure of wv_rb.hip: persistent
//                  workgroups, the x window in LDS by LDS-DMA, raw x kept (packed) in registers as the residual, both stencils from
//                  the accumulators, u never leaves the CU.  Half the bytes of the f32 kernel and 1/16 of its matrix time.
//   conv16_kernel  dense causal Conv1d as an implicit GEMM straight from global memory (no LDS, no barrier): the downsample unit
//                  ELU -> 1x1 -> depth-wise(2r, stride r) with the two convolutions composed into one [M][2r][K] weight (twice the
//                  flops, which this pipe has to spare, and no stencil epilogue at all), and the SpecBlock's 1x1 + add (ks = 1).
//   conv16s_kernel the same conv for stride >= 4 with the x window staged through LDS (skewed LDS-DMA copy).
//   spec16_kernel  whole SpecBlock in one launch: the windowed DFT with both operands split in two f16 terms, log-magnitude, the 1x1
//                  and the add; the spectrogram stays in LDS.  All five scales of the detector incl. spec_post.
//   head16_kernel  mean-probability output: L2Norm, the composed head GEMM, sigmoid and the time mean; the logits never exist.
//   conv_pre16, f32_to_c8, c8_to_f32   the layout's entry and exit.
#include <atomic>
#include <string>
#include <type_traits>

#include "wv_dev.h"

namespace wv {

namespace {

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H_OOB = 0x7f000000;                               // byte offset beyond any num_records here

#define RH_BARRIER()                                             \
    do {                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("" ::: "memory");                           \
    } while (0)

__device__ __forceinline__ float rh_dpp_next(float v) {        // lane i <- lane i+1 (wave_shl:1), lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// 5-tap stencil from the accumulators for a PAIR of rows (accumulator registers r, r + 1: two adjacent channels; a lane's NT = 2
// consecutive columns in its two accumulators): y[e] = bias + sum_i w[i] * H[2q + e + i] as packed-f32 multiply-adds (v_pk_fma_f32: one
// instruction for both rows), taps in ascending order like wv_rb.hip's stencil.  The columns past the lane's own come from the next
// lane and the one after it as shifted copies (wave_shl:1, compiler-visible DPP moves: it keeps the read-after-write distance itself).
// Table row of a channel pair (pack_rh_table): {w0a, w0b, w1a, w1b | w2a, w2b, w3a, w3b | w4a, w4b, ba, bb}.
__device__ __forceinline__ f32x2 rh_pair_next(f32x2 v) { return f32x2{rh_dpp_next(v.x), rh_dpp_next(v.y)}; }
__device__ __forceinline__ void rh_stencil2(const f32x16 (&acc)[2], int r, const float* tp, f32x2 (&y)[2]) {
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(tp), q1 = *reinterpret_cast<const f32x4*>(tp + 4), q2 = *reinterpret_cast<const f32x4*>(tp + 8);
    const f32x2 w0{q0.x, q0.y}, w1{q0.z, q0.w}, w2{q1.x, q1.y}, w3{q1.z, q1.w}, w4{q2.x, q2.y}, bb{q2.z, q2.w};
    const f32x2 a0{acc[0][r], acc[0][r + 1]}, a1{acc[1][r], acc[1][r + 1]};
    const f32x2 n0 = rh_pair_next(a0), n1 = rh_pair_next(a1);    // columns 2q + 2, 2q + 3
    const f32x2 m0 = rh_pair_next(n0), m1 = rh_pair_next(n1);    // columns 2q + 4, 2q + 5
    f32x2 v0 = __builtin_elementwise_fma(w0, a0, bb), v1 = __builtin_elementwise_fma(w0, a1, bb);
    v0 = __builtin_elementwise_fma(w1, a1, v0); v1 = __builtin_elementwise_fma(w1, n0, v1);
    v0 = __builtin_elementwise_fma(w2, n0, v0); v1 = __builtin_elementwise_fma(w2, n1, v1);
    v0 = __builtin_elementwise_fma(w3, n1, v0); v1 = __builtin_elementwise_fma(w3, m0, v1);
    v0 = __builtin_elementwise_fma(w4, m0, v0); v1 = __builtin_elementwise_fma(w4, m1, v1);
    y[0] = v0; y[1] = v1;
}

// log2(e)-domain ELU.  The block's two activations feed matrix products, so their common factor can live in the weights: the kernel
// keeps a' = log2(e) * ELU(t) (computed from t' = log2(e) * t as med3(t', log2(e) * 2^t' - log2(e), 0): one v_exp_f32, one fma, one
// v_med3 -- the plain form needs a multiply in front of the exponential and a subtraction behind it) and the host packs W / log2(e).
// exp(t) - 1 >= t on both sides of zero, so the median picks t' for t > 0 and the exponential branch for t < 0, as elu1 does.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float elu_l2(float tl) { return __builtin_amdgcn_fmed3f(tl, fmaf(__builtin_amdgcn_exp2f(tl), LOG2E, -LOG2E), 0.f); }
__device__ __forceinline__ f32x2 elu_l2(f32x2 tl) {            // two at once: the multiply-add behind the exponentials is one packed instruction
    const f32x2 e{__builtin_amdgcn_exp2f(tl.x), __builtin_amdgcn_exp2f(tl.y)};
    const f32x2 m = __builtin_elementwise_fma(e, f32x2{LOG2E, LOG2E}, f32x2{-LOG2E, -LOG2E});
    return f32x2{__builtin_amdgcn_fmed3f(tl.x, m.x, 0.f), __builtin_amdgcn_fmed3f(tl.y, m.y, 0.f)};
}

// C channels, NG column groups of 32*NT columns (overlapping by the stencil's 4), WPS waves per SIMD.  One wave = SPW 32-row strips x
// one column group (SPW = 2: the 384 / 768-channel layers of the generator's decoder, whose B fragments then serve two strips).
// RESIDENT: both weight strips of a wave stay in registers for the kernel's life (C <= 128: 2 * C/16 fragments); wider layers stream
// them through a ring, A_AHEAD chunks in front of their MFMAs.
template <int C_, int NG_, int NT_, int WPS_, int RING_ = 8, int PD_ = 2, bool RES_ = (C_ <= 128), int SPW_ = 1>
struct RH {
    static constexpr int C = C_, NG = NG_, NT = NT_, WPS = WPS_, PD = PD_, SPW = SPW_;
    static constexpr int WM = C / (32 * SPW), NWAVES = WM * NG, NTHREADS = 64 * NWAVES;
    static constexpr int GS = 32 * NT - 4, WD = NG * GS + 4, TTO = WD - 8;
    static constexpr int G = C / 8, NCH = C / 16;
    static constexpr int PIECES = G * WD;                                  // 16-byte (group, column) pieces of a window
    static constexpr int NI = (PIECES + 64 * NWAVES - 1) / (64 * NWAVES);  // LDS-DMA instructions per wave and window
    static constexpr bool RESIDENT = RES_;
    static constexpr int NA = RESIDENT ? 2 * NCH : RING_, AD = NA - 1;
    static constexpr size_t WBYTES = (size_t)PIECES * 16;
    static constexpr int TABF = C / 2 * 12;                                 // floats of one stencil table: 12 per channel pair
    static constexpr size_t SMEM = WBYTES + (size_t)2 * TABF * sizeof(float);
    static_assert(C % (32 * SPW) == 0 && NT == 2 && NTHREADS <= 1024 && NTHREADS >= C && (2 * NCH) % NA == 0 && (RESIDENT || AD <= NCH) &&
                  SMEM <= 160 * 1024, "geometry");
};

// OUT: 1 = Y, 2 = Yact, 3 = both.  Weights arrive divided by log2(e), the first stencil's taps and bias multiplied by it (pack_rh).
template <class R, int OUT>
__global__ __launch_bounds__(R::NTHREADS) __attribute__((amdgpu_waves_per_eu(R::WPS, R::WPS))) void rh_kernel(RhArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = R::NT, C = R::C, WD = R::WD, NCH = R::NCH, G = R::G, NA = R::NA, AD = R::AD, SPW = R::SPW;
    h16* S = reinterpret_cast<h16*>(smem_raw);                   // [G][WD][8]
    float* tab = reinterpret_cast<float*>(smem_raw + R::WBYTES); // [2][C / 2][12]: taps and bias per channel pair (pack_rh_table)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = (wave % R::WM) * SPW, grp = wave / R::WM;  // first of this wave's SPW strips
    const int h = lane >> 5, q = lane & 31;
    const int T = p.T, ntiles = p.ntiles, num_t = p.num_t;
    const int clip_bytes = G * T * 16;
    const float pl2 = p.pre_scale * LOG2E;

    for (int i = tid; i < R::TABF; i += R::NTHREADS) { tab[i] = p.tab1[i]; tab[R::TABF + i] = p.tab2[i]; }

    // ---- A fragments: wq16[chunk][C][2][8]; lane (q, h) of strip: 16 bytes at ((32*strip + q) * 2 + h) * 16, chunk as scalar offset
    const __amdgpu_buffer_rsrc_t rW1 = uniform_rsrc(p.w1.wq, NCH * C * 32);
    const __amdgpu_buffer_rsrc_t rW2 = uniform_rsrc(p.w2.wq, NCH * C * 32);
    const int avoff = ((32 * strip + q) * 2 + h) * 16;
    h16x8 ar[NA][SPW];
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rw, int c, h16x8 (&dst)[SPW]) {
        const int so = c * C * 32;
#pragma unroll
        for (int sw = 0; sw < SPW; ++sw) dst[sw] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, avoff + sw * 32 * 32, so, 0));
    };

    // ---- this lane's place in a window (wv_rb.hip): rows 32*strip + 8j + 4h + rr (j, rr = 0..3) = group 4*strip + j, halves 4h + rr;
    // B / u columns co + e, x columns co + 8 + e
    const int co = R::GS * grp + NT * q;
    const bool own = NT * q < R::GS && co < R::TTO;
    const bool uw = NT * q < R::GS;
    const int row0 = 32 * strip + 4 * h;
    const h16* Bf = S + (size_t)(h * WD + co) * 8;               // B fragments of this lane's k-half, its columns (chunk c: + 2c*WD*8)
    h16* Urow = S + (size_t)(4 * strip * WD + co) * 8 + 4 * h;    // (group 4*strip, column co), this lane's 4 halves (group j: + j*WD*8)
    h16* Xrow = Urow + 8 * 8;
    const float* Wrow1 = tab + (row0 / 2) * 12;                 // channel pair row0 / 2 (row0 is a multiple of 4)
    const float* Wrow2 = Wrow1 + R::TABF;
    const bool hthread = tid < G * 8;                            // the 8 halo columns in front: one 16-byte piece per thread
    const int hpiece = (tid >> 3) * WD + (tid & 7);

    // ---- window refill by LDS-DMA: piece p = (group, column) lands at S + 16 p; instruction i of this wave copies pieces
    // (i * NWAVES + wave) * 64 + lane.  Columns outside [0, T) read an out-of-range offset = zeros (the causal padding).
    int pk[R::NI];
#pragma unroll
    for (int i = 0; i < R::NI; ++i) {
        const int pi = (i * R::NWAVES + wave) * 64 + lane;
        pk[i] = pi < R::PIECES ? (((pi / WD) << 16) | (pi % WD)) : -1;
    }
    auto refill = [&](int t) {
        if (t < 0) return;
        const int b = t / num_t, tt = t - b * num_t;
        const int tw0 = tt * R::TTO - 8;
        const __amdgpu_buffer_rsrc_t rX = uniform_rsrc(reinterpret_cast<const h16*>(p.X) + (size_t)b * G * T * 8, clip_bytes);
#pragma unroll
        for (int i = 0; i < R::NI; ++i) {
            const int p0 = (i * R::NWAVES + wave) * 64;
            const int k_ = pk[i];
            const int tx = tw0 + (k_ & 0xffff);
            const int vo = (tx >= 0 && tx < T) ? ((k_ >> 16) * T + tx) * 16 : H_OOB;
            h16* dst = S + (size_t)p0 * 8;
            if (p0 < R::PIECES) {
                if (k_ >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    refill(tile);
    if constexpr (R::RESIDENT) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) { load_a(rW1, c, ar[c]); load_a(rW2, c, ar[NCH + c]); }
    } else {
#pragma unroll
        for (int c = 0; c < AD; ++c) load_a(rW1, c, ar[c % NA]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RH_BARRIER();

    h16x4 X[SPW][4][NT];                                         // raw x where this lane's outputs lie: the residual operand
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / num_t, tt = tile - b * num_t;
        const int to0 = tt * R::TTO;
        // ================= activation pass: X = x (raw), S = log2(e) * ELU(c * x) in place ============
        if (hthread) {
            h16x8* hp = reinterpret_cast<h16x8*>(S + (size_t)hpiece * 8);
            h16x8 v = *hp;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (h16)elu_l2(fmaf((float)v[i], pl2, 0.f));
            *hp = v;
        }
        if (own) {
#pragma unroll
            for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < NT; ++e) X[sw][j][e] = *reinterpret_cast<const h16x4*>(Xrow + (size_t)((4 * sw + j) * WD + e) * 8);
#pragma unroll
            for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < NT; ++e) {
                        h16x4 v;
#pragma unroll
                        for (int rr = 0; rr < 4; rr += 2) {
                            // (fma with a zero addend: one v_fma_mix_f32 straight from the f16 half)
                            const f32x2 a = elu_l2(f32x2{fmaf((float)X[sw][j][e][rr], pl2, 0.f), fmaf((float)X[sw][j][e][rr + 1], pl2, 0.f)});
                            v[rr] = (h16)a.x; v[rr + 1] = (h16)a.y;
                        }
                        *reinterpret_cast<h16x4*>(Xrow + (size_t)((4 * sw + j) * WD + e) * 8) = v;
                    }
        }
        RH_BARRIER();                                            // B1: window complete
        // ================= GEMM 1: H1 = W1 @ S =======================================================
        f32x16 acc[SPW][NT];
        auto gemm = [&](auto g0c, const __amdgpu_buffer_rsrc_t& rw, const __amdgpu_buffer_rsrc_t& rw_next) {
            constexpr int g0 = decltype(g0c)::value;
#pragma unroll
            for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
                for (int e = 0; e < NT; ++e)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[sw][e][r] = 0.f;
            constexpr int PD = R::PD, NB = PD + 1;
            h16x8 bq[NB][NT];
#pragma unroll
            for (int c = 0; c < PD && c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < NT; ++e) bq[c % NB][e] = *reinterpret_cast<const h16x8*>(Bf + (size_t)(2 * c * WD + e) * 8);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (!R::RESIDENT) {
                    if (c + AD < NCH) load_a(rw, c + AD, ar[(g0 + c + AD) % NA]);
                    else load_a(rw_next, c + AD - NCH, ar[(g0 + c + AD) % NA]);
                }
                if (c + PD < NCH) {
#pragma unroll
                    for (int e = 0; e < NT; ++e)
                        bq[(c + PD) % NB][e] = *reinterpret_cast<const h16x8*>(Bf + (size_t)(2 * (c + PD) * WD + e) * 8);
                }
#pragma unroll
                for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
                    for (int e = 0; e < NT; ++e) acc[sw][e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[(g0 + c) % NA][sw], bq[c % NB][e], acc[sw][e], 0, 0, 0);
                // one sequence point per chunk over all accumulator chains: left alone the compiler runs the chains one after the
                // other over the whole GEMM and parks the other chains' operands in scratch
#pragma unroll
                for (int sw = 0; sw < SPW; ++sw) {
                    if constexpr (NT == 4) asm volatile("" : "+v"(acc[sw][0]), "+v"(acc[sw][1]), "+v"(acc[sw][2]), "+v"(acc[sw][3]));
                    else asm volatile("" : "+v"(acc[sw][0]), "+v"(acc[sw][1]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        gemm(std::integral_constant<int, 0>{}, rW1, rW2);
        RH_BARRIER();                                            // B2: every wave has read the window (u overwrites it)
        // ================= epilogue 1: u' = log2(e) * ELU(DW5(H1) + b1) -> S (taps and bias carry the log2(e)) ============
#pragma unroll
        for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h16x4 uh[NT];
#pragma unroll
                for (int rr = 0; rr < 4; rr += 2) {
                    f32x2 y[2];
                    rh_stencil2(acc[sw], 4 * j + rr, Wrow1 + ((rr + 8 * j + 32 * sw) / 2) * 12, y);
#pragma unroll
                    for (int e = 0; e < NT; ++e) {
                        const f32x2 u = elu_l2(y[e]);
                        uh[e][rr] = (h16)u.x; uh[e][rr + 1] = (h16)u.y;
                    }
                }
                if (uw) {
#pragma unroll
                    for (int e = 0; e < NT; ++e) *reinterpret_cast<h16x4*>(Urow + (size_t)((4 * sw + j) * WD + e) * 8) = uh[e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (tt == 0 && grp == 0 && NT * q < 4) {                 // u at times < 0 is the second conv's zero padding
            h16x4 z;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) z[rr] = (h16)0.f;
#pragma unroll
            for (int j = 0; j < 4 * SPW; ++j)
#pragma unroll
                for (int e = 0; e < NT; ++e) *reinterpret_cast<h16x4*>(Urow + (size_t)(j * WD + e) * 8) = z;
        }
        RH_BARRIER();                                            // B3: u complete
        // ================= GEMM 2: H2 = W2 @ u ========================================================
        gemm(std::integral_constant<int, NCH>{}, rW2, rW1);
        RH_BARRIER();                                            // B4: every wave has read u (the next window overwrites it)
        // ================= epilogue 2: y = x + s * (DW5(H2) + b2) -> HBM; refill x ======================
        {
            const int next = tile + gridDim.x < ntiles ? tile + gridDim.x : -1;
            const size_t bo = (size_t)b * G * T * 8;
            const __amdgpu_buffer_rsrc_t rY = uniform_rsrc((OUT & 1) ? reinterpret_cast<h16*>(p.Y) + bo : reinterpret_cast<const h16*>(p.X), (OUT & 1) ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rA = uniform_rsrc((OUT & 2) ? reinterpret_cast<h16*>(p.Yact) + bo : reinterpret_cast<const h16*>(p.X), (OUT & 2) ? clip_bytes : 0);
            int voff[NT];
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                const int t = to0 + co + e;
                voff[e] = (own && t < T) ? ((4 * strip) * T + t) * 16 + 8 * h : H_OOB;
            }
            const int jstep = T * 16;
            refill(next);                                        // ahead of the stores below (one in-order queue)
#pragma unroll
            for (int sw = 0; sw < SPW; ++sw)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h16x4 yh[NT];
                    float yy[4][NT];
#pragma unroll
                    for (int rr = 0; rr < 4; rr += 2) {
                        f32x2 v[2];
                        rh_stencil2(acc[sw], 4 * j + rr, Wrow2 + ((rr + 8 * j + 32 * sw) / 2) * 12, v);
#pragma unroll
                        for (int e = 0; e < NT; ++e) {
                            if constexpr (OUT == 1) {
                                // y = f16(v * s + x) in ONE instruction per element: the f16 residual half is a source operand and the f16 result
                                // lands in its half of the output register (the compiler's own choice here is two conversions, a
                                // packed fma and a packed conversion)
                                unsigned* yo = reinterpret_cast<unsigned*>(&yh[e]) + (rr >> 1);
                                const unsigned xi = reinterpret_cast<const unsigned*>(&X[sw][j][e])[rr >> 1];
                                asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(*yo) : "v"(v[e].x), "v"(p.out_scale), "v"(xi));
                                asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(*yo) : "v"(v[e].y), "v"(p.out_scale), "v"(xi));
                            } else {
                                yy[rr][e] = fmaf(v[e].x, p.out_scale, (float)X[sw][j][e][rr]);
                                yy[rr + 1][e] = fmaf(v[e].y, p.out_scale, (float)X[sw][j][e][rr + 1]);
                                yh[e][rr] = (h16)yy[rr][e]; yh[e][rr + 1] = (h16)yy[rr + 1][e];
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < NT; ++e) {
                        const int off = voff[e] == H_OOB ? H_OOB : voff[e] + (4 * sw + j) * jstep;
                        if constexpr ((OUT & 1) != 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, yh[e]), rY, off, 0, 0);
                        if constexpr ((OUT & 2) != 0) {
                            h16x4 v;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)elu1(yy[rr][e] * p.act_scale);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rA, off, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // the refill is older than this epilogue's stores: wait until only those are outstanding, then meet the other waves
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * SPW * NT * ((OUT & 1) + (OUT >> 1))) : "memory");
        RH_BARRIER();                                            // B0: the next window has landed
    }
}

int cu_count16() {
    static std::atomic<int> cached[32];
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 31;
    int n = cached[d].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t pr;
        n = hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
        cached[d].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <class R, int OUT>
hipError_t rh_launch(RhArgs a, hipStream_t s) {
    a.num_t = (a.T + R::TTO - 1) / R::TTO;
    const long long nt = (long long)a.num_t * a.B;
    if (nt > 0x7fffffffLL) return hipErrorInvalidValue;
    a.ntiles = (int)nt;
    static std::atomic<unsigned> attr{0};
    if (R::SMEM > 64 * 1024) {
        int d = 0; (void)hipGetDevice(&d);
        const unsigned bit = 1u << (d & 31);
        if (!(attr.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rh_kernel<R, OUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::SMEM);
            if (e != hipSuccess) return e;
            attr.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const int per_cu = std::max(1, std::min((int)(160 * 1024 / R::SMEM), 4 * R::WPS / R::NWAVES));
    const int grid = (int)std::min<long long>(nt, (long long)cu_count16() * per_cu);
    std::string name;
    if (prof::enabled()) name = "resblock16<" + std::to_string(R::C) + "," + std::to_string(R::WD) + ">";
    const double C = a.C, Bd = a.B, T = a.T;
    prof::Scope ps(s, name.c_str(), 2.0 * 2.0 * Bd * C * (C * T + 5.0 * T), 2.0 * Bd * C * T * (1.0 + ((OUT & 1) ? 1.0 : 0.0) + ((OUT & 2) ? 1.0 : 0.0)));
    hipLaunchKernelGGL((rh_kernel<R, OUT>), dim3((unsigned)grid), dim3(R::NTHREADS), R::SMEM, s, a);
    return hipGetLastError();
}

template <class R>
hipError_t rh_pick_out(const RhArgs& a, hipStream_t s) {
    if (a.Y && a.Yact) return rh_launch<R, 3>(a, s);
    if (a.Y) return rh_launch<R, 1>(a, s);
    return rh_launch<R, 2>(a, s);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ------------------------------------------------------------------------------------------------------------------------------
// Dense causal Conv1d on c8 activations as an implicit GEMM, operands straight from global memory:
//     y[m][to] = out_scale * ( bias[m] + sum_{i < ks} sum_k W[m][i][k] * x[k][to * stride + i - pad] ) + resid[m][to]
// (x = 0 outside [0, Tin)).  Chunk ch = kc * ks + i (16 channels kc, tap i): A fragment = wq16[ch][m][h][8], B fragment = the 16-byte piece
// (group 2 kc + h, time to * stride + i - pad) of x.  A wave owns 64 rows x 64 output times (2 x 2 MFMA tiles); the four waves of a
// workgroup share either the columns (WGM = 4: the B pieces of one wave are L1 hits for the other three) or the rows.  Loads run
// D chunks ahead of their MFMAs in a register ring; there is no LDS and no barrier.
// FLAT: the columns of all clips in one run (column n = clip n / Tout, time n % Tout) -- layers with few outputs per clip (50 frames
// after the last downsample) would otherwise leave most of a 64-column wave tile and of the workgroup idle; offsets then span the whole
// tensor (the launcher checks they fit the sentinel), and the workgroups are dealt so that one XCD keeps the same row blocks of W in
// its L2 while x streams past.
template <int WGM, int WGN, bool FLAT>
__global__ __launch_bounds__(256) void conv16_kernel(Conv16Args p) {
    constexpr int D = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;
    const int h = lane >> 5, q = lane & 31;
    const int Tin = p.Tin, Tout = p.Tout, Gk = p.w.Kp / 8, Gm = (p.M + 15) / 16 * 2, Mp = p.w.Mp, NKC = p.w.Kp / 16, nch = p.w.nchunks;
    int bclip = 0, m0, n0;                                       // per-clip mode: the clip, first row, first time; flat: first row, first flat column
    if constexpr (FLAT) {
        const int num_m = (p.M + 64 * WGM - 1) / (64 * WGM);
        const unsigned L = blockIdx.x;
        unsigned mb, cb;
        if (num_m % 8 == 0) { const unsigned xcd = L & 7, j = L >> 3, per = num_m / 8; mb = xcd + 8 * (j % per); cb = j / per; }
        else { mb = L % num_m; cb = L / num_m; }
        m0 = (mb * WGM + wm) * 64;
        n0 = (cb * WGN + wn) * 64;
        if (m0 >= p.M || n0 >= p.B * Tout) return;
    } else {
        const int ncol = (Tout + 64 * WGN - 1) / (64 * WGN);
        bclip = blockIdx.x / ncol;
        const int ct = blockIdx.x - bclip * ncol;
        m0 = (blockIdx.y * WGM + wm) * 64;
        n0 = (ct * WGN + wn) * 64;
        if (m0 >= p.M || n0 >= Tout) return;
    }
    const size_t xclip = (size_t)Gk * Tin * 8, yclip = (size_t)Gm * Tout * 8;      // halves per clip
    const __amdgpu_buffer_rsrc_t rX = FLAT ? uniform_rsrc(p.X, (int)(xclip * 2 * p.B)) : uniform_rsrc(reinterpret_cast<const h16*>(p.X) + bclip * xclip, (int)(xclip * 2));
    const __amdgpu_buffer_rsrc_t rW = uniform_rsrc(p.w.wq, nch * Mp * 32);
    int avoff[2], tin0[2], xb[2], cb_[2], to_[2];
    bool colok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) avoff[mt] = (m0 + 32 * mt + q < Mp) ? ((m0 + 32 * mt + q) * 2 + h) * 16 : H_OOB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int n = n0 + 32 * e + q;
        if constexpr (FLAT) { cb_[e] = n / Tout; to_[e] = n - cb_[e] * Tout; colok[e] = n < p.B * Tout; xb[e] = cb_[e] * (int)(xclip * 2); }
        else { cb_[e] = bclip; to_[e] = n; colok[e] = n < Tout; xb[e] = 0; }
        tin0[e] = to_[e] * p.stride - p.pad;
    }
    const int nreal = p.ks * NKC;

    h16x8 ra[D][2], rb[D][2];
    int ich = 0, ii = 0, ikc = 0;                                // the next chunk to load = (tap ii, channel chunk ikc); wave-uniform
    auto issue = [&](h16x8 (&a)[2], h16x8 (&bb)[2]) {
        const int so_a = ich * Mp * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, avoff[mt], so_a, 0));
        const int so_b = 2 * ikc * Tin * 16;                     // (the tap goes into the per-lane offset: that one alone is range-checked and must not be negative)
        const bool real = ich < nreal;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = tin0[e] + ii;
            const int vo = (real && colok[e] && t >= 0 && t < Tin) ? xb[e] + (h * Tin + t) * 16 : H_OOB;
            bb[e] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, so_b, 0));
        }
        ++ich;
        if (++ii == p.ks) { ii = 0; ++ikc; }                     // taps innermost: a lane's ks pieces of one channel group are contiguous (stride 1) or share cache lines
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][e][r] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) issue(ra[d], rb[d]);             // nchunks is a multiple of D (zero-padded weights)
    for (int ch0 = 0; ch0 < nch; ch0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            h16x8 a[2] = {ra[d][0], ra[d][1]}, bb[2] = {rb[d][0], rb[d][1]};
            if (ch0 + d + D < nch) issue(ra[d], rb[d]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bb[e], acc[mt][e], 0, 0, 0);
        }
    }
    // ---- epilogue: rows m0 + 32 mt + 8 j + 4 h + rr, columns n0 + 32 e + q.  up > 0: row = (phase, channel), column l -> time l * up + phase
    const int upr = p.up > 0 ? p.up : 1, Mo = p.up > 0 ? p.M / p.up : p.M, To = Tout * upr;
    const size_t yclip_o = (size_t)((Mo + 15) / 16 * 2) * To * 8;
    const int ybytes = FLAT ? (int)(yclip_o * 2 * p.B) : (int)(yclip_o * 2);
    const size_t ybase = FLAT ? 0 : bclip * yclip_o;
    const __amdgpu_buffer_rsrc_t rR = uniform_rsrc(p.resid ? reinterpret_cast<const h16*>(p.resid) + ybase : reinterpret_cast<const h16*>(p.X), p.resid ? ybytes : 0);
    const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(p.Y ? reinterpret_cast<h16*>(p.Y) + ybase : reinterpret_cast<const h16*>(p.X), p.Y ? ybytes : 0);
    const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(p.Yact ? reinterpret_cast<h16*>(p.Yact) + ybase : reinterpret_cast<const h16*>(p.X), p.Yact ? ybytes : 0);
    const int bw = p.film ? Mo / p.bands : 1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mrow = m0 + 32 * mt + 8 * j + 4 * h;      // first of this lane's 4 rows
            int ph = 0, mch = mrow;                              // up: row = (block, phase, channel in block)
            if (p.up > 0) { const int blk = mrow / (upr * p.up_mb), rem = mrow - blk * upr * p.up_mb; ph = rem / p.up_mb; mch = blk * p.up_mb + rem - ph * p.up_mb; }
            float bias[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) bias[rr] = (p.bias && mrow + rr < p.M) ? p.bias[mch + rr] : 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int to = to_[e];
                const bool ok = colok[e] && mrow < p.M;
                const int off = ok ? (FLAT ? cb_[e] * (int)(yclip_o * 2) : 0) + ((mch >> 3) * To + to * upr + ph) * 16 + 8 * h : H_OOB;
                float y[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) y[rr] = (acc[mt][e][4 * j + rr] + bias[rr]) * p.out_scale;
                if (p.film && ok) {
                    const float* fl = p.film + (size_t)cb_[e] * p.film_stride + 2 * (mch / bw);
                    const float gam = fl[0], bet = fl[1];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) y[rr] = fmaf(gam, y[rr], bet);
                }
                if (p.resid) {
                    const h16x4 rv = __builtin_bit_cast(h16x4, __builtin_amdgcn_raw_buffer_load_b64(rR, off, 0, 0));
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) y[rr] += (float)rv[rr];
                }
                if (p.Y) {
                    h16x4 v;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)y[rr];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rY, off, 0, 0);
                }
                if (p.Yact) {
                    h16x4 v;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)elu1(y[rr] * p.act_scale);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rA, off, 0, 0);
                }
                if (p.Yf32 && colok[e]) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (mrow + rr < p.M) p.Yf32[((size_t)cb_[e] * p.M + mrow + rr) * Tout + to] = y[rr];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Whole SpecBlock in one launch (modules/seanet.py:463-511, CausalSTFT modules/conv.py:1036-1086):
//     y = x + out_scale * ( W @ P ),   P[f][t] = (log(max(|STFT(wav)|, 1e-5)) - mean) / std
// GEMM 1 = the windowed DFT as a matrix product on the f16 pipe with BOTH operands split in two f16 terms (v = hi + lo, lo carried times
// 2^11): hi*hi in one accumulator, hi*lo + lo*hi in a second one scaled back by 2^-11 -- 22 bits of each operand, three matrix
// instructions where the f32 pipe needs sixteen.  (With the basis rounded to f16 once, strong bins leak into weak ones at ~1e-4 of the
// frame's level: the real-only DC and Nyquist bins of noise-like frames are often that weak, and their log-magnitude moved by up to 0.6
// -- measured, tools/h16time.py's first version.)  A frame's B fragment is 8 CONSECUTIVE samples, so the
// tile's waveform window sits in LDS as f16 in 8 / hop copies shifted by hop samples each (frame t reads copy (t hop) mod 8, a 16-byte
// aligned piece).  The cos rows f = 0 .. n_fft/2 - 1 and the sin rows are separate A tiles, so that a lane holds re and im of the same
// bin in the same register slot; sin row 0 (identically zero) carries the Nyquist bin's cos row instead.  P is written to LDS in the c8
// layout -- the B operand of GEMM 2 = the 1x1 -- and never leaves the CU.  Epilogue 2 adds x (c8 f16 from HBM) and stores y and / or
// ELU(act_scale * y).  (The exact path's two side rows, sin_0 and sin_{F-1} of a reference-built basis, are rounding-level and dropped.)
// N = n_fft = rows of the 1x1 (the detector's scales: 64, 128, 256, 512; 1024 = spec_post, whose output goes on in f32), HOP in {1, 2, 4, 8 k}.  A wave's unit is 64 rows x 64 frames
// (one cos/sin tile pair in GEMM 1, two row tiles in GEMM 2).
// M = rows of the 1x1 = channels of the stream: n_fft (generator / detector: every scale has as many channels as DFT points) or n_fft / 2
// (the locator: 32 / 64 / 128 channels at n_fft 64 / 128 / 256).  GEMM 2 then has half the row units (some waves sit it out).
template <int N_, int HOP_, int M_ = N_>
struct SP {
    static constexpr int N = N_, HOP = HOP_, M = M_;
    static constexpr int MT2 = M >= 64 ? 2 : 1, NP2 = M / (32 * MT2);   // GEMM 2: row tiles per unit, row units
    static constexpr int NP = N / 64;                            // 64-row units
    static constexpr int NQ = NP >= 4 ? 1 : 4 / NP;              // 64-frame units per tile
    static constexpr int BN = 64 * NQ, PASSES = NP * NQ / 4;
    static constexpr int NPL = HOP >= 8 ? 1 : 8 / HOP;           // shifted copies of the window
    static constexpr int WLEN = (BN - 1) * HOP + N;              // samples a tile's frames cover
    static constexpr int PP = (WLEN + 7 + 7) / 8;                // 16-byte pieces per copy (the last copy starts 7 samples in)
    static constexpr int PSP = (PP + 13) / 16 * 16 + 2;          // copy stride in pieces, = 2 (mod 16): the 8 copies x 2 pieces a 16-lane group reads are 16 different banks
    static constexpr int Fp = N / 2 + 16, G2 = Fp / 8, NC1 = N / 16, NC2 = Fp / 16;
    static constexpr size_t WIN = (size_t)NPL * PSP * 16;        // bytes of one (hi or lo) window
    static constexpr size_t SMEM = 2 * WIN + (size_t)G2 * BN * 16;
    static constexpr int PASSES2 = (NP2 * NQ + 3) / 4;
    static_assert(N % 64 == 0 && (HOP == 1 || HOP == 2 || HOP == 4 || HOP % 8 == 0) && NP * NQ % 4 == 0 && (M == N || 2 * M == N) && M % 32 == 0, "geometry");
};

template <class R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void spec16_kernel(Spec16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = R::N, HOP = R::HOP, BN = R::BN, NP = R::NP;
    h16* Whi = reinterpret_cast<h16*>(smem_raw);
    h16* Wlo = reinterpret_cast<h16*>(smem_raw + R::WIN);
    h16* P16 = reinterpret_cast<h16*>(smem_raw + 2 * R::WIN);    // [G2][BN][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int T = p.T, Tf = p.Tf;
    const int ntile = (Tf + BN - 1) / BN;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x - b * ntile) * BN;

    const __amdgpu_buffer_rsrc_t rC = uniform_rsrc(p.cosw.wq, R::NC1 * (N / 2) * 32);
    const __amdgpu_buffer_rsrc_t rS = uniform_rsrc(p.sinw.wq, R::NC1 * (N / 2) * 32);
    const __amdgpu_buffer_rsrc_t rCl = uniform_rsrc(p.cosl.wq, R::NC1 * (N / 2) * 32);
    const __amdgpu_buffer_rsrc_t rSl = uniform_rsrc(p.sinl.wq, R::NC1 * (N / 2) * 32);
    // basis fragments: a ring NA1 chunks deep (the short GEMMs of the fine scales would otherwise pay one L2 round trip per chunk); the
    // first pass's leading chunks are requested before the window is built
    constexpr int NA1 = 4, AD1 = NA1 - 1;
    h16x8 ac[NA1], as[NA1], lc[NA1], ls[NA1];
    int avoff = ((32 * (wave % NP) + r) * 2 + h) * 16;
    auto lda = [&](int c, int slot) {
        const int so = c * (N / 2) * 32;
        ac[slot] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rC, avoff, so, 0));
        as[slot] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rS, avoff, so, 0));
        lc[slot] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rCl, avoff, so, 0));
        ls[slot] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rSl, avoff, so, 0));
    };
#pragma unroll
    for (int c = 0; c < AD1 && c < R::NC1; ++c) lda(c, c % NA1);

    // ---- the tile's waveform window, split and copied NPL times
    {
        const float* wb = p.wav + (size_t)b * T;
        const int base = t0 * HOP - (N - 1);                     // wav index of window sample 0 (causal: n_fft - 1 zeros in front of the clip, conv.py:1060)
        if constexpr (R::NPL > 1) {
            // every sample lands in NPL copies: fetch the window once (coalesced) into the P16 area, build the copies from there
            float* tmp = reinterpret_cast<float*>(P16);
            static_assert((size_t)(R::WLEN + 16) * 4 <= (size_t)R::G2 * R::BN * 16, "staging area");
            for (int i = tid; i < R::WLEN + 16; i += 256) {
                const int s = base + i;
                tmp[i] = (s >= 0 && s < T) ? wb[s] : 0.f;
            }
            RH_BARRIER();
            for (int i = tid; i < R::NPL * R::PP; i += 256) {
                const int pi = i / R::PP, v = i - pi * R::PP;
                h16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * v + pi * HOP + j;
                    const float x = k < R::WLEN + 16 ? tmp[k] : 0.f;
                    hi[j] = (h16)x;
                    lo[j] = (h16)((x - (float)hi[j]) * 2048.f);
                }
                *reinterpret_cast<h16x8*>(Whi + (size_t)(pi * R::PSP + v) * 8) = hi;
                *reinterpret_cast<h16x8*>(Wlo + (size_t)(pi * R::PSP + v) * 8) = lo;
            }
        } else {
            for (int i = tid; i < R::PP; i += 256) {
                h16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int s = base + 8 * i + j;
                    const float x = (s >= 0 && s < T) ? wb[s] : 0.f;
                    hi[j] = (h16)x;
                    lo[j] = (h16)((x - (float)hi[j]) * 2048.f);
                }
                *reinterpret_cast<h16x8*>(Whi + (size_t)i * 8) = hi;
                *reinterpret_cast<h16x8*>(Wlo + (size_t)i * 8) = lo;
            }
        }
    }
    RH_BARRIER();

    // ================= GEMM 1 + log-magnitude -> P16 =================
    for (int pass = 0; pass < R::PASSES; ++pass) {
        const int u = wave + 4 * pass, mp = u % NP, nq = u / NP;
        if (pass > 0) {
            avoff = ((32 * mp + r) * 2 + h) * 16;
#pragma unroll
            for (int c = 0; c < AD1 && c < R::NC1; ++c) lda(c, c % NA1);
        }
        int boff[2];                                             // piece index of chunk 0's B fragment, per frame tile
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int off = (64 * nq + 32 * e + r) * HOP;
            boff[e] = ((off & 7) / HOP) * R::PSP + (off >> 3) + h;
        }
        f32x16 are[2], aim[2], lre[2], lim[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) { are[e][i] = 0.f; aim[e][i] = 0.f; lre[e][i] = 0.f; lim[e][i] = 0.f; }
#pragma unroll
        for (int c = 0; c < R::NC1; ++c) {
            if (c + AD1 < R::NC1) lda(c + AD1, (c + AD1) % NA1);
            h16x8 bh[2], bl[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bh[e] = *reinterpret_cast<const h16x8*>(Whi + (size_t)(boff[e] + 2 * c) * 8);
                bl[e] = *reinterpret_cast<const h16x8*>(Wlo + (size_t)(boff[e] + 2 * c) * 8);
            }
            const h16x8 a_c = ac[c % NA1], a_s = as[c % NA1], l_c = lc[c % NA1], l_s = ls[c % NA1];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                are[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_c, bh[e], are[e], 0, 0, 0);
                aim[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_s, bh[e], aim[e], 0, 0, 0);
                lre[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_c, bl[e], lre[e], 0, 0, 0);
                lim[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_s, bl[e], lim[e], 0, 0, 0);
                lre[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l_c, bh[e], lre[e], 0, 0, 0);
                lim[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l_s, bh[e], lim[e], 0, 0, 0);
            }
            asm volatile("" : "+v"(are[0]), "+v"(are[1]), "+v"(aim[0]), "+v"(aim[1]));
            asm volatile("" : "+v"(lre[0]), "+v"(lre[1]), "+v"(lim[0]), "+v"(lim[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int col = 64 * nq + 32 * e + r;
            float nyq = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h16x4 v;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float re = fmaf(lre[e][4 * j + rr], 1.f / 2048.f, are[e][4 * j + rr]);
                    float im = fmaf(lim[e][4 * j + rr], 1.f / 2048.f, aim[e][4 * j + rr]);
                    if (j == 0 && rr == 0 && mp == 0 && h == 0) { nyq = im; im = 0.f; }      // bin 0: no imaginary part; its sin slot carried the Nyquist bin
                    v[rr] = (h16)stft_logmag(re, im, p.c1, p.c0);
                }
                *reinterpret_cast<h16x4*>(P16 + (size_t)((4 * mp + j) * BN + col) * 8 + 4 * h) = v;
            }
            if (mp == 0 && h == 0) {                             // rows n_fft/2 (Nyquist) .. Fp - 1 (zero: the 1x1's padded columns)
                h16x8 z;
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = (h16)0.f;
                *reinterpret_cast<h16x8*>(P16 + (size_t)((N / 16 + 1) * BN + col) * 8) = z;
                z[0] = (h16)stft_logmag(nyq, 0.f, p.c1, p.c0);
                *reinterpret_cast<h16x8*>(P16 + (size_t)((N / 16) * BN + col) * 8) = z;
            }
        }
    }
    RH_BARRIER();
    // ================= GEMM 2: W @ P, + x =================
    constexpr int M = R::M, MT2 = R::MT2;
    const __amdgpu_buffer_rsrc_t rW = uniform_rsrc(p.pw.wq, p.pw.nchunks * M * 32);
    constexpr int Gm = M / 8;
    const size_t yclip = (size_t)Gm * Tf * 8;
    const __amdgpu_buffer_rsrc_t rR = uniform_rsrc(reinterpret_cast<const h16*>(p.resid) + b * yclip, (int)(yclip * 2));
    const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(p.Y ? reinterpret_cast<h16*>(p.Y) + b * yclip : reinterpret_cast<const h16*>(p.resid), p.Y ? (int)(yclip * 2) : 0);
    const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(p.Yact ? reinterpret_cast<h16*>(p.Yact) + b * yclip : reinterpret_cast<const h16*>(p.resid), p.Yact ? (int)(yclip * 2) : 0);
    for (int pass = 0; pass < R::PASSES2; ++pass) {
        const int u = wave + 4 * pass, mp = u % R::NP2, nq = u / R::NP2;
        if (u >= R::NP2 * R::NQ) break;                          // (wave-uniform; no barrier follows)
        int avw[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) avw[mt] = ((32 * MT2 * mp + 32 * mt + r) * 2 + h) * 16;
        const h16* Bp = P16 + (size_t)(h * BN + 64 * nq + r) * 8;
        f32x16 acc[MT2][2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][e][i] = 0.f;
        // the residual operand, requested ahead of the matrix loop (nothing it depends on): its latency passes under GEMM 2
        h16x4 xr[MT2][4][2];
        int xoff[MT2][4][2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int mrow = 32 * MT2 * mp + 32 * mt + 8 * j + 4 * h, t = t0 + 64 * nq + 32 * e + r;
                    const int off = t < Tf ? ((mrow >> 3) * Tf + t) * 16 + 8 * h : H_OOB;
                    xoff[mt][j][e] = off;
                    xr[mt][j][e] = __builtin_bit_cast(h16x4, __builtin_amdgcn_raw_buffer_load_b64(rR, off, 0, 0));
                }
        constexpr int NA2 = 4, AD2 = NA2 - 1;
        h16x8 aw[NA2][MT2];
        auto ldw = [&](int c, h16x8 (&d)[MT2]) {
            const int so = c * M * 32;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) d[mt] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, avw[mt], so, 0));
        };
#pragma unroll
        for (int c = 0; c < AD2 && c < R::NC2; ++c) ldw(c, aw[c % NA2]);
#pragma unroll
        for (int c = 0; c < R::NC2; ++c) {
            if (c + AD2 < R::NC2) ldw(c + AD2, aw[(c + AD2) % NA2]);
            h16x8 bb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) bb[e] = *reinterpret_cast<const h16x8*>(Bp + (size_t)(2 * c * BN + 32 * e) * 8);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[c % NA2][mt], bb[e], acc[mt][e], 0, 0, 0);
            if constexpr (MT2 == 2) asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
            else asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mrow = 32 * MT2 * mp + 32 * mt + 8 * j + 4 * h;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int t = t0 + 64 * nq + 32 * e + r;
                    const int off = xoff[mt][j][e];
                    const h16x4 xv = xr[mt][j][e];
                    float y[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) y[rr] = fmaf(acc[mt][e][4 * j + rr], p.out_scale, (float)xv[rr]);
                    if (p.Y) {
                        h16x4 v;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)y[rr];
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rY, off, 0, 0);
                    }
                    if (p.Yact) {
                        h16x4 v;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)elu1(y[rr] * p.act_scale);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rA, off, 0, 0);
                    }
                    if (p.Yf32 && t < Tf) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) p.Yf32[((size_t)b * M + mrow + rr) * Tf + t] = y[rr];
                    }
                }
            }
    }
}

template <class R>
hipError_t spec16_launch(const Spec16Args& a, hipStream_t s) {
    static std::atomic<unsigned> attr{0};
    if (R::SMEM > 64 * 1024) {
        int d = 0; (void)hipGetDevice(&d);
        const unsigned bit = 1u << (d & 31);
        if (!(attr.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec16_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::SMEM);
            if (e != hipSuccess) return e;
            attr.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const long long grid = (long long)((a.Tf + R::BN - 1) / R::BN) * a.B;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    std::string name;
    if (prof::enabled()) name = "spec16<" + std::to_string(R::N) + ",hop" + std::to_string(R::HOP) + (R::M != R::N ? "," + std::to_string(R::M) + "ch" : std::string()) + ">";
    const double Bd = a.B, Tf = a.Tf, Nn = R::N, Mm = R::M;
    prof::Scope ps(s, name.c_str(), Bd * Tf * (2.0 * 2.0 * (Nn + 2.0) * Nn + 2.0 * Mm * (Nn / 2 + 1)),
                   Bd * (4.0 * a.T + 2.0 * Mm * Tf * (1.0 + (a.Y ? 1.0 : 0.0) + (a.Yact ? 1.0 : 0.0) + (a.Yf32 ? 2.0 : 0.0))));
    hipLaunchKernelGGL((spec16_kernel<R>), dim3((unsigned)grid), dim3(256), R::SMEM, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same conv for the strided layers (KS = 2 * stride taps, stride >= 4), with x staged through LDS.  Straight from global memory a
// B fragment load touches one cache line PER LANE (consecutive output times are stride * 16 bytes apart), and the L1's tag rate, not
// the matrix pipe, sets the pace (measured: 515 / 322 / 220 TFLOP/s at stride 4 / 5 / 8).  Here a workgroup owns 256 rows x 128 output
// times; per 16-channel step the x window ((127 stride + KS) times x 2 channel groups) is copied by LDS-DMA -- coalesced, each piece
// fetched once per workgroup -- into a double buffer, skewed by one piece per `stride` pieces so that the fragment reads (lane stride =
// stride + 1 pieces, odd) are conflict-free: the DMA's destination is lane-linear, but each lane may fetch any piece, so the lanes of
// an instruction fetch the pieces whose skewed places they write (the holes fetch zeros).  A fragments stream from L2 through a ring of
// NA taps.  One barrier per 16 channels.  SHORT: layers with at most 64 outputs per clip take two clips per tile (64 columns each).
template <int KS, bool SHORT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv16s_kernel(Conv16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int S = KS / 2, NSUB = SHORT ? 2 : 1, CPS = 128 / NSUB;
    constexpr bool SKEW = (S % 2) == 0;
    constexpr int W = (CPS - 1) * S + KS;                        // window times of one sub-window
    constexpr int WP = SKEW ? W + W / S + 1 : W;                  // ... and its places in LDS
    constexpr int PIECES = 2 * NSUB * WP, ND = (PIECES + 255) / 256;
    constexpr int NA = KS == 10 ? 5 : 8, AD = NA - 1;
    static_assert(KS % NA == 0 && PIECES * 16 * 2 <= 160 * 1024, "geometry");
    h16* S0 = reinterpret_cast<h16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int Tin = p.Tin, Tout = p.Tout, Gk = p.w.Kp / 8, Gm = (p.M + 15) / 16 * 2, Mp = p.w.Mp, NKC = p.w.Kp / 16, nch = p.w.nchunks;
    const size_t xclip = (size_t)Gk * Tin * 8, yclip = (size_t)Gm * Tout * 8;      // halves per clip
    int b0, to0;
    if constexpr (SHORT) { b0 = 2 * blockIdx.x; to0 = 0; }
    else { const int ncol = (Tout + 127) / 128; b0 = blockIdx.x / ncol; to0 = (blockIdx.x - b0 * ncol) * 128; }
    const int m0 = (blockIdx.y * 4 + wave) * 64;
    const bool rows = m0 < p.M;                                  // (all waves copy and meet at the barriers; a wave past M computes nothing)
    const __amdgpu_buffer_rsrc_t rX = SHORT ? uniform_rsrc(p.X, (int)(xclip * 2 * p.B)) : uniform_rsrc(reinterpret_cast<const h16*>(p.X) + b0 * xclip, (int)(xclip * 2));
    const __amdgpu_buffer_rsrc_t rW = uniform_rsrc(p.w.wq, p.w.nchunks * Mp * 32);

    // ---- this thread's ND pieces of a window copy: place q = tid + 256 j of [2 groups][NSUB][WP]
    int dvo[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int q = tid + 256 * j;
        const int gi = q / (NSUB * WP), rem = q - gi * (NSUB * WP), sub = rem / WP, pos = rem - sub * WP;
        const bool hole = SKEW && (pos % (S + 1)) == S;
        const int u = SKEW ? pos - pos / (S + 1) : pos;
        const int t = to0 * S - p.pad + u;
        const bool ok = q < PIECES && !hole && u < W && t >= 0 && t < Tin && (!SHORT || b0 + sub < p.B);
        dvo[j] = ok ? (SHORT ? (b0 + sub) * (int)(xclip * 2) : 0) + (gi * Tin + t) * 16 : H_OOB;
    }
    auto copy = [&](int kc, int buf) {
        const int so = 2 * kc * Tin * 16;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int q0 = 256 * j + 64 * wave;                  // wave-uniform first place of this instruction
            h16* dst = S0 + (size_t)(buf * PIECES + q0) * 8;
            const int vo = dvo[j];
            if (q0 < PIECES) {
                if (q0 + lane < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, 0);
            }
        }
    };
    // ---- B fragment places: column 32 e + r -> (sub-window, local time tl): place (h * NSUB + sub) * WP + tl * (S + skew), + tap (+ 1 past the hole)
    int bpl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = 32 * e + r, sub = SHORT ? col / 64 : 0, tl = SHORT ? col % 64 : col;
        bpl[e] = (h * NSUB + sub) * WP + tl * (S + (SKEW ? 1 : 0));
    }
    int avoff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) avoff[mt] = (rows && m0 + 32 * mt + r < Mp) ? ((m0 + 32 * mt + r) * 2 + h) * 16 : H_OOB;
    h16x8 ar[NA][2];
    auto lda = [&](int chunk, h16x8 (&d)[2]) {
        const int so = chunk * Mp * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) d[mt] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, avoff[mt], so, 0));
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][e][i] = 0.f;

    copy(0, 0);
#pragma unroll
    for (int i = 0; i < AD; ++i) lda(i, ar[i % NA]);
    for (int kc = 0; kc < NKC; ++kc) {
        // everything older than the last AD taps' fragments has landed: this step's window (copied a step ago)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * AD) : "memory");
        RH_BARRIER();                                            // ... in every wave; and every wave has read the other buffer
        if (kc + 1 < NKC) copy(kc + 1, (kc + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);                       // the copy stays in front of this step's fragment loads (the wait above counts on it)
        const h16* Sb = S0 + (size_t)((kc & 1) * PIECES) * 8;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            // chunk kc * KS + i; ring slot = i % NA (KS is a multiple of NA)
            if (kc * KS + i + AD < nch) lda(kc * KS + i + AD, ar[(i + AD) % NA]);   // (only the last step skips any: nothing waits on a count after it)
            const int tp = i + (SKEW ? i / S : 0);
            h16x8 bb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = *reinterpret_cast<const h16x8*>(Sb + (size_t)(bpl[e] + tp) * 8);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[i % NA][mt], bb[e], acc[mt][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!rows) return;
    // ---- epilogue: rows m0 + 32 mt + 8 j + 4 h + rr, columns 32 e + r
    const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(p.Y ? reinterpret_cast<h16*>(p.Y) : reinterpret_cast<const h16*>(p.X), p.Y ? (int)(yclip * 2 * p.B) : 0);
    const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(p.Yact ? reinterpret_cast<h16*>(p.Yact) : reinterpret_cast<const h16*>(p.X), p.Yact ? (int)(yclip * 2 * p.B) : 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mrow = m0 + 32 * mt + 8 * j + 4 * h;
            float bias[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) bias[rr] = (p.bias && mrow + rr < p.M) ? p.bias[mrow + rr] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 32 * e + r;
                const int clip = SHORT ? b0 + col / 64 : b0, to = SHORT ? col % 64 : to0 + col;
                const bool ok = to < Tout && clip < p.B && mrow < p.M;
                const int off = ok ? clip * (int)(yclip * 2) + ((mrow >> 3) * Tout + to) * 16 + 8 * h : H_OOB;
                float y[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) y[rr] = (acc[mt][e][4 * j + rr] + bias[rr]) * p.out_scale;
                if (p.film && ok) {
                    const float* fl = p.film + (size_t)clip * p.film_stride + 2 * (mrow / (p.M / p.bands));
                    const float gam = fl[0], bet = fl[1];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) y[rr] = fmaf(gam, y[rr], bet);
                }
                if (p.Y) {
                    h16x4 v;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)y[rr];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rY, off, 0, 0);
                }
                if (p.Yact) {
                    h16x4 v;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) v[rr] = (h16)elu1(y[rr] * p.act_scale);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rA, off, 0, 0);
                }
                if (p.Yf32 && ok) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (mrow + rr < p.M) p.Yf32[((size_t)clip * p.M + mrow + rr) * Tout + to] = y[rr];
                }
            }
        }
}

template <int KS, bool SHORT>
hipError_t conv16s_launch(const Conv16Args& a, hipStream_t s) {
    constexpr int S = KS / 2, NSUB = SHORT ? 2 : 1, CPS = 128 / NSUB, W = (CPS - 1) * S + KS, WP = (S % 2) == 0 ? W + W / S + 1 : W;
    constexpr int SMEM = 2 * 2 * NSUB * WP * 16;
    static std::atomic<unsigned> attr{0};
    if (SMEM > 64 * 1024) {
        int d = 0; (void)hipGetDevice(&d);
        const unsigned bit = 1u << (d & 31);
        if (!(attr.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv16s_kernel<KS, SHORT>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
            if (e != hipSuccess) return e;
            attr.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    const long long gx = SHORT ? (a.B + 1) / 2 : (long long)((a.Tout + 127) / 128) * a.B;
    if (gx > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv16s_kernel<KS, SHORT>), dim3((unsigned)gx, (unsigned)((a.M + 255) / 256)), dim3(256), SMEM, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------
// The decoder's upsample unit (Conv16Args::up: a two-tap, stride-1 conv over the input frames whose rows are (phase, channel) pairs) with
// the x window staged through LDS.  Straight from global memory (conv16_kernel) every wave streams its own B fragments through the L1,
// whose 64 B per clock hold a 64 x 64 wave tile to about half the matrix pipe's rate (measured 320-680 TFLOP/s on the four units); here
// a workgroup owns 256 rows x 128 input frames, copies the window of 32 channels (4 groups x 129 frames) by LDS-DMA into a double buffer
// -- each piece fetched once per workgroup -- and every wave reads its fragments from there (consecutive frames are consecutive 16-byte
// pieces: conflict-free as they lie).  One barrier per 32 channels (4 chunks x 8 matrix instructions per wave); the A fragments of a
// step are requested two steps ahead into a three-step register ring.  SHORT: at most 64 frames per clip (the first unit: 50) -> two
// clips per tile.  Needs K % 32 == 0.
// The output: lane (frame l) of row (phase p, channel m) belongs at time l * up + p -- stored straight from the accumulators that is one
// 8-byte write per lane at a stride of up * 16 bytes, and those scattered writes took HALF the kernel's time (measured: 510-740 TFLOP/s
// with them, 870-1120 without).  So a workgroup owns a row BLOCK = all phases of up_mb channels (Conv16Args::up_mb, pack_up16), stages
// its tile in LDS in the output's own layout [group][time][8] (one pad piece per 16, which makes the strided writes conflict-free) --
// the window buffers are free by then -- and copies it out in whole contiguous runs, 16 bytes per lane.
template <bool SHORT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv16u_kernel(Conv16Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KCB = 2, GR = 2 * KCB, NCHS = 2 * KCB;         // 16-channel steps, channel groups and (channel step, tap) chunks per barrier
    constexpr int NSUB = SHORT ? 2 : 1, CPS = 128 / NSUB, W = CPS + 1;   // a sub-window's outputs l0 .. l0 + CPS - 1 read frames l0 - 1 .. l0 + CPS - 1
    constexpr int PIECES = GR * NSUB * W, ND = (PIECES + 255) / 256;
    h16* S0 = reinterpret_cast<h16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int Tin = p.Tin, Gk = p.w.Kp / 8, Mp = p.w.Mp, NKC = p.w.Kp / 16;
    const int upr = p.up, Mo = p.M / upr, To = Tin * upr;
    const size_t xclip = (size_t)Gk * Tin * 8, yclip = (size_t)((Mo + 15) / 16 * 2) * To * 8;      // halves per clip
    int b0, to0;
    if constexpr (SHORT) { b0 = 2 * blockIdx.x; to0 = 0; }
    else { const int ncol = (Tin + 127) / 128; b0 = blockIdx.x / ncol; to0 = (blockIdx.x - b0 * ncol) * 128; }
    const int MB = p.up_mb, RB = upr * MB;                       // rows of this workgroup's block (<= 256): (phase, channel in block)
    const int m0 = blockIdx.y * RB + wave * 64;                  // this wave's first row
    const bool rows = wave * 64 < RB;                            // (all waves copy and meet at the barriers; a wave past the block computes nothing)
    const __amdgpu_buffer_rsrc_t rX = SHORT ? uniform_rsrc(p.X, (int)(xclip * 2 * p.B)) : uniform_rsrc(reinterpret_cast<const h16*>(p.X) + b0 * xclip, (int)(xclip * 2));
    const __amdgpu_buffer_rsrc_t rW = uniform_rsrc(p.w.wq, p.w.nchunks * Mp * 32);

    // ---- this thread's ND pieces of a window copy: place q = tid + 256 j of [GR groups][NSUB][W]; group gl of step st is channel group GR * st + gl
    int dvo[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int q = tid + 256 * j;
        const int gl = q / (NSUB * W), rem = q - gl * (NSUB * W), sub = rem / W, pos = rem - sub * W;
        const int t = to0 - 1 + pos;
        const bool ok = q < PIECES && t >= 0 && t < Tin && (!SHORT || b0 + sub < p.B);
        dvo[j] = ok ? (SHORT ? (b0 + sub) * (int)(xclip * 2) : 0) + (gl * Tin + t) * 16 : H_OOB;
    }
    auto copy = [&](int st, int buf) {
        const int so = GR * st * Tin * 16;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int q0 = 256 * j + 64 * wave;                  // wave-uniform first place of this instruction
            h16* dst = S0 + (size_t)(buf * PIECES + q0) * 8;
            const int vo = dvo[j];
            if (q0 < PIECES) {
                if (q0 + lane < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, 0);
            }
        }
    };
    // ---- B fragment places: column 32 e + r -> (sub-window, local frame tl); chunk (channel step kl, tap i), k-half h: group 2 kl + h, frame tl + i
    int bpl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = 32 * e + r, sub = SHORT ? col / 64 : 0, tl = SHORT ? col % 64 : col;
        bpl[e] = (h * NSUB + sub) * W + tl;
    }
    int avoff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) avoff[mt] = (rows && m0 + 32 * mt + r < Mp) ? ((m0 + 32 * mt + r) * 2 + h) * 16 : H_OOB;
    h16x8 ar[3][NCHS][2];                                        // [step mod 3][chunk of the step][row tile]: requested TWO steps ahead
    auto lda = [&](int st, h16x8 (&d)[NCHS][2]) {                 // chunk index of (channel step kc, tap i) = 2 kc + i (pack_up16)
#pragma unroll
        for (int c = 0; c < NCHS; ++c) {
            const int so = (NCHS * st + c) * Mp * 32;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) d[c][mt] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, avoff[mt], so, 0));
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][e][i] = 0.f;

    const int nst = NKC / KCB;                                   // K % 32 == 0 (launcher)
    lda(0, ar[0]);
    copy(0, 0);
    if (nst > 1) lda(1, ar[1]);
    // Order in the queue when step st begins: ... fragments(st) | window(st) | fragments(st + 1).  One L2 round trip under this load is
    // longer than a step (32 matrix instructions per wave), so the fragments run two steps ahead and the wait at the head of a step
    // leaves the youngest set in flight: everything older than those 2 * NCHS loads has landed.
    auto step = [&](int st, auto par) {
        constexpr int P = decltype(par)::value;
        if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NCHS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RH_BARRIER();                                            // this step's window is there in every wave; and every wave has read the other buffer
        if (st + 1 < nst) copy(st + 1, (st + 1) & 1);
        if (st + 2 < nst) lda(st + 2, ar[(P + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
        const h16* Sb = S0 + (size_t)((st & 1) * PIECES) * 8;
#pragma unroll
        for (int c = 0; c < NCHS; ++c) {
            const int kl = c >> 1, i = c & 1;
            h16x8 bb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = *reinterpret_cast<const h16x8*>(Sb + (size_t)(bpl[e] + 2 * kl * NSUB * W + i) * 8);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[P][c][mt], bb[e], acc[mt][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int st = 0; st < nst; st += 3) {
        step(st, std::integral_constant<int, 0>{});
        if (st + 1 < nst) step(st + 1, std::integral_constant<int, 1>{});
        if (st + 2 < nst) step(st + 2, std::integral_constant<int, 2>{});
    }
    // ---- epilogue: the tile through LDS in the output's layout.  Piece (group gl of the block, sub-window, local time ts = tl * up + phase) sits at
    // ((gl * NSUB + sub) * TP + ts + ts / 16) * 16 bytes; a lane writes its 4 channels (8 bytes) of one piece.
    const int TS = CPS * upr, TP = TS + TS / 16 + 1, NG = MB / 8;
    const __amdgpu_buffer_rsrc_t rY = uniform_rsrc(p.Y ? reinterpret_cast<h16*>(p.Y) : reinterpret_cast<const h16*>(p.X), p.Y ? (int)(yclip * 2 * p.B) : 0);
    const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(p.Yact ? reinterpret_cast<h16*>(p.Yact) : reinterpret_cast<const h16*>(p.X), p.Yact ? (int)(yclip * 2 * p.B) : 0);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0 ? !p.Y : !p.Yact) continue;                // (uniform)
        RH_BARRIER();                                            // the window buffers (pass 0) / the staged tile (pass 1) have been read by every wave
        if (rows) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lr = wave * 64 + 32 * mt + 8 * j + 4 * h;                 // row in the block
                    if (lr >= RB) continue;
                    const int ph = lr / MB, ml = lr - ph * MB;
                    float bias[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) bias[rr] = p.bias ? p.bias[blockIdx.y * MB + ml + rr] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = 32 * e + r, sub = SHORT ? col / 64 : 0, tl = SHORT ? col % 64 : col;
                        const int ts = tl * upr + ph;
                        h16x4 v;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const float y = (acc[mt][e][4 * j + rr] + bias[rr]) * p.out_scale;
                            v[rr] = (h16)(pass == 0 ? y : elu1(y * p.act_scale));
                        }
                        *reinterpret_cast<h16x4*>(S0 + (size_t)(((ml >> 3) * NSUB + sub) * TP + ts + (ts >> 4)) * 8 + (ml & 4)) = v;
                    }
                }
        }
        RH_BARRIER();                                            // the tile is staged
        const __amdgpu_buffer_rsrc_t& rO = pass == 0 ? rY : rA;
        const int npieces = NG * NSUB * TS;
        for (int i = tid; i < npieces; i += 256) {
            const int gl = i / (NSUB * TS), rem = i - gl * (NSUB * TS), sub = rem / TS, ts = rem - sub * TS;
            const int clip = SHORT ? b0 + sub : b0, t = (SHORT ? 0 : to0 * upr) + ts;
            const bool ok = t < To && clip < p.B;
            const int off = ok ? clip * (int)(yclip * 2) + ((blockIdx.y * NG + gl) * To + t) * 16 : H_OOB;
            const u32x4 v = *reinterpret_cast<const u32x4*>(S0 + (size_t)((gl * NSUB + sub) * TP + ts + (ts >> 4)) * 8);
            __builtin_amdgcn_raw_buffer_store_b128(v, rO, off, 0, 0);
        }
    }
}

template <bool SHORT>
hipError_t conv16u_launch(const Conv16Args& a, hipStream_t s) {
    constexpr int NSUB = SHORT ? 2 : 1, CPS = 128 / NSUB, W = CPS + 1;
    const int TS = CPS * a.up, TP = TS + TS / 16 + 1;
    const int smem = std::max(2 * 4 * NSUB * W * 16, (a.up_mb / 8) * NSUB * TP * 16);     // the window double buffer, then the staged output tile
    static std::atomic<unsigned> attr{0};
    {
        int d = 0; (void)hipGetDevice(&d);
        const unsigned bit = 1u << (d & 31);
        if (!(attr.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv16u_kernel<SHORT>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e != hipSuccess) return e;
            attr.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    if (smem > 80 * 1024) return hipErrorInvalidValue;
    const long long gx = SHORT ? (a.B + 1) / 2 : (long long)((a.Tin + 127) / 128) * a.B;
    if (gx > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv16u_kernel<SHORT>), dim3((unsigned)gx, (unsigned)(a.M / (a.up * a.up_mb))), dim3(256), smem, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------
// Detector head for the mean-probability output (detector.py:300-310 + core.py:577-580): L2Norm over the latent's channels
// (seanet.py:288-318: y / max(||y||, 1e-12) * sqrt(D)), the composed ConvTranspose1d(k = s = hop) -> Conv1d(O -> nb, 1) as one GEMM per
// frame tile against wc[D][nb * hop], sigmoid, mean over time -- the [B, nb, T] logits never exist.  One workgroup per clip (fixed
// summation order: deterministic); wave w owns bits nb/4 * w ..; the B fragments of a 64-frame tile (D <= 128: 16 of them) stay in
// registers over all of a wave's row tiles, A fragments stream from L2.  hop % 32 == 0, nb % 4 == 0, D % 16 == 0.
struct Head16Args { const float* Y; H16Weight w; const float* bc; float* mean_prob; int B, D, nb, hop, Fr, T; };

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void head16_kernel(Head16Args p) {
    __shared__ __attribute__((aligned(16))) float ys[128 * 64];       // y tile [D][64] f32
    __shared__ __attribute__((aligned(16))) h16 zs[16 * 64 * 8];       // z tile, c8 [D/8][64][8]
    __shared__ float inv[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int b = blockIdx.x, D = p.D, Fr = p.Fr, NC = D / 16, bpw = p.nb / 4, mtb = p.hop / 32;
    const float* Yb = p.Y + (size_t)b * D * Fr;
    const __amdgpu_buffer_rsrc_t rW = uniform_rsrc(p.w.wq, p.w.nchunks * p.w.Mp * 32);
    const int Mp = p.w.Mp;
    float total[8];                                              // per bit of this wave (bpw <= 8), lane-partial
#pragma unroll
    for (int i = 0; i < 8; ++i) total[i] = 0.f;
    for (int f0 = 0; f0 < Fr; f0 += 64) {
        __syncthreads();
        for (int i = tid; i < D * 64; i += 256) {
            const int m = i >> 6, c = i & 63;
            ys[i] = f0 + c < Fr ? Yb[(size_t)m * Fr + f0 + c] : 0.f;
        }
        __syncthreads();
        if (tid < 64) {
            float ss = 0.f;
            for (int m = 0; m < D; ++m) ss = fmaf(ys[m * 64 + tid], ys[m * 64 + tid], ss);
            inv[tid] = sqrtf((float)D) / fmaxf(sqrtf(ss), 1e-12f);
        }
        __syncthreads();
        for (int i = tid; i < (D / 8) * 64; i += 256) {
            const int g = i >> 6, c = i & 63;
            h16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (h16)(ys[(8 * g + j) * 64 + c] * inv[c]);
            *reinterpret_cast<h16x8*>(zs + (size_t)i * 8) = o;
        }
        __syncthreads();
        h16x8 bf[8][2];                                          // B fragments: chunk c, frame tile e (D <= 128)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (c < NC) bf[c][e] = *reinterpret_cast<const h16x8*>(zs + (size_t)((2 * c + h) * 64 + 32 * e + r) * 8);
        for (int bi = 0; bi < bpw; ++bi) {
            const int bit = wave * bpw + bi;
            const float bcv = p.bc[bit];
            float s = 0.f;
            for (int mt = 0; mt < mtb; ++mt) {
                const int m0 = bit * p.hop + 32 * mt;
                const int avoff = ((m0 + r) * 2 + h) * 16;
                h16x8 a[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int so = c * Mp * 32;
                    if (c < NC) a[c] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, avoff, so, 0));
                }
                f32x16 acc[2];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < NC) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], bf[c][e], acc[e], 0, 0, 0);
                    }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = f0 + 32 * e + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int j = 32 * mt + 8 * (i >> 2) + 4 * h + (i & 3);
                        const bool ok = f < Fr && f * p.hop + j < p.T;
                        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-(acc[e][i] + bcv)));   // 1-ulp reciprocal: a full-precision divide is ten more instructions per logit
                        s += ok ? sg : 0.f;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i == bi) total[i] += s;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < bpw) {
            float v = total[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) p.mean_prob[(size_t)b * p.nb + wave * bpw + i] = v / (float)p.T;
        }
    }
}

// conv_pre (SConv1d 1 -> C, k taps, causal; modules/seanet.py:657-663) straight into the c8 layout: a thread owns one time step and
// walks the channel groups (a wave's store of one group is 1 KB contiguous).  KS > 0: the tap count at compile time (the taps of a
// group arrive as a few wide scalar loads instead of one load and one wait per tap).
template <int KS>
__global__ __launch_bounds__(256) void conv_pre16_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         h16* __restrict__ Y, int C, int T, int ks_rt, float in_scale) {
    const int ks = KS > 0 ? KS : ks_rt;
    constexpr int NX = KS > 0 ? KS : 16;
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float* xb = x + (size_t)b * T;
    float xv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int ti = t - (ks - 1) + i;
        xv[i] = (i < ks && ti >= 0) ? xb[ti] * in_scale : 0.f;
    }
    const int G = C / 8;
    for (int g = 0; g < G; ++g) {
        h16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            float y = bias ? bias[c] : 0.f;
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (i < ks) y = fmaf(w[c * ks + i], xv[i], y);
            o[j] = (h16)y;
        }
        *reinterpret_cast<h16x8*>(Y + (((size_t)b * G + g) * T + t) * 8) = o;
    }
}

// [B][C][T] f32 -> c8 f16 with Cp = roundup(C, 16) channels (rows past C are zero), optionally ELU(scale * x) on the way
__global__ __launch_bounds__(256) void f32_to_c8_kernel(const float* __restrict__ X, h16* __restrict__ Y, int C, int Gp, int T, float scale, int elu) {
    const int b = blockIdx.z, g = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    h16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * g + j;
        float v = c < C ? X[((size_t)b * C + c) * T + t] * scale : 0.f;
        if (elu) v = elu1(v);
        o[j] = (h16)v;
    }
    *reinterpret_cast<h16x8*>(Y + (((size_t)b * Gp + g) * T + t) * 8) = o;
}

__global__ __launch_bounds__(256) void c8_to_f32_kernel(const h16* __restrict__ X, float* __restrict__ Y, int C, int Gp, int T) {
    const int b = blockIdx.z, g = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const h16x8 v = *reinterpret_cast<const h16x8*>(X + (((size_t)b * Gp + g) * T + t) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * g + j;
        if (c < C) Y[((size_t)b * C + c) * T + t] = (float)v[j];
    }
}

// L2Norm over the channels of a latent [B][D][Fr] f32 (seanet.py:288-318) written as c8 f16: one thread per (clip, frame).
__global__ __launch_bounds__(256) void l2norm_c8_kernel(const float* __restrict__ X, h16* __restrict__ Y, int D, int Fr, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / Fr, t = i - b * Fr;
    const float* xb = X + (size_t)b * D * Fr + t;
    float ss = 0.f;
    for (int c = 0; c < D; ++c) { const float v = xb[(size_t)c * Fr]; ss = fmaf(v, v, ss); }
    const float inv = sqrtf((float)D) / fmaxf(sqrtf(ss), 1e-12f);
    const int G = (D + 15) / 16 * 2;
    for (int g = 0; g < G; ++g) {
        h16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = 8 * g + j; o[j] = (h16)(c < D ? xb[(size_t)c * Fr] * inv : 0.f); }
        *reinterpret_cast<h16x8*>(Y + (((size_t)b * G + g) * Fr + t) * 8) = o;
    }
}

// Decoder tail on the pre-activated c8 input (the last ResnetBlock writes ELU(s * y)): one thread per output sample walks the channel
// groups; a wave's 16-byte pieces of one (group, tap) are contiguous, each piece is read by KS neighbouring threads (L1 hits).  Taps
// come as wide scalar loads (KS at compile time).  Sums are f32: f16 x f32 multiply-adds straight from the halves.
template <int KS>
__global__ __launch_bounds__(256) void tail16_kernel(const h16* __restrict__ A, const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ x, float* __restrict__ out, int C, int Tin, int T, float out_scale) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int G = (C + 15) / 16 * 2;
    const h16* ab = A + (size_t)b * G * Tin * 8;
    float y = bias ? bias[0] : 0.f;
    for (int g = 0; g < C / 8; ++g) {
        const float* wg = w + (size_t)g * 8 * KS;              // w[c][i], c = 8 g + j
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int ti = t - (KS - 1) + i;
            if (ti < 0 || ti >= Tin) continue;
            const h16x8 v = *reinterpret_cast<const h16x8*>(ab + ((size_t)g * Tin + ti) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) y = fmaf((float)v[j], wg[j * KS + i], y);
        }
    }
    float r = tanhf(y * out_scale);
    if (x) r += x[(size_t)b * T + t];
    out[(size_t)b * T + t] = r;
}

}  // namespace

bool rh_supported(const RhArgs& a) {
    if (!(a.C == 32 || a.C == 64 || a.C == 96 || a.C == 128 || a.C == 192 || a.C == 256 || a.C == 384 || a.C == 512 || a.C == 768)) return false;
    if (!a.X || (!a.Y && !a.Yact) || a.T < 1 || a.B < 1 || !a.w1.wq || !a.w2.wq || !a.tab1 || !a.tab2) return false;
    if (a.w1.M != a.C || a.w1.K != a.C || a.w2.M != a.C || a.w2.K != a.C || a.w1.Mp != a.C || a.w2.Mp != a.C || a.w1.Kp != a.C || a.w2.Kp != a.C) return false;
    if ((long long)a.C * a.T * 2 >= H_OOB) return false;
    return al16(a.X) && (!a.Y || al16(a.Y)) && (!a.Yact || al16(a.Yact)) && al16(a.w1.wq) && al16(a.w2.wq);
}

hipError_t launch_resblock16(const RhArgs& a, hipStream_t s) {
    if (!rh_supported(a)) return hipErrorNotSupported;
    switch (a.C) {                                               // <C, column groups, NT, waves per SIMD, A ring, B ring depth, resident A, strips per wave>
        case 32: return rh_pick_out<RH<32, 8, 2, 4>>(a, s);      // 1 x 8 waves, 484-column windows (the locator's first stage)
        case 64: return rh_pick_out<RH<64, 4, 2, 4, 4, 2, false>>(a, s);   // 2 x 4 waves, 244-column windows, weights streamed (a ring of 4: the packed stencil needs the registers)
#ifndef RH_CFG96
#define RH_CFG96 2
#endif
#ifndef RH_CFG192
#define RH_CFG192 0
#endif
#ifndef RH_CFG384
#define RH_CFG384 0
#endif
#if RH_CFG96 == 1
        case 96: return rh_pick_out<RH<96, 5, 2, 4, 6, 2, false>>(a, s);    // 3 x 5 waves, 304-column windows, one workgroup per CU
#elif RH_CFG96 == 2
        case 96: return rh_pick_out<RH<96, 1, 2, 4, 4, 2, false>>(a, s);    // 3 x 1 waves, 64-column windows, five workgroups per CU
#elif RH_CFG96 == 3
        case 96: return rh_pick_out<RH<96, 4, 2, 4, 6, 2, false>>(a, s);    // 3 x 4 waves, 244-column windows, one workgroup per CU
#else
        case 96: return rh_pick_out<RH<96, 2, 2, 4, 6, 2, false>>(a, s);    // 3 x 2 waves, 124-column windows, two workgroups per CU
#endif
        case 128: return rh_pick_out<RH<128, 2, 2, 4, 4, 2, false>>(a, s);  // 4 x 2 waves, 124-column windows (32 KB), weights streamed
#if RH_CFG192 == 1
        case 192: return rh_pick_out<RH<192, 1, 2, 3, 8, 2, false>>(a, s);  // 6 x 1 waves, 64-column windows, two workgroups per CU
#elif RH_CFG192 == 2
        case 192: return rh_pick_out<RH<192, 2, 2, 3, 4, 1, false, 2>>(a, s);  // 3 x 2 waves of two strips each, 124-column windows, two workgroups per CU
#else
        case 192: return rh_pick_out<RH<192, 2, 2, 3, 8, 2, false>>(a, s);  // 6 x 2 waves, 124-column windows
#endif
        case 256: return rh_pick_out<RH<256, 1, 2, 4, 4, 1>>(a, s); // 8 x 1 waves, 64-column windows (32 KB)
#if RH_CFG384 == 1
        case 384: return rh_pick_out<RH<384, 1, 2, 3, 4, 1, false, 2>>(a, s); // 6 x 1 waves of two strips each, 64-column windows, two workgroups per CU
#else
        case 384: return rh_pick_out<RH<384, 1, 2, 3, 4, 1>>(a, s); // 12 x 1 waves, 64-column windows (48 KB)
#endif
        case 512: return rh_pick_out<RH<512, 1, 2, 4, 4, 1>>(a, s); // 16 x 1 waves, 64-column windows (64 KB)
        default: return rh_pick_out<RH<768, 1, 2, 3, 4, 1, false, 2>>(a, s); // 12 x 1 waves of two strips each, 64-column windows (96 KB)
    }
}

hipError_t launch_conv16(const Conv16Args& a, hipStream_t s) {
    if (!a.X || !a.w.wq || (!a.Y && !a.Yact && !a.Yf32) || a.B < 1 || a.M < 1 || a.Tin < 1 || a.Tout < 1 || a.ks < 1 || a.stride < 1) return hipErrorInvalidValue;
    if (a.w.Kp % 16 || a.w.Mp % 32 || a.w.nchunks % 8 || a.w.nchunks < a.ks * (a.w.Kp / 16) || a.w.M != a.M) return hipErrorInvalidValue;
    if ((long long)a.w.Kp * a.Tin * 2 >= H_OOB || (long long)round_up(a.M, 16) * a.Tout * 2 >= H_OOB || (long long)a.w.nchunks * a.w.Mp * 32 >= H_OOB)
        return hipErrorInvalidValue;
    if (!al16(a.X) || !al16(a.w.wq) || (a.Y && !al16(a.Y)) || (a.Yact && !al16(a.Yact)) || (a.resid && !al16(a.resid))) return hipErrorInvalidValue;
    if (a.up < 0 || (a.up > 0 && (a.M % a.up || a.resid || a.Yf32 || a.stride != 1 || (a.M / a.up) % 16 || a.up_mb < 4 || (a.up_mb & 3) || (a.M / a.up) % a.up_mb)))
        return hipErrorInvalidValue;
    const int Mo = a.up > 0 ? a.M / a.up : a.M, upr = a.up > 0 ? a.up : 1;
    if ((a.Y || a.Yact || a.resid) && (Mo % 16)) return hipErrorInvalidValue;   // c8 outputs: whole 16-channel group pairs
    if (a.film && (a.bands < 1 || Mo % a.bands || (Mo / a.bands) % 4 || a.film_stride < 2 * a.bands)) return hipErrorInvalidValue;
    if ((long long)round_up(Mo, 16) * a.Tout * upr * 2 >= H_OOB) return hipErrorInvalidValue;
    // few outputs per clip: the clips' columns as one run (flat), four waves side by side on the same rows of W (its fragments are L1 hits
    // for three of them); else per-clip column tiles with the waves stacked over the rows
    const long long xbytes = (long long)a.w.Kp * a.Tin * 2 * a.B, ybytes = (long long)round_up(Mo, 16) * a.Tout * upr * 2 * a.B;
    const bool flat = a.Tout < 512 && xbytes < H_OOB && ybytes < H_OOB;
    // the strided layers' own kernel (x through LDS): 2 * stride taps, stride 4 / 5 / 8, no residual, at least 256 rows
    const bool staged = a.ks == 2 * a.stride && a.pad == a.stride && (a.ks == 8 || a.ks == 10 || a.ks == 16) && !a.resid && !a.up && a.M >= 256 && ybytes < H_OOB &&
                        (a.Tout > 64 || xbytes < H_OOB);
    // the upsample form through LDS: two taps, whole 32-channel steps, no FiLM; the clip-spanning offsets of the two-clip tiles must fit the sentinel
    const bool ups = a.up > 0 && a.ks == 2 && a.pad == 1 && a.w.Kp % 32 == 0 && !a.film && a.up_mb % 8 == 0 && a.up * a.up_mb <= 256 && a.up * a.up_mb >= 128 &&
                     ybytes < H_OOB && (a.Tin > 64 || xbytes < H_OOB);
    std::string name;
    if (prof::enabled())
        name = "conv16<k" + std::to_string(a.ks) + ",s" + std::to_string(a.stride) + "," + std::to_string(a.M) + "x" + std::to_string(a.w.K) +
               (a.up ? ",up" + std::to_string(a.up) : std::string()) + ((staged || ups) ? ",lds>" : (flat ? ",flat>" : ">"));
    const double Bd = a.B, M = a.M;
    prof::Scope ps(s, name.c_str(), 2.0 * Bd * M * a.ks * (double)a.w.K * a.Tout,
                   Bd * (2.0 * a.w.Kp * a.Tin + (a.resid ? 2.0 : 0.0) * M * a.Tout + (a.Y ? 2.0 : 0.0) * M * a.Tout + (a.Yact ? 2.0 : 0.0) * M * a.Tout +
                         (a.Yf32 ? 4.0 : 0.0) * M * a.Tout));
    if (ups) return a.Tin <= 64 ? conv16u_launch<true>(a, s) : conv16u_launch<false>(a, s);
    if (staged) {
        const bool sh = a.Tout <= 64;
        if (a.ks == 8) return sh ? conv16s_launch<8, true>(a, s) : conv16s_launch<8, false>(a, s);
        if (a.ks == 10) return sh ? conv16s_launch<10, true>(a, s) : conv16s_launch<10, false>(a, s);
        return sh ? conv16s_launch<16, true>(a, s) : conv16s_launch<16, false>(a, s);
    }
    if (flat) {
        const int wgm = a.M >= 512 ? 1 : (a.M >= 128 ? 2 : 1), wgn = 4 / wgm;
        const long long ncb = ((long long)a.B * a.Tout + 64 * wgn - 1) / (64 * wgn), nmb = (a.M + 64 * wgm - 1) / (64 * wgm);
        if (ncb * nmb > 0x7fffffffLL) return hipErrorInvalidValue;
        dim3 grid((unsigned)(ncb * nmb));
        if (wgm == 1) hipLaunchKernelGGL((conv16_kernel<1, 4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv16_kernel<2, 2, true>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const int wgm = a.M >= 256 ? 4 : (a.M >= 128 ? 2 : 1), wgn = 4 / wgm;
    const int ncol = (a.Tout + 64 * wgn - 1) / (64 * wgn);
    const long long gx = (long long)ncol * a.B;
    if (gx > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)gx, (unsigned)((a.M + 64 * wgm - 1) / (64 * wgm)));
    if (wgm == 4) hipLaunchKernelGGL((conv16_kernel<4, 1, false>), grid, dim3(256), 0, s, a);
    else if (wgm == 2) hipLaunchKernelGGL((conv16_kernel<2, 2, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv16_kernel<1, 4, false>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_spec16(const Spec16Args& a, hipStream_t s) {
    if (!a.wav || !a.resid || (!a.Y && !a.Yact && !a.Yf32) || !a.cosw.wq || !a.sinw.wq || !a.cosl.wq || !a.sinl.wq || !a.pw.wq || a.B < 1 || a.T < 1) return hipErrorInvalidValue;
    const int N = a.n_fft, M = a.pw.M;
    if (a.Tf != (a.T + a.hop - 1) / a.hop || (M != N && 2 * M != N) || a.pw.K != N / 2 + 1 || a.pw.Mp != M || a.pw.Kp != N / 2 + 16 || a.cosw.M != N / 2 || a.cosw.K != N ||
        a.sinw.M != N / 2 || a.sinw.K != N || a.cosw.Mp != N / 2 || a.sinw.Mp != N / 2 || a.cosw.nchunks < N / 16 || a.sinw.nchunks < N / 16 || a.cosl.nchunks < N / 16 || a.sinl.nchunks < N / 16)
        return hipErrorNotSupported;
    if ((long long)M * a.Tf * 2 >= H_OOB || !al16(a.resid) || (a.Y && !al16(a.Y)) || (a.Yact && !al16(a.Yact))) return hipErrorNotSupported;
    if (M == N) {                                               // generator / detector scales
        if (N == 64 && a.hop == 1) return spec16_launch<SP<64, 1>>(a, s);
        if (N == 128 && a.hop == 2) return spec16_launch<SP<128, 2>>(a, s);
        if (N == 256 && a.hop == 8) return spec16_launch<SP<256, 8>>(a, s);
        if (N == 512 && a.hop == 40) return spec16_launch<SP<512, 40>>(a, s);
        if (N == 1024 && a.hop == 320) return spec16_launch<SP<1024, 320>>(a, s);
    } else {                                                    // the locator's: half as many channels as DFT points
        if (N == 64 && a.hop == 1) return spec16_launch<SP<64, 1, 32>>(a, s);
        if (N == 128 && a.hop == 4) return spec16_launch<SP<128, 4, 64>>(a, s);
        if (N == 256 && a.hop == 32) return spec16_launch<SP<256, 32, 128>>(a, s);
    }
    return hipErrorNotSupported;
}

hipError_t launch_head16(const float* Y, const H16Weight& w, const float* bc, float* mean_prob, int B, int D, int nb, int hop, int Fr, int T, hipStream_t s) {
    if (!Y || !w.wq || !bc || !mean_prob || B < 1 || Fr < 1 || T < 1) return hipErrorInvalidValue;
    if (D % 16 || D > 128 || nb % 4 || nb > 32 || hop % 32 || w.M != nb * hop || w.K != D || w.Kp != D || w.Mp != nb * hop || w.nchunks < D / 16) return hipErrorNotSupported;
    prof::Scope ps(s, "head16", 2.0 * B * D * (double)nb * hop * Fr, (double)B * (4.0 * D * Fr + 4.0 * nb));
    Head16Args a{Y, w, bc, mean_prob, B, D, nb, hop, Fr, T};
    hipLaunchKernelGGL(head16_kernel, dim3((unsigned)B), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_pre16(const float* x, const float* w, const float* bias, void* Y, int B, int C, int T, int ks, float in_scale, hipStream_t s) {
    if (!x || !w || !Y || B < 1 || C < 8 || (C % 8) || T < 1 || ks < 1 || ks > 16 || B > 65535) return hipErrorInvalidValue;
    prof::Scope ps(s, "conv_pre16", 2.0 * B * C * ks * (double)T, (double)B * T * (4.0 + 2.0 * C));
    if (ks == 7) hipLaunchKernelGGL(conv_pre16_kernel<7>, dim3((T + 255) / 256, B), dim3(256), 0, s, x, w, bias, reinterpret_cast<h16*>(Y), C, T, ks, in_scale);
    else if (ks == 5) hipLaunchKernelGGL(conv_pre16_kernel<5>, dim3((T + 255) / 256, B), dim3(256), 0, s, x, w, bias, reinterpret_cast<h16*>(Y), C, T, ks, in_scale);
    else hipLaunchKernelGGL(conv_pre16_kernel<0>, dim3((T + 255) / 256, B), dim3(256), 0, s, x, w, bias, reinterpret_cast<h16*>(Y), C, T, ks, in_scale);
    return hipGetLastError();
}

hipError_t launch_f32_to_c8(const float* X, void* Y, int B, int C, int T, float scale, int elu, hipStream_t s) {
    const int Gp = round_up(C, 16) / 8;
    if (!X || !Y || B < 1 || B > 65535 || C < 1 || Gp > 65535 || T < 1) return hipErrorInvalidValue;
    prof::Scope ps(s, "f32_to_c8", 0.0, (double)B * T * (4.0 * C + 16.0 * Gp));
    hipLaunchKernelGGL(f32_to_c8_kernel, dim3((T + 255) / 256, Gp, B), dim3(256), 0, s, X, reinterpret_cast<h16*>(Y), C, Gp, T, scale, elu);
    return hipGetLastError();
}

hipError_t launch_l2norm_c8(const float* X, void* Y, int B, int D, int Fr, hipStream_t s) {
    if (!X || !Y || B < 1 || D < 1 || Fr < 1 || (long long)B * Fr > 0x7fffffffLL) return hipErrorInvalidValue;
    prof::Scope ps(s, "l2norm_c8", 3.0 * B * D * (double)Fr, (double)B * Fr * (4.0 * D + 2.0 * round_up(D, 16)));
    const int total = B * Fr;
    hipLaunchKernelGGL(l2norm_c8_kernel, dim3((total + 255) / 256), dim3(256), 0, s, X, reinterpret_cast<h16*>(Y), D, Fr, total);
    return hipGetLastError();
}

hipError_t launch_tail16(const void* A16, const float* w, const float* bias, const float* x, float* out, int B, int C, int Tin, int T, int ks, float out_scale,
                         hipStream_t s) {
    if (!A16 || !w || !out || B < 1 || B > 65535 || C < 8 || (C % 8) || Tin < 1 || T < 1 || T > Tin) return hipErrorInvalidValue;
    if (ks != 5 && ks != 7 && ks != 3) return hipErrorNotSupported;
    prof::Scope ps(s, "tail16", 2.0 * B * C * ks * (double)T, (double)B * (2.0 * C * Tin + 8.0 * T));
    const dim3 grid((T + 255) / 256, B);
    const h16* A = reinterpret_cast<const h16*>(A16);
    if (ks == 5) hipLaunchKernelGGL(tail16_kernel<5>, grid, dim3(256), 0, s, A, w, bias, x, out, C, Tin, T, out_scale);
    else if (ks == 7) hipLaunchKernelGGL(tail16_kernel<7>, grid, dim3(256), 0, s, A, w, bias, x, out, C, Tin, T, out_scale);
    else hipLaunchKernelGGL(tail16_kernel<3>, grid, dim3(256), 0, s, A, w, bias, x, out, C, Tin, T, out_scale);
    return hipGetLastError();
}

hipError_t launch_c8_to_f32(const void* X, float* Y, int B, int C, int T, hipStream_t s) {
    const int Gp = round_up(C, 16) / 8;
    if (!X || !Y || B < 1 || B > 65535 || C < 1 || Gp > 65535 || T < 1) return hipErrorInvalidValue;
    prof::Scope ps(s, "c8_to_f32", 0.0, (double)B * T * (4.0 * C + 16.0 * Gp));
    hipLaunchKernelGGL(c8_to_f32_kernel, dim3((T + 255) / 256, Gp, B), dim3(256), 0, s, reinterpret_cast<const h16*>(X), Y, C, Gp, T);
    return hipGetLastError();
}

}  // namespace wv
