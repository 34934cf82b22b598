// Whole SEANetResnetBlock in ONE launch, raw in / raw out, for the narrow layers (C = 64, 96, 128, 192;
// modules/seanet.py:245-281 with dws_conv_block :39-116):
//
//     y = x + s * ( DW5( W2 @ ELU( DW5( W1 @ ELU(c*x) ) + b1 ) ) + b2 )
//
// As two K1 launches such a block crosses HBM five times (read x, write u, read u, read x, write y) and both
// launches are bandwidth- or latency-bound (profiles/r02_chunked_batch_infinity_cache.txt: 49-80 TFLOP/s).  Here
// a PERSISTENT workgroup owns all C channels of one time window after the other:
//
//   * x arrives raw, once: every thread fetches its 16-byte pieces of the NEXT window into registers while the
//     current one computes (issued at the start of epilogue 1, consumed a whole tile later), activates them
//     (scale -> ELU, once per element) and writes them into the window buffer S[C][WD] in LDS -- the natural
//     [k][t] layout the B operand is read from.
//   * The weights never touch LDS: a wave owns one 32-row strip of W1 / W2 and streams its A fragments
//     (two 16-byte loads per 16-deep chunk, L2-resident packed layout wq[k/4][m][4]) one chunk ahead into
//     registers.  No A staging, no per-chunk barrier: FOUR barriers per tile (K1 as two launches: 8-24).
//   * u = ELU(DW5(H1) + b1) is written over the same buffer (after a barrier) and is the B operand of the
//     second GEMM; it never leaves the CU.  Both stencils run from the accumulators (a lane holds NT
//     consecutive columns; right neighbours by DPP), as in K1's k5 epilogue.
//   * The residual x is re-read from L2 in epilogue 2 (the window was fetched by this CU one tile ago).
//
// Geometry.  Waves = (C/32 row strips) x (NG column groups of 32*NT columns).  Adjacent groups overlap by the
// stencil's 4 columns, so a window holds WD = NG*(32*NT-4)+4 columns and yields TTO = WD-8 outputs (the
// 8-column halo of the two stencils is recomputed: 3-6 %).  Window column c <-> time to0 - 8 + c for x / H1,
// to0 - 4 + c for u / H2, to0 + c for y.  Zero padding: x is read as 0 outside [0,T) (ELU(0) = 0, the 1x1 has no
// bias, so H1 = 0 there) and u is forced to 0 at times < 0 -- the zero pad SConv1d puts in front of the second
// depth-wise conv.
//
// Accumulation order over k, the stencil's tap order and the ELU are K1's, so the result is bit-identical to
// the block run as two K1 launches.
#include <atomic>
#include <string>

#include "wv_dev.h"

namespace wv {

namespace {

__device__ __forceinline__ float rb_dpp_next(float v) {        // lane i <- lane i+1 (wave_shl:1), lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

template <int NT> struct RbVec;
template <> struct RbVec<4> { typedef f32x4 type; };
template <> struct RbVec<2> { typedef f32x2 type; };

constexpr int RB_OOB = 0x7f000000;                              // byte offset beyond any num_records here
#ifndef RB_SCHED_MASK
#define RB_SCHED_MASK 0
#endif

// C channels, NG column groups, NT 32-column tiles per wave (interleaved: tile e = columns NT*j + e), WPS waves per SIMD
template <int C_, int NG_, int NT_, int WPS_>
struct RB {
    static constexpr int C = C_, NG = NG_, NT = NT_, WPS = WPS_;
    static constexpr int WM = C / 32, NWAVES = WM * NG, NTHREADS = 64 * NWAVES;
    static constexpr int GS = 32 * NT - 4;                      // columns a group contributes
    static constexpr int WD = NG * GS + 4;                      // window columns in LDS
    static constexpr int TTO = WD - 8;                          // outputs per tile
    static constexpr int LD = WD;                               // LDS row stride: pieces of a window are contiguous
    static constexpr int NCH = C / 16;
    static constexpr int W4 = WD / 4, P4 = C * W4;              // 16-byte pieces per row / per window
    static constexpr int RT = NTHREADS / W4;                    // rows the x fetch covers per pass
    static constexpr int XPER = C / RT;                         // pieces a thread carries
    static constexpr size_t SMEM = ((size_t)C * LD + 2 * C * 8) * sizeof(float);
    static_assert(C % 32 == 0 && WD % 4 == 0 && (NT == 2 || NT == 4) && C % RT == 0, "geometry");
};

#define RB_BARRIER()                                             \
    do {                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("" ::: "memory");                           \
    } while (0)

// 5-tap stencil from the accumulators: y[e] = bias + sum_i w[i] * H[NT*q + e + i]  (taps ascending, as K1)
template <int NT>
__device__ __forceinline__ void rb_stencil(const f32x16 (&acc)[NT], int r, const f32x4& w0, const f32x4& w1, float (&y)[NT]) {
    constexpr int NSH = 4 / NT;
    float hh[NT + 4], cur[NT];
#pragma unroll
    for (int e = 0; e < NT; ++e) { cur[e] = acc[e][r]; hh[e] = cur[e]; }
#pragma unroll
    for (int s = 1; s <= NSH; ++s) {
#pragma unroll
        for (int e = 0; e < NT; ++e) { cur[e] = rb_dpp_next(cur[e]); hh[s * NT + e] = cur[e]; }
    }
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        float v = fmaf(w0.x, hh[e], w1.y);
        v = fmaf(w0.y, hh[e + 1], v); v = fmaf(w0.z, hh[e + 2], v);
        v = fmaf(w0.w, hh[e + 3], v); v = fmaf(w1.x, hh[e + 4], v);
        y[e] = v;
    }
}

// one 16-deep chunk of MFMAs: A fragments a0 / a1 from registers, B rows 16c .. 16c+15 of the window
template <class R>
__device__ __forceinline__ void rb_chunk(f32x16 (&acc)[R::NT], const f32x4& a0, const f32x4& a1, const float* Bf, int c, int h) {
    typedef typename RbVec<R::NT>::type bvec;
#define WV_RB_STEP(AV, ROW)                                                                        \
    { const bvec bv = *reinterpret_cast<const bvec*>(Bf + (ROW) * R::LD);                          \
      _Pragma("unroll") for (int e = 0; e < R::NT; ++e)                                            \
          acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, bv[e], acc[e], 0, 0, 0); }
    WV_RB_STEP(a0.x, 16 * c + 4 * h + 0) WV_RB_STEP(a0.y, 16 * c + 4 * h + 1)
    WV_RB_STEP(a0.z, 16 * c + 4 * h + 2) WV_RB_STEP(a0.w, 16 * c + 4 * h + 3)
    WV_RB_STEP(a1.x, 16 * c + 8 + 4 * h + 0) WV_RB_STEP(a1.y, 16 * c + 8 + 4 * h + 1)
    WV_RB_STEP(a1.z, 16 * c + 8 + 4 * h + 2) WV_RB_STEP(a1.w, 16 * c + 8 + 4 * h + 3)
#undef WV_RB_STEP
}

// OUT: 1 = Y, 2 = Yact, 3 = both
template <class R, int OUT>
__global__ __launch_bounds__(R::NTHREADS) __attribute__((amdgpu_waves_per_eu(R::WPS, R::WPS))) void rb_kernel(RbArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef typename RbVec<R::NT>::type ovec;
    typedef unsigned uvec __attribute__((ext_vector_type(R::NT)));
    constexpr int NT = R::NT, C = R::C, LD = R::LD, NCH = R::NCH;
    float* S = smem;
    float* tab = smem + C * LD;                                  // [2][C][8]: taps, bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave % R::WM, grp = wave / R::WM;
    const int h = lane >> 5, q = lane & 31;
    const int T = p.T, ntiles = p.ntiles, num_t = p.num_t;

    for (int i = tid; i < C * 8; i += R::NTHREADS) { tab[i] = p.tab1[i]; tab[C * 8 + i] = p.tab2[i]; }

    // ---- A fragments: wq[k/4][Mp][4]; chunk c, lane half h: a0 = wq[4c + h][m], a1 = wq[4c + h + 2][m].  Buffer loads: one
    // per-lane byte offset per matrix, the chunk as a compile-time scalar offset (per-chunk 64-bit addresses, hoisted out
    // of the tile loop by the compiler, cost 8 registers per chunk)
    constexpr int MP = (C + M_ALIGN - 1) / M_ALIGN * M_ALIGN;    // the packed weights' row count (PwWeight::Mp)
    const __amdgpu_buffer_rsrc_t rW1 = uniform_rsrc(p.pw1.wq, NCH * 4 * MP * 16);
    const __amdgpu_buffer_rsrc_t rW2 = uniform_rsrc(p.pw2.wq, NCH * 4 * MP * 16);
    const int avoff = (h * MP + 32 * strip + q) * 16;
    f32x4 ar[2][2];
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rw, int c, f32x4 (&dst)[2]) {
        dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, avoff, c * 4 * MP * 16, 0));
        dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, avoff, (c * 4 + 2) * MP * 16, 0));
    };

    // ---- x pieces of one window: threads form an [RT rows][W4 pieces] grid (the few left over idle here); piece i of a
    // thread is row xr + i * RT at its own column, so its address is one per-lane offset plus a scalar step and the edge
    // test (T % 4 == 0: a piece is all inside [0,T) or all outside) is per thread, not per piece
    const int xr = tid / R::W4, xc = tid - xr * R::W4;
    const bool xthread = tid < R::RT * R::W4;
    f32x4 xp[R::XPER];
    auto xfetch = [&](int tile) {
        const int b = tile / num_t, tt = tile - b * num_t;
        const int t = tt * R::TTO - 8 + 4 * xc;
        const __amdgpu_buffer_rsrc_t rX = uniform_rsrc(p.X + (size_t)b * C * T, C * T * 4);
        const int voff = (xthread && t >= 0 && t < T) ? (xr * T + t) * 4 : RB_OOB;
#pragma unroll
        for (int i = 0; i < R::XPER; ++i)
            xp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, voff, i * R::RT * T * 4, 0));
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    xfetch(tile);
    load_a(rW1, 0, ar[0]);
    RB_BARRIER();                                                // tables visible

    const float* Bf = S + R::GS * grp + NT * q;                  // this lane's B columns
    const float* Wrow1 = tab + (32 * strip + 4 * h) * 8;
    const float* Wrow2 = Wrow1 + C * 8;
    float* Urow = S + (32 * strip + 4 * h) * LD + R::GS * grp + NT * q;
    const bool uw = NT * q < R::GS;                              // lanes that own u / y columns

    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / num_t, tt = tile - b * num_t;
        const int to0 = tt * R::TTO;
        // ================= activation pass: S = ELU(c * x) ==========================================
        if (xthread) {
            f32x4* Sx = reinterpret_cast<f32x4*>(S) + xr * R::W4 + xc;
#pragma unroll
            for (int i = 0; i < R::XPER; ++i) {
                f32x4 v = xp[i];
                v.x = act(v.x, p.pre_scale, 1); v.y = act(v.y, p.pre_scale, 1);
                v.z = act(v.z, p.pre_scale, 1); v.w = act(v.w, p.pre_scale, 1);
                Sx[i * R::RT * R::W4] = v;
            }
        }
        RB_BARRIER();                                            // B1: window complete
        // ================= GEMM 1: H1 = W1 @ S =======================================================
        f32x16 acc[NT];
#pragma unroll
        for (int e = 0; e < NT; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load_a(rW1, c + 1, ar[(c + 1) & 1]);
            else load_a(rW2, 0, ar[(c + 1) & 1]);            // first chunk of W2 lands behind epilogue 1
            rb_chunk<R>(acc, ar[c & 1][0], ar[c & 1][1], Bf, c, h);
            __builtin_amdgcn_sched_barrier(RB_SCHED_MASK);       // the A loads stay one chunk ahead (hoisted, they cost a register set per chunk)
        }
        RB_BARRIER();                                            // B2: every wave has read the window (u overwrites it)
        {   // next window's x: in flight across epilogue 1, GEMM 2 and epilogue 2
            const int next = tile + gridDim.x;
            if (next < ntiles) xfetch(next);
        }
        // ================= epilogue 1: u = ELU(DW5(H1) + b1) -> S ======================================
        {
            f32x4 w0n = *reinterpret_cast<const f32x4*>(Wrow1), w1n = *reinterpret_cast<const f32x4*>(Wrow1 + 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2);
                const f32x4 w0 = w0n, w1 = w1n;
                if (r + 1 < 16) {
                    const int cn = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
                    w0n = *reinterpret_cast<const f32x4*>(Wrow1 + cn * 8);
                    w1n = *reinterpret_cast<const f32x4*>(Wrow1 + cn * 8 + 4);
                }
                float y[NT];
                rb_stencil<NT>(acc, r, w0, w1, y);
                ovec uv;
#pragma unroll
                for (int e = 0; e < NT; ++e) uv[e] = elu1(y[e] * 1.f);
                if (uw) *reinterpret_cast<ovec*>(Urow + cr * LD) = uv;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tt == 0 && grp == 0 && NT * q < 4) {             // u at times < 0 is the second conv's zero padding
                ovec z;
#pragma unroll
                for (int e = 0; e < NT; ++e) z[e] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) *reinterpret_cast<ovec*>(Urow + ((r & 3) + 8 * (r >> 2)) * LD) = z;
            }
        }
        RB_BARRIER();                                            // B3: u complete
        // ================= GEMM 2: H2 = W2 @ u ========================================================
        // epilogue-2 addressing (range-checked buffers over one clip's [C][T] block, as K1's k5 epilogue)
        const int to = to0 + R::GS * grp + NT * q;
        const bool ovalid = uw && R::GS * grp + NT * q < R::TTO && to < T;
        const int voff0 = ovalid ? ((32 * strip + 4 * h) * T + to) * 4 : RB_OOB;
        const int row_bytes = T * 4, clip_bytes = C * T * 4;
        const size_t bo = (size_t)b * C * T;
        const __amdgpu_buffer_rsrc_t rR = uniform_rsrc(p.X + bo, clip_bytes);
        ovec res4[4];
#pragma unroll
        for (int e = 0; e < NT; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load_a(rW2, c + 1, ar[(NCH + c + 1) & 1]);
            else {
                load_a(rW1, 0, ar[(NCH + c + 1) & 1]);       // next tile's first chunk of W1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (NT == 4) res4[r] = __builtin_bit_cast(ovec, __builtin_amdgcn_raw_buffer_load_b128(rR, voff0 + ((r & 3) + 8 * (r >> 2)) * row_bytes, 0, 0));
                    else res4[r] = __builtin_bit_cast(ovec, __builtin_amdgcn_raw_buffer_load_b64(rR, voff0 + ((r & 3) + 8 * (r >> 2)) * row_bytes, 0, 0));
                }
            }
            rb_chunk<R>(acc, ar[(NCH + c) & 1][0], ar[(NCH + c) & 1][1], Bf, c, h);
            __builtin_amdgcn_sched_barrier(RB_SCHED_MASK);
        }
        RB_BARRIER();                                            // B4: every wave has read u (the next window overwrites it)
        // ================= epilogue 2: y = x + s * (DW5(H2) + b2) -> HBM ================================
        {
            const __amdgpu_buffer_rsrc_t rY = uniform_rsrc((OUT & 1) ? p.Y + bo : p.X, (OUT & 1) ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rA = uniform_rsrc((OUT & 2) ? p.Yact + bo : p.X, (OUT & 2) ? clip_bytes : 0);
            f32x4 w0n = *reinterpret_cast<const f32x4*>(Wrow2), w1n = *reinterpret_cast<const f32x4*>(Wrow2 + 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2);
                const f32x4 w0 = w0n, w1 = w1n;
                if (r + 1 < 16) {
                    const int cn = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
                    w0n = *reinterpret_cast<const f32x4*>(Wrow2 + cn * 8);
                    w1n = *reinterpret_cast<const f32x4*>(Wrow2 + cn * 8 + 4);
                }
                float v[NT];
                rb_stencil<NT>(acc, r, w0, w1, v);
                const ovec rr = res4[r & 3];
                if (r + 4 < 16) {
                    const int o4 = voff0 + (((r + 4) & 3) + 8 * ((r + 4) >> 2)) * row_bytes;
                    if constexpr (NT == 4) res4[r & 3] = __builtin_bit_cast(ovec, __builtin_amdgcn_raw_buffer_load_b128(rR, o4, 0, 0));
                    else res4[r & 3] = __builtin_bit_cast(ovec, __builtin_amdgcn_raw_buffer_load_b64(rR, o4, 0, 0));
                }
                ovec y;
#pragma unroll
                for (int e = 0; e < NT; ++e) y[e] = fmaf(v[e], p.out_scale, rr[e]);
                const int off = voff0 + cr * row_bytes;
                if constexpr ((OUT & 1) != 0) {
                    if constexpr (NT == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, y), rY, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, y), rY, off, 0, 0);
                }
                if constexpr ((OUT & 2) != 0) {
                    ovec a;
#pragma unroll
                    for (int e = 0; e < NT; ++e) a[e] = elu1(y[e] * p.act_scale);
                    if constexpr (NT == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, a), rA, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, a), rA, off, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int cu_count() {                                                 // per device, cached
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
hipError_t rb_launch(RbArgs a, hipStream_t s) {
    a.num_t = (a.T + R::TTO - 1) / R::TTO;
    const long long nt = (long long)a.num_t * a.B;
    if (nt > 0x7fffffffLL) return hipErrorInvalidValue;
    a.ntiles = (int)nt;
    static std::atomic<unsigned> attr{0};
    if (R::SMEM > 64 * 1024) {
        int d = 0; (void)hipGetDevice(&d);
        const unsigned bit = 1u << (d & 31);
        if (!(attr.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rb_kernel<R, OUT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::SMEM);
            if (e != hipSuccess) return e;
            attr.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    // persistent grid: as many workgroups as the chip holds at once (LDS- and wave-limited), each walks tiles with that stride
    const int per_cu = std::max(1, std::min((int)(160 * 1024 / R::SMEM), 4 * R::WPS / R::NWAVES));
    const int grid = (int)std::min<long long>(nt, (long long)cu_count() * per_cu);
    std::string name;
    if (prof::enabled()) name = "resblock<" + std::to_string(R::C) + "," + std::to_string(R::WD) + ">";
    const double C = a.C, Bd = a.B, T = a.T;
    prof::Scope ps(s, name.c_str(), 2.0 * 2.0 * Bd * C * (C * T + 5.0 * T),
                   4.0 * Bd * C * T * (1.0 + ((OUT & 1) ? 1.0 : 0.0) + ((OUT & 2) ? 1.0 : 0.0)));
    hipLaunchKernelGGL((rb_kernel<R, OUT>), dim3((unsigned)grid), dim3(R::NTHREADS), R::SMEM, s, a);
    return hipGetLastError();
}

template <class R>
hipError_t rb_pick_out(const RbArgs& a, hipStream_t s) {
    if (a.Y && a.Yact) return rb_launch<R, 3>(a, s);
    if (a.Y) return rb_launch<R, 1>(a, s);
    return rb_launch<R, 2>(a, s);
}

}  // namespace

bool rb_supported(const RbArgs& a) {
    if (!(a.C == 64 || a.C == 96 || a.C == 128 || a.C == 192)) return false;
    if (!a.X || (!a.Y && !a.Yact) || a.T < 4 || (a.T & 3) || a.B < 1 || !a.pw1.wq || !a.pw2.wq || !a.tab1 || !a.tab2) return false;
    if (a.pw1.M != a.C || a.pw1.K != a.C || a.pw2.M != a.C || a.pw2.K != a.C) return false;
    if (a.pw1.Mp != round_up(a.C, M_ALIGN) || a.pw2.Mp != a.pw1.Mp) return false;
    if ((long long)a.C * a.T * 4 >= RB_OOB) return false;       // 32-bit buffer offsets inside one clip
    return aligned16(a.X) && (!a.Y || aligned16(a.Y)) && (!a.Yact || aligned16(a.Yact));
}

hipError_t launch_resblock(const RbArgs& a, hipStream_t s) {
    if (!rb_supported(a)) return hipErrorNotSupported;
    switch (a.C) {
        case 64: return rb_pick_out<RB<64, 2, 4, 2>>(a, s);      // 2 x 2 waves, 252-column windows, two workgroups per CU
        case 96: return rb_pick_out<RB<96, 4, 2, 3>>(a, s);      // 3 x 4 waves, 244-column windows
        case 128: return rb_pick_out<RB<128, 2, 4, 2>>(a, s);    // 4 x 2 waves, 252-column windows
        default: return rb_pick_out<RB<192, 2, 2, 3>>(a, s);     // 6 x 2 waves, 124-column windows
    }
}

}  // namespace wv
