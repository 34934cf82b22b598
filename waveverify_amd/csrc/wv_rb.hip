// Whole SEANetResnetBlock in ONE launch, raw in / raw out, for the narrow layers (C = 64, 96, 128, 192;
// modules/seanet.py:245-281 with dws_conv_block :39-116):
//
//     y = x + s * ( DW5( W2 @ ELU( DW5( W1 @ ELU(c*x) ) + b1 ) ) + b2 )
//
// As two K1 launches such a block crosses HBM five times (read x, write u, read u, read x, write y) and both
// launches are bandwidth- or latency-bound (profiles/r02_chunked_batch_infinity_cache.txt: 49-80 TFLOP/s).  Here
// a PERSISTENT workgroup owns all C channels of one time window after the other, and x is read from HBM ONCE:
//
//   * The next window's x is copied into the window buffer S[C][WD] by LDS-DMA (buffer_load ... lds), all of it issued in
//     one burst when the second GEMM has finished with the buffer and BEFORE epilogue 2's stores, whose VALU work and
//     store issue cover its latency.  (Refill loads interleaved with the stores doubled the time of that epilogue; a
//     residual re-read from global memory went past L2 -- 2.3x the x bytes fetched, profiles/r03_resblock_traffic.txt.)
//   * The activation pass reads x back from S in the MFMA accumulator layout (a lane takes exactly the elements where
//     its own outputs lie: 16 rows of its wave's 32-row strip x NT consecutive columns), keeps them raw in registers --
//     the residual operand of epilogue 2 -- and writes ELU(c*x) in place: the natural [k][t] layout the B operand is
//     read from.  The 8 halo columns in front are activated in place by a few extra 16-byte pieces.
//   * The weights never touch LDS: a wave owns one 32-row strip of W1 / W2 and streams its A fragments (two
//     16-byte buffer loads per 16-deep chunk, L2-resident packed layout wq[k/4][m][4]) into a register ring.
//     No A staging, no per-chunk barrier: FOUR barriers per tile (K1 as two launches: 8-24).
//   * u = ELU(DW5(H1) + b1) is written over the same buffer (after a barrier) and is the B operand of the
//     second GEMM; it never leaves the CU.  Both stencils run from the accumulators (a lane holds NT
//     consecutive columns; the right neighbours' taps are DPP operands of the multiply-adds themselves).
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

template <int NT> struct RbVec;
template <> struct RbVec<4> { typedef f32x4 type; };
template <> struct RbVec<2> { typedef f32x2 type; };

constexpr int RB_OOB = 0x7f000000;                              // byte offset beyond any num_records here
// geometry per channel count: <C, column groups, 32-column tiles per wave, waves per SIMD, parts the window refill is issued in>
// (measured, tools/rbbench.py, one box, interleaved: the refill started inside GEMM 2 in 3-4 parts gains 4 % at C = 96 and 1-2 % at
//  C = 128, nothing at C = 64, and costs 8 % at C = 192 in four parts where two are even)
#ifndef RB_CFG64
#define RB_CFG64 RB<64, 2, 4, 2, 1>                             // 2 x 2 waves, 252-column windows, two workgroups per CU
#endif
#ifndef RB_CFG96
#define RB_CFG96 RB<96, 4, 2, 3, 3>                             // 3 x 4 waves, 244-column windows
#endif
#ifndef RB_CFG128
#define RB_CFG128 RB<128, 2, 4, 2, 4>                           // 4 x 2 waves, 252-column windows
#endif
#ifndef RB_CFG192
#define RB_CFG192 RB<192, 2, 2, 3, 2>                           // 6 x 2 waves, 124-column windows
#endif
#ifndef RB_PD
#define RB_PD 3                                                 // B rows in flight ahead of their MFMAs
#endif
#ifndef RB_AD
#define RB_AD 3                                                 // A chunks in flight ahead of their MFMAs (ring of RB_AD + 1; 1 or 3: the refill issued inside GEMM 2 sits in the
                                                                // same in-order queue, so the A loads behind it must not be needed soon)
#endif

// Diagnostic builds only (tools/variant.sh ... -DRB_STAMP): per-phase shader-clock totals of every wave, written over the head of
// the output tensor at the end (tools/rbbench.py --stamps reads them back; the output is garbage then).
#ifdef RB_STAMP
#define RB_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define RB_T(i) do {} while (0)
#endif

// C channels, NG column groups, NT 32-column tiles per wave (interleaved: tile e = columns NT*j + e), WPS waves per SIMD;
// PARTS: the next window's refill is issued in this many parts, all but the last inside GEMM 2 (1: all of it after GEMM 2)
template <int C_, int NG_, int NT_, int WPS_, int PARTS_>
struct RB {
    static constexpr int C = C_, NG = NG_, NT = NT_, WPS = WPS_, PARTS = PARTS_;
    static constexpr int WM = C / 32, NWAVES = WM * NG, NTHREADS = 64 * NWAVES;
    static constexpr int GS = 32 * NT - 4;                      // columns a group contributes
    static constexpr int WD = NG * GS + 4;                      // window columns in LDS
    static constexpr int TTO = WD - 8;                          // outputs per tile
    static constexpr int LD = WD;                               // LDS row stride
    static constexpr int NCH = C / 16;
    static constexpr int W4 = WD / 4, P4 = C * W4;              // 16-byte pieces per row / per window
    static constexpr int RPI = 64 / W4;                         // whole window rows one LDS-DMA instruction copies
    static constexpr int NI = (C / RPI + NWAVES - 1) / NWAVES;  // LDS-DMA instructions per wave and window
    static constexpr size_t SMEM = ((size_t)C * LD + 2 * C * 8) * sizeof(float);
    static_assert(C % 32 == 0 && WD % 4 == 0 && (NT == 2 || NT == 4) && NTHREADS >= 2 * C && RPI >= 1 && C % RPI == 0 && 16 % RPI == 0 &&
                  PARTS >= 1 && NCH % PARTS == 0, "geometry");
    static_assert((2 * NCH) % (RB_AD + 1) == 0, "the A ring must close over a tile");
};

#define RB_BARRIER()                                             \
    do {                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                            \
        asm volatile("" ::: "memory");                           \
    } while (0)

__device__ __forceinline__ float rb_dpp_next(float v) {        // lane i <- lane i+1 (wave_shl:1), lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// 5-tap stencil from the accumulators: y[e] = bias + sum_i w[i] * H[NT*q + e + i] (taps ascending, as K1), for a PAIR of rows
// (accumulator registers r, r + 1 = two adjacent channels) as packed-f32 multiply-adds: one
// v_pk_fma_f32 does both rows, the neighbour lanes' columns come as shifted copies (compiler-visible DPP moves).  Taps in the same
// ascending order, every multiply-add fused: the values of round 3's per-row stencil (whose neighbour taps were DPP operands of inline-asm
// v_fmac_f32_dpp) bit for bit, 3.5 (NT = 4) / 4.5 (NT = 2) vector instructions per output instead of 5 / 6 -- measured even on time (+-1 %,
// profiles/r04_rb_packed_stencil.txt) -- and no inline-asm cross-lane read is left for the compiler's hazard recogniser to miss.  tp: the pair's table row {w0a, w0b, w1a, w1b | w2a, w2b,
// w3a, w3b | w4a, w4b, ba, bb} (built from the [C][8] table when the kernel copies it to LDS).
__device__ __forceinline__ f32x2 rb_pair_next(f32x2 v) { return f32x2{rb_dpp_next(v.x), rb_dpp_next(v.y)}; }
template <int NT>
__device__ __forceinline__ void rb_stencil2(const f32x16 (&acc)[NT], int r, const float* tp, f32x2 (&y)[NT]) {
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(tp), q1 = *reinterpret_cast<const f32x4*>(tp + 4), q2 = *reinterpret_cast<const f32x4*>(tp + 8);
    const f32x2 w[5] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}, {q2.x, q2.y}};
    const f32x2 bb{q2.z, q2.w};
    f32x2 col[NT + 4];                                           // the lane's NT columns, then the 4 that follow them
#pragma unroll
    for (int e = 0; e < NT; ++e) col[e] = f32x2{acc[e][r], acc[e][r + 1]};
    if constexpr (NT == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) col[4 + e] = rb_pair_next(col[e]);
    } else {
        col[2] = rb_pair_next(col[0]); col[3] = rb_pair_next(col[1]);
        col[4] = rb_pair_next(col[2]); col[5] = rb_pair_next(col[3]);
    }
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        f32x2 v = bb;
#pragma unroll
        for (int i = 0; i < 5; ++i) v = __builtin_elementwise_fma(w[i], col[e + i], v);
        y[e] = v;
    }
}

// One GEMM of the block: acc = W @ S over all C rows of the window.  A fragments: global chunk g in ar[g % NA], loaded AD chunks
// ahead at the head of a chunk; the chunks after the last wrap into the OTHER matrix (rw_next) for the GEMM that follows.  B rows: one
// ds_read per 2-deep step (a lane's NT columns of row k; lane half h owns k in [4h,4h+4) U [8+4h,12+4h) of every 16), issued PD steps
// ahead of the MFMAs that consume them into a ring of PD + 1 row vectors.  The order is pinned (scheduling barrier per step): left to
// itself the compiler issues each read right in front of its MFMAs and sinks the A loads to the end of the chunk, so that both
// latencies are exposed once per step / chunk.
// at_chunk(c) runs at the head of chunk c >= 1 (GEMM 2 frees the window's rows there, in quarters, for the next window's x).
template <class R, int G0, class LoadA, class AtChunk>
__device__ __forceinline__ void rb_gemm(f32x16 (&acc)[R::NT], f32x4 (&ar)[RB_AD + 1][2], const float* Bf, LoadA&& load_a,
                                        const __amdgpu_buffer_rsrc_t& rw, const __amdgpu_buffer_rsrc_t& rw_next, AtChunk&& at_chunk) {
    typedef typename RbVec<R::NT>::type bvec;
    constexpr int NS = 8 * R::NCH, PD = RB_PD, NB = PD + 1, AD = RB_AD, NA = AD + 1;
    bvec bq[NB];
    auto row_of = [](int s) { return 16 * (s >> 3) + ((s & 7) < 4 ? (s & 7) : 4 + (s & 7)); };     // + 4h, folded into Bf
#pragma unroll
    for (int s = 0; s < PD; ++s) bq[s % NB] = *reinterpret_cast<const bvec*>(Bf + row_of(s) * R::LD);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = s >> 3, j = s & 7;
        if (j == 0) {
            if (c > 0) at_chunk(c);
            if (c + AD < R::NCH) load_a(rw, c + AD, ar[(G0 + c + AD) % NA]);
            else load_a(rw_next, c + AD - R::NCH, ar[(G0 + c + AD) % NA]);
        }
        if (s + PD < NS) bq[(s + PD) % NB] = *reinterpret_cast<const bvec*>(Bf + row_of(s + PD) * R::LD);
        const f32x4 av = ar[(G0 + c) % NA][j >> 2];
        const float a = av[j & 3];
        const bvec bv = bq[s % NB];
#pragma unroll
        for (int e = 0; e < R::NT; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[e], acc[e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// OUT: 1 = Y, 2 = Yact, 3 = both; 5 = Y and the four tensors a training step keeps for the backward (RbArgs::sv_*)
template <class R, int OUT>
__global__ __launch_bounds__(R::NTHREADS) __attribute__((amdgpu_waves_per_eu(R::WPS, R::WPS))) void rb_kernel(RbArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef typename RbVec<R::NT>::type ovec;
    typedef unsigned uvec __attribute__((ext_vector_type(R::NT)));
    constexpr int NT = R::NT, C = R::C, LD = R::LD, NCH = R::NCH, AD = RB_AD, NA = AD + 1;
    float* S = smem;
    float* tab = smem + C * LD;                                  // [2][C][8]: taps, bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave % R::WM, grp = wave / R::WM;
    const int h = lane >> 5, q = lane & 31;
    const int T = p.T, ntiles = p.ntiles, num_t = p.num_t;
    const int row_bytes = T * 4, clip_bytes = C * T * 4;

    // the stencil tables [C][8] (taps, bias, 1, 0) into LDS in channel-PAIR layout [C / 2][12] (rb_stencil2), the two tables C * 8 floats apart
    for (int i = tid; i < C * 8; i += R::NTHREADS) {
        const int m = i >> 3, j = i & 7;
        if (j < 6) {
            const int d = (m >> 1) * 12 + 2 * (j < 5 ? j : 5) + (m & 1);
            tab[d] = p.tab1[i]; tab[C * 8 + d] = p.tab2[i];
        }
    }

    // ---- A fragments: wq[k/4][Mp][4]; chunk c, lane half h: a0 = wq[4c + h][m], a1 = wq[4c + h + 2][m].  Buffer loads: one
    // per-lane byte offset, the chunk as a compile-time scalar offset (per-chunk 64-bit addresses, hoisted out of the tile loop by
    // the compiler, cost 8 registers per chunk)
    constexpr int MP = (C + M_ALIGN - 1) / M_ALIGN * M_ALIGN;    // the packed weights' row count (PwWeight::Mp)
    const __amdgpu_buffer_rsrc_t rW1 = uniform_rsrc(p.pw1.wq, NCH * 4 * MP * 16);
    const __amdgpu_buffer_rsrc_t rW2 = uniform_rsrc(p.pw2.wq, NCH * 4 * MP * 16);
    const int avoff = (h * MP + 32 * strip + q) * 16;
    f32x4 ar[NA][2];
    auto load_a = [&](const __amdgpu_buffer_rsrc_t& rw, int c, f32x4 (&dst)[2]) {
        dst[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, avoff, c * 4 * MP * 16, 0));
        dst[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, avoff, (c * 4 + 2) * MP * 16, 0));
    };

    // ---- this lane's place in a window: rows 32*strip + 4h + (r&3) + 8(r>>2), r = 0..15; x / S columns co + 8 .. (its outputs are
    // columns co .. co + NT - 1 of the tile), B / u columns co ..
    const int co = R::GS * grp + NT * q;                         // first output (= u, H) column of this lane
    const bool own = NT * q < R::GS && co < R::TTO;              // lanes that own outputs (and x columns inside the window)
    const bool uw = NT * q < R::GS;                              // lanes that own u columns
    const int row0 = 32 * strip + 4 * h;
    const float* Bf = S + co + 4 * h * LD;                       // B rows of this lane's half, its columns
    const float* Wrow1 = tab + (row0 / 2) * 12;                  // pair table row of channel row0 (a multiple of 4)
    const float* Wrow2 = Wrow1 + C * 8;
    float* Urow = S + row0 * LD + co;
    float* Xrow = Urow + 8;
    // halo: the 8 columns in front of the outputs, 2 pieces per row, threads 0 .. 2C-1 activate them
    const bool hthread = tid < 2 * C;
    const int hrow = tid >> 1, hpc = tid & 1;

    ovec X[16];                                                  // raw x where this lane's outputs lie: the residual operand
    // ---- window refill by LDS-DMA.  One instruction copies a block of RPI whole rows of the window (RPI * W4 <= 64 sixteen-byte pieces:
    // lane l takes piece l of the block, the few lanes past it are switched off); the LDS image is the window itself (LD = WD), so a
    // lane's destination is its piece's place and its source is one per-lane offset plus the block's rows as a SCALAR offset.  Block rb
    // belongs to wave rb % NWAVES.  The first and last windows of a clip test the lane's column against [0,T) (T % 4 == 0: a piece is
    // all inside or all outside; outside reads an out-of-range offset = zeros, the causal padding).
    const int xrow = lane / R::W4, xc4 = lane - xrow * R::W4;    // this lane's row inside a block, its piece inside the row
    const bool xlane = lane < R::RPI * R::W4;
    auto refill = [&](int t, int row0_, int row1_) {             // rows [row0_, row1_) of tile t's window -> S (t < 0: none)
        if (t < 0) return;
        const int b = t / num_t, tt = t - b * num_t;
        const int tw0 = tt * R::TTO - 8;
        const __amdgpu_buffer_rsrc_t rX = uniform_rsrc(p.X + (size_t)b * C * T, clip_bytes);
        const int tx = tw0 + 4 * xc4;
        const int vo = (tx >= 0 && tx < T) ? (xrow * T + tx) * 4 : RB_OOB;
        if (xlane) {
#pragma unroll
            for (int i = 0; i < R::NI; ++i) {
                const int rb = i * R::NWAVES + wave;              // this wave's i-th block (wave-uniform test: a scalar branch)
                const int r_ = rb * R::RPI, so = r_ * row_bytes;  // (plain locals: dependent expressions as builtin arguments make hipcc's
                float* dst = S + r_ * LD;                         //  host pass drop the kernel's stub)
                if (r_ >= row0_ && r_ < row1_)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, 0);
            }
        }
    };
    auto x_off = [&](int to0) { const int t = to0 + co; return (own && t < T) ? (row0 * T + t) * 4 : RB_OOB; };
    // training forward: which of this lane's columns of the GEMM outputs / of u belong to this tile.  Accumulator column c of GEMM 1 is
    // time to0 - 8 + c: this tile's share is c in [8, WD), each column owned by one group (the last group also keeps its 4 overlap
    // columns); u and GEMM 2's output at column c are time to0 - 4 + c: c in [4, WD - 4), the lanes that write u.
    const bool h0own = (OUT & 4) && co >= 8 && co < R::WD && (NT * q < R::GS || (grp == R::NG - 1 && NT * q < R::GS + 4));
    const bool u_own = (OUT & 4) && uw && co >= 4 && co < R::WD - 4;
    auto sv_off = [&](bool mine, int t) { return (mine && t < T) ? (row0 * T + t) * 4 : RB_OOB; };
    const float oscale = p.out_scale_ptr ? p.out_scale * p.out_scale_ptr[0] : p.out_scale;

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    refill(tile, 0, C);
#pragma unroll
    for (int c = 0; c < AD; ++c) load_a(rW1, c, ar[c % NA]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RB_BARRIER();                                                // tables visible, first window landed

#ifdef RB_STAMP
    unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / num_t, tt = tile - b * num_t;
        const int to0 = tt * R::TTO;
        // ================= activation pass: X = x (raw), S = ELU(c * x) in place ======================
        if (hthread) {
            f32x4* hp = reinterpret_cast<f32x4*>(S + hrow * LD + 4 * hpc);
            f32x4 v = *hp;
            v.x = elu1(v.x * p.pre_scale); v.y = elu1(v.y * p.pre_scale); v.z = elu1(v.z * p.pre_scale); v.w = elu1(v.w * p.pre_scale);
            *hp = v;
        }
        if (own) {
#pragma unroll
            for (int r = 0; r < 16; ++r) X[r] = *reinterpret_cast<const ovec*>(Xrow + ((r & 3) + 8 * (r >> 2)) * LD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ovec v;
#pragma unroll
                for (int e = 0; e < NT; ++e) v[e] = elu1(X[r][e] * p.pre_scale);
                *reinterpret_cast<ovec*>(Xrow + ((r & 3) + 8 * (r >> 2)) * LD) = v;
            }
        }
        RB_T(0);
        RB_BARRIER();                                            // B1: window complete
        RB_T(1);
        // ================= GEMM 1: H1 = W1 @ S =======================================================
        f32x16 acc[NT];
#pragma unroll
        for (int e = 0; e < NT; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
        rb_gemm<R, 0>(acc, ar, Bf, load_a, rW1, rW2, [](int) {});   // the first chunk(s) of W2 land behind epilogue 1
        RB_T(2);
        RB_BARRIER();                                            // B2: every wave has read the window (u overwrites it)
        RB_T(3);
        // ================= epilogue 1: u = ELU(DW5(H1) + b1) -> S ======================================
        {
            const size_t bo1 = (size_t)b * C * T;
            const __amdgpu_buffer_rsrc_t rH0 = uniform_rsrc((OUT & 4) ? p.sv_h0 + bo1 : p.X, (OUT & 4) ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rU = uniform_rsrc((OUT & 4) ? p.sv_u + bo1 : p.X, (OUT & 4) ? clip_bytes : 0);
            const int vh0 = sv_off(h0own, to0 - 8 + co), vu = sv_off(u_own, to0 - 4 + co);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                    // rows r, r + 1: channels cr, cr + 1
                const int cr = (r & 3) + 8 * (r >> 2);
                f32x2 y2[NT];
                rb_stencil2<NT>(acc, r, Wrow1 + (cr / 2) * 12, y2);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    ovec uv;
#pragma unroll
                    for (int e = 0; e < NT; ++e) uv[e] = elu1(y2[e][k] * 1.f);
                    if (uw) *reinterpret_cast<ovec*>(Urow + (cr + k) * LD) = uv;
                    if constexpr ((OUT & 4) != 0) {
                        ovec hv, ur;                             // the first half's output BEFORE the second half's ELU: what its backward differentiates
#pragma unroll
                        for (int e = 0; e < NT; ++e) { hv[e] = acc[e][r + k]; ur[e] = y2[e][k]; }
                        const int o0 = vh0 == RB_OOB ? RB_OOB : vh0 + (cr + k) * row_bytes, o1 = vu == RB_OOB ? RB_OOB : vu + (cr + k) * row_bytes;
                        if constexpr (NT == 4) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, hv), rH0, o0, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, ur), rU, o1, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, hv), rH0, o0, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, ur), rU, o1, 0, 0);
                        }
                    }
                }
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
        RB_T(4);
        RB_BARRIER();                                            // B3: u complete
        RB_T(5);
        // ================= GEMM 2: H2 = W2 @ u ========================================================
#pragma unroll
        for (int e = 0; e < NT; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
        // The next window's x goes into S while this GEMM still runs: once every wave has passed chunk c (a barrier at each quarter), the
        // rows below 16c of u are dead, and the DMA instructions that lie wholly inside them are issued -- under matrix work, ahead of
        // epilogue 2's stores (loads and stores share one in-order queue: refill loads issued next to the stores doubled that epilogue).
        const int next = tile + gridDim.x < ntiles ? tile + gridDim.x : -1;
        rb_gemm<R, NCH>(acc, ar, Bf, load_a, rW2, rW1, [&](int c) {   // ... and the next tile's first chunk(s) of W1 behind epilogue 2
            if (R::PARTS > 1 && c % (NCH / R::PARTS) == 0) {
                constexpr int per = NCH / R::PARTS;
                RB_BARRIER();
                refill(next, 16 * (c - per), 16 * c);
            }
        });
        RB_T(6);
        RB_BARRIER();                                            // B4: every wave has read u (the next window overwrites it)
        RB_T(7);
        // ================= epilogue 2: y = x + s * (DW5(H2) + b2) -> HBM; refill x ======================
        {
            // range-checked buffers over one clip's [C][T] block (as K1's k5 epilogue): no mask, no branch, 32-bit offsets
            const size_t bo = (size_t)b * C * T;
            const __amdgpu_buffer_rsrc_t rY = uniform_rsrc((OUT & 1) ? p.Y + bo : p.X, (OUT & 1) ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rA = uniform_rsrc((OUT & 2) ? p.Yact + bo : p.X, (OUT & 2) ? clip_bytes : 0);
            const int voff0 = x_off(to0);
            const __amdgpu_buffer_rsrc_t rH1 = uniform_rsrc((OUT & 4) ? p.sv_h1 + bo : p.X, (OUT & 4) ? clip_bytes : 0);
            const __amdgpu_buffer_rsrc_t rV = uniform_rsrc((OUT & 4) ? p.sv_v + bo : p.X, (OUT & 4) ? clip_bytes : 0);
            const int vh1 = sv_off(u_own, to0 - 4 + co);
            // what is left of the next window (the rows the last quarter of GEMM 2 still read), ahead of the stores below
            refill(next, R::PARTS > 1 ? 16 * (NCH - NCH / R::PARTS) : 0, C);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int cr0 = (r & 3) + 8 * (r >> 2);
                f32x2 v2[NT];
                rb_stencil2<NT>(acc, r, Wrow2 + (cr0 / 2) * 12, v2);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int cr = cr0 + k;
                    ovec y;
#pragma unroll
                    for (int e = 0; e < NT; ++e) y[e] = fmaf(v2[e][k], oscale, X[r + k][e]);
                    const int off = voff0 + cr * row_bytes;
                    if constexpr ((OUT & 4) != 0) {
                        ovec hv, vv;
#pragma unroll
                        for (int e = 0; e < NT; ++e) { hv[e] = acc[e][r + k]; vv[e] = v2[e][k]; }
                        const int o1 = vh1 == RB_OOB ? RB_OOB : vh1 + cr * row_bytes;
                        if constexpr (NT == 4) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, hv), rH1, o1, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, vv), rV, off, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, hv), rH1, o1, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, vv), rV, off, 0, 0);
                        }
                    }
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
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        RB_T(8);
        // the refill is older than this epilogue's stores: wait until only those are outstanding, then meet the other waves
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(16 * ((OUT & 1) + ((OUT >> 1) & 1) + ((OUT & 4) ? 2 : 0))) : "memory");
        RB_BARRIER();                                            // B0: the next window has landed
    }
#ifdef RB_STAMP
    if (lane == 0 && (OUT & 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = 0; i < 9; ++i) p.Y[((size_t)blockIdx.x * R::NWAVES + wave) * 9 + i] = (float)ph[i];
    }
#endif
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
    if (prof::enabled()) name = std::string((OUT & 4) ? "resblock_train<" : "resblock<") + std::to_string(R::C) + "," + std::to_string(R::WD) + ">";
    const double C = a.C, Bd = a.B, T = a.T;
    prof::Scope ps(s, name.c_str(), 2.0 * 2.0 * Bd * C * (C * T + 5.0 * T),
                   4.0 * Bd * C * T * (1.0 + ((OUT & 1) ? 1.0 : 0.0) + ((OUT & 2) ? 1.0 : 0.0) + ((OUT & 4) ? 4.0 : 0.0)));
    hipLaunchKernelGGL((rb_kernel<R, OUT>), dim3((unsigned)grid), dim3(R::NTHREADS), R::SMEM, s, a);
    return hipGetLastError();
}

template <class R>
hipError_t rb_pick_out(const RbArgs& a, hipStream_t s) {
    if (a.sv_h0) return rb_launch<R, 5>(a, s);
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
    if (a.sv_h0 || a.sv_u || a.sv_h1 || a.sv_v) {               // the training form: all four, with Y alone
        if (!a.sv_h0 || !a.sv_u || !a.sv_h1 || !a.sv_v || !a.Y || a.Yact) return false;
        if (!aligned16(a.sv_h0) || !aligned16(a.sv_u) || !aligned16(a.sv_h1) || !aligned16(a.sv_v)) return false;
    }
    return aligned16(a.X) && (!a.Y || aligned16(a.Y)) && (!a.Yact || aligned16(a.Yact));
}

hipError_t launch_resblock(const RbArgs& a, hipStream_t s) {
    if (!rb_supported(a)) return hipErrorNotSupported;
    switch (a.C) {
        case 64: return rb_pick_out<RB_CFG64>(a, s);
        case 96: return rb_pick_out<RB_CFG96>(a, s);
        case 128: return rb_pick_out<RB_CFG128>(a, s);
        default: return rb_pick_out<RB_CFG192>(a, s);
    }
}

}  // namespace wv
