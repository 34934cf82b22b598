// gfx950 (MI355X / CDNA4) kernels of the WaveVerify embed/detect hot path.
//
// Everything dense (1x1 convs, DFT basis, detector head) runs through ONE tiled GEMM core on
// the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain, so results
// are float32-faithful to the reference's fp32 convs); the stencils that surround each GEMM
// in the reference (scale -> ELU prologue, causal depth-wise conv / conv-transpose, FiLM,
// residual, L2-norm, log-magnitude) are fused into the GEMM's operand loader or its epilogue so
// every activation crosses HBM once per fused unit.
//
// Layout: activations [B, C, T] float32, time innermost.  GEMM roles: A = weights W^T packed
// [Kp][Mp] (k-major, so an MFMA A-fragment is 32 consecutive floats), B = activations
// [K][time] (k-major too), D[m][t].  64-lane wavefronts, 4 waves per workgroup.
#include "wv_kernels.h"
#include "wv_dev.h"

namespace wv {


// ------------------------------------------------------------------------------------------
// GEMM core.  Tile BM x BN per workgroup, WM x WN waves, each wave MT x NT MFMA tiles of 32x32.
// Per K-chunk (BK rows): operands are fetched RAW into registers one chunk ahead (16-byte loads
// where the tile is aligned), the MFMAs of the current chunk run while those loads are in
// flight, and only then are the raw values transformed (scale -> ELU, stencil taps, ...) and
// committed to the other LDS stage -- so no s_waitcnt on global memory sits in front of the
// matrix work.  One barrier per chunk.  Fragments are ds_read_b32: lane l reads row k+(l>>5),
// column base+(l&31): 32 consecutive floats per half-wave, conflict-free.
// ------------------------------------------------------------------------------------------
template <int BM_, int BN_, int WM_, int WN_>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NTHREADS = 64 * WM * WN;
    static constexpr int MT = BM / (32 * WM);
    static constexpr int NT = BN / (32 * WN);
    static constexpr int A_VEC = BK * BM / 4;                 // float4 per stage
    static constexpr int A_PER = (A_VEC + NTHREADS - 1) / NTHREADS;
    static constexpr int B_TPR = BN / 4;                      // threads per k-row (4 columns each)
    static constexpr int BKSTEP = NTHREADS / B_TPR;           // k rows per pass
    static constexpr int B_PER = (BK + BKSTEP - 1) / BKSTEP;
    static constexpr int STAGE = BK * (BM + BN);              // floats per LDS stage
    static_assert(MT >= 1 && NT >= 1, "tile too small for the wave grid");
    static_assert(BM % 4 == 0 && NTHREADS % B_TPR == 0 && BKSTEP >= 1, "staging map");
};

template <class T>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[T::MT][T::NT]) {
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// LA: float4 load4(int k, int m)                      m multiple of 4, in [0,BM)
// LB: NRAW; init(col4); fetch(k, raw[4*NRAW]); finish(k, raw, out[4])   4 consecutive columns
struct NoSide { __device__ __forceinline__ void operator()(int, const float*) const {} };

// SD: optional per-chunk hook side(c, Bs) with the staged B chunk Bs[BK][BN] (valid until the
// chunk's barrier); used by the STFT for its two vector-side rows.
template <class T, class LA, class LB, class SD = NoSide>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[T::MT][T::NT], const LA& la, LB& lb,
                                              int nchunks, float* smem, SD&& side = SD()) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int bcol = (tid % T::B_TPR) * 4, bk0 = tid / T::B_TPR;
    float4 ra[T::A_PER];
    float rb[T::B_PER][4 * LB::NRAW];
    lb.init(bcol);

    auto fetch = [&](int c) {
#pragma unroll
        for (int r = 0; r < T::A_PER; ++r) {
            const int e4 = tid + r * T::NTHREADS;
            if (T::A_VEC % T::NTHREADS == 0 || e4 < T::A_VEC)
                ra[r] = la.load4(c * BK + e4 / (T::BM / 4), (e4 % (T::BM / 4)) * 4);
        }
#pragma unroll
        for (int r = 0; r < T::B_PER; ++r)
            if (BK % T::BKSTEP == 0 || bk0 + r * T::BKSTEP < BK)
                lb.fetch(c * BK + bk0 + r * T::BKSTEP, rb[r]);
    };
    auto commit = [&](int c, float* buf) {
        float* As = buf;
        float* Bs = buf + BK * T::BM;
#pragma unroll
        for (int r = 0; r < T::A_PER; ++r) {
            const int e4 = tid + r * T::NTHREADS;
            if (T::A_VEC % T::NTHREADS == 0 || e4 < T::A_VEC) *reinterpret_cast<float4*>(As + e4 * 4) = ra[r];
        }
#pragma unroll
        for (int r = 0; r < T::B_PER; ++r) {
            if (!(BK % T::BKSTEP == 0 || bk0 + r * T::BKSTEP < BK)) continue;
            float o[4];
            lb.finish(c * BK + bk0 + r * T::BKSTEP, rb[r], o);
            *reinterpret_cast<float4*>(Bs + (bk0 + r * T::BKSTEP) * T::BN + bcol) =
                make_float4(o[0], o[1], o[2], o[3]);
        }
    };

    fetch(0);
    commit(0, smem);
    __syncthreads();
    const int arow = (lane >> 5), acol = wm * T::MT * 32 + (lane & 31);
    const int bcol_f = wn * T::NT * 32 + (lane & 31);
    for (int c = 0; c < nchunks; ++c) {
        const float* As = smem + (c & 1) * T::STAGE;
        const float* Bs = As + BK * T::BM;
        if (c + 1 < nchunks) fetch(c + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[T::MT], b[T::NT];
#pragma unroll
            for (int i = 0; i < T::MT; ++i) a[i] = As[(kk + arow) * T::BM + acol + i * 32];
#pragma unroll
            for (int j = 0; j < T::NT; ++j) b[j] = Bs[(kk + arow) * T::BN + bcol_f + j * 32];
#pragma unroll
            for (int i = 0; i < T::MT; ++i)
#pragma unroll
                for (int j = 0; j < T::NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) commit(c + 1, smem + ((c + 1) & 1) * T::STAGE);
        side(c, Bs);
        __syncthreads();
    }
}

// Visit every accumulator element with its (row, col) inside the workgroup tile.
// C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <class T, class F>
__device__ __forceinline__ void for_each_acc(const f32x16 (&acc)[T::MT][T::NT], F&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * T::MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * T::NT * 32 + j * 32 + (lane & 31);
                f(row, col, acc[i][j][r]);
            }
}

struct WLoader {                       // packed Wt[Kp][Mp], rows 16-byte aligned
    const float* wt; int Mp, m0;
    __device__ __forceinline__ float4 load4(int k, int m) const {
        return *reinterpret_cast<const float4*>(wt + (size_t)k * Mp + m0 + m);
    }
};

// B operand = rows of a [K][ld] matrix, columns c0+col .. (zero outside [0,ncols) and k >= K),
// optional scale -> ELU applied at commit time.  16-byte loads when the tile is aligned.
struct RowLoader {
    static constexpr int NRAW = 1;
    const float* base; int K, ld, ncols, c0; float scale; int elu;
    const float* p; int c; bool full, vec;
    __device__ __forceinline__ void init(int col4) {
        c = c0 + col4;
        full = c >= 0 && c + 3 < ncols;
        vec = full && ((ld & 3) == 0) && ((c & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
        p = base + c;
    }
    __device__ __forceinline__ void fetch(int k, float (&raw)[4]) const {
        if (k < K && vec) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * ld);
            raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                raw[i] = (k < K && c + i >= 0 && c + i < ncols) ? p[(size_t)k * ld + i] : 0.f;
        }
    }
    __device__ __forceinline__ void finish(int, const float (&raw)[4], float (&o)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act(raw[i], scale, elu);      // act(0) == 0
    }
};

// ------------------------------------------------------------------------------------------
// K1  pw_dw:  Y = epi( DWconv(W @ producer(X)) + b )
// One workgroup: (m-tile, time-tile, clip).  The GEMM produces H[BM][BN] for the input-time
// window the output tile needs (halo = (ks-1)*d - (s-1) on the left, recomputed per tile; the
// window start is 4-aligned so X rows are read with 16-byte loads).  Row-strip tiles: every wave
// owns 32 output channels x the whole window, so the depth-wise stencil never leaves the wave:
// accumulator rows go through wave-private LDS strips (no workgroup barrier) and come back with
// time on the lanes for the stencil + FiLM / residual epilogue (PwDwEpi).
// Zero padding: X is staged as 0 outside [0,Tin) and the 1x1 has no bias, so H is 0 there,
// which is exactly the zero pad SConv1d inserts between the 1x1 and the DW conv.
// producer(X) is act(s*X) (ResnetBlock halves, downsample, SpecBlock add) or the depth-wise
// ConvTranspose of act(s*X) (upsample unit, identity stencil): see pw_dw_kernel.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// f32 GEMM core for the row-strip tiles, k-inner LDS layout.  LDS holds float4 fragments
// [kq][row] = 4 consecutive k of one row/column, so a lane fetches the operands of FOUR MFMA
// steps with one ds_read_b128 (the generic core needs four ds_read_b32): 10 LDS reads per
// 16-deep chunk and wave instead of 40.  The k index is only a summation index, so the two
// lane halves may take any disjoint k sets as long as A and B agree: half h owns fragments
// kq = h and kq = h + 2, i.e. k in [4h, 4h+4) U [8+4h, 12+4h).
// A fragments are pre-packed on the host (wq) and never touch LDS: a wave's 32 weight rows are
// private to it, so every lane loads its two fragments per chunk straight from global memory.
// Only the B operand, which all waves share, is staged (see RowPairLoader / ConvTrPair).
// ------------------------------------------------------------------------------------------
template <class T>
struct QT {
    static constexpr int KQ = BK / 4;                         // fragments along k per chunk
    static constexpr int NB = KQ * T::BN;                     // B fragments (f32x4) per LDS stage
    static constexpr int CG = T::BN / 4;                      // column groups of 4
    static constexpr int NBT = (BK / 2) * CG;                 // B micro-tiles: 2 k rows x 4 columns
    static constexpr int B_PER = (NBT + T::NTHREADS - 1) / T::NTHREADS;
    static_assert(T::NTHREADS % CG == 0, "a thread keeps its column group");
};

template <class T, class LB>
__device__ __forceinline__ void gemm_mainloop_q(f32x16 (&acc)[1][T::NT], const f32x4* __restrict__ wq,
                                                int Mp, int m0, LB& lb, int nchunks, f32x4* smem) {
    using Q = QT<T>;
    static_assert(T::WN == 1 && T::MT == 1, "row-strip tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid % Q::CG;
    const int h = lane >> 5, i31 = lane & 31;
    // A fragments: a wave's 32 weight rows are private to it, so they skip LDS altogether -- each
    // lane loads the two fragments it multiplies with straight from the packed weights (L2-resident),
    // one chunk ahead.  Only the B operand (shared by all waves) is staged.
    const f32x4* wa = wq + m0 + 32 * wave + i31;
    f32x4 an0, an1;
    float rb[Q::B_PER][LB::NRAW];
    lb.init(cg);

    auto fetch = [&](int c) {
        an0 = wa[(size_t)(c * Q::KQ + h) * Mp];
        an1 = wa[(size_t)(c * Q::KQ + h + 2) * Mp];
#pragma unroll
        for (int r = 0; r < Q::B_PER; ++r) {
            const int idx = tid + r * T::NTHREADS;
            if (Q::NBT % T::NTHREADS == 0 || idx < Q::NBT) lb.fetch2(c * BK + (idx / Q::CG) * 2, rb[r]);
        }
    };
    auto commit = [&](int c, f32x4* buf) {
        float* Bf = reinterpret_cast<float*>(buf);
#pragma unroll
        for (int r = 0; r < Q::B_PER; ++r) {
            const int idx = tid + r * T::NTHREADS;
            if (!(Q::NBT % T::NTHREADS == 0 || idx < Q::NBT)) continue;
            const int kp = idx / Q::CG;                       // k rows 2kp, 2kp+1 of the chunk
            const int kq = kp >> 1, kh = kp & 1;
            float o[8];
            lb.finish2(c * BK + 2 * kp, rb[r], o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = q_slot(4 * cg + j);
                f32x2 v{o[j], o[4 + j]};
                *reinterpret_cast<f32x2*>(Bf + ((size_t)(kq * T::BN + slot) * 4 + 2 * kh)) = v;
            }
        }
    };
    int bslot[T::NT];
#pragma unroll
    for (int j = 0; j < T::NT; ++j) bslot[j] = q_slot(32 * j + i31);
    fetch(0);
    commit(0, smem);
    f32x4 a0 = an0, a1 = an1;
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const f32x4* Bs = smem + (c & 1) * Q::NB;
        if (c + 1 < nchunks) fetch(c + 1);
        {
            f32x4 b0[T::NT], b1[T::NT];
#pragma unroll
            for (int j = 0; j < T::NT; ++j) {
                b0[j] = Bs[h * T::BN + bslot[j]];
                b1[j] = Bs[(h + 2) * T::BN + bslot[j]];
            }
#define WV_QSTEP(AV, BQ, COMP)                                                                     \
    _Pragma("unroll") for (int j = 0; j < T::NT; ++j)                                              \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BQ[j].COMP, acc[0][j], 0, 0, 0);
            WV_QSTEP(a0.x, b0, x) WV_QSTEP(a0.y, b0, y) WV_QSTEP(a0.z, b0, z) WV_QSTEP(a0.w, b0, w)
            WV_QSTEP(a1.x, b1, x) WV_QSTEP(a1.y, b1, y) WV_QSTEP(a1.z, b1, z) WV_QSTEP(a1.w, b1, w)
#undef WV_QSTEP
        }
        if (c + 1 < nchunks) commit(c + 1, smem + ((c + 1) & 1) * Q::NB);
        a0 = an0; a1 = an1;
        __syncthreads();
    }
}

// K1 epilogue of the round-1 core.  begin() runs BEFORE the GEMM:
// it fills the per-row table (taps, bias, FiLM gamma/beta) in a dedicated LDS region and issues
// the first residual loads, so none of the epilogue's global-memory latency is exposed after
// the matrix phase.  finish() spills the wave's 32 x BN accumulator strip two rows at a time
// into double-buffered wave-private LDS strips and applies stencil + bias (+FiLM | residual).
template <class T, int KS, int RP_ = 8>
struct PwDwEpi {
    static constexpr int HLD = T::BN + 4;
    static constexpr int RP = RP_;                            // residual rows in flight per lane
    static constexpr int WLD = KS ? 8 : 20;                   // per-row table: taps, bias, gamma, beta
    static constexpr int FLOATS = T::BM * WLD;                // LDS floats this epilogue owns (the row table)
    static constexpr int STRIP_FLOATS = T::WM * 4 * HLD;      // wave-private strips: alias the GEMM stages
    int M, m0, b, to0, lane, wave, half, q, o, to;
    bool act_lane, vec;
    const float* Rb; float* Yb; float* Wl; float* Hw;
    float4 res[RP ? RP : 1];                                  // RP == 0: a unit without residual operand

    __device__ __forceinline__ int row_of(int r) const { return 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half; }
    __device__ __forceinline__ float4 load_res(const PwDwArgs& p, int r) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gm = m0 + row_of(r);
        if (Rb && act_lane && gm < M) {
            const float* rp = Rb + (size_t)gm * p.Tout + to;
            if (vec) v = *reinterpret_cast<const float4*>(rp);
            else {
                v.x = rp[0];
                if (to + 1 < p.Tout) v.y = rp[1];
                if (to + 2 < p.Tout) v.z = rp[2];
                if (to + 3 < p.Tout) v.w = rp[3];
            }
        }
        return v;
    }
    // epi_smem: FLOATS floats not aliased with the GEMM stages.  A workgroup barrier must follow
    // before finish() (the GEMM main loop has several).
    // strip_smem may alias the GEMM stage buffers: it is only touched in finish(), and both main
    // loops end with a workgroup barrier after their last LDS read.
    __device__ __forceinline__ void begin(const PwDwArgs& p, float* epi_smem, float* strip_smem, int m0_, int b_, int to0_) {
        M = p.pw.M; m0 = m0_; b = b_; to0 = to0_;
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        half = lane >> 5; q = lane & 31;
        Wl = epi_smem;
        Hw = strip_smem + wave * (4 * HLD);
        Yb = p.Y + (size_t)b * M * p.Tout;
        Rb = p.resid ? p.resid + (size_t)b * M * p.Tout : nullptr;
        o = 4 * q; to = to0 + o;
        act_lane = o < p.tto && to < p.Tout;
        vec = act_lane && to + 3 < p.Tout && o + 3 < p.tto && (p.Tout & 3) == 0;
        if (KS) {
            const int bw = p.film ? (M / p.bands) : 1;
            const float* filmb = p.film ? p.film + (size_t)b * p.film_stride : nullptr;
            for (int m = tid; m < T::BM; m += T::NTHREADS) {
                const int gm = m0 + m;
                float v[8] = {0, 0, 0, 0, 0, 0, 1.f, 0};
                if (gm < M) {
#pragma unroll
                    for (int i = 0; i < KS; ++i) v[i] = p.dw_w[(size_t)gm * KS + i];
                    v[5] = p.dw_b ? p.dw_b[gm] : 0.f;
                    if (filmb) { const int band = gm / bw; v[6] = filmb[2 * band]; v[7] = filmb[2 * band + 1]; }
                }
                *reinterpret_cast<float4*>(Wl + m * 8) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(Wl + m * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if constexpr (RP > 0) {
#pragma unroll
                for (int r = 0; r < RP; ++r) res[r] = load_res(p, r);
            }
        } else {
            // generic stencil (strided downsample etc.): taps [0,16), bias 16, FiLM gamma 17, beta 18
            const int bw = p.film ? (M / p.bands) : 1;
            const float* filmb = p.film ? p.film + (size_t)b * p.film_stride : nullptr;
            for (int m = tid; m < T::BM; m += T::NTHREADS) {
                const int gm = m0 + m;
                float* row = Wl + m * WLD;
                for (int i = 0; i < 16; ++i) row[i] = (gm < M && i < p.ks) ? p.dw_w[(size_t)gm * p.ks + i] : 0.f;
                row[16] = (gm < M && p.dw_b) ? p.dw_b[gm] : 0.f;
                float gam = 1.f, bet = 0.f;
                if (filmb && gm < M) { const int band = gm / bw; gam = filmb[2 * band]; bet = filmb[2 * band + 1]; }
                row[17] = gam; row[18] = bet; row[19] = 0.f;
            }
        }
    }

    __device__ __forceinline__ void finish(f32x16 (&acc)[1][T::NT], const PwDwArgs& p) {
        if (KS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* strip = Hw + (r & 1) * 2 * HLD;           // double-buffered 2-row strip
#pragma unroll
                for (int j = 0; j < T::NT; ++j) strip[half * HLD + 32 * j + q] = acc[0][j][r];
                const int row = row_of(r), gm = m0 + row;
                float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (RP > 0) {
                    rr = res[r % RP];
                    if (r + RP < 16) res[r % RP] = load_res(p, r + RP);
                }
                if (act_lane && gm < M) {
                    const float4 h0 = *reinterpret_cast<const float4*>(strip + half * HLD + o);
                    const float4 h1 = *reinterpret_cast<const float4*>(strip + half * HLD + o + 4);
                    const float4 w0 = *reinterpret_cast<const float4*>(Wl + row * 8);
                    const float4 w1 = *reinterpret_cast<const float4*>(Wl + row * 8 + 4);
                    const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    const float rv[4] = {rr.x, rr.y, rr.z, rr.w};
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = w1.y;                                          // bias
                        v = fmaf(w0.x, h[e], v); v = fmaf(w0.y, h[e + 1], v); v = fmaf(w0.z, h[e + 2], v);
                        v = fmaf(w0.w, h[e + 3], v); v = fmaf(w1.x, h[e + 4], v);
                        v = fmaf(v, w1.z, w1.w);                                 // FiLM (1, 0 when off)
                        if (RP > 0 && Rb) v = fmaf(v, p.out_scale, rv[e]);
                        y[e] = v;
                    }
                    const size_t yo = (size_t)gm * p.Tout + to;
                    if (p.Y) {
                        float* yp = Yb + yo;
                        if (vec) *reinterpret_cast<float4*>(yp) = make_float4(y[0], y[1], y[2], y[3]);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (o + e < p.tto && to + e < p.Tout) yp[e] = y[e];
                        }
                    }
                    if (p.Yact) {                                   // second output: the consumer's ELU(s*y)
                        float* ya = p.Yact + (size_t)b * M * p.Tout + yo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = elu1(y[e] * p.act_scale);
                        if (vec) *reinterpret_cast<float4*>(ya) = make_float4(y[0], y[1], y[2], y[3]);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (o + e < p.tto && to + e < p.Tout) ya[e] = y[e];
                        }
                    }
                }
            }
        } else {
            const int ks = p.ks;
            const int no = (p.tto + 31) / 32;                 // consecutive outputs per lane
            const bool pair2 = no == 2 && p.stride == 2 && ks == 4 && p.dil == 1 && (p.off & 1) == 0 &&
                               (p.Tout & 1) == 0 && (to0 & 1) == 0;   // the r = 2 downsample: 8-byte LDS reads + stores
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* strip = Hw + (r & 1) * 2 * HLD;
#pragma unroll
                for (int j = 0; j < T::NT; ++j) strip[half * HLD + 32 * j + q] = acc[0][j][r];
                const int row = row_of(r), gm = m0 + row;
                if (gm >= M) continue;
                const float* wt = Wl + row * WLD;
                const float bias = wt[16], gam = wt[17], bet = wt[18];
                const float* hrow = strip + half * HLD + p.off;
                float* yrow = Yb + (size_t)gm * p.Tout + to0;
                float* arow = p.Yact ? p.Yact + ((size_t)b * M + gm) * p.Tout + to0 : nullptr;
                const float* rrow = Rb ? Rb + (size_t)gm * p.Tout + to0 : nullptr;
                if (pair2) {
                    const int o0 = 2 * q;
                    if (o0 + 1 < p.tto && to0 + o0 + 1 < p.Tout) {
                        const f32x2 ha = *reinterpret_cast<const f32x2*>(hrow + 2 * o0);
                        const f32x2 hb = *reinterpret_cast<const f32x2*>(hrow + 2 * o0 + 2);
                        const f32x2 hc = *reinterpret_cast<const f32x2*>(hrow + 2 * o0 + 4);
                        float y0 = fmaf(wt[3], hb.y, fmaf(wt[2], hb.x, fmaf(wt[1], ha.y, fmaf(wt[0], ha.x, bias))));
                        float y1 = fmaf(wt[3], hc.y, fmaf(wt[2], hc.x, fmaf(wt[1], hb.y, fmaf(wt[0], hb.x, bias))));
                        y0 = fmaf(y0, gam, bet); y1 = fmaf(y1, gam, bet);
                        if (rrow) { y0 = fmaf(y0, p.out_scale, rrow[o0]); y1 = fmaf(y1, p.out_scale, rrow[o0 + 1]); }
                        if (p.Y) *reinterpret_cast<f32x2*>(yrow + o0) = f32x2{y0, y1};
                        if (arow) *reinterpret_cast<f32x2*>(arow + o0) = f32x2{elu1(y0 * p.act_scale), elu1(y1 * p.act_scale)};
                        continue;
                    }
                }
                for (int e = 0; e < no; ++e) {
                    const int oo = q * no + e;
                    if (oo >= p.tto || to0 + oo >= p.Tout) break;
                    const float* h = hrow + oo * p.stride;
                    float y = bias;
                    for (int i = 0; i < ks; ++i) y = fmaf(wt[i], h[i * p.dil], y);
                    y = fmaf(y, gam, bet);
                    if (rrow) y = fmaf(y, p.out_scale, rrow[oo]);
                    if (p.Y) yrow[oo] = y;
                    if (arow) arow[oo] = elu1(y * p.act_scale);
                }
            }
        }
    }
};

// RM < 0: the B operand is act(s*X) (ResnetBlock / downsample / spec add).  RM >= 0: the upsample
// unit -- B is the depth-wise ConvTranspose of act(s*X) built by ConvTrPair<RM>, the stencil is the
// identity (taps 0,0,0,0,1, exact) and the "DW bias" is the 1x1 bias.
// RES = false: instantiation for units without a residual operand (first half of a ResnetBlock): it
// does not carry the residual-prefetch registers.
template <class T, int KS, int RM = -1, bool RES = true>
__global__ __launch_bounds__(T::NTHREADS, (T::NTHREADS >= 192 && T::BN == 128) ? 3 : 2) void pw_dw_kernel(PwDwArgs p) {
    // Row-strip tile: WN == 1, every wave owns 32 channel rows x the whole BN-column window, so
    // the depth-wise stencil never crosses a wave: accumulators are spilled two rows at a time
    // into a wave-private LDS strip (no workgroup barrier), read back with time on the lanes and
    // written with 16-byte stores.  KS == 5 is the ResnetBlock fast path (k5, stride 1, dil 1).
    static_assert(T::WN == 1 && T::MT == 1, "row-strip tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HLD = T::BN + 4;
    const TileId tile = decode_tile(p);
    if (!tile.valid) return;
    const int m0 = tile.m_tile * T::BM;
    const int b = tile.b;
    const int K = p.pw.K;
    const int to0 = tile.t_tile * p.tto;
    const int ti0 = to0 * p.stride - p.pad - p.off;

    if (p.stagger > 0) {
        // De-phase co-resident workgroups: the first generation (one per resident slot) starts
        // together and, having identical work, would stay in lockstep -- all in the MFMA phase or
        // all in the HBM-bound epilogue at once.  Delaying 1/3 and 2/3 of that first generation
        // lets one workgroup's epilogue overlap another's matrix phase for the whole launch.
        const unsigned lin = blockIdx.x;
        if (lin < (unsigned)p.first_gen) {
            // consecutive ids round-robin over the 8 XCDs, then over an XCD's 32 CUs: ids that
            // differ by 256 share a CU, so lin / 256 enumerates a CU's resident slots
            const int slot = lin / 256;
            for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    // residual rows in flight per lane: 8 on the 128 x 128 tile; 4 on the others, where the 16
    // registers saved buy one more wave per SIMD (or remove the spill of the 96-row tile)
    PwDwEpi<T, KS, (RM < 0 && RES) ? ((T::BN == 64 || T::BM <= 96) ? 4 : 8) : 0> epi;
    constexpr int STAGES_F = 2 * QT<T>::NB * 4;              // floats in the two B stages (A skips LDS)
    epi.begin(p, smem + STAGES_F, smem, m0, b, to0);
    static_assert(PwDwEpi<T, KS>::STRIP_FLOATS <= STAGES_F, "strips alias the stages");
    f32x16 acc[1][T::NT];
    zero_acc<T>(acc);
    if constexpr (RM < 0) {
        RowPairLoader lb{p.X + (size_t)b * K * p.Tin, K, p.Tin, p.Tin, ti0, p.pre_scale, p.pre_elu, nullptr, 0, false, false};
        gemm_mainloop_q<T>(acc, reinterpret_cast<const f32x4*>(p.pw.wq), p.pw.Mp, m0, lb, p.pw.Kp / BK,
                           reinterpret_cast<f32x4*>(smem));
    } else {
        ConvTrPair<RM> lb{p.X + (size_t)b * K * p.Tin, p.ct_w, p.ct_wt, K, p.pw.Kp, p.Tin, p.Tout, ti0, p.ratio, p.pre_scale, p.pre_elu, 0, 0, {}, {}};
        gemm_mainloop_q<T>(acc, reinterpret_cast<const f32x4*>(p.pw.wq), p.pw.Mp, m0, lb, p.pw.Kp / BK,
                           reinterpret_cast<f32x4*>(smem));
    }
    epi.finish(acc, p);
}

// ------------------------------------------------------------------------------------------
// K2  dw_pw:  Y = epi( W @ producer(X) + b ), producer fused into the B-operand loader.
//   MODE 0: act(s*X)   MODE 1: causal DW conv k of act(s*X)
// (the upsample unit -- DW ConvTranspose producer -- runs on the K1 kernel, see ConvTrPair)
// ------------------------------------------------------------------------------------------
struct ConvLoader {                    // MODE 1 (conv_post only: tiny layer, computed at commit)
    static constexpr int NRAW = 1;
    const float* Xb; const float* dw_w; int K, Tin, t0, ks; float scale; int elu; int t;
    __device__ __forceinline__ void init(int col4) { t = t0 + col4; }
    __device__ __forceinline__ void fetch(int, float (&)[4]) const {}
    __device__ __forceinline__ void finish(int k, const float (&)[4], float (&o)[4]) const {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = 0.f;
            if (k < K && t + c < Tin) {
                const float* xr = Xb + (size_t)k * Tin;
                const float* w = dw_w + (size_t)k * ks;
                for (int i = 0; i < ks; ++i) {
                    const int ti = t + c - (ks - 1) + i;
                    if (ti >= 0) v = fmaf(w[i], act(xr[ti], scale, elu), v);
                }
            }
            o[c] = v;
        }
    }
};

template <class T, class LB>
__device__ __forceinline__ void dw_pw_body(const DwPwArgs& p, LB& lb, float* smem, int m0, int t0, int b) {
    constexpr int HLD = T::BN + 4;
    const int M = p.pw.M;
    {
        f32x16 acc[T::MT][T::NT];
        zero_acc<T>(acc);
        WLoader la{p.pw.wt, p.pw.Mp, m0};
        gemm_mainloop<T>(acc, la, lb, p.pw.Kp / BK, smem);
        float* Hs = smem;                                // [BM][HLD], aliases the stages
        for_each_acc<T>(acc, [&](int row, int col, float v) { Hs[row * HLD + col] = v; });
    }
    __syncthreads();
    float* Hs = smem;
    float* Yb = p.Y + (size_t)b * M * p.Tout;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncol = min(T::BN, p.Tout - t0);
    float* inv = Hs + T::BM * HLD;                       // [BN] column scale (L2-norm only)
    if (p.l2norm) {
        // L2Norm over channels (seanet.py:288-318): y = v / max(||v||_2, 1e-12) * sqrt(M).
        // The host guarantees a single m-tile (M <= BM).
        for (int c = threadIdx.x; c < T::BN; c += NT_) {
            float ss = 0.f;
            for (int m = 0; m < M; ++m) {
                const float v = Hs[m * HLD + c] + (p.bias ? p.bias[m] : 0.f);
                ss = fmaf(v, v, ss);
            }
            inv[c] = sqrtf((float)M) / fmaxf(sqrtf(ss), 1e-12f);
        }
        __syncthreads();
    }
    // Full, 16-byte aligned tiles: float4 rows, LPR lanes per row, U passes batched so the
    // read-modify-write of an accumulate layer has U independent loads in flight per lane.
    if (ncol == T::BN && (p.Tout & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Y) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(p.Yact) & 15) == 0) {
        constexpr int LPR = T::BN / 4, RPP = 64 / LPR, RPW = T::BM / 4, NP = RPW / RPP;
        constexpr int U = NP % 8 == 0 ? 8 : (NP % 4 == 0 ? 4 : (NP % 2 == 0 ? 2 : 1));
        static_assert(RPW % RPP == 0, "rows per wave");
        const int sub = lane / LPR, c4 = (lane % LPR) * 4;
        f32x4 sc{1.f, 1.f, 1.f, 1.f};
        if (p.l2norm) sc = *reinterpret_cast<const f32x4*>(inv + c4);
        for (int p0 = 0; p0 < NP; p0 += U) {
            f32x4 y[U];
            if (p.accumulate) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int gm = m0 + wave * RPW + (p0 + u) * RPP + sub;
                    y[u] = gm < M ? *reinterpret_cast<const f32x4*>(Yb + (size_t)gm * p.Tout + t0 + c4)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = wave * RPW + (p0 + u) * RPP + sub, gm = m0 + row;
                if (gm >= M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(Hs + row * HLD + c4);
                if (p.accumulate) {
                    v.x = fmaf(p.out_scale, v.x, y[u].x); v.y = fmaf(p.out_scale, v.y, y[u].y);
                    v.z = fmaf(p.out_scale, v.z, y[u].z); v.w = fmaf(p.out_scale, v.w, y[u].w);
                } else {
                    const float bias = p.bias ? p.bias[gm] : 0.f;
                    v.x += bias; v.y += bias; v.z += bias; v.w += bias;
                }
                if (p.l2norm) { v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
                *reinterpret_cast<f32x4*>(Yb + (size_t)gm * p.Tout + t0 + c4) = v;
                if (p.Yact)
                    *reinterpret_cast<f32x4*>(p.Yact + ((size_t)b * M + gm) * p.Tout + t0 + c4) =
                        f32x4{elu1(v.x * p.act_scale), elu1(v.y * p.act_scale), elu1(v.z * p.act_scale), elu1(v.w * p.act_scale)};
            }
        }
        return;
    }
    for (int m = wave; m < T::BM; m += 4) {              // ragged tiles: wave-uniform row, time on the lanes
        const int gm = m0 + m;
        if (gm >= M) break;
        const float bias = p.bias ? p.bias[gm] : 0.f;
        float* yrow = Yb + (size_t)gm * p.Tout + t0;
        const float* h = Hs + m * HLD;
        for (int c = lane; c < ncol; c += 64) {
            float v = h[c];
            if (p.accumulate) v = fmaf(p.out_scale, v, yrow[c]);
            else v += bias;
            if (p.l2norm) v *= inv[c];
            yrow[c] = v;
            if (p.Yact) p.Yact[((size_t)b * M + gm) * p.Tout + t0 + c] = elu1(v * p.act_scale);
        }
    }
}

template <class T, int MODE>
__global__ __launch_bounds__(NT_) void dw_pw_kernel(DwPwArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.x * T::BM;
    const int t0 = blockIdx.y * T::BN;
    const int b = blockIdx.z;
    const int K = p.pw.K;
    const float* Xb = p.X + (size_t)b * K * p.Tin;
    if (MODE == 0) {
        RowLoader lb{Xb, K, p.Tin, p.Tin, t0, p.pre_scale, p.pre_elu, nullptr, 0, false, false};
        dw_pw_body<T>(p, lb, smem, m0, t0, b);
    } else {
        ConvLoader lb{Xb, p.dw_w, K, p.Tin, t0, p.ks, p.pre_scale, p.pre_elu, 0};
        dw_pw_body<T>(p, lb, smem, m0, t0, b);
    }
}

// ------------------------------------------------------------------------------------------
// K3a  causal STFT -> log-magnitude -> affine.  GEMM with A = interleaved (cos,sin) basis and
// B = frames gathered on the fly from the waveform (left zero history only at true t < 0).
// An accumulator register pair (2q, 2q+1) of one lane is (re, im) of one bin, so the
// magnitude needs no cross-lane traffic.
// ------------------------------------------------------------------------------------------
struct FrameLoader {
    static constexpr int NRAW = 1;
    const float* wb; int T, Tf, n_fft, hop, t0;
    int base[4]; bool inb[4];
    __device__ __forceinline__ void init(int col4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = t0 + col4 + c;
            inb[c] = t < Tf;
            base[c] = t * hop - (n_fft - 1);
        }
    }
    __device__ __forceinline__ void fetch(int k, float (&raw)[4]) const {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = base[c] + k;
            raw[c] = (inb[c] && k < n_fft && idx >= 0 && idx < T) ? wb[idx] : 0.f;
        }
    }
    __device__ __forceinline__ void finish(int, const float (&raw)[4], float (&o)[4]) const {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = raw[c];
    }
};

// the two vector-side rows (sin_0, sin_{F-1}): thread `col` of the first m-tile keeps the two dot
// products of its frame, fed from the staged B chunk
template <int BN>
struct StftSide {
    const float* s0; const float* s1; int n_fft; bool on; int col; float d0, d1;
    __device__ __forceinline__ void operator()(int c, const float* Bs) {
        if (!on) return;
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            const int k = min(c * BK + kk, n_fft - 1);            // staged rows past n_fft are zero
            const float v = Bs[kk * BN + col];
            d0 = fmaf(s0[k], v, d0);
            d1 = fmaf(s1[k], v, d1);
        }
    }
};

template <class T>
__global__ __launch_bounds__(NT_) void stft_logmag_kernel(StftArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.x * T::BM;
    const int t0 = blockIdx.y * T::BN;
    const int b = blockIdx.z;
    f32x16 acc[T::MT][T::NT];
    zero_acc<T>(acc);
    WLoader la{p.basis_t, p.Mp, m0};
    FrameLoader lb{p.wav + (size_t)b * p.T, p.T, p.Tf, p.n_fft, p.hop, t0, {}, {}};
    StftSide<T::BN> side{p.side, p.side + p.n_fft, p.n_fft, blockIdx.x == 0 && (int)threadIdx.x < T::BN,
                         (int)threadIdx.x % T::BN, 0.f, 0.f};
    gemm_mainloop<T>(acc, la, lb, (p.n_fft + BK - 1) / BK, smem, side);
    float* sd = smem;                                        // [2][BN] (stages are free after the last barrier)
    if (side.on) { sd[side.col] = side.d0; sd[T::BN + side.col] = side.d1; }
    if (blockIdx.x == 0) __syncthreads();

    float* Pb = p.P + (size_t)b * p.F * p.Tf;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int row = m0 + wm * T::MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * T::NT * 32 + j * 32 + (lane & 31);
                const int t = t0 + col;
                if (row >= p.n_fft || t >= p.Tf) continue;
                const float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                auto logmag = [&](float re, float im) { return stft_logmag(re, im, p.c1, p.c0); };
                if (row == 0) {                          // (cos_0, cos_{F-1}) + the vector-side sin rows
                    Pb[t] = logmag(v0, sd[col]);
                    Pb[(size_t)(p.F - 1) * p.Tf + t] = logmag(v1, sd[T::BN + col]);
                } else {                                 // row = 2f: (re, im) of bin f
                    Pb[(size_t)(row >> 1) * p.Tf + t] = logmag(v0, v1);
                }
            }
}

// ------------------------------------------------------------------------------------------
// K4  conv_pre: Conv1d(1 -> C, k) on x*in_scale (seanet.py:657-664).  HBM-bound: one read of
// x, C coalesced row writes.
// ------------------------------------------------------------------------------------------
constexpr int MAX_KS = 16;

__global__ __launch_bounds__(NT_) void conv_pre_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ Y, float* __restrict__ Yact,
                                                       float act_scale, int C, int T, int ks, float in_scale) {
    // a thread owns 4 consecutive samples: one 16-byte store per channel row (1 KB per wave)
    const int b = blockIdx.y;
    const int t = (blockIdx.x * NT_ + threadIdx.x) * 4;
    if (t >= T) return;
    const float* xb = x + (size_t)b * T;
    float xv[MAX_KS + 3];                                    // x[t-(ks-1) .. t+3], scaled, 0 outside [0,T)
#pragma unroll
    for (int i = 0; i < MAX_KS + 3; ++i) {
        const int ti = t - (ks - 1) + i;
        xv[i] = (i < ks + 3 && ti >= 0 && ti < T) ? xb[ti] * in_scale : 0.f;
    }
    float* yb = Y + (size_t)b * C * T + t;
    float* ab = Yact ? Yact + (size_t)b * C * T + t : nullptr;      // second output: ELU(act_scale * y)
    const bool vec = (T & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (reinterpret_cast<uintptr_t>(Yact) & 15) == 0;
    for (int c = 0; c < C; ++c) {
        const float bc = bias ? bias[c] : 0.f;
        float y[4] = {bc, bc, bc, bc};
#pragma unroll
        for (int i = 0; i < MAX_KS; ++i)
            if (i < ks) {
                const float wi = w[c * ks + i];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaf(wi, xv[i + e], y[e]);
            }
        float* yr = yb + (size_t)c * T;
        if (vec) *reinterpret_cast<f32x4*>(yr) = f32x4{y[0], y[1], y[2], y[3]};
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e < T) yr[e] = y[e];
        }
        if (ab) {
            float* ar = ab + (size_t)c * T;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = elu1(y[e] * act_scale);
            if (vec) *reinterpret_cast<f32x4*>(ar) = f32x4{y[0], y[1], y[2], y[3]};
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + e < T) ar[e] = y[e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K5  decoder tail: tanh(out_scale*(Conv1d(C -> 1, k)(ELU(s*H)) + b)) (+ x).  HBM-bound read of
// H[B,C,Tin].  Each wave owns 64 consecutive input columns (ks-1 of them halo); a lane
// activates its own column once and gets its ks-1 left neighbours by wave shuffles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT_) void tail_kernel(const float* __restrict__ H,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ x,
                                                   float* __restrict__ out, int C, int Tin, int T,
                                                   int ks, float pre_scale, float out_scale) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = 64 - (ks - 1);
    const int t = (blockIdx.x * 4 + wave) * per_wave - (ks - 1) + lane;   // this lane's column
    const bool inb = t >= 0 && t < Tin;
    const float* hb = H + (size_t)b * C * Tin + (inb ? t : 0);
    float y = 0.f;
    for (int c0 = 0; c0 < C; c0 += 8) {                        // 8 channel rows in flight per lane
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = hb[(size_t)min(c0 + u, C - 1) * Tin];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            if (c >= C) break;
            const float e = inb ? elu1(v[u] * pre_scale) : 0.f;
            const float* wc = w + c * ks;
            y = fmaf(wc[ks - 1], e, y);
            for (int i = 1; i < ks; ++i) {
                const float ei = __shfl_up(e, i);              // lanes < i get garbage; they are halo lanes
                y = fmaf(wc[ks - 1 - i], ei, y);
            }
        }
    }
    if (lane >= ks - 1 && t < T) {
        float v = tanhf((y + (bias ? bias[0] : 0.f)) * out_scale);
        if (x) v += x[(size_t)b * T + t];
        out[(size_t)b * T + t] = v;
    }
}

// The same tail for ks = 5 and 16-byte aligned rows: a lane loads FOUR consecutive samples per channel (1 KB per wave and row instead
// of 256 B), the four older ones come from the previous lane; a wave makes 252 outputs, a workgroup 1008.  Taps are summed in the order of
// tail_kernel (newest sample first), so results are bit-identical.
__global__ __launch_bounds__(NT_) void tail5_vec_kernel(const float* __restrict__ H, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ x, float* __restrict__ out, int C, int Tin, int T,
                                                        float pre_scale, float out_scale) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (blockIdx.x * 4 + wave) * 252;                  // first output time of this wave
    const int c0 = base - 4 + 4 * lane;                              // this lane's four input times c0 .. c0 + 3 (lane 0: the halo)
    const bool inb = c0 >= 0 && c0 + 3 < Tin;
    const float* hb = H + (size_t)b * C * Tin + (inb ? c0 : 0);
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch0 = 0; ch0 < C; ch0 += 4) {                           // 4 channel rows in flight per lane
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(hb + (size_t)min(ch0 + u, C - 1) * Tin);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = ch0 + u;
            if (c >= C) break;
            float e[8];                                              // e[0..3] = previous lane's samples, e[4..7] = own
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float ev = 0.f;
                if (inb) ev = elu1(v[u][k] * pre_scale);
                else if (c0 + k >= 0 && c0 + k < Tin) ev = elu1(H[((size_t)b * C + c) * Tin + c0 + k] * pre_scale);
                e[4 + k] = ev;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = __shfl_up(e[4 + k], 1);
            const float* wc = w + c * 5;
            const float w0 = wc[0], w1 = wc[1], w2 = wc[2], w3 = wc[3], w4 = wc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // output at time c0 + j: e[j+4] newest ... e[j] oldest
                y[j] = fmaf(w4, e[j + 4], y[j]);
                y[j] = fmaf(w3, e[j + 3], y[j]);
                y[j] = fmaf(w2, e[j + 2], y[j]);
                y[j] = fmaf(w1, e[j + 1], y[j]);
                y[j] = fmaf(w0, e[j], y[j]);
            }
        }
    }
    if (lane >= 1) {
        const float bv = bias ? bias[0] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = c0 + j;
            if (t < T) {
                float v = tanhf((y[j] + bv) * out_scale);
                if (x) v += x[(size_t)b * T + t];
                out[(size_t)b * T + t] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K6  detector / locator head.  The reference's ConvTranspose1d(k = s = hop) -> trim ->
// Conv1d(O -> nb, 1) is one GEMM per frame against the composed weight wc[D][nb*hop].
// Workgroup = (bit, clip): rows = frames of the clip, cols = the hop samples of that bit.
// sigmoid + mean over time are reduced in-register, so for detect() the [B,nb,T] logits
// never reach HBM (core.py:577-580); summation order is fixed -> deterministic.
// ------------------------------------------------------------------------------------------
struct ZLoader {                       // A operand: rows = frames of one clip, Z[b][k][f]
    const float* Zb; int D, Fr, f0;
    __device__ __forceinline__ float4 load4(int k, int m) const {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = f0 + m + i;
            v[i] = (k < D && f < Fr) ? Zb[(size_t)k * Fr + f] : 0.f;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

template <class T>
__global__ __launch_bounds__(NT_) void head_kernel(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bit = blockIdx.x, b = blockIdx.y;
    const float bc = p.bc[bit];
    float psum = 0.f;
    const int nchunks = (p.D + BK - 1) / BK;
    for (int f0 = 0; f0 < p.Fr; f0 += T::BM) {
        for (int j0 = 0; j0 < p.hop; j0 += T::BN) {
            f32x16 acc[T::MT][T::NT];
            zero_acc<T>(acc);
            ZLoader la{p.Z + (size_t)b * p.D * p.Fr, p.D, p.Fr, f0};
            RowLoader lb{p.wc + (size_t)bit * p.hop, p.D, p.nb * p.hop, p.hop, j0, 1.f, 0,
                         nullptr, 0, false, false};
            gemm_mainloop<T>(acc, la, lb, nchunks, smem);
            for_each_acc<T>(acc, [&](int row, int col, float v) {
                const int f = f0 + row, j = j0 + col;
                const int t = f * p.hop + j;
                if (f < p.Fr && j < p.hop && t < p.T) {
                    const float lg = v + bc;
                    if (p.logits) p.logits[((size_t)b * p.nb + bit) * p.T + t] = lg;
                    psum += sigmoidf_(lg);
                }
            });
        }
    }
    if (p.mean_prob) {
        float* red = smem;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = psum;
        __syncthreads();
        if (threadIdx.x == 0)
            p.mean_prob[(size_t)b * p.nb + bit] = (red[0] + red[1] + red[2] + red[3]) / (float)p.T;
    }
}

// ------------------------------------------------------------------------------------------
// K7  message MLP + FiLM gammas/betas (seanet.py:831-846, 905-912).  One workgroup per clip.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT_) void film_kernel(FilmArgs p) {
    __shared__ float h0[NT_], h1[NT_];
    const int b = blockIdx.x, e = threadIdx.x;
    const float* msg = p.msg + (size_t)(p.msg_rows == 1 ? 0 : b) * p.msg_dim;
    if (e < p.E) {
        float v = p.b0[e];
        for (int i = 0; i < p.msg_dim; ++i) v = fmaf(p.w0[e * p.msg_dim + i], msg[i], v);
        h0[e] = v;                                     // no ReLU after the first Linear
    }
    __syncthreads();
    float* cur = h0;
    float* nxt = h1;
    for (int l = 0; l < p.n_layers; ++l) {
        if (e < p.E) {
            const float* w = p.wl + ((size_t)l * p.E + e) * p.E;
            float v = p.bl[l * p.E + e];
            for (int i = 0; i < p.E; ++i) v = fmaf(w[i], cur[i], v);
            nxt[e] = fmaxf(v, 0.f);
        }
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int o = e; o < p.n_out; o += NT_) {
        float v = p.bf[o];
        for (int i = 0; i < p.E; ++i) v = fmaf(p.wf[(size_t)o * p.E + i], cur[i], v);
        p.film[(size_t)b * p.n_out + o] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Per-launch profiler (HIP events on the launch stream)
// ------------------------------------------------------------------------------------------
}  // namespace wv
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
namespace wv {
namespace prof {
namespace {
// All state below is guarded by g_mu (launchers may be called from several host threads).  Events are
// never recorded on a capturing stream (a captured event cannot be synchronised on later), and the
// record list is drained when it grows past MAX_RECS so a long profiled run keeps bounded memory.
struct Rec { hipEvent_t a, b; int key; };
struct Agg { std::string name; long long launches = 0; double ms = 0, flops = 0, bytes = 0; };
constexpr size_t MAX_RECS = 8192;
std::mutex g_mu;
std::atomic<bool> g_on{false};
thread_local const char* g_role = "";
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::vector<Agg> g_agg;
std::map<std::string, int> g_index;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void drain_locked() {
    for (Rec& r : g_recs) {
        float ms = 0.f;
        if (r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess)
            g_agg[r.key].ms += ms;
        g_pool.push_back(r.a);
        if (r.b) g_pool.push_back(r.b);
    }
    g_recs.clear();
}
}  // namespace
void enable(bool on) { g_on.store(on); }
bool enabled() { return g_on.load(); }
void reset() { std::lock_guard<std::mutex> lk(g_mu); drain_locked(); g_agg.clear(); g_index.clear(); }
void set_role(const char* role) { g_role = role ? role : ""; }
int collect(Entry* out, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    int n = 0;
    for (const Agg& a : g_agg) {
        if (n < cap) out[n] = Entry{a.name.c_str(), a.launches, a.ms, a.flops, a.bytes};
        ++n;
    }
    return n;
}
Scope::Scope(hipStream_t st, const char* kernel, double flops, double bytes) : s(st) {
    if (!g_on.load(std::memory_order_relaxed)) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_recs.size() >= MAX_RECS) drain_locked();
    std::string key = std::string(kernel) + "|" + g_role;
    auto it = g_index.find(key);
    int k;
    if (it == g_index.end()) { k = (int)g_agg.size(); g_index[key] = k; Agg a; a.name = key; g_agg.push_back(a); }
    else k = it->second;
    g_agg[k].launches += 1; g_agg[k].flops += flops; g_agg[k].bytes += bytes;
    ev_a = get_event(); ev_b = get_event(); key_ = k;
    (void)hipEventRecord((hipEvent_t)ev_a, s);
    armed = true;
}
Scope::~Scope() {
    if (!armed) return;
    (void)hipEventRecord((hipEvent_t)ev_b, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(Rec{(hipEvent_t)ev_a, (hipEvent_t)ev_b, key_});
}
}  // namespace prof

template <class T>
static std::string tile_name(const char* base) {
    return std::string(base) + "<" + std::to_string(T::BM) + "," + std::to_string(T::BN) + "," +
           std::to_string(T::WM) + "," + std::to_string(T::WN) + ">";
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------
template <class T>
static constexpr size_t stage_bytes() { return 2 * (size_t)T::STAGE * sizeof(float); }

// Per-kernel launch attributes are per DEVICE (one process may drive several GPUs): `done` is a bit
// mask of the devices on which the attribute has been set (idempotent, so a lost race only repeats it).
static int cur_dev() { int d = 0; (void)hipGetDevice(&d); return d & 31; }
template <class K>
static hipError_t set_smem(K kernel, size_t bytes, std::atomic<unsigned>& done) {
    if (bytes <= 64 * 1024) return hipSuccess;
    const unsigned bit = 1u << cur_dev();
    if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

static int gcd_(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// Time-tile geometry of K1: the H window of a tile starts at ti0 = to0*s - pad - off with `off`
// chosen so that ti0 is a multiple of 4 for every tile (16-byte X loads); tto outputs per tile.
bool pw_dw_geometry(PwDwArgs& a, int BN) {
    const int span = (a.ks - 1) * a.dil + 1;
    a.off = (4 - (a.pad % 4)) % 4;
    int tto = (BN - a.off - span) / a.stride + 1;
    const int q = 4 / gcd_(a.stride, 4);
    tto -= tto % q;
    if (tto < 1) {                                  // cannot align: plain window, scalar loads
        a.off = 0;
        tto = (BN - span) / a.stride + 1;
    }
    a.tto = tto;
    return tto >= 1;
}

template <class T, int KS>
static hipError_t run_pw_dw_ks(PwDwArgs a, hipStream_t s) {
    if (!pw_dw_geometry(a, T::BN)) return hipErrorInvalidValue;
    const size_t eb = (size_t)PwDwEpi<T, KS>::FLOATS * sizeof(float);
    const size_t smem = 2 * (size_t)QT<T>::NB * 16 + eb;       // two B stages + the epilogue's row table
    static std::atomic<unsigned> attr_f32{0};
    {
        hipError_t e = set_smem(pw_dw_kernel<T, KS>, smem, attr_f32);
        if (e != hipSuccess) return e;
    }
    static std::atomic<int> per_cu_dev[32];               // resident workgroups per CU, per device (0 = unknown)
    int per_cu = per_cu_dev[cur_dev()].load(std::memory_order_relaxed);
    if (per_cu < 1) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pw_dw_kernel<T, KS>, T::NTHREADS, smem) != hipSuccess || n < 1)
            n = 1;
        per_cu = n;
        per_cu_dev[cur_dev()].store(n, std::memory_order_relaxed);
    }
    // de-phase the first generation of workgroups (see kernel); off for tiny grids
    a.stagger = 2;                                         // 2 x s_sleep(127) per resident slot
    a.first_gen = 256 * per_cu;
    a.num_m = (a.pw.M + T::BM - 1) / T::BM;
    a.num_t = (a.Tout + a.tto - 1) / a.tto;
    const long long n_act = (long long)a.num_t * a.B;
    const long long nblk = ((n_act + 7) / 8) * 8 * a.num_m;
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblk);
    if (nblk < 4LL * a.first_gen) a.stagger = 0;
    static const std::string name = tile_name<T>(KS ? "pw_dw_k5" : "pw_dw");
    const double M = a.pw.M, K = a.pw.K, Bd = a.B;
    if constexpr (KS == 5) if (a.ct_w) {
        // upsample unit: ConvTranspose producer in the B loader, identity stencil (see pw_dw_kernel)
        static const std::string cname = tile_name<T>("convtr_pw");
        prof::Scope pc(s, cname.c_str(), 2.0 * Bd * a.Tout * K * (M + 2.0), 4.0 * Bd * (K * a.Tin + M * a.Tout));
        static std::atomic<unsigned> attr_ct4{0}, attr_ct2{0}, attr_ct1{0}, attr_ct0{0};
        {
            hipError_t e = set_smem(pw_dw_kernel<T, 5, 4>, smem, attr_ct4);
            if (e == hipSuccess) e = set_smem(pw_dw_kernel<T, 5, 2>, smem, attr_ct2);
            if (e == hipSuccess) e = set_smem(pw_dw_kernel<T, 5, 1>, smem, attr_ct1);
            if (e == hipSuccess) e = set_smem(pw_dw_kernel<T, 5, 0>, smem, attr_ct0);
            if (e != hipSuccess) return e;
        }
        if (a.ratio % 4 == 0) hipLaunchKernelGGL((pw_dw_kernel<T, 5, 4>), grid, dim3(T::NTHREADS), smem, s, a);
        else if (a.ratio == 2) hipLaunchKernelGGL((pw_dw_kernel<T, 5, 2>), grid, dim3(T::NTHREADS), smem, s, a);
        else if (a.ratio == 1) hipLaunchKernelGGL((pw_dw_kernel<T, 5, 1>), grid, dim3(T::NTHREADS), smem, s, a);
        else hipLaunchKernelGGL((pw_dw_kernel<T, 5, 0>), grid, dim3(T::NTHREADS), smem, s, a);
        return hipGetLastError();
    }
    if constexpr (KS == 5) if (a.spec_add && a.resid) {
        // same code as the residual unit; a separate instantiation (RM = -2) so that profiles keep
        // this HBM-bound launch apart from the matrix-bound ResnetBlock halves
        static const std::string sname = tile_name<T>("spec_add");
        prof::Scope pa(s, sname.c_str(), 2.0 * Bd * M * K * a.Tin, 4.0 * Bd * (K * a.Tin + 2.0 * M * a.Tout));
        static std::atomic<unsigned> attr_sa{0};
        {
            hipError_t e = set_smem(pw_dw_kernel<T, 5, -2>, smem, attr_sa);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((pw_dw_kernel<T, 5, -2>), grid, dim3(T::NTHREADS), smem, s, a);
        return hipGetLastError();
    }
    static const std::string name_nr = tile_name<T>("pw_dw_k5_nr");    // no-residual instantiation
    const bool nores = KS == 5 && !a.resid;
    prof::Scope ps(s, nores ? name_nr.c_str() : name.c_str(), 2.0 * Bd * M * (K * a.Tin + (double)a.ks * a.Tout),
                   4.0 * Bd * (K * a.Tin + M * a.Tout * (a.resid ? 2.0 : 1.0)));
    if constexpr (KS == 5) if (nores) {
        static std::atomic<unsigned> attr_nr{0};
        {
            hipError_t e = set_smem(pw_dw_kernel<T, 5, -1, false>, smem, attr_nr);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((pw_dw_kernel<T, 5, -1, false>), grid, dim3(T::NTHREADS), smem, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((pw_dw_kernel<T, KS>), grid, dim3(T::NTHREADS), smem, s, a);
    return hipGetLastError();
}

template <class T>
static hipError_t run_pw_dw(const PwDwArgs& a, hipStream_t s) {
    if (a.ks == 5 && a.stride == 1 && a.dil == 1 && a.pad == 4) return run_pw_dw_ks<T, 5>(a, s);
    return run_pw_dw_ks<T, 0>(a, s);
}

static int pick_bm(int M) {
    if (M <= 32) return 32;
    if (M <= 64) return 64;
    if (M <= 96) return 96;
    if (M % 128 == 0) return 128;
    if (M == 192) return 64;           // measured: 3 x 64 (two-wave workgroups) beats 2 x 96 by 3-4 %
    if (M % 96 == 0) return 96;
    return 128;
}

hipError_t launch_pw_dw(const PwDwArgs& a, hipStream_t s) {
    if (a.ct_w && (!a.ct_wt || (reinterpret_cast<uintptr_t>(a.ct_wt) & 7) || a.ratio < 1 || a.Tout != a.Tin * a.ratio || a.ks != 5 || a.stride != 1 || a.dil != 1 || a.pad != 4 ||
                   (reinterpret_cast<uintptr_t>(a.ct_w) & 15)))
        return hipErrorInvalidValue;
    if (a.ks < 1 || a.ks > 16 || a.stride < 1 || a.dil < 1 || a.pad < 0 || a.pw.Mp % M_ALIGN || a.pw.Kp % BK || !a.pw.wq)
        return hipErrorInvalidValue;
    if (k1_supported(a)) {                                   // the LDS-DMA core (wv_k1.hip)
        const hipError_t e = launch_k1(a, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.res_mode || a.Yraw || a.Ysum) return hipErrorNotSupported;   // only the LDS-DMA core has the training epilogues: the caller runs them unfused
    const int need = (a.ks - 1) * a.dil + 1;            // H columns one output needs
    bool narrow = a.Tin + a.pad + 3 <= 64 && need + 3 <= 64;
    if (!narrow && need + 3 <= 64) {
        // pick the window width that computes the fewest columns for this Tout (tile quantisation:
        // e.g. Tout = 400 needs 4 x 128 columns with 124-output tiles but only 7 x 64 with 60-output ones)
        PwDwArgs g128 = a, g64 = a;
        if (pw_dw_geometry(g128, 128) && pw_dw_geometry(g64, 64)) {
            const long long c128 = (long long)((a.Tout + g128.tto - 1) / g128.tto) * 128;
            const long long c64 = (long long)((a.Tout + g64.tto - 1) / g64.tto) * 64;
            if (c64 * 100 < c128 * 95) narrow = true;
        }
    }
    const int bm = pick_bm(a.pw.M);
    if (narrow) {
        switch (bm) {
            case 32: return run_pw_dw<Tile<32, 64, 1, 1>>(a, s);
            case 64: return run_pw_dw<Tile<64, 64, 2, 1>>(a, s);
            case 96: return run_pw_dw<Tile<96, 64, 3, 1>>(a, s);
            default: return run_pw_dw<Tile<128, 64, 4, 1>>(a, s);
        }
    }
    switch (bm) {
        case 32: return run_pw_dw<Tile<32, 128, 1, 1>>(a, s);
        case 64: return run_pw_dw<Tile<64, 128, 2, 1>>(a, s);
        case 96: return run_pw_dw<Tile<96, 128, 3, 1>>(a, s);
        default: return run_pw_dw<Tile<128, 128, 4, 1>>(a, s);
    }
}

template <class T, int MODE>
static hipError_t run_dw_pw_mode(const DwPwArgs& a, hipStream_t s) {
    size_t smem = stage_bytes<T>();
    const size_t hb = ((size_t)T::BM * (T::BN + 4) + T::BN) * sizeof(float);
    if (hb > smem) smem = hb;
    static std::atomic<unsigned> attr_done{0};
    {
        hipError_t e = set_smem(dw_pw_kernel<T, MODE>, smem, attr_done);
        if (e != hipSuccess) return e;
    }
    if (a.l2norm && a.pw.M > T::BM) return hipErrorInvalidValue;
    dim3 grid((a.pw.M + T::BM - 1) / T::BM, (a.Tout + T::BN - 1) / T::BN, a.B);
    static const std::string name = tile_name<T>(MODE == 0 ? "pw" : "dwconv_pw");
    const double M = a.pw.M, K = a.pw.K, Bd = a.B;
    const double stencil = a.mode == 1 ? 2.0 * a.ks : 0.0;
    prof::Scope ps(s, name.c_str(), Bd * a.Tout * (2.0 * M * K + stencil * K),
                   4.0 * Bd * (K * a.Tin + M * a.Tout * (a.accumulate ? 2.0 : 1.0)));
    hipLaunchKernelGGL((dw_pw_kernel<T, MODE>), grid, dim3(NT_), smem, s, a);
    return hipGetLastError();
}

template <class T>
static hipError_t run_dw_pw(const DwPwArgs& a, hipStream_t s) {
    if (a.mode == 0) return run_dw_pw_mode<T, 0>(a, s);
    return run_dw_pw_mode<T, 1>(a, s);
}

hipError_t launch_dw_pw(const DwPwArgs& a, hipStream_t s) {
    if (a.pw.Mp % M_ALIGN || a.pw.Kp % BK || a.mode < 0 || a.mode > 1 || a.Tout != a.Tin) return hipErrorInvalidValue;
    int bm = pick_bm(a.pw.M);
    if (a.l2norm) bm = a.pw.M <= 64 ? 64 : 128;
    if (a.Tout <= 64) {
        if (bm <= 64) return run_dw_pw<Tile<64, 64, 2, 2>>(a, s);
        return run_dw_pw<Tile<128, 64, 2, 2>>(a, s);
    }
    switch (bm) {
        case 32: return run_dw_pw<Tile<32, 128, 1, 4>>(a, s);
        case 64: return run_dw_pw<Tile<64, 128, 1, 4>>(a, s);
        case 96: return run_dw_pw<Tile<96, 128, 1, 4>>(a, s);
        default: return run_dw_pw<Tile<128, 128, 1, 4>>(a, s);
    }
}

template <class T>
static hipError_t run_stft(const StftArgs& a, hipStream_t s) {
    dim3 grid((a.n_fft + T::BM - 1) / T::BM, (a.Tf + T::BN - 1) / T::BN, a.B);
    static const std::string name = tile_name<T>("stft_logmag");
    prof::Scope ps(s, name.c_str(), 2.0 * a.B * (2.0 * a.F) * a.n_fft * a.Tf,
                   4.0 * a.B * ((double)a.T + (double)a.F * a.Tf));
    hipLaunchKernelGGL(stft_logmag_kernel<T>, grid, dim3(NT_), stage_bytes<T>(), s, a);
    return hipGetLastError();
}

hipError_t launch_stft_logmag(const StftArgs& a_in, hipStream_t s) {
    StftArgs a = a_in;
    a.c1 = 0.5f * 0.69314718055994531f * a.inv_std;             // see stft_logmag (wv_dev.h)
    a.c0 = -a.mean * a.inv_std;
    if (a.Mp % M_ALIGN || a.Mp < a.n_fft || a.n_fft < 4 || (a.n_fft & 1) || a.F != a.n_fft / 2 + 1 || !a.side)
        return hipErrorInvalidValue;
    {
        const hipError_t e = launch_stft_k1(a, s);                 // the LDS-DMA core; not supported: <= 64 or odd frame counts
        if (e != hipErrorNotSupported) return e;
    }
    if (a.Tf <= 64) return run_stft<Tile<128, 64, 2, 2>>(a, s);
    if (a.n_fft <= 64) return run_stft<Tile<64, 128, 1, 4>>(a, s);
    // n_fft rows: pick the tile height that pads them least
    if (round_up(a.n_fft, 96) < round_up(a.n_fft, 128)) return run_stft<Tile<96, 128, 1, 4>>(a, s);
    return run_stft<Tile<128, 128, 1, 4>>(a, s);
}

hipError_t launch_conv_pre(const float* x, const float* w, const float* bias, float* Y, float* Yact, float act_scale,
                           int B, int C, int T, int ks, float in_scale, hipStream_t s) {
    if (ks < 1 || ks > MAX_KS) return hipErrorInvalidValue;
    dim3 grid((T + 4 * NT_ - 1) / (4 * NT_), B);
    prof::Scope ps(s, "conv_pre", 2.0 * B * C * ks * (double)T, 4.0 * B * (double)T * (1.0 + C * (Yact ? 2.0 : 1.0)));
    hipLaunchKernelGGL(conv_pre_kernel, grid, dim3(NT_), 0, s, x, w, bias, Y, Yact, act_scale, C, T, ks, in_scale);
    return hipGetLastError();
}

hipError_t launch_tail(const float* H, const float* w, const float* bias, const float* x,
                       float* out, int B, int C, int Tin, int T, int ks, float pre_scale,
                       float out_scale, hipStream_t s) {
    if (ks < 1 || ks > 32 || T > Tin) return hipErrorInvalidValue;
    if (ks == 5 && (Tin & 3) == 0 && (reinterpret_cast<uintptr_t>(H) & 15) == 0) {
        prof::Scope ps(s, "tail", 2.0 * B * C * ks * (double)T, 4.0 * B * ((double)C * Tin + 2.0 * T));
        hipLaunchKernelGGL(tail5_vec_kernel, dim3((T + 1007) / 1008, B), dim3(NT_), 0, s, H, w, bias, x, out, C, Tin, T, pre_scale, out_scale);
        return hipGetLastError();
    }
    const int per_block = 4 * (64 - (ks - 1));
    dim3 grid((T + per_block - 1) / per_block, B);
    prof::Scope ps(s, "tail", 2.0 * B * C * ks * (double)T, 4.0 * B * ((double)C * Tin + 2.0 * T));
    hipLaunchKernelGGL(tail_kernel, grid, dim3(NT_), 0, s, H, w, bias, x, out, C, Tin, T, ks,
                       pre_scale, out_scale);
    return hipGetLastError();
}

hipError_t launch_head(const HeadArgs& a, hipStream_t s) {
    using T = Tile<64, 64, 2, 2>;
    dim3 grid(a.nb, a.B);
    static const std::string name = tile_name<T>("head");
    prof::Scope ps(s, name.c_str(), 2.0 * a.B * a.D * (double)a.nb * a.hop * a.Fr,
                   4.0 * a.B * ((double)a.D * a.Fr + (a.logits ? (double)a.nb * a.T : 0.0)));
    hipLaunchKernelGGL(head_kernel<T>, grid, dim3(NT_), stage_bytes<T>(), s, a);
    return hipGetLastError();
}

hipError_t launch_film(const FilmArgs& a, hipStream_t s) {
    if (a.E > NT_ || a.E < 1) return hipErrorInvalidValue;
    prof::Scope ps(s, "film", 2.0 * a.B * a.E * (a.msg_dim + a.n_layers * a.E + a.n_out), 4.0 * a.B * (a.msg_dim + a.n_out));
    hipLaunchKernelGGL(film_kernel, dim3(a.B), dim3(NT_), 0, s, a);
    return hipGetLastError();
}

}  // namespace wv
