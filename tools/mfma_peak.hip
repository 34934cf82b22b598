// Ceiling microbenchmark for the exact-f32 matrix instruction on MI355X (gfx950).
//
// Settles what K1's GEMM core can reach at best: a loop of v_mfma_f32_32x32x2_f32 on random data with
//   V0  operands held in registers (no LDS, no barrier, no memory),
//   V1  + B fragments re-read from LDS with ds_read_b128 (K1's k-inner layout: one read feeds 4 MFMAs),
//   V2  + one workgroup barrier per 16-deep chunk (K1's pipeline step),
//   V3  + A fragments streamed from global memory (L2-resident weights) one chunk ahead,
//   V4  + the B operand really staged: 2 x 16-byte activation loads per thread and chunk from a 0.8 GB
//       [B][K][T] array, scale -> ELU and four ds_write_b64 AFTER the chunk's MFMAs (K1 round-1 order),
//   V5  the same work software-pipelined: loads two chunks ahead (second register set), the commit of
//       chunk c+1 placed between the MFMAs of chunk c (sched_group_barrier interleave),
//   V6  V4 + tile turnover: every `chunks` chunks a cold prologue (first loads not prefetched) and a
//       stencil-like epilogue (accumulators through wave-private LDS strips, residual load, 16-byte
//       stores of the 32 x 128 strip) -- a model of one K1 workgroup per tile,
//   V7  V5 + the same tile turnover,
//   V8  the round-2 core: BOTH operands by LDS-DMA (global_load_lds_dwordx4, no staging registers, no
//       VALU), B kept in its natural [k][t] layout: one ds_read_b128 = 4 consecutive columns of one k row
//       feeds 4 MFMAs whose column tiles are INTERLEAVED (tile e holds columns 4j+e), two LDS stages,
//   V9  V8 with the B operand through registers (scale -> ELU, two ds_write_b128 in natural layout),
//   V10 V8 + tile turnover with the round-2 epilogue: a lane already holds 4 consecutive columns of a
//       row (acc[0..3][r]), the stencil's right neighbours come by DPP wave_shl:1 -- no LDS strips,
//   V11 V9 + the same tile turnover,
//   V12/V13  ablations of V4: commit without ELU / without the global B loads,
// at 1..4 waves per SIMD (256-thread workgroups, 1..4 workgroups per CU), NT = 4 column tiles per
// wave (64 accumulator registers, K1's 32 x 128 row strip).  Every run lasts >= 1 ms.  The in-kernel
// clock (s_memtime / s_memrealtime) is reported beside the rate (MI355X_MICROARCH.md, DVFS item 6).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip && tools/bin/mfma_peak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int NT = 4;        // column tiles per wave
constexpr int BN = 32 * NT;  // columns of the B stage
constexpr int KQ = 4;        // float4 fragments along k per 16-deep chunk
constexpr int XT = 2048, XB = 256;   // activation array [XB][K][XT]

__device__ __forceinline__ int q_slot(int n) { return (n & ~3) | ((n & 3) ^ ((n >> 3) & 3)); }
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : (__expf(x) - 1.f); }

struct Args {
    const f32x4* wq; const f32x4* bsrc; const float* X; float* Y; const float* R;
    float* out; unsigned long long* clk; int chunks, tiles, Mp, K;
};

template <int MODE_, int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_loop(Args p) {
    constexpr int MODE = MODE_ >= 12 ? 4 : MODE_;
    constexpr bool NO_ELU = MODE_ == 12, NO_LOAD = MODE_ == 13;
    constexpr int HLD = BN + 4;
    __shared__ __attribute__((aligned(16))) f32x4 Bs[2 * KQ * BN];
    __shared__ __attribute__((aligned(16))) float strips[4 * 4 * HLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i31 = lane & 31;
    const int chunks = p.chunks, Mp = p.Mp;
    if (MODE < 4) {   // B stage: random values, written once (V1..V3 re-read them every chunk)
        for (int i = tid; i < 2 * KQ * BN; i += 256) Bs[i] = p.bsrc[(blockIdx.x % 64) * 2 * KQ * BN + i];
        __syncthreads();
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const f32x4* wa = p.wq + 32 * wave + i31;
    f32x4 a0 = wa[(size_t)h * Mp], a1 = wa[(size_t)(h + 2) * Mp];
    f32x4 b0[NT], b1[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { b0[j] = Bs[h * BN + 32 * j + i31]; b1[j] = Bs[(h + 2) * BN + 32 * j + i31]; }
    // B staging map of K1: thread = 2(k) x 4(t) micro-tile
    const int cg = tid % (BN / 4), kp = tid / (BN / 4);          // kp 0..7: rows 2kp, 2kp+1 of the chunk
    int bslot[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bslot[j] = MODE >= 4 ? q_slot(32 * j + i31) : 32 * j + i31;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    for (int t = 0; t < p.tiles; ++t) {
        // tile -> (clip, window): 3 consecutive workgroups share a window (three m-tiles of one layer)
        const int tile = (blockIdx.x / 3) + t * (gridDim.x / 3 + 1);
        const int b = tile % XB, tt = (tile / XB) % (XT / BN);
        const float* Xb = p.X + ((size_t)b * p.K) * XT + tt * BN + 4 * cg;
        f32x4 ra[2] = {{0.5f, 0.25f, -0.5f, 1.f}, {0.1f, -0.2f, 0.3f, 0.4f}}, rb[2];   // raw sets (V5/V7 use both)
        auto fetchB = [&](int c, f32x4 (&r)[2]) {
            if (NO_LOAD) { asm volatile("" : "+v"(r[0]), "+v"(r[1])); return; }
            r[0] = *reinterpret_cast<const f32x4*>(Xb + (size_t)(c * 16 + 2 * kp) * XT);
            r[1] = *reinterpret_cast<const f32x4*>(Xb + (size_t)(c * 16 + 2 * kp + 1) * XT);
        };
        auto commitB = [&](int c, const f32x4 (&r)[2]) {
            float* Bf = reinterpret_cast<float*>(Bs + (c & 1) * KQ * BN);
            const int kq = kp >> 1, kh = kp & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 v{NO_ELU ? r[0][j] : elu1(0.87f * r[0][j]), NO_ELU ? r[1][j] : elu1(0.87f * r[1][j])};
                *reinterpret_cast<f32x2*>(Bf + ((size_t)(kq * BN + q_slot(4 * cg + j)) * 4 + 2 * kh)) = v;
            }
        };
        if (MODE >= 4) {
            if (MODE >= 6 || t == 0) {                            // cold prologue
                fetchB(0, ra);
                commitB(0, ra);
                if (MODE == 5 || MODE == 7) fetchB(1 % chunks, ra);
                __syncthreads();
            }
        }
        f32x4 res[4];
        if (MODE >= 6) {                                          // first residual rows in flight
#pragma unroll
            for (int r = 0; r < 4; ++r)
                res[r] = *reinterpret_cast<const f32x4*>(p.R + ((size_t)b * 128 + 32 * wave + r) * XT + tt * BN + 4 * i31 + 0 * h);
        }
        for (int c = 0; c < chunks; ++c) {
            f32x4 an0 = a0, an1 = a1;
            if (MODE >= 3) {
                const int cc = (c + 1) % chunks;
                an0 = wa[(size_t)(cc * KQ + h) * Mp];
                an1 = wa[(size_t)(cc * KQ + h + 2) * Mp];
            }
            if (MODE == 4 || MODE == 6) { if (c + 1 < chunks) fetchB(c + 1, ra); }
            if (MODE == 5 || MODE == 7) { if (c + 2 < chunks) { if (c & 1) fetchB(c + 2, ra); else fetchB(c + 2, rb); } }
            if (MODE >= 1) {
                const f32x4* S = Bs + (c & 1) * KQ * BN;
#pragma unroll
                for (int j = 0; j < NT; ++j) { b0[j] = S[h * BN + bslot[j]]; b1[j] = S[(h + 2) * BN + bslot[j]]; }
            }
#define STEP(AV, BQ, COMP)                                                                         \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                 \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BQ[j].COMP, acc[j], 0, 0, 0);
            if (MODE == 5 || MODE == 7) {
                // commit of chunk c+1 (loaded during chunk c-1) between the MFMAs of chunk c
                STEP(a0.x, b0, x) STEP(a0.y, b0, y)
                if (c + 1 < chunks) { if (c & 1) commitB(c + 1, rb); else commitB(c + 1, ra); }
                STEP(a0.z, b0, z) STEP(a0.w, b0, w)
                STEP(a1.x, b1, x) STEP(a1.y, b1, y) STEP(a1.z, b1, z) STEP(a1.w, b1, w)
#pragma unroll
                for (int g = 0; g < 8; ++g) {                     // 1 MFMA : 8 VALU : 1 DS write, then the rest
                    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 7, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            } else {
                STEP(a0.x, b0, x) STEP(a0.y, b0, y) STEP(a0.z, b0, z) STEP(a0.w, b0, w)
                STEP(a1.x, b1, x) STEP(a1.y, b1, y) STEP(a1.z, b1, z) STEP(a1.w, b1, w)
                if ((MODE == 4 || MODE == 6) && c + 1 < chunks) commitB(c + 1, ra);
            }
#undef STEP
            if (MODE >= 3) { a0 = an0; a1 = an1; }
            if (MODE >= 2) __syncthreads();
            if (MODE == 0) asm volatile("" : "+v"(a0), "+v"(a1));   // keep the loop from collapsing
        }
        if (MODE >= 6) {
            // epilogue model: 16 steps of 2 rows through a double-buffered wave-private strip,
            // 5-tap stencil with time on the lanes, residual add, 16-byte stores
            float* Hw = strips + wave * 4 * HLD;
            const int q = lane & 31, o = 4 * q;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* strip = Hw + (r & 1) * 2 * HLD;
#pragma unroll
                for (int j = 0; j < NT; ++j) strip[h * HLD + 32 * j + q] = acc[j][r];
                const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                const f32x4 rr = res[r & 3];
                if (r + 4 < 16)
                    res[r & 3] = *reinterpret_cast<const f32x4*>(p.R + ((size_t)b * 128 + row) * XT + tt * BN + o);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(strip + h * HLD + o);
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(strip + h * HLD + o + 4);
                const float hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.1f;
#pragma unroll
                    for (int i = 0; i < 5; ++i) v = fmaf(0.2f + 0.1f * i, hh[e + i], v);
                    y[e] = fmaf(v, 0.5f, rr[e]);
                }
                if (q < 31) *reinterpret_cast<f32x4*>(p.Y + ((size_t)b * 128 + row) * XT + tt * BN + o) = y;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            __syncthreads();
        }
    }
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.clk[2 * blockIdx.x] = t1 - t0;
        p.clk[2 * blockIdx.x + 1] = r1 - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    p.out[(size_t)blockIdx.x * 256 + tid] = s;
}


__device__ __forceinline__ float dpp_next_lane(float v) {      // lane i <- lane i+1 (wave_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// MODE 8..11: t-inner B layout.  LDS stage = A fragments [KQ][128] f32x4 + B rows [16][128] floats.
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_loop2(Args p) {
    constexpr int STAGE4 = KQ * 128 + 16 * 32;                  // f32x4 per stage (A 512 + B 512)
    __shared__ __attribute__((aligned(16))) f32x4 smem[2 * STAGE4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i31 = lane & 31;
    const int chunks = p.chunks, Mp = p.Mp;
    constexpr bool REGB = (MODE == 9 || MODE == 11), TILES = MODE >= 10;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int cg = tid % 32, kp = tid / 32;                      // register path: rows 2kp, 2kp+1, columns 4cg..
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    for (int t = 0; t < p.tiles; ++t) {
        const int tile = (blockIdx.x / 3) + t * (gridDim.x / 3 + 1);
        const int b = tile % XB, tt = (tile / XB) % (XT / BN);
        const float* Xw = p.X + ((size_t)b * p.K) * XT + tt * BN;
        auto dma = [&](int c, int st) {
            f32x4* S = smem + st * STAGE4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = (2 * wave + i) * 64 + lane;
                const f32x4* src = p.wq + (size_t)(c * KQ + idx / 128) * Mp + (idx % 128);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(S + (2 * wave + i) * 64), 16, 0, 0);
            }
            if (!REGB) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = (2 * wave + i) * 64 + lane;
                    const float* src = Xw + (size_t)(c * 16 + idx / 32) * XT + 4 * (idx % 32);
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(S + KQ * 128 + (2 * wave + i) * 64), 16, 0, 0);
                }
            }
        };
        f32x4 ra[2];
        auto fetchB = [&](int c) {
            ra[0] = *reinterpret_cast<const f32x4*>(Xw + (size_t)(c * 16 + 2 * kp) * XT + 4 * cg);
            ra[1] = *reinterpret_cast<const f32x4*>(Xw + (size_t)(c * 16 + 2 * kp + 1) * XT + 4 * cg);
        };
        auto commitB = [&](int st) {
            f32x4* Bq = smem + st * STAGE4 + KQ * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = elu1(0.87f * ra[i][j]);
                Bq[(2 * kp + i) * 32 + cg] = v;
            }
        };
        if (TILES || t == 0) {
            dma(0, 0);
            if (REGB) { fetchB(0); commitB(0); }
            __syncthreads();
        }
        f32x4 res[4];
        if (TILES) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                res[r] = *reinterpret_cast<const f32x4*>(p.R + ((size_t)b * 128 + 32 * wave + r) * XT + tt * BN + 4 * i31);
        }
        for (int c = 0; c < chunks; ++c) {
            const f32x4* S = smem + (c & 1) * STAGE4;
            const int cn = TILES ? c + 1 : (c + 1) % chunks;
            if (cn < chunks) { dma(cn, (c + 1) & 1); if (REGB) fetchB(cn); }
            const f32x4 a0 = S[h * 128 + 32 * wave + i31], a1 = S[(h + 2) * 128 + 32 * wave + i31];
            const f32x4* Bq = S + KQ * 128 + i31;
#define STEP2(AV, ROW)                                                                             \
    { const f32x4 bv = Bq[(ROW) * 32];                                                             \
      _Pragma("unroll") for (int e = 0; e < NT; ++e)                                               \
          acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, bv[e], acc[e], 0, 0, 0); }
            STEP2(a0.x, 4 * h + 0) STEP2(a0.y, 4 * h + 1) STEP2(a0.z, 4 * h + 2) STEP2(a0.w, 4 * h + 3)
            STEP2(a1.x, 8 + 4 * h + 0) STEP2(a1.y, 8 + 4 * h + 1) STEP2(a1.z, 8 + 4 * h + 2) STEP2(a1.w, 8 + 4 * h + 3)
#undef STEP2
            if (REGB && cn < chunks) commitB((c + 1) & 1);
            __syncthreads();
        }
        if (TILES) {
            // round-2 epilogue: lane holds columns 4q..4q+3 of a row; right neighbours by DPP
            const int q = i31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                const f32x4 rr = res[r & 3];
                if (r + 4 < 16)
                    res[r & 3] = *reinterpret_cast<const f32x4*>(p.R + ((size_t)b * 128 + row) * XT + tt * BN + 4 * q);
                float hh[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { hh[e] = acc[e][r]; hh[4 + e] = dpp_next_lane(acc[e][r]); }
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.1f;
#pragma unroll
                    for (int i = 0; i < 5; ++i) v = fmaf(0.2f + 0.1f * i, hh[e + i], v);
                    y[e] = fmaf(v, 0.5f, rr[e]);
                }
                if (q < 31) *reinterpret_cast<f32x4*>(p.Y + ((size_t)b * 128 + row) * XT + tt * BN + 4 * q) = y;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
    }
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.clk[2 * blockIdx.x] = t1 - t0;
        p.clk[2 * blockIdx.x + 1] = r1 - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    p.out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE, int WPS>
static void run(Args a, int chunks, int total_chunks) {
    a.chunks = chunks; a.tiles = total_chunks / chunks;
    const int grid = 256 * WPS;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        if constexpr (MODE >= 8 && MODE <= 11) hipLaunchKernelGGL((mfma_loop2<MODE, WPS>), dim3(grid), dim3(256), 0, 0, a);
        else hipLaunchKernelGGL((mfma_loop<MODE, WPS>), dim3(grid), dim3(256), 0, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(2 * grid);
    CHECK(hipMemcpy(h.data(), a.clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(grid);
    for (int i = 0; i < grid; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;   // 100 MHz real-time clock
    std::sort(ghz.begin(), ghz.end());
    const double flop = (double)grid * 4 /*waves*/ * a.tiles * chunks * 8.0 * NT * 4096.0;
    const char* names[] = {"registers only", "+ds_read_b128 B", "+barrier/chunk", "+A from global", "+B staged (r1 order)",
                           "+B staged (pipelined)", "V4 + tile turnover", "V5 + tile turnover",
                           "DMA A+B, t-inner", "DMA A, reg B (ELU)", "V8 + tiles, DPP epi", "V9 + tiles, DPP epi",
                           "V4 without ELU", "V4 without B loads"};
    printf("V%d %-22s K/tile=%4d waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)  clock %.2f GHz\n", MODE,
           names[MODE], chunks * 16, WPS, best, flop / best / 1e9, 100.0 * flop / best / 1e9 / 157.3, ghz[grid / 2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int Mp = 128, K = 768, Kq = K / 4;
    const bool quick = argc > 1;
    std::vector<float> hw((size_t)Kq * Mp * 4), hb((size_t)64 * 2 * KQ * BN * 4);
    srand(1);
    for (auto& v : hw) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    for (auto& v : hb) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    Args a{};
    f32x4 *wq, *bsrc; float *X, *Y, *R;
    CHECK(hipMalloc(&wq, hw.size() * 4)); CHECK(hipMalloc(&bsrc, hb.size() * 4));
    CHECK(hipMalloc(&a.out, (size_t)1024 * 256 * 4)); CHECK(hipMalloc(&a.clk, 2 * 1024 * 8));
    const size_t nx = (size_t)XB * K * XT, ny = (size_t)XB * 128 * XT;
    CHECK(hipMalloc(&X, nx * 4)); CHECK(hipMalloc(&Y, ny * 4)); CHECK(hipMalloc(&R, ny * 4));
    {
        std::vector<float> hx((size_t)1 << 24);
        for (auto& v : hx) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
        for (size_t o = 0; o < nx; o += hx.size()) CHECK(hipMemcpy(X + o, hx.data(), std::min(hx.size(), nx - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < ny; o += hx.size()) CHECK(hipMemcpy(R + o, hx.data(), std::min(hx.size(), ny - o) * 4, hipMemcpyHostToDevice));
    }
    CHECK(hipMemcpy(wq, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(bsrc, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    a.wq = wq; a.bsrc = bsrc; a.X = X; a.Y = Y; a.R = R; a.Mp = Mp; a.K = K;
    // a few seconds of back-to-back launches first, so the clock has settled (DVFS)
    { Args w = a; w.chunks = 48; w.tiles = 40;
      for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((mfma_loop<0, 2>), dim3(512), dim3(256), 0, 0, w); }
    CHECK(hipDeviceSynchronize());
    const int TC = 48 * 24;
#define ROW(M, CH) run<M, 1>(a, CH, TC); run<M, 2>(a, CH, TC); run<M, 3>(a, CH, TC); run<M, 4>(a, CH, TC);
    if (!quick) { ROW(0, 48) ROW(1, 48) ROW(2, 48) ROW(3, 48) ROW(4, 48) ROW(5, 48) ROW(6, 48) ROW(7, 48) ROW(6, 24) ROW(7, 24) ROW(6, 8) ROW(7, 8) }
    ROW(12, 48) ROW(13, 48)
    ROW(8, 48) ROW(9, 48)
    ROW(10, 48) ROW(11, 48) ROW(10, 24) ROW(11, 24) ROW(10, 8) ROW(11, 8)
    return 0;
}
