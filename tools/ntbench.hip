// Microbenchmark of the time-contracting dW GEMM (waveverify_amd/csrc/wv_train.hip gemm_nt_kernel) with ablations:
//   MODE 0 = full kernel, 1 = no matrix work (loads + LDS commit only), 2 = no global loads (LDS + MFMA only)
//   hipcc --offload-arch=gfx950 -O3 -o ntbench tools/ntbench.hip && ./ntbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RB, int MODE, int XCD = 0>
__device__ __forceinline__ void nt_body(const float* __restrict__ dh, const float* __restrict__ x, float* __restrict__ part, float s, int elu,
                                                 int B, int M, int K, int T, int TC, int Sx) {
    constexpr int TS = 32, LD = TS + 1, ROWS = 64 * RB, NV = ROWS * TS / 4 / 256;
    __shared__ float As[ROWS * LD], Bs[ROWS * LD];
    int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z, S = gridDim.z;
    if (XCD) {                                   // 1-D grid: the tiles of one split sit on one XCD (ids L, L + 8, ... share an L2)
        const int gx = (M + ROWS - 1) / ROWS, gy = (K + ROWS - 1) / ROWS, nt = gx * gy;
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3, tile = slot % nt;
        S = Sx; split = (slot / nt) * 8 + xcd;
        if (split >= S) return;
        bx = tile % gx; by = tile / gx;
    }
    const int m0 = bx * ROWS, k0 = by * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wk = wave & 1;
    const int i31 = lane & 31, hh = lane >> 5;
    const bool vec = (T & 3) == 0;
    f32x16 acc[RB][RB];
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < RB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NV], rb[NV];
    auto fetch = [&](const float* dhb, const float* xb, int t0, int te) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256, row = idx / (TS / 4), t = t0 + (idx % (TS / 4)) * 4;
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 2) { a4[0] = (float)t; b4[0] = (float)row; }
            else if (vec && t + 3 < te) {
                if (m0 + row < M) a4 = *reinterpret_cast<const f32x4*>(dhb + (size_t)(m0 + row) * T + t);
                if (k0 + row < K) b4 = *reinterpret_cast<const f32x4*>(xb + (size_t)(k0 + row) * T + t);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (t + e < te) {
                        if (m0 + row < M) a4[e] = dhb[(size_t)(m0 + row) * T + t + e];
                        if (k0 + row < K) b4[e] = xb[(size_t)(k0 + row) * T + t + e];
                    }
            }
            ra[v] = a4; rb[v] = b4;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256, row = idx / (TS / 4), c = (idx % (TS / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[row * LD + c + e] = ra[v][e];
                const float xv = rb[v][e] * s;
                Bs[row * LD + c + e] = (!elu || xv > 0.f) ? xv : (__expf(xv) - 1.f);
            }
        }
    };
    const int nch = (T + TC - 1) / TC;
    for (int item = split; item < B * nch; item += S) {
        const int b = item / nch, tb = (item - b * nch) * TC, te = min(T, tb + TC);
        const float* dhb = dh + (size_t)b * M * T;
        const float* xb = x + (size_t)b * K * T;
        fetch(dhb, xb, tb, te);
        for (int t0 = tb; t0 < te; t0 += TS) {
            __syncthreads();
            commit();
            __syncthreads();
            if (t0 + TS < te) fetch(dhb, xb, t0 + TS, te);
            if (MODE == 1) { acc[0][0][0] += As[lane * LD + wave] * Bs[lane * LD + wave]; continue; }
#pragma unroll
            for (int kk = 0; kk < TS; kk += 2) {
                float av[RB], bv[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) av[i] = As[(32 * (RB * wm + i) + i31) * LD + kk + hh];
#pragma unroll
                for (int j = 0; j < RB; ++j) bv[j] = Bs[(32 * (RB * wk + j) + i31) * LD + kk + hh];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < RB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* P = part + (size_t)split * M * K;
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < RB; ++j)
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * (RB * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * hh, k = k0 + 32 * (RB * wk + j) + i31;
                if (m < M && k < K) P[(size_t)m * K + k] = acc[i][j][r];
            }
}


template <int RB, int MODE, int XCD = 0>
__global__ __launch_bounds__(256) void nt_kernel(const float* __restrict__ dh, const float* __restrict__ x, float* __restrict__ part, float s, int elu,
                                                 int B, int M, int K, int T, int TC, int Sx) { nt_body<RB, MODE, XCD>(dh, x, part, s, elu, B, M, K, T, TC, Sx); }
#define WRAP(NAME, RB, W) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void NAME(const float* __restrict__ dh, const float* __restrict__ x, \
    float* __restrict__ part, float s, int elu, int B, int M, int K, int T, int TC, int Sx) { nt_body<RB, 0, 0>(dh, x, part, s, elu, B, M, K, T, TC, Sx); }
WRAP(nt_r2_w3, 2, 3)
WRAP(nt_r2_w4, 2, 4)
WRAP(nt_r1_w5, 1, 5)
WRAP(nt_r1_w6, 1, 6)
WRAP(nt_r1_w8, 1, 8)

// branch-free fetch: rows clamped (rows past M / K only feed accumulator rows that are never stored), time overrun clamped and the dh
// operand zeroed at commit -- all loads of a step issue back to back, one wait at the commit
template <int RB>
__device__ __forceinline__ void nt3_body(const float* __restrict__ dh, const float* __restrict__ x, float* __restrict__ part, float s, int elu,
                                         int B, int M, int K, int T, int TC) {
    constexpr int TS = 32, LD = TS + 1, ROWS = 64 * RB, NV = ROWS * TS / 4 / 256;
    __shared__ float As[ROWS * LD], Bs[ROWS * LD];
    const int m0 = blockIdx.x * ROWS, k0 = blockIdx.y * ROWS, split = blockIdx.z, S = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wk = wave & 1;
    const int i31 = lane & 31, hh = lane >> 5;
    const int r0 = tid >> 3, tq = (tid & 7) * 4;
    f32x16 acc[RB][RB];
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < RB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    unsigned offa[NV], offb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        offa[v] = (unsigned)min(m0 + r0 + 32 * v, M - 1) * (unsigned)T;
        offb[v] = (unsigned)min(k0 + r0 + 32 * v, K - 1) * (unsigned)T;
    }
    f32x4 ra[NV], rb[NV];
    auto fetch = [&](const float* dhb, const float* xb, int t0, int tb, int te) {
        const int t = t0 + tq, tc = t < te ? t : tb;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            ra[v] = *reinterpret_cast<const f32x4*>(dhb + offa[v] + tc);
            rb[v] = *reinterpret_cast<const f32x4*>(xb + offb[v] + tc);
        }
    };
    auto commit = [&](int t0, int te) {
        const float keep = (t0 + tq < te) ? 1.f : 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int row = r0 + 32 * v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[row * LD + tq + e] = ra[v][e] * keep;
                const float xv = rb[v][e] * s;
                Bs[row * LD + tq + e] = (!elu || xv > 0.f) ? xv : (__expf(xv) - 1.f);
            }
        }
    };
    const int nch = (T + TC - 1) / TC;
    for (int item = split; item < B * nch; item += S) {
        const int b = item / nch, tb = (item - b * nch) * TC, te = min(T, tb + TC);
        const float* dhb = dh + (size_t)b * M * T;
        const float* xb = x + (size_t)b * K * T;
        fetch(dhb, xb, tb, tb, te);
        for (int t0 = tb; t0 < te; t0 += TS) {
            __syncthreads();
            commit(t0, te);
            __syncthreads();
            if (t0 + TS < te) fetch(dhb, xb, t0 + TS, tb, te);
#pragma unroll
            for (int kk = 0; kk < TS; kk += 2) {
                float av[RB], bv[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) av[i] = As[(32 * (RB * wm + i) + i31) * LD + kk + hh];
#pragma unroll
                for (int j = 0; j < RB; ++j) bv[j] = Bs[(32 * (RB * wk + j) + i31) * LD + kk + hh];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < RB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* P = part + (size_t)split * M * K;
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < RB; ++j)
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * (RB * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * hh, k = k0 + 32 * (RB * wk + j) + i31;
                if (m < M && k < K) P[(size_t)m * K + k] = acc[i][j][r];
            }
}
#define WRAP3(NAME, RB, W) __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void NAME(const float* __restrict__ dh, const float* __restrict__ x, \
    float* __restrict__ part, float s, int elu, int B, int M, int K, int T, int TC, int Sx) { nt3_body<RB>(dh, x, part, s, elu, B, M, K, T, TC); }
WRAP3(nt3_r2_w3, 2, 3)
WRAP3(nt3_r2_w4, 2, 4)
WRAP3(nt3_r1_w4, 1, 4)
WRAP3(nt3_r1_w6, 1, 6)
WRAP3(nt3_r1_w8, 1, 8)

template <typename KF>
static float runk(KF kf, int RB, const float* dh, const float* x, float* part, int B, int M, int K, int T, int S, int TC) {
    const int R = 64 * RB;
    dim3 g((M + R - 1) / R, (K + R - 1) / R, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kf, g, dim3(256), 0, 0, dh, x, part, 1.f, 1, B, M, K, T, TC, S);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kf, g, dim3(256), 0, 0, dh, x, part, 1.f, 1, B, M, K, T, TC, S);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}
template <int RB, int MODE, int XCD = 0>
static float run(const float* dh, const float* x, float* part, int B, int M, int K, int T, int S, int TC, int elu) {
    const int R = 64 * RB;
    dim3 g((M + R - 1) / R, (K + R - 1) / R, S);
    if (XCD) g = dim3(g.x * g.y * ((S + 7) / 8 * 8), 1, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((nt_kernel<RB, MODE, XCD>), g, dim3(256), 0, 0, dh, x, part, 1.f, elu, B, M, K, T, TC, S);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((nt_kernel<RB, MODE, XCD>), g, dim3(256), 0, 0, dh, x, part, 1.f, elu, B, M, K, T, TC, S);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}


// flat-stream variant: the (item, 32-sample step) sequence of a split is one stream, the register prefetch runs PF steps ahead of the
// matrix work (also across item boundaries)
template <int RB, int PF>
__global__ __launch_bounds__(256) void nt2_kernel(const float* __restrict__ dh, const float* __restrict__ x, float* __restrict__ part, float s, int elu,
                                                  int B, int M, int K, int T, int TC) {
    constexpr int TS = 32, LD = TS + 1, ROWS = 64 * RB, NV = ROWS * TS / 4 / 256;
    __shared__ float As[ROWS * LD], Bs[ROWS * LD];
    const int m0 = blockIdx.x * ROWS, k0 = blockIdx.y * ROWS, split = blockIdx.z, S = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wk = wave & 1;
    const int i31 = lane & 31, hh = lane >> 5;
    const bool vec = (T & 3) == 0;
    f32x16 acc[RB][RB];
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < RB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[PF][NV], rb[PF][NV];
    const int nch = (T + TC - 1) / TC, nitems = B * nch;
    int nsteps = 0;
    for (int item = split; item < nitems; item += S) {
        const int tb = (item % nch) * TC, te = min(T, tb + TC);
        nsteps += (te - tb + TS - 1) / TS;
    }
    int f_item = split, f_t0 = (split % nch) * TC;                 // fetch cursor
    auto fetch = [&](f32x4* pa, f32x4* pb) {
        if (f_item >= nitems) return;
        const int b = f_item / nch, tb = (f_item - b * nch) * TC, te = min(T, tb + TC);
        const float* dhb = dh + (size_t)b * M * T;
        const float* xb = x + (size_t)b * K * T;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256, row = idx / (TS / 4), t = f_t0 + (idx % (TS / 4)) * 4;
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if (vec && t + 3 < te) {
                if (m0 + row < M) a4 = *reinterpret_cast<const f32x4*>(dhb + (size_t)(m0 + row) * T + t);
                if (k0 + row < K) b4 = *reinterpret_cast<const f32x4*>(xb + (size_t)(k0 + row) * T + t);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (t + e < te) {
                        if (m0 + row < M) a4[e] = dhb[(size_t)(m0 + row) * T + t + e];
                        if (k0 + row < K) b4[e] = xb[(size_t)(k0 + row) * T + t + e];
                    }
            }
            pa[v] = a4; pb[v] = b4;
        }
        f_t0 += TS;
        if (f_t0 >= te) { f_item += S; f_t0 = (f_item % nch) * TC; }
    };
    auto step = [&](f32x4* pa, f32x4* pb) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256, row = idx / (TS / 4), c = (idx % (TS / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[row * LD + c + e] = pa[v][e];
                const float xv = pb[v][e] * s;
                Bs[row * LD + c + e] = (!elu || xv > 0.f) ? xv : (__expf(xv) - 1.f);
            }
        }
        __syncthreads();
        fetch(pa, pb);
#pragma unroll
        for (int kk = 0; kk < TS; kk += 2) {
            float av[RB], bv[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) av[i] = As[(32 * (RB * wm + i) + i31) * LD + kk + hh];
#pragma unroll
            for (int j = 0; j < RB; ++j) bv[j] = Bs[(32 * (RB * wk + j) + i31) * LD + kk + hh];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < RB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(ra[p], rb[p]);
    for (int n = 0; n < nsteps; n += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (n + p < nsteps) step(ra[p], rb[p]);
    }
    float* P = part + (size_t)split * M * K;
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < RB; ++j)
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * (RB * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * hh, k = k0 + 32 * (RB * wk + j) + i31;
                if (m < M && k < K) P[(size_t)m * K + k] = acc[i][j][r];
            }
}

template <int RB, int PF>
static float run2(const float* dh, const float* x, float* part, int B, int M, int K, int T, int S, int TC, int elu) {
    const int R = 64 * RB;
    dim3 g((M + R - 1) / R, (K + R - 1) / R, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((nt2_kernel<RB, PF>), g, dim3(256), 0, 0, dh, x, part, 1.f, elu, B, M, K, T, TC);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((nt2_kernel<RB, PF>), g, dim3(256), 0, 0, dh, x, part, 1.f, elu, B, M, K, T, TC);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    struct Case { int M, K, T, RB, S, TC; };
    const int B = 64;
    std::vector<Case> cases = {{256, 256, 2000, 0, 0, 512}, {128, 128, 8000, 0, 0, 512}, {384, 384, 2000, 0, 0, 512},
                               {512, 512, 400, 0, 0, 512}, {768, 768, 400, 0, 0, 512}, {192, 192, 8000, 0, 0, 512}, {1024, 1024, 50, 0, 0, 512}, {1536, 1536, 50, 0, 0, 512}};
    size_t nmax = (size_t)B * 512 * 16000;
    float *dh, *x, *part;
    hipMalloc(&dh, nmax * 4); hipMalloc(&x, nmax * 4); hipMalloc(&part, (size_t)256 << 20);
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    for (size_t o = 0; o < nmax; o += h.size()) { hipMemcpy(dh + o, h.data(), std::min(h.size(), nmax - o) * 4, hipMemcpyHostToDevice); hipMemcpy(x + o, h.data(), std::min(h.size(), nmax - o) * 4, hipMemcpyHostToDevice); }
    for (auto c : cases) {
        const int items = B * ((c.T + 511) / 512);
        auto Sof = [&](int R, int target) { const int tiles = ((c.M + R - 1) / R) * ((c.K + R - 1) / R); return std::max(1, std::min(items, target / tiles)); };
        const double gf = 2.0 * c.M * c.K * c.T * B * 1e-9;
        printf("M=%d K=%d T=%d:", c.M, c.K, c.T);
        float t;
        t = run<2, 0>(dh, x, part, B, c.M, c.K, c.T, Sof(128, 1024), 512, 1); printf("  r2 %.0f (%.0f)", t, gf / t * 1e3);
        t = run<1, 0>(dh, x, part, B, c.M, c.K, c.T, Sof(64, 1024), 512, 1); printf("  r1 %.0f (%.0f)", t, gf / t * 1e3);
        t = runk(nt3_r2_w3, 2, dh, x, part, B, c.M, c.K, c.T, Sof(128, 1024), 512); printf("  n3r2w3 %.0f (%.0f)", t, gf / t * 1e3);
        t = runk(nt3_r2_w4, 2, dh, x, part, B, c.M, c.K, c.T, Sof(128, 1024), 512); printf("  n3r2w4 %.0f (%.0f)", t, gf / t * 1e3);
        t = runk(nt3_r1_w4, 1, dh, x, part, B, c.M, c.K, c.T, Sof(64, 1024), 512); printf("  n3r1w4 %.0f (%.0f)", t, gf / t * 1e3);
        t = runk(nt3_r1_w6, 1, dh, x, part, B, c.M, c.K, c.T, Sof(64, 2048), 512); printf("  n3r1w6 %.0f (%.0f)", t, gf / t * 1e3);
        t = runk(nt3_r1_w8, 1, dh, x, part, B, c.M, c.K, c.T, Sof(64, 2048), 512); printf("  n3r1w8 %.0f (%.0f)\n", t, gf / t * 1e3);
    }
    return 0;
}
