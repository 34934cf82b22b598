// Do f16 MFMAs (v_mfma_f32_32x32x16_f16) and ordinary VALU work overlap on one SIMD of gfx950?  (tools/coissue.hip asked the same of
// the f32 matrix instruction: they serialise.)  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a bare MFMA loop, waves
// 4-7 (their SIMD partners) a loop of independent v_fma_f32 / v_exp_f32.  Also: ONE wave per SIMD issuing both kinds interleaved.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/coissue16 tools/coissue16.hip && tools/bin/coissue16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: both, 1: MFMA waves only, 2: VALU waves only, 3: one wave per SIMD does both (interleaved)
__global__ __launch_bounds__(512) void k(float* out, int iters, int valu_per_iter, int kind) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (MODE == 3) {
        if (wave >= 4) return;
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        h16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + threadIdx.x * 1e-3f); b[i] = (_Float16)(0.25f - threadIdx.x * 1e-3f); }
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float c0 = 0.999f, c1 = 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                for (int n = 0; n < valu_per_iter; ++n) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = kind ? __builtin_amdgcn_exp2f(v[i]) * c0 : fmaf(v[i], c0, c1);
                }
            }
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if (wave < 4) {
        if (MODE == 2) return;
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        h16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + threadIdx.x * 1e-3f); b[i] = (_Float16)(0.25f - threadIdx.x * 1e-3f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        if (MODE == 1) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float c0 = 0.999f, c1 = 1e-3f;
        for (int it = 0; it < iters; ++it) {
            for (int n = 0; n < valu_per_iter; ++n) {          // valu_per_iter x 8 independent VALU ops per 32 MFMAs
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = kind ? __builtin_amdgcn_exp2f(v[i]) * c0 : fmaf(v[i], c0, c1);
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> static float run(float* out, int iters, int vpi, int kind) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, vpi, kind);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    return best;
}
int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 512 * 4));
    const int iters = 4000;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters, 0, 0);
    CHECK(hipDeviceSynchronize());
    const float tm = run<1>(out, iters, 0, 0);
    printf("f16 MFMA waves alone: %.3f ms  (%.1f TFLOP/s; %.1f cycles per MFMA at 2.4 GHz)\n", tm, 256.0 * 4 * iters * 32 * 32768.0 / tm / 1e9,
           tm * 1e-3 * 2.4e9 / (iters * 32.0));
    for (int kind = 0; kind < 2; ++kind)
        for (int vpi : {1, 2, 4, 8, 16, 32}) {
            const float tv = run<2>(out, iters, vpi, kind), tb = run<0>(out, iters, vpi, kind);
            printf("%s: %3d VALU per 32 MFMAs (%.2f per MFMA): VALU alone %.3f ms, both %.3f ms  -> serial would be %.3f, overlap would be %.3f\n",
                   kind ? "v_exp+v_mul" : "v_fma", vpi * 8 * (kind ? 2 : 1), vpi * 8 * (kind ? 2 : 1) / 32.0, tv, tb, tm + tv, tm > tv ? tm : tv);
        }
    for (int kind = 0; kind < 2; ++kind)
        for (int vpi : {1, 2, 4}) {
            const float tb = run<3>(out, iters, vpi, kind);
            printf("one wave per SIMD, interleaved, %s: %3d VALU per 4 MFMAs: %.3f ms (MFMA alone %.3f)\n", kind ? "v_exp+v_mul" : "v_fma", vpi * 8 * (kind ? 2 : 1), tb, tm);
        }
    return 0;
}
