// Temporal augmentations of the training step on the device:
//   LocalizationAugmentation.forward  (utils/localization_augmentation.py:212-321)
//   SequenceAugmentation.forward      (utils/seq_augmentation.py:100-273)
//   AudioWatermarking._apply_augmentations (model/watermarking.py:487-519): the two, back to back.
// The reference walks the batch in Python, clip by clip and segment by segment, mutating three tensors with
// slice assignments, then flips / rolls / re-gathers all three -- and the training step moves them GPU -> CPU
// -> GPU around it (watermarking.py:540).  Everything the reference does is a per-sample SELECT over an index
// map, so here it is one bandwidth-bound pass:
//   * the host draws the plan with the reference's own RNG call order (waveverify_amd/augment.py) -- a table
//     plan[B][nseg] of codes: 0 keep, 1 revert to the original, 2 zero, 3 + j take clip j's original;
//   * the sequence augmentation is a map t -> ts on the time axis (reverse, circular shift, segment
//     permutation, chunk swap), composed in front of the table lookup;
//   * one thread produces 4 consecutive output samples of all three outputs (watermarked', original', mask).
// Pure copies and constants: results are bit-identical to the reference's.
#include <hip/hip_runtime.h>

#include "../../include/waveverify_hip.h"
#include "wv_kernels.h"

namespace wv {

struct SeqMap {              // out[t] = in[src(t)]
    int mode;                // WV_SEQ_*
    int a, b, c;             // roll: a = shift;  permutation: a = segment size;  chunk swap: a = start 1, b = start 2, c = size
    const int* perm;         // permutation: source segment of every output segment
    int T;                   // input length
    __device__ __forceinline__ int src(int t) const {
        switch (mode) {
            case WV_SEQ_REVERSE: return T - 1 - t;                                 // torch.flip
            case WV_SEQ_ROLL: { const int s = t - a; return s < 0 ? s + T : s; }     // torch.roll(shifts=a), 0 < a < T
            case WV_SEQ_PERMUTE: { const int g = t / a; return perm[g] * a + (t - g * a); }
            case WV_SEQ_CHUNK_SWAP:
                if (t >= a && t < a + c) return b + (t - a);
                if (t >= b && t < b + c) return a + (t - b);
                return t;
            default: return t;
        }
    }
};

struct AugArgs {
    const float* orig; const float* wm; const int* plan;
    float* wm_out; float* orig_out; float* mask_out;
    int B, C, T, T_out, nseg, seg_len;
    SeqMap sm;
};

__global__ __launch_bounds__(256) void aug_kernel(AugArgs p) {
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int bc = blockIdx.y, b = bc / p.C, ch = bc - b * p.C;
    if (t0 >= p.T_out) return;
    const size_t row_in = (size_t)bc * p.T, row_out = (size_t)bc * p.T_out;
    float w[4], o[4], m[4];
    int ts[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ts[e] = p.sm.src(min(t0 + e, p.T_out - 1));
    // the four sources are usually one aligned run (forward or reversed) inside one segment: 16-byte loads, one lookup
    const int lo = min(ts[0], ts[3]);
    const bool run = (p.T & 3) == 0 && (lo & 3) == 0 && ((ts[1] - ts[0] == 1 && ts[2] - ts[0] == 2 && ts[3] - ts[0] == 3) ||
                                                         (ts[0] - ts[1] == 1 && ts[0] - ts[2] == 2 && ts[0] - ts[3] == 3));
    if (run && lo / p.seg_len == (lo + 3) / p.seg_len) {
        const int code = p.plan ? p.plan[b * p.nseg + lo / p.seg_len] : 0;
        const bool rev = ts[0] > ts[3];
        const float4 xo4 = *reinterpret_cast<const float4*>(p.orig + row_in + lo);
        float4 xw4 = xo4, xu4 = xo4;
        float mk = 0.f;
        if (code == 0) { xw4 = *reinterpret_cast<const float4*>(p.wm + row_in + lo); mk = 1.f; }
        else if (code == 2) { xw4 = make_float4(0.f, 0.f, 0.f, 0.f); xu4 = xw4; }
        else if (code >= 3) { xw4 = *reinterpret_cast<const float4*>(p.orig + ((size_t)(code - 3) * p.C + ch) * p.T + lo); xu4 = xw4; }
        const float wv_[4] = {xw4.x, xw4.y, xw4.z, xw4.w}, ov_[4] = {xu4.x, xu4.y, xu4.z, xu4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { w[e] = wv_[rev ? 3 - e : e]; o[e] = ov_[rev ? 3 - e : e]; m[e] = mk; }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int code = p.plan ? p.plan[b * p.nseg + ts[e] / p.seg_len] : 0;
            const float xo = p.orig[row_in + ts[e]];
            float xw = p.wm[row_in + ts[e]], xu = xo, mk = 1.f;
            if (code == 1) { xw = xo; mk = 0.f; }                                        // revert (:128-133)
            else if (code == 2) { xw = 0.f; xu = 0.f; mk = 0.f; }                        // zeros (:151-155)
            else if (code >= 3) {                                                        // another clip's original (:157-193)
                const float xj = p.orig[((size_t)(code - 3) * p.C + ch) * p.T + ts[e]];
                xw = xj; xu = xj; mk = 0.f;
            }
            w[e] = xw; o[e] = xu; m[e] = mk;
        }
    }
    if (t0 + 3 < p.T_out && (p.T_out & 3) == 0) {
        *reinterpret_cast<float4*>(p.wm_out + row_out + t0) = make_float4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<float4*>(p.orig_out + row_out + t0) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p.mask_out + row_out + t0) = make_float4(m[0], m[1], m[2], m[3]);
    } else {
        for (int e = 0; e < 4 && t0 + e < p.T_out; ++e) {
            p.wm_out[row_out + t0 + e] = w[e]; p.orig_out[row_out + t0 + e] = o[e]; p.mask_out[row_out + t0 + e] = m[e];
        }
    }
}

struct SeqArgs {
    const float* in[3]; float* out[3];
    int rows, T, T_out;
    SeqMap sm;
};

__global__ __launch_bounds__(256) void seq_kernel(SeqArgs p) {
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int row = blockIdx.y;
    if (t0 >= p.T_out) return;
    int ts[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ts[e] = p.sm.src(min(t0 + e, p.T_out - 1));
    const bool vec = t0 + 3 < p.T_out && (p.T_out & 3) == 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!p.in[k]) continue;
        const float* x = p.in[k] + (size_t)row * p.T;
        float* y = p.out[k] + (size_t)row * p.T_out;
        const float v0 = x[ts[0]], v1 = x[ts[1]], v2 = x[ts[2]], v3 = x[ts[3]];
        if (vec) *reinterpret_cast<float4*>(y + t0) = make_float4(v0, v1, v2, v3);
        else {
            const float v[4] = {v0, v1, v2, v3};
            for (int e = 0; e < 4 && t0 + e < p.T_out; ++e) y[t0 + e] = v[e];
        }
    }
}

// Backward of wv_aug_localize_sequence towards the WATERMARKED input: the forward is a select, so the gradient of an output sample
// goes to the watermarked sample it was copied from and nowhere else (reverted / zeroed / substituted segments carry none; the
// segment permutation drops the clip's tail).  inv = the inverse of the forward's sequence map (out[t] = in[src(t)] <=> t = inv.src(ts)).
struct AugBwdArgs {
    const float* d_out; const int* plan; float* d_wm;
    int B, C, T, T_out, nseg, seg_len;
    SeqMap inv;
};
__global__ __launch_bounds__(256) void aug_bwd_kernel(AugBwdArgs p) {
    const int ts = blockIdx.x * 256 + threadIdx.x;
    const int bc = blockIdx.y, b = bc / p.C;
    if (ts >= p.T) return;
    float g = 0.f;
    if (ts < p.T_out || p.inv.mode != WV_SEQ_PERMUTE) {
        const int code = p.plan ? p.plan[b * p.nseg + ts / p.seg_len] : 0;
        if (code == 0) g = p.d_out[(size_t)bc * p.T_out + p.inv.src(ts)];
    }
    p.d_wm[(size_t)bc * p.T + ts] = g;
}

static bool seq_ok(int mode, int a, int b, int c, const int* perm, int T, int T_out) {
    switch (mode) {
        case WV_SEQ_IDENTITY: case WV_SEQ_REVERSE: return T_out == T;
        case WV_SEQ_ROLL: return T_out == T && a > 0 && a < T;
        case WV_SEQ_PERMUTE: return perm && a > 0 && T_out > 0 && T_out % a == 0 && T_out <= T;
        case WV_SEQ_CHUNK_SWAP: return T_out == T && c > 0 && a >= 0 && b >= 0 && a + c <= T && b + c <= T && (a + c <= b || b + c <= a);
        default: return false;
    }
}

}  // namespace wv

extern "C" {

int wv_aug_localize_sequence(const float* original, const float* watermarked, const int* plan, int nseg, int seg_len,
                             int seq_mode, int seq_a, int seq_b, int seq_c, const int* perm,
                             float* wm_out, float* orig_out, float* mask_out, int B, int C, int T, int T_out, void* stream) {
    if (!original || !watermarked || !wm_out || !orig_out || !mask_out || B < 1 || C < 1 || T < 1) return WV_EINVAL;
    if (plan && (seg_len < 1 || nseg != (T + seg_len - 1) / seg_len)) return WV_EINVAL;
    if (!wv::seq_ok(seq_mode, seq_a, seq_b, seq_c, perm, T, T_out)) return WV_EINVAL;
    if ((long long)B * C > 65535) return WV_EINVAL;
    wv::AugArgs a{original, watermarked, plan, wm_out, orig_out, mask_out, B, C, T, T_out, nseg, plan ? seg_len : 1,
                  wv::SeqMap{seq_mode, seq_a, seq_b, seq_c, perm, T}};
    hipStream_t s = (hipStream_t)stream;
    wv::prof::Scope ps(s, "augment", 0.0, 4.0 * B * C * (2.0 * T + 3.0 * T_out));
    hipLaunchKernelGGL(wv::aug_kernel, dim3((T_out + 1023) / 1024, B * C), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

int wv_aug_sequence(const float* in0, const float* in1, const float* in2, float* out0, float* out1, float* out2,
                    int seq_mode, int seq_a, int seq_b, int seq_c, const int* perm, int rows, int T, int T_out, void* stream) {
    if (rows < 1 || T < 1 || rows > 65535 || (!in0 && !in1 && !in2)) return WV_EINVAL;
    if ((in0 && !out0) || (in1 && !out1) || (in2 && !out2)) return WV_EINVAL;
    if (!wv::seq_ok(seq_mode, seq_a, seq_b, seq_c, perm, T, T_out)) return WV_EINVAL;
    wv::SeqArgs a{{in0, in1, in2}, {out0, out1, out2}, rows, T, T_out, wv::SeqMap{seq_mode, seq_a, seq_b, seq_c, perm, T}};
    const int n = (in0 ? 1 : 0) + (in1 ? 1 : 0) + (in2 ? 1 : 0);
    hipStream_t s = (hipStream_t)stream;
    wv::prof::Scope ps(s, "augment_seq", 0.0, 4.0 * rows * n * ((double)T_out * 2.0));
    hipLaunchKernelGGL(wv::seq_kernel, dim3((T_out + 1023) / 1024, rows), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

int wv_aug_backward(const float* d_out, const int* plan, int nseg, int seg_len, int inv_mode, int inv_a, int inv_b, int inv_c, const int* inv_perm,
                    float* d_wm, int B, int C, int T, int T_out, void* stream) {
    if (!d_out || !d_wm || B < 1 || C < 1 || T < 1 || (long long)B * C > 65535) return WV_EINVAL;
    if (plan && (seg_len < 1 || nseg != (T + seg_len - 1) / seg_len)) return WV_EINVAL;
    // the inverse map runs over the OUTPUT axis: it is a map of length T_out (identity / reverse / roll / chunk swap: T_out == T)
    if (!wv::seq_ok(inv_mode, inv_a, inv_b, inv_c, inv_perm, T_out, T_out)) return WV_EINVAL;
    wv::AugBwdArgs a{d_out, plan, d_wm, B, C, T, T_out, nseg, plan ? seg_len : 1, wv::SeqMap{inv_mode, inv_a, inv_b, inv_c, inv_perm, T_out}};
    hipLaunchKernelGGL(wv::aug_bwd_kernel, dim3((T + 255) / 256, B * C), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? WV_OK : WV_EHIP;
}

}  // extern "C"
