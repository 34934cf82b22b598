// Host side of libwaveverify_hip.so: parameter table (reference state-dict grammar), weight
// packing, the launch plan of the three nets, and the extern "C" API of
// include/waveverify_hip.h.  No torch here: raw device pointers + a hipStream_t.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/waveverify_hip.h"
#include "wv_kernels.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(WV_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct Param {
    std::string name;
    std::vector<int64_t> shape;   // reference layout
    bool wn = false;
    bool optional = false;        // message MLP / FiLM of a detector/locator encoder
    std::vector<float> data;
    bool set = false;
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

using wv::PwWeight;

struct ResBlock {
    PwWeight pw1, pw2;
    const float *dw1_w, *dw1_b, *dw2_w, *dw2_b;
    const float *tab1 = nullptr, *tab2 = nullptr;   // [C][8] taps + bias rows of the fused-block kernel (ks == 5 only)
    float pre_scale, out_scale;
    int ks, dil1, dil2;
};
struct SpecLayer {
    const float* basis_t; const float* basis_q; const float* side; int n_fft, hop, F, Mp; float mean, inv_std;
    PwWeight pw; float scale;
    const float* id_taps;      // [C][5] = (0,0,0,0,1): the spec add runs on K1 with an identity stencil
};
struct DownLayer { PwWeight pw; const float *dw_w, *dw_b; int ratio; float pre_scale; };
struct UpLayer {
    const float* ct_w; int ratio; PwWeight pw; const float* pw_b; float pre_scale;
    const float* id_taps;      // [M][5] = (0,0,0,0,1): identity stencil of the K1 form
    const float* ct_wt;        // ct_w transposed [2r][Kp]
    std::vector<ResBlock> res;
};

}  // namespace

struct wv_model {
    wv_config cfg{};
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    std::unordered_map<std::string, std::vector<float>> stft_override;
    bool finalized = false;
    std::vector<void*> dev;                      // owned device allocations

    // ---- encoder plan
    const float *pre_w = nullptr, *pre_b = nullptr;
    std::vector<std::vector<ResBlock>> enc_blocks;
    std::vector<SpecLayer> specs;                // n_scales + 1 (last = spec_post)
    std::vector<DownLayer> downs;
    const float* post_dw_w = nullptr; PwWeight post_pw; const float* post_b = nullptr;
    bool has_film = false;
    const float *f_w0 = nullptr, *f_b0 = nullptr, *f_wl = nullptr, *f_bl = nullptr,
                *f_wf = nullptr, *f_bf = nullptr;
    // ---- decoder plan
    PwWeight dec_pw0; const float *dec_dw0_w = nullptr, *dec_dw0_b = nullptr;
    std::vector<UpLayer> ups;
    const float *last_w = nullptr, *last_b = nullptr;
    float dec_post = 1.f;
    // ---- head plan
    const float *head_wc = nullptr, *head_bc = nullptr;
    int head_nb = 0;
    // ---- f16 mode (wv_h16.hip): A-fragment weights per encoder stage -- the ResnetBlocks' 1x1 pairs (divided by log2(e), their first
    // stencil table times log2(e): RhArgs), the SpecBlock's 1x1 over the zero-padded spectrum rows, the downsample unit's 1x1 and
    // depth-wise conv composed into one [M][2r][K] conv -- and per decoder stage: the upsample unit as one 2-tap conv over (phase, channel)
    // rows (pack_up16) and its ResnetBlocks; the decoder's first conv pair as one composed conv
    struct H16Block { wv::H16Weight w1, w2; const float *tab1 = nullptr, *tab2 = nullptr; };
    struct H16Stage { std::vector<H16Block> blocks; wv::H16Weight spec, down, cosw, sinw, cosl, sinl, post, head; };   // post / head: the last entry only (conv_post as one composed conv, the head GEMM)
    struct H16Up { wv::H16Weight up; int mb = 0; std::vector<H16Block> blocks; };
    std::vector<H16Stage> h16;                    // n_strides stages; one more when spec_post runs on the f16 pipe too (only its spec / cos / sin / post / head members)
    wv::H16Weight h16_dec_head;
    std::vector<H16Up> h16_ups;                   // generator only; empty = no f16 decoder plan

    ~wv_model() { for (void* p : dev) (void)hipFree(p); }
};

namespace {

int hop_of(const wv_config& c) { int h = 1; for (int i = 0; i < c.n_strides; ++i) h *= c.strides[i]; return h; }
int ratio_enc(const wv_config& c, int s) { return c.strides[c.n_strides - 1 - s]; }   // seanet.py:646

void add(wv_model* m, const std::string& name, std::vector<int64_t> shape, bool wn, bool optional = false) {
    Param p; p.name = name; p.shape = std::move(shape); p.wn = wn; p.optional = optional;
    m->index[name] = (int)m->params.size();
    m->params.push_back(std::move(p));
}

void add_resblock(wv_model* m, const std::string& pre, int dim, int k, bool zero_init) {
    if (zero_init) add(m, pre + ".res_scale_param", {1}, false);
    const int idx[2][2] = {{1, 2}, {4, 5}};
    for (auto& pd : idx) {
        add(m, pre + ".block." + std::to_string(pd[0]) + ".conv.conv.weight", {dim, dim, 1}, true);
        add(m, pre + ".block." + std::to_string(pd[1]) + ".conv.conv.bias", {dim}, false);
        add(m, pre + ".block." + std::to_string(pd[1]) + ".conv.conv.weight", {dim, 1, k}, true);
    }
}

// Key grammar: modules/seanet.py:657-846 (encoder), :1067-1204 (decoder), detector.py:209-218.
void build_param_table(wv_model* m) {
    const wv_config& c = m->cfg;
    const int C0 = c.channels_enc, S = c.n_strides;
    const bool zi = c.zero_init != 0;
    add(m, "encoder.conv_pre.1.conv.conv.bias", {C0}, false);
    add(m, "encoder.conv_pre.1.conv.conv.weight", {C0, 1, c.kernel_size}, true);
    int mult = 1;
    for (int s = 0; s < S; ++s) {
        for (int j = 0; j < c.n_residual_enc; ++j)
            add_resblock(m, "encoder.blocks." + std::to_string(s) + "." + std::to_string(j),
                         mult * C0, c.residual_kernel_size, zi);
        mult *= 2;
    }
    mult = 1;
    for (int s = 0; s < S; ++s) {
        const std::string pre = "encoder.spec_blocks." + std::to_string(s);
        if (zi) add(m, pre + ".scale_param", {1}, false);
        add(m, pre + ".layer.conv.conv.weight", {mult * C0, mult * c.n_fft_base / 2 + 1, 1}, true);
        mult *= 2;
    }
    mult = 1;
    for (int s = 0; s < S; ++s) {
        const int C = mult * C0, r = ratio_enc(c, s);
        const std::string pre = "encoder.downsample." + std::to_string(s);
        add(m, pre + ".2.conv.conv.weight", {2 * C, C, 1}, true);
        add(m, pre + ".3.conv.conv.bias", {2 * C}, false);
        add(m, pre + ".3.conv.conv.weight", {2 * C, 1, 2 * r}, true);
        mult *= 2;
    }
    const int Ctop = mult * C0;
    if (zi) add(m, "encoder.spec_post.scale_param", {1}, false);
    add(m, "encoder.spec_post.layer.conv.conv.weight", {Ctop, mult * c.n_fft_base / 2 + 1, 1}, true);
    add(m, "encoder.conv_post.1.conv.conv.weight", {Ctop, 1, c.last_kernel_size}, true);
    add(m, "encoder.conv_post.2.conv.conv.bias", {c.dimension}, false);
    add(m, "encoder.conv_post.2.conv.conv.weight", {c.dimension, Ctop, 1}, true);
    // message MLP + FiLM exist in every SEANetEncoder but only the generator's forward uses them
    const bool opt = c.kind != WV_KIND_GENERATOR;
    const int E = c.embedding_dim;
    add(m, "encoder.msg_embedding.0.weight", {E, c.msg_dimension}, false, opt);
    add(m, "encoder.msg_embedding.0.bias", {E}, false, opt);
    for (int l = 0; l < c.embedding_layers; ++l) {
        const std::string pre = "encoder.msg_embedding." + std::to_string(1 + 2 * l);
        add(m, pre + ".weight", {E, E}, false, opt);
        add(m, pre + ".bias", {E}, false, opt);
    }
    for (int s = 0; s < S; ++s)
        for (int b = 0; b < c.freq_bands; ++b)
            for (const char* nm : {"gamma", "beta"}) {
                const std::string pre = "encoder.film_layers." + std::to_string(s) + "." +
                                        std::to_string(b) + "." + nm + "_layer";
                add(m, pre + ".weight", {1, E}, false, opt);
                add(m, pre + ".bias", {1}, false, opt);
            }
    if (c.kind == WV_KIND_GENERATOR) {
        const int Cd = c.channels_dec;
        int dm = 1 << S;
        add(m, "decoder.model.0.conv.conv.weight", {dm * Cd, c.dimension, 1}, true);
        add(m, "decoder.model.1.conv.conv.bias", {dm * Cd}, false);
        add(m, "decoder.model.1.conv.conv.weight", {dm * Cd, 1, c.kernel_size}, true);
        int n = 2;
        for (int i = 0; i < S; ++i) {
            const int C = dm * Cd, r = c.strides[i];
            add(m, "decoder.model." + std::to_string(n + 2) + ".convtr.convtr.weight", {C, 1, 2 * r}, true);
            add(m, "decoder.model." + std::to_string(n + 3) + ".conv.conv.bias", {C / 2}, false);
            add(m, "decoder.model." + std::to_string(n + 3) + ".conv.conv.weight", {C / 2, C, 1}, true);
            for (int j = 0; j < c.n_residual_dec; ++j)
                add_resblock(m, "decoder.model." + std::to_string(n + 4 + j), C / 2,
                             c.residual_kernel_size, zi);
            n += 4 + c.n_residual_dec;
            dm /= 2;
        }
        add(m, "decoder.model." + std::to_string(n + 2) + ".conv.conv.bias", {1}, false);
        add(m, "decoder.model." + std::to_string(n + 2) + ".conv.conv.weight", {1, Cd, c.last_kernel_size}, true);
    } else {
        const int nb = c.kind == WV_KIND_DETECTOR ? c.nbits : 1;
        add(m, "reverse_convolution.weight", {c.dimension, c.output_dim, hop_of(c)}, false);
        add(m, "reverse_convolution.bias", {c.output_dim}, false);
        add(m, "last_layer.weight", {nb, c.output_dim, 1}, false);
        add(m, "last_layer.bias", {nb}, false);
    }
}

// ------------------------------------------------------------------------------ packing
struct Uploader {
    wv_model* m;
    int err = WV_OK;
    const float* up(const std::vector<float>& h) {
        if (err != WV_OK) return nullptr;
        void* d = nullptr;
        const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(float);
        if (hipMalloc(&d, bytes) != hipSuccess) { err = fail(WV_EHIP, "hipMalloc failed while packing weights"); return nullptr; }
        m->dev.push_back(d);
        if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            err = fail(WV_EHIP, "hipMemcpy failed while packing weights");
            return nullptr;
        }
        return static_cast<const float*>(d);
    }
    wv::H16Weight h16(const std::vector<float>& pw, const float* dw, int M, int K, int ks) {
        wv::H16Weight w;
        const std::vector<uint16_t> q = wv::pack_h16(pw.data(), dw, M, K, ks, &w);
        return h16_up(q, w);
    }
    wv_model::H16Block h16_block(const std::string& pre, int C) {
        wv_model::H16Block b;
        wv::H16Weight w;
        { const std::vector<uint16_t> q = wv::pack_rh_pw(host(pre + ".block.1.conv.conv.weight").data(), C, &w); b.w1 = h16_up(q, w); }
        { const std::vector<uint16_t> q = wv::pack_rh_pw(host(pre + ".block.4.conv.conv.weight").data(), C, &w); b.w2 = h16_up(q, w); }
        b.tab1 = up(wv::pack_rh_table(host(pre + ".block.2.conv.conv.weight").data(), host(pre + ".block.2.conv.conv.bias").data(), C, wv::RH_LOG2E));
        b.tab2 = up(wv::pack_rh_table(host(pre + ".block.5.conv.conv.weight").data(), host(pre + ".block.5.conv.conv.bias").data(), C, 1.0));
        return b;
    }
    wv::H16Weight h16_up(const std::vector<uint16_t>& q, wv::H16Weight w) {
        if (err != WV_OK) return w;
        void* d = nullptr;
        if (hipMalloc(&d, q.size() * sizeof(uint16_t)) != hipSuccess) { err = fail(WV_EHIP, "hipMalloc failed while packing f16 weights"); return w; }
        m->dev.push_back(d);
        if (hipMemcpy(d, q.data(), q.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) {
            err = fail(WV_EHIP, "hipMemcpy failed while packing f16 weights");
            return w;
        }
        w.wq = d;
        return w;
    }
    const std::vector<float>& host(const std::string& name) {
        static const std::vector<float> empty;
        auto it = m->index.find(name);
        if (it == m->index.end()) { err = fail(WV_ENOKEY, "internal: unknown key " + name); return empty; }
        return m->params[it->second].data;
    }
    const float* plain(const std::string& name) { return up(host(name)); }
    float scalar_or(const std::string& name, float dflt) {
        auto it = m->index.find(name);
        return it == m->index.end() ? dflt : m->params[it->second].data[0];
    }
    // W[M][K] (a 1x1 conv weight [M,K,1]) -> Wt[Kp][Mp], zero padded
    PwWeight pw_from(const std::vector<float>& w, int M, int K) {
        PwWeight p; p.M = M; p.K = K; p.Kp = wv::round_up(K, wv::BK); p.Mp = wv::round_up(M, wv::M_ALIGN);
        std::vector<float> t((size_t)p.Kp * p.Mp, 0.f);
        for (int mm = 0; mm < M; ++mm)
            for (int k = 0; k < K; ++k) t[(size_t)k * p.Mp + mm] = w[(size_t)mm * K + k];
        p.wt = up(t);
        std::vector<float> q((size_t)wv::round_up(K, 32) * p.Mp, 0.f);   // wq[roundup(K,32)/4][Mp][4]
        for (int mm = 0; mm < M; ++mm)
            for (int k = 0; k < K; ++k) q[((size_t)(k / 4) * p.Mp + mm) * 4 + (k & 3)] = w[(size_t)mm * K + k];
        p.wq = up(q);
        return p;
    }
    PwWeight pw(const std::string& name) {
        auto it = m->index.find(name);
        const Param& P = m->params[it->second];
        return pw_from(P.data, (int)P.shape[0], (int)P.shape[1]);
    }
};

// CausalSTFT basis exactly as modules/conv.py:1003-1020 forms it: float32 angle
// (-2*pi/n_fft) * k * n, cos / sin of that float32 angle, times torch.hann_window (periodic).
std::vector<float> make_basis(int n_fft) {
    const int F = n_fft / 2 + 1;
    std::vector<float> w((size_t)2 * F * n_fft);
    const float c = (float)(-2.0 * M_PI / n_fft);
    const float wc = (float)(2.0 * M_PI / n_fft);
    for (int k = 0; k < F; ++k) {
        volatile float ck = c * (float)k;
        for (int n = 0; n < n_fft; ++n) {
            volatile float ang = ck * (float)n;
            volatile float wa = (float)n * wc;
            const float win = 0.5f - 0.5f * (float)std::cos((double)wa);
            w[(size_t)k * n_fft + n] = (float)std::cos((double)ang) * win;
            w[(size_t)(F + k) * n_fft + n] = (float)std::sin((double)ang) * win;
        }
    }
    return w;
}

// [2F][n_fft] reference layout -> basis_t[Kp][Mp] with column 2f = cos row f, 2f+1 = sin row f
const float* pack_basis(Uploader& U, const std::vector<float>& basis, int n_fft, int* Mp_out, const float** side_out,
                        const float** q_out) {
    std::vector<float> bt, side;
    wv::pack_stft_basis(basis.data(), n_fft, bt, side, Mp_out);
    *side_out = U.up(side);
    *q_out = U.up(wv::pack_stft_q(bt, n_fft, *Mp_out));
    return U.up(bt);
}

ResBlock pack_resblock(Uploader& U, const std::string& pre, int idx, float rs, int ks, int dil1) {
    ResBlock r{};
    r.pw1 = U.pw(pre + ".block.1.conv.conv.weight");
    r.dw1_w = U.plain(pre + ".block.2.conv.conv.weight");
    r.dw1_b = U.plain(pre + ".block.2.conv.conv.bias");
    r.pw2 = U.pw(pre + ".block.4.conv.conv.weight");
    r.dw2_w = U.plain(pre + ".block.5.conv.conv.weight");
    r.dw2_b = U.plain(pre + ".block.5.conv.conv.bias");
    r.pre_scale = (float)std::pow(1.0 + idx * (double)rs * rs, -0.5);      // seanet.py:183
    r.out_scale = rs * U.scalar_or(pre + ".res_scale_param", 1.f);         // seanet.py:272-274
    r.ks = ks; r.dil1 = dil1; r.dil2 = 1;                                   // dilations=[base**j, 1]
    if (ks == 5 && U.err == WV_OK) {
        const int C = r.pw1.M;
        r.tab1 = U.up(wv::pack_rb_table(U.host(pre + ".block.2.conv.conv.weight").data(), U.host(pre + ".block.2.conv.conv.bias").data(), C));
        r.tab2 = U.up(wv::pack_rb_table(U.host(pre + ".block.5.conv.conv.weight").data(), U.host(pre + ".block.5.conv.conv.bias").data(), C));
    }
    return r;
}

int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }
int round_up_int(int x, int a) { return (x + a - 1) / a * a; }

int pack_model(wv_model* m) {
    const wv_config& c = m->cfg;
    Uploader U{m};
    std::vector<float> head_wc_host;                              // composed head weight [D][nb * hop] (detector / locator), for the f16 packer
    const int S = c.n_strides, C0 = c.channels_enc;
    const float rs = c.res_scale_enc;
    m->pre_w = U.plain("encoder.conv_pre.1.conv.conv.weight");
    m->pre_b = U.plain("encoder.conv_pre.1.conv.conv.bias");
    int mult = 1, stride = 1;
    for (int s = 0; s <= S; ++s) {
        const bool post = s == S;
        const std::string pre = post ? "encoder.spec_post" : "encoder.spec_blocks." + std::to_string(s);
        if (!post) {
            std::vector<ResBlock> blocks;
            for (int j = 1; j <= c.n_residual_enc; ++j)                    // idx = j (seanet.py:684)
                blocks.push_back(pack_resblock(U, "encoder.blocks." + std::to_string(s) + "." + std::to_string(j - 1),
                                               j, rs, c.residual_kernel_size, ipow(c.dilation_base, j)));
            m->enc_blocks.push_back(std::move(blocks));
        }
        SpecLayer sp{};
        sp.n_fft = mult * c.n_fft_base; sp.hop = stride; sp.F = sp.n_fft / 2 + 1;
        auto ov = m->stft_override.find(pre + ".spec.weight");
        sp.basis_t = pack_basis(U, ov != m->stft_override.end() ? ov->second : make_basis(sp.n_fft),
                                sp.n_fft, &sp.Mp, &sp.side, &sp.basis_q);
        const int mi = post ? WV_MAX_STRIDES : s;                           // spec_post uses [-1]
        sp.mean = post ? c.spec_means[WV_MAX_STRIDES] : c.spec_means[s];
        sp.inv_std = 1.f / (post ? c.spec_stds[WV_MAX_STRIDES] : c.spec_stds[s]);
        (void)mi;
        sp.pw = U.pw(pre + ".layer.conv.conv.weight");
        sp.scale = U.scalar_or(pre + ".scale_param", 1.f) * rs;            // seanet.py:500-502
        {
            std::vector<float> taps((size_t)sp.pw.M * 5, 0.f);
            for (int mm = 0; mm < sp.pw.M; ++mm) taps[(size_t)mm * 5 + 4] = 1.f;
            sp.id_taps = U.up(taps);
        }
        m->specs.push_back(sp);
        if (!post) {
            DownLayer d{};
            const std::string dp = "encoder.downsample." + std::to_string(s);
            d.pw = U.pw(dp + ".2.conv.conv.weight");
            d.dw_w = U.plain(dp + ".3.conv.conv.weight");
            d.dw_b = U.plain(dp + ".3.conv.conv.bias");
            d.ratio = ratio_enc(c, s);
            d.pre_scale = (float)std::pow(1.0 + c.n_residual_enc * (double)rs * rs, -0.5);   // seanet.py:739
            m->downs.push_back(d);
            stride *= d.ratio;
            mult *= 2;
        }
    }
    m->post_dw_w = U.plain("encoder.conv_post.1.conv.conv.weight");
    m->post_pw = U.pw("encoder.conv_post.2.conv.conv.weight");
    m->post_b = U.plain("encoder.conv_post.2.conv.conv.bias");
    (void)C0;

    // message MLP + FiLM (only when the tensors were provided; mandatory for the generator)
    m->has_film = m->params[m->index["encoder.msg_embedding.0.weight"]].set;
    if (m->has_film) {
        const int E = c.embedding_dim;
        m->f_w0 = U.plain("encoder.msg_embedding.0.weight");
        m->f_b0 = U.plain("encoder.msg_embedding.0.bias");
        std::vector<float> wl, bl, wf, bf;
        for (int l = 0; l < c.embedding_layers; ++l) {
            const std::string pre = "encoder.msg_embedding." + std::to_string(1 + 2 * l);
            const auto& w = U.host(pre + ".weight"); const auto& b = U.host(pre + ".bias");
            wl.insert(wl.end(), w.begin(), w.end()); bl.insert(bl.end(), b.begin(), b.end());
        }
        for (int s = 0; s < S; ++s)
            for (int b = 0; b < c.freq_bands; ++b)
                for (const char* nm : {"gamma", "beta"}) {
                    const std::string pre = "encoder.film_layers." + std::to_string(s) + "." +
                                            std::to_string(b) + "." + nm + "_layer";
                    const auto& w = U.host(pre + ".weight"); const auto& bb = U.host(pre + ".bias");
                    wf.insert(wf.end(), w.begin(), w.end()); bf.push_back(bb[0]);
                }
        m->f_wl = U.up(wl); m->f_bl = U.up(bl); m->f_wf = U.up(wf); m->f_bf = U.up(bf);
        (void)E;
    }

    if (c.kind == WV_KIND_GENERATOR) {
        const float rd = c.res_scale_dec;
        m->dec_pw0 = U.pw("decoder.model.0.conv.conv.weight");
        m->dec_dw0_w = U.plain("decoder.model.1.conv.conv.weight");
        m->dec_dw0_b = U.plain("decoder.model.1.conv.conv.bias");
        m->dec_post = (float)std::pow(1.0 + c.n_residual_dec * (double)rd * rd, -0.5);     // seanet.py:1104
        int n = 2;
        for (int i = 0; i < S; ++i) {
            UpLayer u{};
            u.ratio = c.strides[i];
            u.ct_w = U.plain("decoder.model." + std::to_string(n + 2) + ".convtr.convtr.weight");
            u.pw = U.pw("decoder.model." + std::to_string(n + 3) + ".conv.conv.weight");
            u.pw_b = U.plain("decoder.model." + std::to_string(n + 3) + ".conv.conv.bias");
            {
                std::vector<float> taps((size_t)u.pw.M * 5, 0.f);
                for (int mm = 0; mm < u.pw.M; ++mm) taps[(size_t)mm * 5 + 4] = 1.f;
                u.id_taps = U.up(taps);
                const std::vector<float>& cw = U.host("decoder.model." + std::to_string(n + 2) + ".convtr.convtr.weight");
                u.ct_wt = U.up(wv::pack_ct_wt(cw.data(), u.pw.K, u.pw.Kp, u.ratio));
            }
            u.pre_scale = i > 0 ? m->dec_post : 1.f;
            for (int j = 0; j < c.n_residual_dec; ++j)                     // idx = j (seanet.py:1159)
                u.res.push_back(pack_resblock(U, "decoder.model." + std::to_string(n + 4 + j), j, rd,
                                              c.residual_kernel_size, ipow(c.dilation_base, j)));
            m->ups.push_back(std::move(u));
            n += 4 + c.n_residual_dec;
        }
        m->last_w = U.plain("decoder.model." + std::to_string(n + 2) + ".conv.conv.weight");
        m->last_b = U.plain("decoder.model." + std::to_string(n + 2) + ".conv.conv.bias");
    } else {
        // compose ConvTranspose1d(D->O, k=s=hop) with Conv1d(O->nb, 1) (detector.py:300-310)
        const int D = c.dimension, O = c.output_dim, hop = hop_of(c);
        const int nb = c.kind == WV_KIND_DETECTOR ? c.nbits : 1;
        const auto& w1 = U.host("reverse_convolution.weight");   // [D][O][hop]
        const auto& b1 = U.host("reverse_convolution.bias");
        const auto& w2 = U.host("last_layer.weight");            // [nb][O]
        const auto& b2 = U.host("last_layer.bias");
        std::vector<float> wc((size_t)D * nb * hop), bc(nb);
        for (int d = 0; d < D; ++d)
            for (int nn = 0; nn < nb; ++nn)
                for (int j = 0; j < hop; ++j) {
                    double acc = 0;
                    for (int o = 0; o < O; ++o) acc += (double)w2[(size_t)nn * O + o] * w1[((size_t)d * O + o) * hop + j];
                    wc[((size_t)d * nb + nn) * hop + j] = (float)acc;
                }
        for (int nn = 0; nn < nb; ++nn) {
            double acc = b2[nn];
            for (int o = 0; o < O; ++o) acc += (double)w2[(size_t)nn * O + o] * b1[o];
            bc[nn] = (float)acc;
        }
        m->head_wc = U.up(wc); m->head_bc = U.up(bc); m->head_nb = nb;
        head_wc_host = wc;
    }
    // f16 mode: only for layer shapes the f16 kernels cover (else the *_forward_f16 entry points report WV_ESTATE)
    auto rh_c = [](int C) { return C == 32 || C == 64 || C == 96 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512 || C == 768; };
    bool h16_ok = c.residual_kernel_size == 5 && c.dilation_base == 1 && c.kernel_size <= 16 && c.last_kernel_size <= 16;
    for (int s = 0, C = C0; s < S && h16_ok; ++s, C *= 2) {
        h16_ok = rh_c(C) || m->enc_blocks[s].empty();
        // a scale the one-launch SpecBlock does not take stages its f16 spectrogram (roundup(F, 16) rows) in the intermediate buffer of C rows of f32
        if (C != m->specs[s].n_fft && round_up_int(m->specs[s].F, 16) > 2 * C) h16_ok = false;
        if (C % 8) h16_ok = false;
    }
    if (h16_ok) {
        int C = C0;
        for (int s = 0; s < S && U.err == WV_OK; ++s, C *= 2) {
            wv_model::H16Stage st;
            for (int j = 0; j < c.n_residual_enc; ++j)
                st.blocks.push_back(U.h16_block("encoder.blocks." + std::to_string(s) + "." + std::to_string(j), C));
            const int F = m->specs[s].F, r = ratio_enc(c, s);
            st.spec = U.h16(U.host("encoder.spec_blocks." + std::to_string(s) + ".layer.conv.conv.weight"), nullptr, C, F, 1);
            const std::string dp = "encoder.downsample." + std::to_string(s);
            st.down = U.h16(U.host(dp + ".2.conv.conv.weight"), U.host(dp + ".3.conv.conv.weight").data(), 2 * C, C, 2 * r);
            {
                const int n_fft = m->specs[s].n_fft;
                auto ov = m->stft_override.find("encoder.spec_blocks." + std::to_string(s) + ".spec.weight");
                const std::vector<float> basis = ov != m->stft_override.end() ? ov->second : make_basis(n_fft);
                std::vector<uint16_t> q4[4];
                wv::H16Weight w4[4];
                wv::pack_stft16(basis.data(), n_fft, q4, w4);
                st.cosw = U.h16_up(q4[0], w4[0]); st.sinw = U.h16_up(q4[1], w4[1]); st.cosl = U.h16_up(q4[2], w4[2]); st.sinl = U.h16_up(q4[3], w4[3]);
            }
            m->h16.push_back(std::move(st));
        }
        if (U.err == WV_OK && C % 16 == 0) {
            // spec_post on the f16 pipe as well: in one launch where the scale is one of spec16's (the default generator / detector: 1024 points,
            // hop 320; the default locator: 256 points, hop 32, 128 channels), else the exact STFT kernel + the 1x1 on the f16 pipe
            wv_model::H16Stage st;
            const int n_fft = m->specs[S].n_fft;
            st.spec = U.h16(U.host("encoder.spec_post.layer.conv.conv.weight"), nullptr, C, m->specs[S].F, 1);
            if ((C == n_fft && C == 1024 && m->specs[S].hop == 320) || (2 * C == n_fft && C == 128 && m->specs[S].hop == 32)) {
                auto ov = m->stft_override.find("encoder.spec_post.spec.weight");
                const std::vector<float> basis = ov != m->stft_override.end() ? ov->second : make_basis(n_fft);
                std::vector<uint16_t> q4[4];
                wv::H16Weight w4[4];
                wv::pack_stft16(basis.data(), n_fft, q4, w4);
                st.cosw = U.h16_up(q4[0], w4[0]); st.sinw = U.h16_up(q4[1], w4[1]); st.cosl = U.h16_up(q4[2], w4[2]); st.sinl = U.h16_up(q4[3], w4[3]);
            }
            // conv_post (ELU -> depth-wise k -> 1x1 + bias, seanet.py:797-823) as one dense conv: W[m][i][k] = pw[m][k] * dw[k][i];
            // the head's composed weight wc[D][nb * hop] transposed into A fragments [nb * hop][D]
            {
                wv::H16Weight w;
                const std::vector<uint16_t> q = wv::pack_h16(U.host("encoder.conv_post.2.conv.conv.weight").data(), U.host("encoder.conv_post.1.conv.conv.weight").data(),
                                                             c.dimension, C, c.last_kernel_size, &w, true);
                st.post = U.h16_up(q, w);
                if (c.kind != WV_KIND_GENERATOR) {
                    const int D = c.dimension, rows = m->head_nb * hop_of(c);
                    std::vector<float> wt((size_t)rows * D);
                    for (int d = 0; d < D; ++d)
                        for (int r = 0; r < rows; ++r) wt[(size_t)r * D + d] = head_wc_host[(size_t)d * rows + r];
                    st.head = U.h16(wt, nullptr, rows, D, 1);
                }
            }
            m->h16.push_back(std::move(st));
        }
        // the generator's decoder (seanet.py:1067-1226): needs the latent from the f16 conv_post above
        bool dec_ok = c.kind == WV_KIND_GENERATOR && (int)m->h16.size() > S && c.dimension % 16 == 0 && (c.last_kernel_size == 3 || c.last_kernel_size == 5 || c.last_kernel_size == 7);
        for (size_t i = 0; i < m->ups.size() && dec_ok; ++i) dec_ok = (m->ups[i].res.empty() || rh_c(m->ups[i].pw.M)) && m->ups[i].pw.M % 16 == 0 && m->ups[i].pw.K % 16 == 0;
        if (dec_ok && U.err == WV_OK) {
            // decoder.model.0 (1x1, D -> C) and .1 (depth-wise k) as one composed conv: W[m][i][k] = dw[m][i] * pw[m][k]
            m->h16_dec_head = U.h16(U.host("decoder.model.0.conv.conv.weight"), U.host("decoder.model.1.conv.conv.weight").data(), m->dec_pw0.M, c.dimension, c.kernel_size);
            int n = 2;
            for (size_t i = 0; i < m->ups.size() && U.err == WV_OK; ++i) {
                wv_model::H16Up hu;
                const UpLayer& u = m->ups[i];
                wv::H16Weight w;
                hu.mb = wv::up16_block(u.pw.M, u.ratio) ? wv::up16_block(u.pw.M, u.ratio) : u.pw.M;
                const std::vector<uint16_t> q = wv::pack_up16(U.host("decoder.model." + std::to_string(n + 3) + ".conv.conv.weight").data(),
                                                              U.host("decoder.model." + std::to_string(n + 2) + ".convtr.convtr.weight").data(), u.pw.M, u.pw.K, u.ratio, hu.mb, &w);
                hu.up = U.h16_up(q, w);
                for (int j = 0; j < c.n_residual_dec; ++j) hu.blocks.push_back(U.h16_block("decoder.model." + std::to_string(n + 4 + j), u.pw.M));
                m->h16_ups.push_back(std::move(hu));
                n += 4 + c.n_residual_dec;
            }
        }
    }
    return U.err;
}

// ------------------------------------------------------------------------------ workspace
size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout {
    size_t act = 0;        // floats per activation buffer
    size_t spec = 0;       // floats of the STFT scratch
    size_t film = 0, latent = 0;
    size_t off_r0 = 0, off_r1 = 0, off_a0 = 0, off_a1 = 0, off_u = 0, off_p = 0, off_film = 0, off_lat = 0, total = 0;
};

WsLayout layout(const wv_model* m, int B, int T) {
    const wv_config& c = m->cfg;
    WsLayout L;
    size_t Tl = (size_t)T, C = (size_t)c.channels_enc, mx = C * Tl, sp = 0;
    int mult = 1;
    size_t stride = 1;
    for (int s = 0; s <= c.n_strides; ++s) {
        const size_t F = (size_t)mult * c.n_fft_base / 2 + 1;
        const size_t Tf = ((size_t)T + stride - 1) / stride;
        sp = std::max(sp, F * Tf);
        if (s < c.n_strides) {
            const int r = ratio_enc(c, s);
            Tl = (Tl + r - 1) / r; C *= 2; mx = std::max(mx, C * Tl);
            stride *= r; mult *= 2;
        }
    }
    const size_t Fr = Tl;
    if (c.kind == WV_KIND_GENERATOR) {
        size_t Cd = (size_t)c.channels_dec << c.n_strides, Td = Fr;
        mx = std::max(mx, Cd * Td);
        for (int i = 0; i < c.n_strides; ++i) { Td *= c.strides[i]; Cd /= 2; mx = std::max(mx, Cd * Td); }
    }
    L.act = (size_t)B * mx; L.spec = (size_t)B * sp;
    L.film = (size_t)B * c.n_strides * c.freq_bands * 2;
    L.latent = (size_t)B * c.dimension * Fr;
    size_t o = 0;
    L.off_r0 = o; o += align_up(L.act * 4);      // residual stream, raw (ping-pong)
    L.off_r1 = o; o += align_up(L.act * 4);
    L.off_a0 = o; o += align_up(L.act * 4);      // the same stream pre-activated for its consumer (ping-pong)
    L.off_a1 = o; o += align_up(L.act * 4);
    L.off_u = o; o += align_up(L.act * 4);       // ResnetBlock intermediate (activated)
    L.off_p = o; o += align_up(L.spec * 4);
    L.off_film = o; o += align_up(L.film * 4);
    L.off_lat = o; o += align_up(L.latent * 4);
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------ forward
#define LAUNCH(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(WV_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// The residual stream of a net: `raw` = x, `act` = ELU(s * x) for the NEXT consumer's scale s (written
// by the producer's epilogue, PwDwArgs::Yact, so that the consumer stages it by LDS-DMA), either may be
// null.  r[2] / a[2] are the ping-pong buffers behind them, u the ResnetBlock intermediate.
// Does a ResnetBlock of C channels want its input pre-activated by the producer (second output, one more HBM
// write pass)?  Blocks the one-launch kernel takes (wv_rb.hip: C in {64, 96, 128, 192}, k = 5, dilation 1, T % 4 == 0)
// read x raw and activate it on the way into LDS.  Blocks run as two K1 launches: narrow ones (one m-tile,
// bandwidth-bound) apply scale -> ELU themselves while staging through registers, from C = 129 up the producer-side
// copy wins (measured, one box, interleaved: thresholds 129 / 97 / 65 / 0 -> 81.2 / 82.1 / 82.5 / 83.7 ms per step).
inline bool fused_block(const wv_config& c, int C, int T) {
    return (C == 64 || C == 96 || C == 128 || C == 192) && c.residual_kernel_size == 5 && c.dilation_base == 1 && (T & 3) == 0 &&
           (long long)C * T * 4 < 0x7f000000LL;
}
inline bool wants_act_copy(const wv_config& c, int C, int T) { return C >= 129 && !fused_block(c, C, T); }

struct Stream {
    float* r[2]; float* a[2]; float* u;
    float* raw = nullptr; float* act = nullptr;
    float* other_raw() const { return raw == r[0] ? r[1] : r[0]; }
    float* other_act() const { return act == a[0] ? a[1] : a[0]; }
};

// SEANetResnetBlock (seanet.py:245-281): one launch for the narrow layers (wv_rb.hip), else two K1 launches.  next_scale > 0: also write ELU(next_scale*y);
// want_raw: write y itself (needed when y is a later residual / raw operand).
int run_resblock(const ResBlock& r, Stream& st, bool want_raw, float next_scale, int B, int T, hipStream_t s,
                 const char* role) {
    wv::prof::set_role(role);
    if (st.raw && r.tab1 && r.dil1 == 1 && r.dil2 == 1) {
        // narrow layers: the whole block in one launch, raw in / raw out, the intermediate stays in LDS (wv_rb.hip)
        wv::RbArgs f{};
        f.X = st.raw; f.pre_scale = r.pre_scale; f.pw1 = r.pw1; f.pw2 = r.pw2; f.tab1 = r.tab1; f.tab2 = r.tab2;
        f.Y = want_raw ? st.other_raw() : nullptr;
        f.Yact = next_scale > 0.f ? st.other_act() : nullptr;
        f.out_scale = r.out_scale; f.act_scale = next_scale; f.B = B; f.C = r.pw1.M; f.T = T;
        const hipError_t e = wv::launch_resblock(f, s);
        if (e == hipSuccess) { st.raw = f.Y; st.act = f.Yact; return WV_OK; }
        if (e != hipErrorNotSupported) return fail(WV_EHIP, std::string("launch_resblock: ") + hipGetErrorString(e));
    }
    wv::PwDwArgs a{};
    if (st.act) { a.X = st.act; a.pre_scale = 1.f; a.pre_elu = 0; }
    else { a.X = st.raw; a.pre_scale = r.pre_scale; a.pre_elu = 1; }
    a.pw = r.pw1; a.dw_w = r.dw1_w; a.dw_b = r.dw1_b; a.Y = nullptr; a.Yact = st.u; a.act_scale = 1.f;
    a.B = B; a.Tin = T; a.Tout = T; a.ks = r.ks; a.stride = 1; a.dil = r.dil1; a.pad = (r.ks - 1) * r.dil1;
    a.out_scale = 1.f; a.bands = 1;
    LAUNCH(wv::launch_pw_dw(a, s));
    wv::PwDwArgs b{};
    float* yr = want_raw ? st.other_raw() : nullptr;
    float* ya = next_scale > 0.f ? st.other_act() : nullptr;
    b.X = st.u; b.pw = r.pw2; b.dw_w = r.dw2_w; b.dw_b = r.dw2_b; b.resid = st.raw; b.Y = yr; b.Yact = ya;
    b.act_scale = next_scale;
    b.B = B; b.Tin = T; b.Tout = T; b.ks = r.ks; b.stride = 1; b.dil = r.dil2; b.pad = (r.ks - 1) * r.dil2;
    b.pre_scale = 1.f; b.pre_elu = 0; b.out_scale = r.out_scale; b.bands = 1;
    LAUNCH(wv::launch_pw_dw(b, s));
    st.raw = yr; st.act = ya;
    return WV_OK;
}

int run_film(wv_model* m, const float* msg, int msg_rows, float* film, int B, hipStream_t st) {
    const wv_config& c = m->cfg;
    if (!m->has_film) return fail(WV_ESTATE, "this model was finalized without message-MLP / FiLM tensors");
    if (msg_rows != B && msg_rows != 1) return fail(WV_EINVAL, "msg_rows must be B or 1");
    wv::prof::set_role("enc.film");
    wv::FilmArgs f{};
    f.msg = msg; f.msg_rows = msg_rows; f.msg_dim = c.msg_dimension; f.E = c.embedding_dim;
    f.n_layers = c.embedding_layers; f.n_out = c.n_strides * c.freq_bands * 2;
    f.w0 = m->f_w0; f.b0 = m->f_b0; f.wl = m->f_wl; f.bl = m->f_bl; f.wf = m->f_wf; f.bf = m->f_bf;
    f.film = film; f.B = B;
    LAUNCH(wv::launch_film(f, st));
    return WV_OK;
}

Stream make_stream(char* ws, const WsLayout& L) {
    Stream st{};
    st.r[0] = (float*)(ws + L.off_r0); st.r[1] = (float*)(ws + L.off_r1);
    st.a[0] = (float*)(ws + L.off_a0); st.a[1] = (float*)(ws + L.off_a1);
    st.u = (float*)(ws + L.off_u);
    return st;
}

// SEANetEncoder.forward (modules/seanet.py:883-976). Result in `latent` [B, dimension, Fr].
// first_stage > 0: the stages before it ran elsewhere (the f16 mode); their raw output [B, C, Tl] is in the stream's buffer r[0];
// post_spec_done: so did spec_post (only conv_post is left).
int run_encoder(wv_model* m, const float* x, const float* msg, int msg_rows, float* latent, int B,
                int T, char* ws, const WsLayout& L, hipStream_t st, int* Fr_out, int first_stage = 0, bool post_spec_done = false) {
    const wv_config& c = m->cfg;
    Stream sm = make_stream(ws, L);
    float* P = (float*)(ws + L.off_p);
    float* film = nullptr;
    if (msg) {
        film = (float*)(ws + L.off_film);
        int rc = run_film(m, msg, msg_rows, film, B, st);
        if (rc) return rc;
    }
    int Tl = T, C = c.channels_enc;
    sm.raw = sm.r[0];
    if (first_stage > 0) {
        sm.act = nullptr;
        for (int s = 0; s < first_stage; ++s) { Tl = (Tl + ratio_enc(c, s) - 1) / ratio_enc(c, s); C *= 2; }
    } else {
        wv::prof::set_role("enc.conv_pre");
        sm.act = (m->enc_blocks[0].empty() || !wants_act_copy(c, c.channels_enc, T)) ? nullptr : sm.a[0];   // also ELU(c1 * y) for the first ResnetBlock
        LAUNCH(wv::launch_conv_pre(x, m->pre_w, m->pre_b, sm.raw, sm.act, sm.act ? m->enc_blocks[0][0].pre_scale : 0.f,
                                   B, c.channels_enc, T, c.kernel_size, 1.f / c.wav_std, st));
    }
    const int film_stride = c.n_strides * c.freq_bands * 2;
    for (int s = first_stage; s <= c.n_strides; ++s) {
        const bool post = s == c.n_strides;
        if (!post) {
            const std::vector<ResBlock>& blocks = m->enc_blocks[s];
            for (size_t j = 0; j < blocks.size(); ++j) {
                // the last block feeds the SpecBlock add, which takes y raw (as its residual operand)
                const bool last = j + 1 == blocks.size();
                int rc = run_resblock(blocks[j], sm, true, (last || !wants_act_copy(c, C, Tl)) ? 0.f : blocks[j + 1].pre_scale, B, Tl, st,
                                      "enc.resblock");
                if (rc) return rc;
            }
        }
        if (post && post_spec_done) break;
        wv::prof::set_role("enc.spec");
        const SpecLayer& sp = m->specs[s];
        wv::StftArgs sa{};
        sa.wav = x; sa.basis_t = sp.basis_t; sa.basis_q = sp.basis_q; sa.side = sp.side; sa.P = P; sa.B = B; sa.T = T;
        sa.Tf = (T + sp.hop - 1) / sp.hop; sa.n_fft = sp.n_fft; sa.hop = sp.hop; sa.F = sp.F; sa.Mp = sp.Mp;
        sa.mean = sp.mean; sa.inv_std = sp.inv_std;
        if (sa.Tf != Tl) return fail(WV_EINVAL, "internal: STFT frame count != feature length");
        bool fused_spec = false;
        {
            // the scales whose whole spectrum is one tile (n_fft = C <= 128): STFT -> log-magnitude -> 1x1 -> add in ONE launch, the
            // spectrogram never leaves LDS (wv_k1.hip, stft_k1_kernel<.., true>)
            wv::SpecAddArgs q{};
            q.pw = sp.pw; q.resid = sm.raw; q.out_scale = sp.scale;
            q.Y = post ? sm.raw : nullptr;
            q.Yact = post ? nullptr : sm.other_act(); q.act_scale = post ? 0.f : m->downs[s].pre_scale;
            const hipError_t e = wv::launch_stft_spec(sa, q, st);
            if (e == hipSuccess) {
                if (!post) { sm.act = q.Yact; sm.raw = nullptr; }
                fused_spec = true;
            } else if (e != hipErrorNotSupported) return fail(WV_EHIP, std::string("launch_stft_spec: ") + hipGetErrorString(e));
        }
        if (!fused_spec) {
        LAUNCH(wv::launch_stft_logmag(sa, st));
        // x += scale * (W @ P)  (seanet.py:500-502).  Runs on K1 with an identity stencil (taps 0,0,0,0,1
        // are exact: fmaf(0,h,0) = 0, fmaf(1,h,0) = h) and x as the residual operand.  Before a
        // downsample only ELU(s * x') is consumed, so only that is written; spec_post keeps x' raw
        // (in place: every element is read and then written by the same lane).
        const float down_scale = post ? 0.f : m->downs[s].pre_scale;
        if (sp.pw.M < 33 || (Tl & 3)) {       // tiny or ragged layers: the plain 1x1 kernel
            wv::DwPwArgs k2{};
            k2.X = P; k2.pw = sp.pw; k2.Y = sm.raw; k2.B = B; k2.Tin = Tl; k2.Tout = Tl; k2.mode = 0;
            k2.pre_scale = 1.f; k2.pre_elu = 0; k2.accumulate = 1; k2.out_scale = sp.scale;
            k2.Yact = post ? nullptr : sm.other_act(); k2.act_scale = down_scale;    // the downsample's ELU(s * x')
            LAUNCH(wv::launch_dw_pw(k2, st));
            sm.act = k2.Yact;
        } else {
            wv::PwDwArgs acc{};
            acc.X = P; acc.pw = sp.pw; acc.dw_w = sp.id_taps; acc.dw_b = nullptr; acc.resid = sm.raw;
            acc.Y = post ? sm.raw : nullptr;
            acc.Yact = post ? nullptr : sm.other_act(); acc.act_scale = down_scale;
            acc.B = B; acc.Tin = Tl; acc.Tout = Tl; acc.ks = 5; acc.stride = 1; acc.dil = 1; acc.pad = 4;
            acc.pre_scale = 1.f; acc.pre_elu = 0; acc.out_scale = sp.scale; acc.bands = 1;
            acc.spec_add = 1;
            LAUNCH(wv::launch_pw_dw(acc, st));
            if (!post) { sm.act = acc.Yact; sm.raw = nullptr; }
        }
        }
        if (post) break;
        const DownLayer& d = m->downs[s];
        wv::prof::set_role(film ? "enc.down_film" : "enc.down");
        wv::PwDwArgs a{};
        if (sm.act) { a.X = sm.act; a.pre_scale = 1.f; a.pre_elu = 0; }
        else { a.X = sm.raw; a.pre_scale = d.pre_scale; a.pre_elu = 1; }
        const bool next_has_blocks = s + 1 < c.n_strides && !m->enc_blocks[s + 1].empty() && wants_act_copy(c, 2 * C, (Tl + d.ratio - 1) / d.ratio);
        float* yr = sm.raw ? sm.other_raw() : sm.r[0];
        float* ya = next_has_blocks ? (sm.act ? sm.other_act() : sm.a[0]) : nullptr;
        a.pw = d.pw; a.dw_w = d.dw_w; a.dw_b = d.dw_b; a.Y = yr; a.Yact = ya;
        a.act_scale = next_has_blocks ? m->enc_blocks[s + 1][0].pre_scale : 0.f;
        a.B = B; a.Tin = Tl; a.Tout = (Tl + d.ratio - 1) / d.ratio;
        a.ks = 2 * d.ratio; a.stride = d.ratio; a.dil = 1; a.pad = d.ratio;  // (k-1) - (s-1) = r
        a.out_scale = 1.f;
        a.bands = c.freq_bands; a.film_stride = film_stride;
        a.film = film ? film + (size_t)s * c.freq_bands * 2 : nullptr;
        if (film && (2 * C) % c.freq_bands) return fail(WV_EINVAL, "channels not divisible by freq_bands");
        LAUNCH(wv::launch_pw_dw(a, st));
        sm.raw = yr; sm.act = ya;
        Tl = a.Tout; C *= 2;
    }
    wv::prof::set_role("enc.conv_post");
    wv::DwPwArgs cp{};                       // conv_post: ELU -> DW k -> 1x1 + bias -> L2Norm
    cp.X = sm.raw; cp.dw_w = m->post_dw_w; cp.pw = m->post_pw; cp.bias = m->post_b; cp.Y = latent;
    cp.B = B; cp.Tin = Tl; cp.Tout = Tl; cp.mode = 1; cp.ks = c.last_kernel_size;
    cp.pre_scale = 1.f; cp.pre_elu = 1; cp.l2norm = 1;
    if (c.dimension > 128) return fail(WV_EINVAL, "dimension > 128 not supported by the fused L2-norm epilogue");
    LAUNCH(wv::launch_dw_pw(cp, st));
    *Fr_out = Tl;
    return WV_OK;
}

int check_common(const wv_model* m, int B, int T, const void* ws, size_t ws_bytes, WsLayout* L) {
    if (!m) return fail(WV_EINVAL, "null model");
    if (!m->finalized) return fail(WV_ESTATE, "wv_model_finalize has not been called");
    if (B < 1 || T < 1) return fail(WV_EINVAL, "B and T must be >= 1");
    *L = layout(m, B, T);
    if (!ws || ws_bytes < L->total)
        return fail(WV_ENOMEM, "workspace too small: need " + std::to_string(L->total) + " bytes");
    return WV_OK;
}

}  // namespace

// ================================================================================ C API
extern "C" {

const char* wv_last_error(void) { return g_err.c_str(); }
#ifndef WV_SRC_HASH
#define WV_SRC_HASH "unhashed"
#endif
const char* wv_version(void) { return "waveverify_hip 0.3 (gfx950, f32 MFMA) src " WV_SRC_HASH; }

int wv_config_default(int kind, wv_config* c) {
    if (!c || kind < 0 || kind > 2) return fail(WV_EINVAL, "bad kind / null cfg");
    std::memset(c, 0, sizeof(*c));
    c->kind = kind;
    c->dimension = 128; c->msg_dimension = 16; c->channels_enc = 64; c->channels_dec = 96;
    c->n_fft_base = 64; c->n_residual_enc = 2; c->n_residual_dec = 3;
    c->n_strides = 4; const int st[4] = {8, 5, 4, 2};
    for (int i = 0; i < 4; ++i) c->strides[i] = st[i];
    c->kernel_size = 5; c->last_kernel_size = 5; c->residual_kernel_size = 5; c->dilation_base = 1;
    c->zero_init = 1; c->nbits = 16; c->output_dim = 32; c->embedding_dim = 64;
    c->embedding_layers = 2; c->freq_bands = 4;
    c->res_scale_enc = c->res_scale_dec = 0.5773502691896258f; c->wav_std = 0.1122080159f;
    const float mu[5] = {-4.554f, -4.315f, -4.021f, -3.726f, -3.477f};
    const float sd[5] = {2.830f, 2.837f, 2.817f, 2.796f, 2.871f};
    for (int i = 0; i < 5; ++i) { c->spec_means[i] = mu[i]; c->spec_stds[i] = sd[i]; }
    c->spec_means[WV_MAX_STRIDES] = mu[4]; c->spec_stds[WV_MAX_STRIDES] = sd[4];   // spec_post = [-1]
    if (kind == WV_KIND_LOCATOR) {            // model/locator.py:84-115
        c->dimension = 64; c->channels_enc = 32; c->n_residual_enc = 1;
        c->n_strides = 2; c->strides[0] = 8; c->strides[1] = 4; c->strides[2] = c->strides[3] = 0;
    }
    return WV_OK;
}

int wv_model_create(const wv_config* cfg, wv_model** out) {
    if (!cfg || !out) return fail(WV_EINVAL, "null argument");
    if (cfg->kind < 0 || cfg->kind > 2) return fail(WV_EINVAL, "bad kind");
    if (cfg->n_strides < 1 || cfg->n_strides > WV_MAX_STRIDES) return fail(WV_EINVAL, "bad n_strides");
    for (int i = 0; i < cfg->n_strides; ++i)
        if (cfg->strides[i] < 1 || 2 * cfg->strides[i] > 16) return fail(WV_EINVAL, "stride out of range (1..8)");
    if (cfg->channels_enc < 1 || cfg->dimension < 1 || cfg->n_fft_base < 2 || (cfg->n_fft_base & 1))
        return fail(WV_EINVAL, "bad channel / fft sizes");
    if (cfg->embedding_dim > 256 || cfg->freq_bands < 1) return fail(WV_EINVAL, "bad embedding_dim / freq_bands");
    if (cfg->kernel_size > 16 || cfg->last_kernel_size > 16 || cfg->residual_kernel_size > 16)
        return fail(WV_EINVAL, "kernel sizes above 16 are not supported");
    auto* m = new wv_model();
    m->cfg = *cfg;
    build_param_table(m);
    *out = m;
    return WV_OK;
}

void wv_model_destroy(wv_model* m) { delete m; }

int wv_model_num_params(const wv_model* m) { return m ? (int)m->params.size() : 0; }

int wv_model_param_info(const wv_model* m, int i, char* name_out, int name_cap, int64_t* shape_out,
                        int* ndim_out, int* is_wn) {
    if (!m || i < 0 || i >= (int)m->params.size()) return fail(WV_EINVAL, "index out of range");
    const Param& p = m->params[i];
    if (name_out && name_cap > 0) { std::strncpy(name_out, p.name.c_str(), name_cap - 1); name_out[name_cap - 1] = 0; }
    if (shape_out) for (int d = 0; d < 4; ++d) shape_out[d] = d < (int)p.shape.size() ? p.shape[d] : 1;
    if (ndim_out) *ndim_out = (int)p.shape.size();
    if (is_wn) *is_wn = p.wn ? 1 : 0;
    return WV_OK;
}

int wv_model_set_param(wv_model* m, const char* name, const float* data, int64_t numel) {
    if (!m || !name || !data) return fail(WV_EINVAL, "null argument");
    if (m->finalized) return fail(WV_ESTATE, "model already finalized");
    auto it = m->index.find(name);
    if (it == m->index.end()) return fail(WV_ENOKEY, std::string("unknown parameter: ") + name);
    Param& p = m->params[it->second];
    if (numel != p.numel())
        return fail(WV_EINVAL, std::string("size mismatch for ") + name + ": got " + std::to_string(numel) +
                                   ", expected " + std::to_string(p.numel()));
    p.data.assign(data, data + numel);
    p.set = true;
    return WV_OK;
}

int wv_model_set_param_wn(wv_model* m, const char* name, const float* g, int64_t gn, const float* v, int64_t vn) {
    if (!m || !name || !g || !v) return fail(WV_EINVAL, "null argument");
    auto it = m->index.find(name);
    if (it == m->index.end()) return fail(WV_ENOKEY, std::string("unknown parameter: ") + name);
    Param& p = m->params[it->second];
    if (!p.wn) return fail(WV_EINVAL, std::string(name) + " is not weight-normed in the reference");
    if (vn != p.numel() || gn != p.shape[0]) return fail(WV_EINVAL, std::string("size mismatch for ") + name);
    // w = g * v / ||v||, norm over all dims but 0 (modules/conv.py:73-74)
    const int64_t rows = p.shape[0], inner = vn / rows;
    std::vector<float> w((size_t)vn);
    for (int64_t r = 0; r < rows; ++r) {
        float ss = 0.f;
        for (int64_t i = 0; i < inner; ++i) ss += v[r * inner + i] * v[r * inner + i];
        const float sc = g[r] / std::sqrt(ss);
        for (int64_t i = 0; i < inner; ++i) w[(size_t)(r * inner + i)] = v[r * inner + i] * sc;
    }
    return wv_model_set_param(m, name, w.data(), vn);
}

int wv_model_set_stft_basis(wv_model* m, const char* name, const float* data, int64_t numel) {
    if (!m || !name || !data) return fail(WV_EINVAL, "null argument");
    if (m->finalized) return fail(WV_ESTATE, "model already finalized");
    const std::string n(name);
    int mult = 1; bool ok = false;
    for (int s = 0; s <= m->cfg.n_strides; ++s) {
        const std::string key = (s == m->cfg.n_strides ? std::string("encoder.spec_post")
                                                       : "encoder.spec_blocks." + std::to_string(s)) + ".spec.weight";
        const int n_fft = mult * m->cfg.n_fft_base;
        if (key == n) {
            if (numel != (int64_t)(n_fft + 2) * n_fft) return fail(WV_EINVAL, "size mismatch for " + n);
            ok = true;
        }
        mult *= 2;
    }
    if (!ok) return fail(WV_ENOKEY, "unknown STFT buffer: " + n);
    m->stft_override[n].assign(data, data + numel);
    return WV_OK;
}

int wv_model_finalize(wv_model* m) {
    if (!m) return fail(WV_EINVAL, "null model");
    if (m->finalized) return WV_OK;
    int n_film_set = 0, n_film = 0;
    for (const Param& p : m->params) {
        if (p.optional) { ++n_film; n_film_set += p.set; continue; }
        if (!p.set) return fail(WV_ENOKEY, "missing parameter: " + p.name);
    }
    if (n_film_set != 0 && n_film_set != n_film)
        return fail(WV_ENOKEY, "message-MLP / FiLM tensors must be given all or none");
    int rc = pack_model(m);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    m->finalized = true;
    for (Param& p : m->params) { p.data.clear(); p.data.shrink_to_fit(); }
    return WV_OK;
}

size_t wv_workspace_bytes(const wv_model* m, int B, int T) {
    if (!m || B < 1 || T < 1) return 0;
    return layout(m, B, T).total;
}

int wv_encoder_forward(wv_model* m, const float* x, const float* msg, int msg_rows, float* latent,
                       int B, int T, void* ws, size_t ws_bytes, void* stream) {
    WsLayout L;
    int rc = check_common(m, B, T, ws, ws_bytes, &L);
    if (rc) return rc;
    if (!x || !latent) return fail(WV_EINVAL, "null tensor");
    int Fr = 0;
    return run_encoder(m, x, msg, msg_rows, latent, B, T, (char*)ws, L, (hipStream_t)stream, &Fr);
}

int wv_generator_forward(wv_model* m, const float* x, const float* msg, int msg_rows, float* out,
                         int add_input, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    WsLayout L;
    int rc = check_common(m, B, T, ws, ws_bytes, &L);
    if (rc) return rc;
    if (m->cfg.kind != WV_KIND_GENERATOR) return fail(WV_ESTATE, "not a generator model");
    if (!x || !msg || !out) return fail(WV_EINVAL, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    const wv_config& c = m->cfg;
    char* w = (char*)ws;
    float* latent = (float*)(w + L.off_lat);
    int Fr = 0;
    rc = run_encoder(m, x, msg, msg_rows, latent, B, T, w, L, st, &Fr);
    if (rc) return rc;
    // SEANetDecoder.forward (modules/seanet.py:1212-1226).  Every unit's output is consumed through an
    // ELU, so producers write the activated copy the next unit wants; the raw stream exists only where
    // it is a residual operand (inside a stage) or the tail's input.
    Stream sm = make_stream(w, L);
    wv::prof::set_role("dec.head");
    wv::PwDwArgs h{};
    h.X = latent; h.pw = m->dec_pw0; h.dw_w = m->dec_dw0_w; h.dw_b = m->dec_dw0_b;
    h.Y = nullptr; h.Yact = sm.a[0]; h.act_scale = m->ups.empty() ? 1.f : m->ups[0].pre_scale;
    h.B = B; h.Tin = Fr; h.Tout = Fr; h.ks = c.kernel_size; h.stride = 1; h.dil = 1; h.pad = c.kernel_size - 1;
    h.pre_scale = 1.f; h.pre_elu = 0; h.out_scale = 1.f; h.bands = 1;
    LAUNCH(wv::launch_pw_dw(h, st));
    sm.raw = nullptr; sm.act = sm.a[0];
    int Tl = Fr;
    for (size_t i = 0; i < m->ups.size(); ++i) {
        const UpLayer& u = m->ups[i];
        const bool last_up = i + 1 == m->ups.size();
        // what the stage's final output is consumed as: the next upsample's ELU(dec_post * y), or y by the tail
        const float stage_next = last_up ? 0.f : m->ups[i + 1].pre_scale;
        wv::prof::set_role("dec.upsample");
        // [Scale] -> ELU -> DW ConvTranspose(2r, r), trimmed -> 1x1 + bias (seanet.py:1147-1170): the K1
        // kernel with the ConvTranspose built in its B-operand loader (on the pre-activated input) and an
        // identity stencil.
        wv::PwDwArgs a{};
        a.X = sm.act; a.ct_w = u.ct_w; a.ct_wt = u.ct_wt; a.ratio = u.ratio; a.pw = u.pw; a.dw_w = u.id_taps; a.dw_b = u.pw_b;
        const bool has_blocks = !u.res.empty();
        a.Y = (has_blocks || last_up) ? sm.r[0] : nullptr;
        const bool blocks_act = has_blocks && wants_act_copy(c, u.pw.M, Tl * u.ratio);
        a.Yact = has_blocks ? (blocks_act ? sm.other_act() : nullptr) : (last_up ? nullptr : sm.other_act());
        a.act_scale = has_blocks ? u.res[0].pre_scale : stage_next;
        a.B = B; a.Tin = Tl; a.Tout = Tl * u.ratio; a.ks = 5; a.stride = 1; a.dil = 1; a.pad = 4;
        a.pre_scale = 1.f; a.pre_elu = 0; a.out_scale = 1.f; a.bands = 1;
        LAUNCH(wv::launch_pw_dw(a, st));
        sm.raw = a.Y; sm.act = a.Yact;
        Tl = a.Tout;
        for (size_t j = 0; j < u.res.size(); ++j) {
            const bool last = j + 1 == u.res.size();
            const float next = last ? stage_next : (wants_act_copy(c, u.pw.M, Tl) ? u.res[j + 1].pre_scale : 0.f);
            rc = run_resblock(u.res[j], sm, !last || last_up, next, B, Tl, st, "dec.resblock");
            if (rc) return rc;
        }
    }
    wv::prof::set_role("dec.tail");
    LAUNCH(wv::launch_tail(sm.raw, m->last_w, m->last_b, add_input ? x : nullptr, out, B, c.channels_dec,
                           Tl, T, c.last_kernel_size, m->dec_post, c.wav_std, st));
    return WV_OK;
}

// The encoder stages of the f16 mode (wv_h16.hip): conv_pre, then per stage the ResnetBlocks (one launch each), the SpecBlock (one launch
// where the scale is one of spec16's, else the exact path's STFT kernel and the 1x1 + add on the f16 pipe) and the downsample unit (one
// composed conv; FiLM in its epilogue when `film` is given: the generator).  What happens behind the last stage:
//   tail = H16_TAIL_F32      the last downsample (or spec_post, when it runs on the f16 pipe: *post_done) writes f32 [B, C, Tl] into the stream
//                            buffer r[0], where run_encoder(first_stage = n_strides) picks up (logits outputs: conv_post and the head exact)
//   tail = H16_TAIL_LATENT   spec_post -> ELU -> conv_post as one composed conv on the f16 pipe, the latent BEFORE its L2Norm as f32
//                            [B, D, Fr] at the workspace's latent slot; *latent_done (falls back to H16_TAIL_F32 when that plan is missing)
enum { H16_TAIL_F32 = 0, H16_TAIL_LATENT = 1 };
static int run_encoder_stages_f16(wv_model* m, const float* x, const float* film, int B, int T, char* ws, const WsLayout& L, hipStream_t st, int tail,
                                  bool* post_done, bool* latent_done, int* Fr_out) {
    const wv_config& c = m->cfg;
    const int S = c.n_strides;
    void* R[2] = {ws + L.off_r0, ws + L.off_r1};
    void* A0 = ws + L.off_a0;
    float* P = (float*)(ws + L.off_p);
    void* P16 = ws + L.off_u;
    int cur = 0, Tl = T, C = c.channels_enc;
    const int film_stride = c.n_strides * c.freq_bands * 2;
    wv::prof::set_role("enc16.conv_pre");
    LAUNCH(wv::launch_conv_pre16(x, m->pre_w, m->pre_b, R[0], B, C, T, c.kernel_size, 1.f / c.wav_std, st));
    for (int s = 0; s < S; ++s) {
        const wv_model::H16Stage& hs = m->h16[s];
        wv::prof::set_role("enc16.resblock");
        for (size_t j = 0; j < m->enc_blocks[s].size(); ++j) {
            const ResBlock& r = m->enc_blocks[s][j];
            wv::RhArgs a{};
            a.X = R[cur]; a.pre_scale = r.pre_scale; a.w1 = hs.blocks[j].w1; a.w2 = hs.blocks[j].w2; a.tab1 = hs.blocks[j].tab1; a.tab2 = hs.blocks[j].tab2;
            a.Y = R[cur ^ 1]; a.Yact = nullptr; a.out_scale = r.out_scale; a.act_scale = 0.f; a.B = B; a.C = C; a.T = Tl;
            const hipError_t e = wv::launch_resblock16(a, st);
            if (e != hipSuccess) return fail(e == hipErrorNotSupported ? WV_ESTATE : WV_EHIP, std::string("launch_resblock16: ") + hipGetErrorString(e));
            cur ^= 1;
        }
        wv::prof::set_role("enc16.spec");
        const SpecLayer& sp = m->specs[s];
        wv::StftArgs sa{};
        sa.wav = x; sa.basis_t = sp.basis_t; sa.basis_q = sp.basis_q; sa.side = sp.side; sa.P = P; sa.B = B; sa.T = T;
        sa.Tf = (T + sp.hop - 1) / sp.hop; sa.n_fft = sp.n_fft; sa.hop = sp.hop; sa.F = sp.F; sa.Mp = sp.Mp;
        sa.mean = sp.mean; sa.inv_std = sp.inv_std;
        if (sa.Tf != Tl) return fail(WV_EINVAL, "internal: STFT frame count != feature length");
        const DownLayer& d = m->downs[s];
        // x' = x + scale * (W @ P); only ELU(c * x') is consumed.  One launch where the scale is one of the fused kernel's (the default
        // detector's / generator's four), else the exact path's STFT kernel -> P in HBM -> f16 copy -> the 1x1 + add as a k = 1 conv
        wv::Spec16Args f{};
        f.wav = x; f.cosw = hs.cosw; f.sinw = hs.sinw; f.cosl = hs.cosl; f.sinl = hs.sinl; f.pw = hs.spec; f.resid = R[cur]; f.Y = nullptr; f.Yact = A0;
        f.out_scale = sp.scale; f.act_scale = d.pre_scale; f.c1 = 0.5f * 0.6931471805599453f * sp.inv_std; f.c0 = -sp.mean * sp.inv_std;
        f.B = B; f.T = T; f.Tf = Tl; f.n_fft = sp.n_fft; f.hop = sp.hop;
        const hipError_t fe = (C == sp.n_fft || 2 * C == sp.n_fft) ? wv::launch_spec16(f, st) : hipErrorNotSupported;
        if (fe != hipSuccess && fe != hipErrorNotSupported) return fail(WV_EHIP, std::string("launch_spec16: ") + hipGetErrorString(fe));
        if (fe == hipErrorNotSupported) {
            if ((size_t)B * round_up_int(sp.F, 16) * Tl * 2 > L.act * 4) return fail(WV_ESTATE, "f16 mode: the spectrogram of this scale does not fit its staging buffer");
            LAUNCH(wv::launch_stft_logmag(sa, st));
            LAUNCH(wv::launch_f32_to_c8(P, P16, B, sp.F, Tl, 1.f, 0, st));
            wv::Conv16Args q{};
            q.X = P16; q.w = hs.spec; q.bias = nullptr; q.resid = R[cur]; q.Y = nullptr; q.Yact = A0; q.Yf32 = nullptr;
            q.out_scale = sp.scale; q.act_scale = d.pre_scale; q.B = B; q.M = C; q.Tin = Tl; q.Tout = Tl; q.ks = 1; q.stride = 1; q.pad = 0;
            LAUNCH(wv::launch_conv16(q, st));
        }
        wv::prof::set_role(film ? "enc16.down_film" : "enc16.down");
        const bool last = s + 1 == S, post16 = (int)m->h16.size() > S;
        wv::Conv16Args g{};
        g.X = A0; g.w = hs.down; g.bias = d.dw_b; g.resid = nullptr; g.Y = (last && !post16) ? nullptr : R[cur ^ 1]; g.Yact = nullptr;
        g.Yf32 = (last && !post16) ? (float*)R[0] : nullptr; g.out_scale = 1.f; g.act_scale = 0.f;
        g.B = B; g.M = 2 * C; g.Tin = Tl; g.Tout = (Tl + d.ratio - 1) / d.ratio; g.ks = 2 * d.ratio; g.stride = d.ratio; g.pad = d.ratio;
        if (film) {
            if ((2 * C) % c.freq_bands) return fail(WV_EINVAL, "channels not divisible by freq_bands");
            g.film = film + (size_t)s * c.freq_bands * 2; g.bands = c.freq_bands; g.film_stride = film_stride;
        }
        LAUNCH(wv::launch_conv16(g, st));
        cur ^= 1; Tl = g.Tout; C *= 2;
    }
    *post_done = false; *latent_done = false; *Fr_out = Tl;
    if ((int)m->h16.size() > S) {
        // spec_post (seanet.py:781-795) on the f16 pipe as well: x' = x + scale * (W @ P) from the c8 buffer R[cur], out as ELU(x') in c8 (for
        // the composed conv_post) or as f32 [B, C, Tl] in the stream buffer r[0] (for the exact conv_post; when x sits in R[0] the f32
        // result, twice the bytes, goes through R[1] and is copied over).
        wv::prof::set_role("enc16.spec_post");
        const SpecLayer& sp = m->specs[S];
        const wv_model::H16Stage& hs = m->h16[S];
        if ((T + sp.hop - 1) / sp.hop != Tl) return fail(WV_EINVAL, "internal: STFT frame count != feature length");
        const bool latent = tail == H16_TAIL_LATENT && hs.post.wq && c.last_kernel_size <= 16;
        float* f32out = latent ? nullptr : (float*)R[cur ^ 1];
        bool done = false;
        if (hs.cosw.wq) {
            wv::Spec16Args f{};
            f.wav = x; f.cosw = hs.cosw; f.sinw = hs.sinw; f.cosl = hs.cosl; f.sinl = hs.sinl; f.pw = hs.spec; f.resid = R[cur]; f.Y = nullptr;
            f.Yact = latent ? A0 : nullptr; f.Yf32 = f32out;
            f.out_scale = sp.scale; f.act_scale = latent ? 1.f : 0.f; f.c1 = 0.5f * 0.6931471805599453f * sp.inv_std; f.c0 = -sp.mean * sp.inv_std;
            f.B = B; f.T = T; f.Tf = Tl; f.n_fft = sp.n_fft; f.hop = sp.hop;
            const hipError_t fe = wv::launch_spec16(f, st);
            if (fe == hipSuccess) done = true;
            else if (fe != hipErrorNotSupported) return fail(WV_EHIP, std::string("launch_spec16 (post): ") + hipGetErrorString(fe));
        }
        if (!done && (size_t)B * round_up_int(sp.F, 16) * Tl * 2 <= L.act * 4 && (size_t)B * sp.F * Tl <= L.spec) {
            // any other scale: the exact path's STFT kernel -> P in HBM -> f16 copy -> the 1x1 + add as a k = 1 conv
            wv::StftArgs sa{};
            sa.wav = x; sa.basis_t = sp.basis_t; sa.basis_q = sp.basis_q; sa.side = sp.side; sa.P = P; sa.B = B; sa.T = T;
            sa.Tf = Tl; sa.n_fft = sp.n_fft; sa.hop = sp.hop; sa.F = sp.F; sa.Mp = sp.Mp; sa.mean = sp.mean; sa.inv_std = sp.inv_std;
            LAUNCH(wv::launch_stft_logmag(sa, st));
            LAUNCH(wv::launch_f32_to_c8(P, P16, B, sp.F, Tl, 1.f, 0, st));
            wv::Conv16Args q{};
            q.X = P16; q.w = hs.spec; q.bias = nullptr; q.resid = R[cur]; q.Y = nullptr; q.Yact = latent ? A0 : nullptr; q.Yf32 = f32out;
            q.out_scale = sp.scale; q.act_scale = latent ? 1.f : 0.f; q.B = B; q.M = C; q.Tin = Tl; q.Tout = Tl; q.ks = 1; q.stride = 1; q.pad = 0;
            LAUNCH(wv::launch_conv16(q, st));
            done = true;
        }
        if (done && latent) {
            wv::prof::set_role("enc16.conv_post");
            wv::Conv16Args g{};
            g.X = A0; g.w = hs.post; g.bias = m->post_b; g.resid = nullptr; g.Y = nullptr; g.Yact = nullptr; g.Yf32 = (float*)(ws + L.off_lat);
            g.out_scale = 1.f; g.act_scale = 0.f; g.B = B; g.M = c.dimension; g.Tin = Tl; g.Tout = Tl; g.ks = c.last_kernel_size; g.stride = 1;
            g.pad = c.last_kernel_size - 1;
            LAUNCH(wv::launch_conv16(g, st));
            *post_done = true; *latent_done = true;
            return WV_OK;
        }
        if (done) {
            if ((cur ^ 1) != 0) LAUNCH(hipMemcpyAsync(R[0], R[1], (size_t)B * C * Tl * sizeof(float), hipMemcpyDeviceToDevice, st));
            *post_done = true;
            return WV_OK;
        }
        LAUNCH(wv::launch_c8_to_f32(R[cur], (float*)R[cur ^ 1], B, C, Tl, st));     // the exact path's spec_post takes it from here
        if ((cur ^ 1) != 0) LAUNCH(hipMemcpyAsync(R[0], R[1], (size_t)B * C * Tl * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return WV_OK;
}

static int run_head_model(wv_model* m, const float* x, float* logits, float* mean_prob, int B, int T,
                          void* ws, size_t ws_bytes, void* stream, bool f16 = false) {
    WsLayout L;
    int rc = check_common(m, B, T, ws, ws_bytes, &L);
    if (rc) return rc;
    if (!x || (!logits && !mean_prob)) return fail(WV_EINVAL, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* latent = (float*)(w + L.off_lat);
    int Fr = 0;
    bool post_done = false;
    if (f16) {
        if (m->h16.empty()) return fail(WV_ESTATE, "this model has no f16 plan (ResnetBlock stages of 32 ... 768 channels, k = 5, dilation 1)");
        // mean probabilities only: conv_post and the head run on the f16 pipe as well (L2Norm + composed head GEMM + sigmoid + time mean)
        const wv_config& c = m->cfg;
        const int D = c.dimension, hop = hop_of(c);
        const bool head16 = !logits && (int)m->h16.size() > c.n_strides && m->h16.back().head.wq && D % 16 == 0 && D <= 128 && m->head_nb % 4 == 0 && hop % 32 == 0;
        bool latent_done = false;
        rc = run_encoder_stages_f16(m, x, nullptr, B, T, w, L, st, head16 ? H16_TAIL_LATENT : H16_TAIL_F32, &post_done, &latent_done, &Fr);
        if (rc) return rc;
        if (latent_done) {
            wv::prof::set_role("head16");
            const hipError_t e2 = wv::launch_head16(latent, m->h16.back().head, m->head_bc, mean_prob, B, D, m->head_nb, hop, Fr, T, st);
            if (e2 != hipSuccess) return fail(WV_EHIP, std::string("launch_head16: ") + hipGetErrorString(e2));
            return WV_OK;
        }
    }
    rc = run_encoder(m, x, nullptr, 0, latent, B, T, w, L, st, &Fr, f16 ? m->cfg.n_strides : 0, post_done);
    if (rc) return rc;
    wv::prof::set_role("head");
    wv::HeadArgs h{};
    h.Z = latent; h.wc = m->head_wc; h.bc = m->head_bc; h.logits = logits; h.mean_prob = mean_prob;
    h.B = B; h.D = m->cfg.dimension; h.nb = m->head_nb; h.hop = hop_of(m->cfg); h.Fr = Fr; h.T = T;
    LAUNCH(wv::launch_head(h, st));
    return WV_OK;
}

int wv_detector_forward(wv_model* m, const float* x, float* logits, float* mean_prob, int B, int T,
                        void* ws, size_t ws_bytes, void* stream) {
    if (m && m->cfg.kind != WV_KIND_DETECTOR) return fail(WV_ESTATE, "not a detector model");
    return run_head_model(m, x, logits, mean_prob, B, T, ws, ws_bytes, stream);
}

int wv_detector_forward_f16(wv_model* m, const float* x, float* logits, float* mean_prob, int B, int T,
                            void* ws, size_t ws_bytes, void* stream) {
    if (m && m->cfg.kind != WV_KIND_DETECTOR) return fail(WV_ESTATE, "not a detector model");
    return run_head_model(m, x, logits, mean_prob, B, T, ws, ws_bytes, stream, true);
}

int wv_locator_forward(wv_model* m, const float* x, float* logits, int B, int T, void* ws,
                       size_t ws_bytes, void* stream) {
    if (m && m->cfg.kind != WV_KIND_LOCATOR) return fail(WV_ESTATE, "not a locator model");
    return run_head_model(m, x, logits, nullptr, B, T, ws, ws_bytes, stream);
}

int wv_locator_forward_f16(wv_model* m, const float* x, float* logits, int B, int T, void* ws,
                           size_t ws_bytes, void* stream) {
    if (m && m->cfg.kind != WV_KIND_LOCATOR) return fail(WV_ESTATE, "not a locator model");
    return run_head_model(m, x, logits, nullptr, B, T, ws, ws_bytes, stream, true);
}

// Generator.forward in the f16-operand mode (generator.py:360-423; seanet.py:883-976, 1067-1226): the encoder stages above with FiLM in
// the downsample convs' epilogues, the latent's L2Norm, then the decoder on the same kernels -- the first conv pair as one composed
// conv, every upsample unit as one 2-tap conv over (phase, channel) rows, the ResnetBlocks in one launch each, the tail (f32 sums, tanh)
// on the pre-activated c8 stream.  Message MLP and FiLM scalars are the exact path's (f32).
int wv_generator_forward_f16(wv_model* m, const float* x, const float* msg, int msg_rows, float* out,
                             int add_input, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    WsLayout L;
    int rc = check_common(m, B, T, ws, ws_bytes, &L);
    if (rc) return rc;
    if (m->cfg.kind != WV_KIND_GENERATOR) return fail(WV_ESTATE, "not a generator model");
    if (!x || !msg || !out) return fail(WV_EINVAL, "null tensor");
    if (m->h16.empty() || m->h16_ups.empty()) return fail(WV_ESTATE, "this model has no f16 plan (stages of 32 ... 768 channels, k = 5, dilation 1, the default spec_post)");
    hipStream_t st = (hipStream_t)stream;
    const wv_config& c = m->cfg;
    char* w = (char*)ws;
    float* latent = (float*)(w + L.off_lat);
    float* film = (float*)(w + L.off_film);
    rc = run_film(m, msg, msg_rows, film, B, st);
    if (rc) return rc;
    int Fr = 0;
    bool post_done = false, latent_done = false;
    rc = run_encoder_stages_f16(m, x, film, B, T, w, L, st, H16_TAIL_LATENT, &post_done, &latent_done, &Fr);
    if (rc) return rc;
    if (!latent_done) return fail(WV_ESTATE, "f16 generator: spec_post / conv_post are not on the f16 pipe for this configuration");
    void* R[2] = {w + L.off_r0, w + L.off_r1};
    void* A0 = w + L.off_a0;
    wv::prof::set_role("dec16.l2norm");
    LAUNCH(wv::launch_l2norm_c8(latent, A0, B, c.dimension, Fr, st));
    wv::prof::set_role("dec16.head");
    {
        wv::Conv16Args g{};
        g.X = A0; g.w = m->h16_dec_head; g.bias = m->dec_dw0_b; g.resid = nullptr; g.Y = nullptr; g.Yact = R[0]; g.Yf32 = nullptr;
        g.out_scale = 1.f; g.act_scale = m->ups[0].pre_scale; g.B = B; g.M = m->dec_pw0.M; g.Tin = Fr; g.Tout = Fr; g.ks = c.kernel_size; g.stride = 1;
        g.pad = c.kernel_size - 1;
        LAUNCH(wv::launch_conv16(g, st));
    }
    void* act = R[0];                                            // the activated stream the next unit consumes
    int Tl = Fr;
    for (size_t i = 0; i < m->ups.size(); ++i) {
        const UpLayer& u = m->ups[i];
        const wv_model::H16Up& hu = m->h16_ups[i];
        const bool last_up = i + 1 == m->ups.size();
        const float stage_next = last_up ? m->dec_post : m->ups[i + 1].pre_scale;    // the next upsample's / the tail's ELU(dec_post * y)
        wv::prof::set_role("dec16.upsample");
        void* free1 = act == R[0] ? R[1] : R[0];
        wv::Conv16Args g{};
        g.X = act; g.w = hu.up; g.bias = u.pw_b; g.resid = nullptr; g.Yf32 = nullptr; g.out_scale = 1.f;
        g.B = B; g.M = u.pw.M * u.ratio; g.Tin = Tl; g.Tout = Tl; g.ks = 2; g.stride = 1; g.pad = 1; g.up = u.ratio; g.up_mb = hu.mb;
        if (u.res.empty()) { g.Y = nullptr; g.Yact = free1; g.act_scale = stage_next; }
        else { g.Y = free1; g.Yact = nullptr; g.act_scale = 0.f; }
        LAUNCH(wv::launch_conv16(g, st));
        Tl *= u.ratio;
        void* bufs[3] = {R[0], R[1], A0};
        void* curb = free1;
        wv::prof::set_role("dec16.resblock");
        for (size_t j = 0; j < u.res.size(); ++j) {
            const ResBlock& r = u.res[j];
            const bool last = j + 1 == u.res.size();
            void* dst = nullptr;
            for (void* bb : bufs) if (bb != curb) { dst = bb; break; }
            wv::RhArgs a{};
            a.X = curb; a.pre_scale = r.pre_scale; a.w1 = hu.blocks[j].w1; a.w2 = hu.blocks[j].w2; a.tab1 = hu.blocks[j].tab1; a.tab2 = hu.blocks[j].tab2;
            a.Y = last ? nullptr : dst; a.Yact = last ? dst : nullptr; a.out_scale = r.out_scale; a.act_scale = last ? stage_next : 0.f;
            a.B = B; a.C = u.pw.M; a.T = Tl;
            const hipError_t e = wv::launch_resblock16(a, st);
            if (e != hipSuccess) return fail(e == hipErrorNotSupported ? WV_ESTATE : WV_EHIP, std::string("launch_resblock16 (decoder): ") + hipGetErrorString(e));
            curb = dst;
        }
        act = curb;
    }
    wv::prof::set_role("dec16.tail");
    {
        const hipError_t e = wv::launch_tail16(act, m->last_w, m->last_b, add_input ? x : nullptr, out, B, c.channels_dec, Tl, T, c.last_kernel_size, c.wav_std, st);
        if (e != hipSuccess) return fail(e == hipErrorNotSupported ? WV_ESTATE : WV_EHIP, std::string("launch_tail16: ") + hipGetErrorString(e));
    }
    return WV_OK;
}

int wv_profile_enable(int on) { wv::prof::enable(on != 0); return WV_OK; }
int wv_profile_reset(void) { wv::prof::reset(); return WV_OK; }
int wv_profile_collect(int index, char* name_out, int name_cap, int64_t* launches, double* total_ms,
                       double* flops, double* bytes) {
    static thread_local std::vector<wv::prof::Entry> snap;
    if (index < 0) {                                   // refresh the snapshot, return its size
        snap.resize(4096);
        int n = wv::prof::collect(snap.data(), (int)snap.size());
        snap.resize(std::min<int>(n, 4096));
        return (int)snap.size();
    }
    if (index >= (int)snap.size()) return fail(WV_EINVAL, "profile index out of range");
    const wv::prof::Entry& e = snap[index];
    if (name_out && name_cap > 0) { std::strncpy(name_out, e.name, name_cap - 1); name_out[name_cap - 1] = 0; }
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.ms;
    if (flops) *flops = e.flops;
    if (bytes) *bytes = e.bytes;
    return WV_OK;
}

int wv_model_film(wv_model* m, const float* msg, int msg_rows, float* film, int B, void* stream) {
    if (!m || !m->finalized) return fail(WV_ESTATE, "model not finalized");
    if (!msg || !film || B < 1) return fail(WV_EINVAL, "bad argument");
    return run_film(m, msg, msg_rows, film, B, (hipStream_t)stream);
}

}  // extern "C"
