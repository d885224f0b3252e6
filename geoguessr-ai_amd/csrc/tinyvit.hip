// TinyViT encoder runtime: parameter table, bf16 weight cache, workspace plan, whole-network forward and
// backward as a static schedule of libgg kernels on one HIP stream (no tracing compiler, no per-op host
// round trips).  Architecture: timm TinyVit (TinyViT, Wu et al. 2022) as instantiated by the reference through
// timm.create_model(name, num_classes=0, global_pool="avg") -- models/tinyvit.py:48-53,135; SURVEY.md App. A.
// Activations are NHWC bf16 ([tokens, channels]), so every 1x1 conv / Linear is a plain GEMM and the
// BHWC<->BCHW permutes of TinyVitBlock vanish.
#include <string>
#include <vector>
#include <map>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "graph.h"
#include "../../include/gg.h"

namespace {

struct TensorInfo { std::string name; int64_t offset; int64_t numel; int ndim; int64_t shape[4]; int kind; };
static const float kAttnScale = 0.17677669529663687f;   // head_dim 32 ^ -0.5; the expanded bias tables are divided by it
struct DenseW { int t_w = -1, t_b = -1; int N = 0, K = 0, Kp = 0, Np = 0, taps = 1, cin = 0; int64_t wn = 0, wt = 0;
                int64_t wn3 = -1, wt3 = -1; };     // fp32_split mode: bf16 planes [3][N][Kp] / [3][Kp][Np] of the cached f32 W / W^T (gg_gemm_nt_split3's B operand)
struct BNP { int t_g = -1, t_b = -1; int64_t rm = 0, rv = 0; int cnt = 0; int C = 0; };
struct DwW { int t_w = -1; int C = 0; int64_t taps = 0; };
struct LNP { int t_g = -1, t_b = -1; int C = 0; };
struct ConvBNDense { DenseW w; BNP bn; };
struct ConvBNDw { DwW w; BNP bn; };
struct MBConvL { ConvBNDense c1; ConvBNDw c2; ConvBNDense c3; };
struct MergeL { ConvBNDense c1; ConvBNDw c2; ConvBNDense c3; };
struct BlockL { int t_ab = -1; int64_t bias_full = 0; LNP ln1; DenseW qkv, proj; LNP ln2; DenseW fc1, fc2; ConvBNDw local; };
struct StageL { MergeL merge; std::vector<BlockL> blocks; int C, heads, ws, res; };

struct Model {
    GgTinyVitCfg cfg;
    int es = 2;             // bytes per activation / cached-weight element: 2 (bf16) or 4 (f32, reference-precision mode)
    bool f32 = false;
    bool split = false;     // act_dtype 3 ("fp32_split"): f32 storage and arithmetic; the Linears of the transformer blocks (forward and data gradients) run as fp32-accurate
                            // split products on the bf16 MFMA -- the activation operand split while the kernel stages it, the weight as cached planes
    struct PlaneOf { int64_t w, planes; int rows, ld; };      // cached f32 matrix at wcache offset w ([rows][ld]) -> its bf16 planes [3][rows][ld]
    std::vector<PlaneOf> plane_of;
    std::vector<TensorInfo> tensors;
    int64_t param_floats = 0, buffer_floats = 0, wcache_bytes = 0;
    int num_counters = 0;
    ConvBNDense pe1, pe2;
    std::vector<MBConvL> mb;
    StageL stages[3];
    LNP head;
    int res0;   // spatial size of stage 0 (img/4)
};

static int add_tensor(Model& m, const std::string& name, std::initializer_list<int64_t> shape, int kind) {
    TensorInfo t;
    t.name = name; t.kind = kind; t.ndim = (int)shape.size(); t.numel = 1;
    int i = 0;
    for (int j = 0; j < 4; ++j) t.shape[j] = 1;
    for (auto s : shape) { t.shape[i++] = s; t.numel *= s; }
    if (kind == GG_KIND_PARAM) { t.offset = m.param_floats; m.param_floats += gg_align(t.numel, 8); }
    else if (kind == GG_KIND_BUFFER) { t.offset = m.buffer_floats; m.buffer_floats += gg_align(t.numel, 8); }
    else { t.offset = m.num_counters++; }
    m.tensors.push_back(t);
    return (int)m.tensors.size() - 1;
}
static int64_t wc_alloc(Model& m, int64_t bytes) {
    int64_t o = m.wcache_bytes;
    m.wcache_bytes += gg_align(bytes, 256);
    return o;
}
static void make_dense(Model& m, DenseW& w, const std::string& wname, int N, int cin, int taps, const std::string* bname, bool conv) {
    w.N = N; w.cin = cin; w.taps = taps; w.K = cin * taps; w.Kp = (int)gg_align(w.K, 8); w.Np = (int)gg_align(N, 8);
    if (conv) { const int ks = taps == 9 ? 3 : 1; w.t_w = add_tensor(m, wname, {N, cin, ks, ks}, GG_KIND_PARAM); }
    else w.t_w = add_tensor(m, wname, {N, cin}, GG_KIND_PARAM);
    if (bname) w.t_b = add_tensor(m, *bname, {N}, GG_KIND_PARAM);
    w.wn = wc_alloc(m, (int64_t)N * w.Kp * m.es);
    w.wt = wc_alloc(m, (int64_t)w.Kp * w.Np * m.es);
}
static void make_bn(Model& m, BNP& bn, const std::string& prefix, int C) {
    bn.C = C;
    bn.t_g = add_tensor(m, prefix + ".bn.weight", {C}, GG_KIND_PARAM);
    bn.t_b = add_tensor(m, prefix + ".bn.bias", {C}, GG_KIND_PARAM);
    bn.rm = m.tensors[add_tensor(m, prefix + ".bn.running_mean", {C}, GG_KIND_BUFFER)].offset;
    bn.rv = m.tensors[add_tensor(m, prefix + ".bn.running_var", {C}, GG_KIND_BUFFER)].offset;
    bn.cnt = (int)m.tensors[add_tensor(m, prefix + ".bn.num_batches_tracked", {}, GG_KIND_COUNTER)].offset;
}
static void make_convbn_dense(Model& m, ConvBNDense& c, const std::string& prefix, int cin, int cout, int taps) {
    make_dense(m, c.w, prefix + ".conv.weight", cout, cin, taps, nullptr, true);
    make_bn(m, c.bn, prefix, cout);
    if (m.split) {      // planes of W: the forward of a ConvNorm's dense convolution (plain epilogue + BatchNorm partials) runs as a split product
        c.w.wn3 = wc_alloc(m, (int64_t)3 * c.w.N * c.w.Kp * 2);
        m.plane_of.push_back({c.w.wn, c.w.wn3, c.w.N, c.w.Kp});
        c.w.wt3 = wc_alloc(m, (int64_t)3 * c.w.Kp * c.w.Np * 2);      // ... and of W^T: the data gradients that go through the plain / Linear epilogues
        m.plane_of.push_back({c.w.wt, c.w.wt3, c.w.Kp, c.w.Np});
    }
}
static void make_convbn_dw(Model& m, ConvBNDw& c, const std::string& prefix, int C) {
    c.w.C = C;
    c.w.t_w = add_tensor(m, prefix + ".conv.weight", {C, 1, 3, 3}, GG_KIND_PARAM);
    c.w.taps = wc_alloc(m, (int64_t)9 * C * 4);
    make_bn(m, c.bn, prefix, C);
}
static void make_ln(Model& m, LNP& l, const std::string& prefix, int C) {
    l.C = C;
    l.t_g = add_tensor(m, prefix + ".weight", {C}, GG_KIND_PARAM);
    l.t_b = add_tensor(m, prefix + ".bias", {C}, GG_KIND_PARAM);
}

static int build_model(const GgTinyVitCfg* cfg, Model& m) {
    GG_CHECK(cfg, "tinyvit: null config");
    m.cfg = *cfg;
    GG_CHECK(cfg->act_dtype == 0 || cfg->act_dtype == 1 || cfg->act_dtype == 3, "tinyvit: act_dtype must be 0 (bf16), 1 (f32) or 3 (f32 storage, split-bf16 Linears)");
    m.f32 = cfg->act_dtype != 0; m.split = cfg->act_dtype == 3; m.es = m.f32 ? 4 : 2;
    const int* d = cfg->embed_dims;
    GG_CHECK(cfg->img_size > 0 && cfg->img_size % 32 == 0, "tinyvit: img_size must be a multiple of 32");
    GG_CHECK(cfg->in_chans == 3, "tinyvit: in_chans must be 3");
    for (int s = 0; s < 4; ++s) {
        GG_CHECK(d[s] > 0 && d[s] % 16 == 0, "tinyvit: embed_dims[%d]=%d must be a multiple of 16", s, d[s]);
        GG_CHECK(cfg->depths[s] > 0, "tinyvit: depths[%d] must be > 0", s);
        if (s > 0) GG_CHECK(d[s] == cfg->num_heads[s] * 32, "tinyvit: head_dim must be 32 (dim %d, heads %d)", d[s], cfg->num_heads[s]);
    }
    m.res0 = cfg->img_size / 4;
    make_convbn_dense(m, m.pe1, "patch_embed.conv1", cfg->in_chans, d[0] / 2, 9);
    make_convbn_dense(m, m.pe2, "patch_embed.conv2", d[0] / 2, d[0], 9);
    const int mid = (int)(d[0] * cfg->mbconv_expand_ratio);
    GG_CHECK(mid % 8 == 0, "tinyvit: mbconv mid channels must be a multiple of 8");
    m.mb.resize(cfg->depths[0]);
    for (int i = 0; i < cfg->depths[0]; ++i) {
        const std::string p = "stages.0.blocks." + std::to_string(i);
        make_convbn_dense(m, m.mb[i].c1, p + ".conv1", d[0], mid, 1);
        make_convbn_dw(m, m.mb[i].c2, p + ".conv2", mid);
        make_convbn_dense(m, m.mb[i].c3, p + ".conv3", mid, d[0], 1);
    }
    int res = m.res0;
    for (int s = 1; s < 4; ++s) {
        StageL& st = m.stages[s - 1];
        const int C = d[s], ws = cfg->window_sizes[s], nh = cfg->num_heads[s];
        res /= 2;
        st.C = C; st.heads = nh; st.ws = ws; st.res = res;
        GG_CHECK(res % ws == 0, "tinyvit: stage %d map %d not divisible by window %d (padding path not built)", s, res, ws);
        GG_CHECK(ws <= 32, "tinyvit: window %d unsupported (max 32x32 tokens)", ws);
        const std::string pm = "stages." + std::to_string(s) + ".downsample";
        make_convbn_dense(m, st.merge.c1, pm + ".conv1", d[s - 1], C, 1);
        make_convbn_dw(m, st.merge.c2, pm + ".conv2", C);
        make_convbn_dense(m, st.merge.c3, pm + ".conv3", C, C, 1);
        const int hid = (int)(C * cfg->mlp_ratio);
        st.blocks.resize(cfg->depths[s]);
        for (int i = 0; i < cfg->depths[s]; ++i) {
            BlockL& b = st.blocks[i];
            const std::string p = "stages." + std::to_string(s) + ".blocks." + std::to_string(i);
            b.t_ab = add_tensor(m, p + ".attn.attention_biases", {nh, ws * ws}, GG_KIND_PARAM);
            if (!m.f32 && ws <= 16) {     // expanded bf16 table of the register-resident kernels; the flash kernels read the compact parameter
                const int np_ = gg_attention_padded_tokens(ws * ws);
                b.bias_full = wc_alloc(m, (int64_t)nh * np_ * np_ * 2);
            } else b.bias_full = -1;
            make_ln(m, b.ln1, p + ".attn.norm", C);
            std::string bn = p + ".attn.qkv.bias";
            make_dense(m, b.qkv, p + ".attn.qkv.weight", 3 * C, C, 1, &bn, false);
            bn = p + ".attn.proj.bias";
            make_dense(m, b.proj, p + ".attn.proj.weight", C, C, 1, &bn, false);
            make_ln(m, b.ln2, p + ".mlp.norm", C);
            bn = p + ".mlp.fc1.bias";
            make_dense(m, b.fc1, p + ".mlp.fc1.weight", hid, C, 1, &bn, false);
            bn = p + ".mlp.fc2.bias";
            make_dense(m, b.fc2, p + ".mlp.fc2.weight", C, hid, 1, &bn, false);
            make_convbn_dw(m, b.local, p + ".local_conv", C);
            if (m.split) {      // bf16 planes of W and W^T of the block's four Linears (B operand of gg_gemm_nt_split3_af32: forward and data gradients)
                for (DenseW* w : {&b.qkv, &b.proj, &b.fc1, &b.fc2}) {
                    w->wn3 = wc_alloc(m, (int64_t)3 * w->N * w->Kp * 2);
                    w->wt3 = wc_alloc(m, (int64_t)3 * w->Kp * w->Np * 2);
                    m.plane_of.push_back({w->wn, w->wn3, w->N, w->Kp});
                    m.plane_of.push_back({w->wt, w->wt3, w->Kp, w->Np});
                }
            }
        }
    }
    make_ln(m, m.head, "head.norm", d[3]);
    return 0;
}

// ------------------------------------------------------------------------------------------- workspace plan
struct Region { std::string name; int64_t offset, bytes; };
struct Plan {
    bool training = true;
    bool dry = true;
    std::vector<Region> regs;
    std::map<std::string, int> index;
    int64_t total = 0;
    int64_t max_transient = 0;
    int ring_next = 0;
    int64_t ring_base = 0;
    static constexpr int RING = 10;
    // Training-time temporaries: tensors that live only between their producer and their one consumer because -- under the trainable mask the plan
    // was made for -- no weight gradient reads them (the input of a frozen Linear / depthwise conv, the activation tensors the fused forward never
    // writes).  They alternate between TRING slots of the largest one; `temps` holds their names (gg_tinyvit_activation_info refuses them).
    const uint8_t* mask = nullptr;       // one byte per tensor of the model, or null = everything trainable
    bool fwd_dw_s1 = true, fwd_dw_s2 = true, fwd_pro = true;      // which forward fusions the executor will take (they decide who writes act1 / act2)
    int64_t max_temp = 0, temp_base = 0, gbytes_seen = 0;
    int temp_next = 0;
    bool alias_G = false;
    static constexpr int TRING = 2;
    std::map<std::string, int> temps;
    bool tr(int t) const { return mask == nullptr || mask[t] != 0; }
    int64_t alloc_temp(const std::string& name, int64_t bytes) {
        if (!training) return alloc(name, bytes, true);
        bytes = gg_align(std::max<int64_t>(bytes, 16), 256);
        max_temp = std::max(max_temp, bytes);
        const int64_t off = dry ? 0 : temp_base + (int64_t)(temp_next % TRING) * max_temp;
        temp_next++;
        temps[name] = 1;
        index[name] = (int)regs.size();
        regs.push_back({name, off, bytes});
        return off;
    }
    // act1 behind a fused forward with trainable taps: never written in forward, re-formed by backward right before the tap gradient reads it -- all
    // such layers share ONE region (they are processed one at a time)
    int64_t max_shared = 0, shared_base = 0;
    int64_t alloc_shared(const std::string& name, int64_t bytes) {
        if (!training) return alloc(name, bytes, true);
        bytes = gg_align(std::max<int64_t>(bytes, 16), 256);
        max_shared = std::max(max_shared, bytes);
        temps[name] = 1;
        index[name] = (int)regs.size();
        regs.push_back({name, shared_base, bytes});
        return shared_base;
    }
    // persistent: always gets its own storage; transient activations share a ring in inference mode
    int64_t alloc(const std::string& name, int64_t bytes, bool transient = true) {
        bytes = gg_align(std::max<int64_t>(bytes, 16), 256);
        int64_t off;
        if (!training && transient) {
            max_transient = std::max(max_transient, bytes);
            off = dry ? 0 : ring_base + (int64_t)(ring_next % RING) * max_transient;
            ring_next++;
        } else {
            off = total;
            total += bytes;
        }
        index[name] = (int)regs.size();
        regs.push_back({name, off, bytes});
        return off;
    }
};

struct Act {   // offsets (bytes) of the saved tensors of one ConvNorm
    int64_t y = 0, stat = 0;
};
struct MBAct { int64_t x, a1, a2, out; Act c1, c2, c3; };
struct MergeAct { int64_t a1, a2, out; Act c1, c2, c3; };
struct BlockAct { int64_t x0, a, mean1, rstd1, qkv, o, lse, x1, x2, b, mean2, rstd2, hpre, h, x3; Act local; };
struct Layout {
    int64_t col1, col2, x_pe; Act pe1, pe2;
    std::vector<MBAct> mb;
    MergeAct merge[3];
    std::vector<BlockAct> blocks[3];
    int64_t pooled, mean_h, rstd_h;
    // scratch
    int64_t statpart, bnscratch, lnscratch, colsum, splitk, attn_ds = -1, G[5];
    int64_t foldw, foldb;        // BatchNorm-backward-folded dgrad weights bf16 [Cin][2*Cout] and bias f32 [Cin]
    int64_t gbytes;
};

static int64_t bn_part_floats(int64_t M, int C, int B, int Ho, int Wo, bool dw) {
    const int rows = dw ? std::max(std::max(gg_dwconv_stat_rows(B, Ho, Wo, C, 1), gg_dwconv_stat_rows(B, Ho, Wo, C, 2)),
                                   std::max(gg_dwconv_tiled_stat_rows(B, Ho), std::max(gg_dwconv_fused_stat_rows(B, Ho, Wo, C, 1), gg_dwconv_fwd_fused_stat_rows(B, Ho, Wo, C, 1))))
                        : gg_gemm_colstats_rows((int)M);
    return (int64_t)gg_stat_rows_capacity(rows) * 2 * C;
}

static void plan_build(const Model& m, int B, Plan& p, Layout& L) {
    const GgTinyVitCfg& c = m.cfg;
    const int* d = c.embed_dims;
    const int H1 = c.img_size / 2, H0 = m.res0;
    const int64_t M1 = (int64_t)B * H1 * H1, M0 = (int64_t)B * H0 * H0;
    int64_t gmax = 0, statmax = 0, bnsmax = 0, lnsmax = 0, csmax = 0;
    const int64_t es = m.es;
    auto track = [&](int64_t elems) { gmax = std::max(gmax, elems * es); };
    auto bnreg = [&](const std::string& n, Act& a, int64_t M, int C, bool dw, int Bn, int Ho, int Wo) {
        a.y = p.alloc(n + ".y", M * C * es);
        a.stat = p.alloc(n + ".stat", 2 * C * 4, false);
        statmax = std::max(statmax, bn_part_floats(M, C, Bn, Ho, Wo, dw) * 4);
        if (dw) statmax = std::max(statmax, (int64_t)gg_stat_rows_capacity(std::max(gg_dwconv_f32_stat_rows(Bn, Ho, Wo, C, 1), gg_dwconv_f32_stat_rows(Bn, Ho, Wo, C, 2))) * 2 * C * 4);
        statmax = std::max(statmax, (int64_t)gg_stat_rows_capacity(4096) * 2 * C * 4);      // stride-2 fused data gradient partials
        bnsmax = std::max(bnsmax, gg_bn_bwd_scratch_floats(M, C) * 4);
        track(M * C);
    };
    L.col1 = p.alloc("patch_embed.col1", M1 * 32 * es);
    bnreg("patch_embed.conv1", L.pe1, M1, d[0] / 2, false, B, H1, H1);
    L.col2 = p.alloc("patch_embed.col2", M0 * m.pe2.w.Kp * es);
    track(M0 * m.pe2.w.Kp); track(M1 * 32);
    bnreg("patch_embed.conv2", L.pe2, M0, d[0], false, B, H0, H0);
    L.x_pe = p.alloc("patch_embed.out", M0 * d[0] * es);
    const int mid = (int)(d[0] * c.mbconv_expand_ratio);
    L.mb.resize(m.mb.size());
    int64_t prev = L.x_pe;
    for (size_t i = 0; i < m.mb.size(); ++i) {
        const std::string n = "stages.0.blocks." + std::to_string(i);
        MBAct& a = L.mb[i];
        a.x = prev;
        // act1 = GELU(BN1(y1)): the fused depthwise forward never writes it (backward re-forms it as a temporary when conv2's taps train); unfused, it
        // is read again only by conv2's weight gradient.  act2 = GELU(BN2(y2)): conv3's input, kept only for conv3's weight gradient.
        const MBConvL& ml = m.mb[i];
        bnreg(n + ".conv1", a.c1, M0, mid, false, B, H0, H0);
        a.a1 = !p.tr(ml.c2.w.t_w) ? p.alloc_temp(n + ".act1", M0 * mid * es) : p.fwd_dw_s1 ? p.alloc_shared(n + ".act1", M0 * mid * es) : p.alloc(n + ".act1", M0 * mid * es);
        bnreg(n + ".conv2", a.c2, M0, mid, true, B, H0, H0);
        a.a2 = !p.tr(ml.c3.w.t_w) ? p.alloc_temp(n + ".act2", M0 * mid * es) : p.alloc(n + ".act2", M0 * mid * es);
        bnreg(n + ".conv3", a.c3, M0, d[0], false, B, H0, H0);
        a.out = p.alloc(n + ".out", M0 * d[0] * es);
        prev = a.out;
    }
    int res = H0;
    int64_t Mprev = M0;
    for (int s = 0; s < 3; ++s) {
        const StageL& st = m.stages[s];
        const int C = st.C;
        const int64_t M = (int64_t)B * st.res * st.res;
        const std::string n = "stages." + std::to_string(s + 1) + ".downsample";
        MergeAct& ma = L.merge[s];
        bnreg(n + ".conv1", ma.c1, Mprev, C, false, B, res, res);
        ma.a1 = !p.tr(st.merge.c2.w.t_w) ? p.alloc_temp(n + ".act1", Mprev * C * es) : p.fwd_dw_s2 ? p.alloc_shared(n + ".act1", Mprev * C * es) : p.alloc(n + ".act1", Mprev * C * es);
        bnreg(n + ".conv2", ma.c2, M, C, true, B, st.res, st.res);
        ma.a2 = !p.tr(st.merge.c3.w.t_w) ? p.alloc_temp(n + ".act2", M * C * es) : p.alloc(n + ".act2", M * C * es);
        bnreg(n + ".conv3", ma.c3, M, C, false, B, st.res, st.res);
        ma.out = p.alloc(n + ".out", M * C * es);
        prev = ma.out;
        const int hid = (int)(C * c.mlp_ratio);
        L.blocks[s].resize(st.blocks.size());
        for (size_t i = 0; i < st.blocks.size(); ++i) {
            const std::string bn = "stages." + std::to_string(s + 1) + ".blocks." + std::to_string(i);
            BlockAct& a = L.blocks[s][i];
            const BlockL& bl = st.blocks[i];
            a.x0 = prev;
            // inputs of frozen Linears / of the frozen local conv are consumed once, right after they are written: ln1 -> qkv, x1 -> local_conv,
            // ln2 -> fc1, GELU(fc1) -> fc2 (the backward of a frozen block needs x0 / x2 / means for the LayerNorms, qkv / o / lse for the attention,
            // local_conv.y for its BatchNorm and the fc1 pre-activation for GELU'): 7 C of the block's 19 C floats per token
            a.a = !p.tr(bl.qkv.t_w) ? p.alloc_temp(bn + ".ln1", M * C * es) : p.alloc(bn + ".ln1", M * C * es);
            a.mean1 = p.alloc(bn + ".mean1", M * 4);
            a.rstd1 = p.alloc(bn + ".rstd1", M * 4);
            a.qkv = p.alloc(bn + ".qkv", M * 3 * C * es);
            a.o = p.alloc(bn + ".attn.out", M * C * es);
            a.lse = p.alloc(bn + ".attn.lse", M * st.heads * 4);
            a.x1 = !p.tr(bl.local.w.t_w) ? p.alloc_temp(bn + ".x1", M * C * es) : p.alloc(bn + ".x1", M * C * es);
            bnreg(bn + ".local_conv", a.local, M, C, true, B, st.res, st.res);
            a.x2 = p.alloc(bn + ".x2", M * C * es);
            a.b = !p.tr(bl.fc1.t_w) ? p.alloc_temp(bn + ".ln2", M * C * es) : p.alloc(bn + ".ln2", M * C * es);
            a.mean2 = p.alloc(bn + ".mean2", M * 4);
            a.rstd2 = p.alloc(bn + ".rstd2", M * 4);
            a.hpre = p.alloc(bn + ".fc1.pre", M * hid * es);
            a.h = !p.tr(bl.fc2.t_w) ? p.alloc_temp(bn + ".fc1.act", M * hid * es) : p.alloc(bn + ".fc1.act", M * hid * es);
            a.x3 = p.alloc(bn + ".out", M * C * es);
            prev = a.x3;
            track(M * hid); track(M * 3 * C);
            lnsmax = std::max(lnsmax, gg_layernorm_bwd_scratch_floats(M, C) * 4);
            csmax = std::max(csmax, gg_colsum_scratch_floats((int)M, hid) * 4);
            csmax = std::max(csmax, gg_colsum_scratch_floats((int)M, 3 * C) * 4);
        }
        res = st.res;
        Mprev = M;
    }
    L.pooled = p.alloc("head.pooled", (int64_t)B * d[3] * 4, false);
    L.mean_h = p.alloc("head.mean", (int64_t)B * 4, false);
    L.rstd_h = p.alloc("head.rstd", (int64_t)B * 4, false);
    lnsmax = std::max(lnsmax, gg_layernorm_bwd_scratch_floats(B, d[3]) * 4);
    L.statpart = p.alloc("scratch.statpart", statmax, false);
    if (p.training) {
        // dw wgrad scratch may exceed the BN scratch
        int64_t dwmax = 0;
        dwmax = std::max(dwmax, gg_dwconv_wgrad_scratch_floats(B, H0, H0, mid, 1) * 4);
        dwmax = std::max(dwmax, gg_dwconv_f32_wgrad_scratch_floats(B, H0, H0, mid, 1) * 4);
        for (int s = 0; s < 3; ++s) {
            const int rin = s == 0 ? H0 : m.stages[s - 1].res;
            dwmax = std::max(dwmax, gg_dwconv_wgrad_scratch_floats(B, rin, rin, m.stages[s].C, 2) * 4);
            dwmax = std::max(dwmax, gg_dwconv_wgrad_scratch_floats(B, m.stages[s].res, m.stages[s].res, m.stages[s].C, 1) * 4);
            dwmax = std::max(dwmax, gg_dwconv_f32_wgrad_scratch_floats(B, rin, rin, m.stages[s].C, 2) * 4);
            dwmax = std::max(dwmax, gg_dwconv_f32_wgrad_scratch_floats(B, m.stages[s].res, m.stages[s].res, m.stages[s].C, 1) * 4);
        }
        L.bnscratch = p.alloc("scratch.bn", std::max(bnsmax, dwmax), false);
        L.lnscratch = p.alloc("scratch.ln", lnsmax, false);
        L.colsum = p.alloc("scratch.colsum", std::max<int64_t>(csmax, 1024), false);
        L.splitk = p.alloc("scratch.splitk", (int64_t)128 << 20, false);      // gg_gemm_tn_f32_splits sizes its slabs against this
        if (m.f32) {     // dS hand-off between the two passes of the flash attention backward (GgAttnArgs.ds_scratch): the largest stage decides
            // only windows beyond 256 tokens use it (the single-pass backward keeps dS on the CU), and only while it stays a small part of the workspace:
            // at most 4 GB and 1/8 of what is planned so far (24 x 24 windows: 16 MB per image; 32 x 32: 50 MB per image -- there both passes recompute)
            int64_t dsmax = 0;
            for (int s = 1; s < 4; ++s) {
                const auto& st = m.stages[s - 1];
                const int nw = B * (st.res / st.ws) * (st.res / st.ws);
                if (!gg_attention_flash_single_pass(st.ws * st.ws, 32, st.ws, 1))
                    dsmax = std::max(dsmax, gg_attention_flash_ds_scratch_floats(nw, st.heads, st.ws * st.ws) * 4);
            }
            if (dsmax > 0 && dsmax <= ((int64_t)4 << 30) && dsmax <= p.total / 8) L.attn_ds = p.alloc("scratch.attn_ds", dsmax, false);
        }
        int64_t fold = (int64_t)d[0] * 2 * mid;
        for (int s = 0; s < 3; ++s) fold = std::max(fold, (int64_t)(s == 0 ? d[0] : m.stages[s - 1].C) * 2 * m.stages[s].C);
        L.foldw = p.alloc("scratch.foldw", fold * es, false);
        L.foldb = p.alloc("scratch.foldb", 4096 * 4, false);
        L.gbytes = gg_align(gmax, 256);
        p.gbytes_seen = L.gbytes;
        // the forward's temporaries are dead when backward starts and the first two gradient ping-pong buffers are untouched until then: they share
        // the ring's slots -- unless backward itself re-forms a temporary (act1 for a trainable depthwise conv behind a fused forward)
        for (int i = 0; i < 5; ++i) {
            if (i < Plan::TRING && p.alias_G && !p.dry) {
                L.G[i] = p.temp_base + (int64_t)i * p.max_temp;
                p.index["scratch.G" + std::to_string(i)] = (int)p.regs.size();
                p.regs.push_back({"scratch.G" + std::to_string(i), L.G[i], L.gbytes});
            } else L.G[i] = p.alloc("scratch.G" + std::to_string(i), L.gbytes, false);
        }
    }
}
// which forward fusions the executor takes (mirrors Exec::exec_init): they decide who writes the MBConv / PatchMerging activation tensors
static void plan_flags(const Model& m, Plan& p) {
    bool fuse_dw = gg_dev_env("GG_FUSE_DW") != nullptr, s1 = gg_dev_env("GG_NO_FUSE_DW_S1") == nullptr, s2 = gg_dev_env("GG_NO_FUSE_DW_S2") == nullptr;
    bool pro = gg_dev_env("GG_NO_PRO") == nullptr;
    if (m.f32 && gg_dev_env("GG_F32_NO_FUSE")) fuse_dw = s1 = s2 = pro = false;
    if (m.f32) fuse_dw = false;
    p.fwd_dw_s1 = fuse_dw || s1; p.fwd_dw_s2 = fuse_dw || s2; p.fwd_pro = pro;
}
static void plan_make(const Model& m, int B, bool training, Plan& p, Layout& L, const uint8_t* mask = nullptr) {
    p.training = training;
    p.dry = true;
    p.mask = training ? mask : nullptr;
    plan_flags(m, p);
    plan_build(m, B, p, L);
    if (training && (p.max_temp > 0 || p.max_shared > 0)) {
        Plan q;                                    // second pass: the temporaries' ring (= the first two gradient buffers) first, then the shared act1 region, then the rest
        q.training = true; q.dry = false; q.mask = p.mask; q.temp_base = 0;
        q.fwd_dw_s1 = p.fwd_dw_s1; q.fwd_dw_s2 = p.fwd_dw_s2; q.fwd_pro = p.fwd_pro;
        q.alias_G = p.max_temp > 0;
        q.max_temp = q.alias_G ? std::max(p.max_temp, p.gbytes_seen) : 0;
        q.shared_base = (int64_t)Plan::TRING * q.max_temp;
        q.total = q.shared_base + p.max_shared;
        Layout L2;
        plan_build(m, B, q, L2);
        p = q; L = L2;
    }
    if (!training) {
        Plan q;
        q.training = false; q.dry = false; q.max_transient = p.max_transient;
        q.ring_base = p.total;                     // persistent regions first, ring after
        Layout L2;
        plan_build(m, B, q, L2);
        q.total = q.ring_base + (int64_t)Plan::RING * q.max_transient;
        p = q; L = L2;
    }
}

// ------------------------------------------------------------------------------------------- execution context
typedef char act_t;        // an activation / cached-weight element of the model's storage type (bf16 or f32): only ever passed on
struct Exec {
    const Model* m; const Layout* L;
    int B; bool training;
    const float* params; float* buffers; int64_t* counters;
    const char* wc; char* ws; hipStream_t st;
    const float* drop;   // [slots][B] or null
    float* grads; const uint8_t* trainable;
    GgStageDoneFn stage_done = nullptr; void* stage_user = nullptr;     // host callback: the gradient of a stage is fully enqueued
    void done(int stage) const { if (stage_done) stage_done(stage, stage_user); }
    // Fusing BatchNorm+GELU of the producer into the depthwise conv's input load removes one [M,C] write+read, but each input is
    // loaded (and transformed) by its three neighbouring columns: the erf work triples and the conv turns VALU-bound
    // (measured at 1024 images: +4.7 ms conv vs -2.7 ms elementwise) -> off by default.
    bool f32 = false;       // reference-precision mode: f32 activations / cached weights, f32 MFMA, every fusion below off (set by exec_init)
    bool fuse_dw = gg_dev_env("GG_FUSE_DW") != nullptr;
    // the stride-2 depthwise conv of PatchMerging stages its input tile in LDS: BatchNorm1 + GELU are applied once per staged element
    // (17x17 inputs per 8x8 outputs = 1.13x), the apply pass and the activation tensor disappear
    bool fuse_dw_s2 = gg_dev_env("GG_NO_FUSE_DW_S2") == nullptr;
    // MBConv.conv2 likewise through the 4-columns-per-thread stride-1 kernel (1.5 BatchNorm+GELU evaluations per input element)
    bool fuse_dw_s1 = gg_dev_env("GG_NO_FUSE_DW_S1") == nullptr;
    // Frozen depthwise taps: the data gradient forms BatchNorm backward's apply step (dy = c0*dz + c1*y + c2) while it loads its
    // input, and (MBConv) emits dz = da*act'(BN(y)) + the reduce sums of the ConvNorm in front: apply and reduce passes and
    // the dy / da tensors disappear.  GG_NO_FUSE_BNBWD=1 / GG_NO_FUSE_BNBWD_EPI=1 restore the separate passes.
    bool fuse_bnbwd = gg_dev_env("GG_NO_FUSE_BNBWD") == nullptr;
    bool fuse_bnbwd_epi = gg_dev_env("GG_NO_FUSE_BNBWD_EPI") == nullptr;
    // Frozen ConvNorm chains: BatchNorm backward's reduce rides in the epilogue of the conv dgrad that produces its input
    // gradient, and its apply step is folded into the weights of the 1x1 dgrad that consumes its output gradient.
    bool fuse_bngemm = gg_dev_env("GG_NO_BNGEMM") == nullptr;
    // MBConv conv3 applies BatchNorm2 + GELU in its A prologue (one N tile: each element is transformed once)
    bool fuse_pro = gg_dev_env("GG_NO_PRO") == nullptr;
    bool fuse_lncol = gg_dev_env("GG_NO_LN_COLSUM") == nullptr;   // norm2's backward leaves local_conv's BatchNorm-backward column sums (frozen blocks)
    bool fuse_lnbn = true;           // fp32: local_conv's BatchNorm apply inside norm2 (off with GG_F32_NO_FUSE)
    const float* P(int t) const { return params + m->tensors[t].offset; }
    float* Gd(int t) const { return grads + m->tensors[t].offset; }
    bool tr(int t) const { return trainable == nullptr || trainable[t] != 0; }
    void exec_init() {
        f32 = m->f32;
        // reference-precision mode: the same fusions where an f32 twin exists (MBConv / PatchMerging forward, the stride-1 data gradients, the
        // GEMM-side BatchNorm epilogue / prologues); GG_F32_NO_FUSE=1 runs every BatchNorm pass on its own (the schedule the fusions are tested against)
        if (f32 && gg_dev_env("GG_F32_NO_FUSE")) fuse_dw = fuse_dw_s2 = fuse_dw_s1 = fuse_bnbwd = fuse_bnbwd_epi = fuse_bngemm = fuse_pro = fuse_lnbn = fuse_lncol = false;
        if (f32) fuse_dw = false;
    }
    act_t* A(int64_t off) const { return reinterpret_cast<act_t*>(ws + off); }
    float* F(int64_t off) const { return reinterpret_cast<float*>(ws + off); }
    const act_t* Wn(const DenseW& w) const { return reinterpret_cast<const act_t*>(wc + w.wn); }
    const act_t* Wt(const DenseW& w) const { return reinterpret_cast<const act_t*>(wc + w.wt); }
    const float* Taps(const DwW& w) const { return reinterpret_cast<const float*>(wc + w.taps); }
    const float* dropv(int slot) const { return drop ? drop + (int64_t)slot * B : nullptr; }
};

// conv dgrad with the BatchNorm-backward reduce of the ConvNorm it feeds as epilogue: dz = (dY . W) * act'(BN(y)), partial
// column sums -> statpart
static int gemm_bnbwd(const Exec& e, const act_t* dY, int64_t ldy, const act_t* Wt, int64_t ldw, act_t* dz, int64_t M, int N, int K,
                      const BNP& bn, const Act& a, int act) {
    GgGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dY; g.lda = ldy; g.B = Wt; g.ldb = ldw; g.C = dz; g.ldc = N; g.M = (int)M; g.N = N; g.K = K;
    g.bn_y = e.A(a.y); g.bn_stat = e.F(a.stat); g.bn_gamma = e.P(bn.t_g); g.bn_beta = e.P(bn.t_b); g.bn_act = act;
    g.colstats = e.F(e.L->statpart);
    return e.f32 ? gg_gemm_nt_f32(&g, e.st) : gg_gemm_nt(&g, e.st);
}
// 1x1-conv dgrad straight from (dz, y): BatchNorm backward's apply step is folded into the weights (gg_bn_bwd_fold_weights)
static int gemm_folded_dgrad(const Exec& e, const DenseW& w, const act_t* dz, const act_t* y, const float* coef, const float* stat,
                             act_t* dx, int64_t M, const act_t* residual) {
    const int Cout = w.N, Cin = w.K;
    if (e.f32) {
        // f32: a doubled contraction would cost real MFMA time (1/16 of the bf16 rate); instead the apply step dy = c0*dz + c1*y + c2 is formed
        // from the two sources while the register-staged GEMM stages its A tile (GgGemmArgs.A2 + a_bn_stat = coef)
        GgGemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = dz; g.lda = Cout; g.A2 = y; g.a_bn_stat = coef; g.B = e.Wt(w); g.ldb = w.Np; g.C = dx; g.ldc = Cin;
        g.M = (int)M; g.N = Cin; g.K = Cout; g.residual = residual; g.ldr = Cin;
        return gg_gemm_nt_f32(&g, e.st);
    }
    GG_TRY(gg_bn_bwd_fold_weights(e.P(w.t_w), coef, stat, Cout, Cin, e.A(e.L->foldw), e.F(e.L->foldb), e.st));
    GgGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dz; g.lda = Cout; g.A2 = y; g.k_split = Cout; g.B = e.A(e.L->foldw); g.ldb = 2 * Cout; g.C = dx; g.ldc = Cin;
    g.M = (int)M; g.N = Cin; g.K = 2 * Cout; g.bias = e.F(e.L->foldb); g.residual = residual; g.ldr = Cin;
    return gg_gemm_nt(&g, e.st);
}
static int gemm(const Exec& e, const act_t* A, int64_t lda, const act_t* Bm, int64_t ldb, void* C, int64_t ldc, int64_t M, int N, int K,
                const float* bias = nullptr, int act = 0, void* preact = nullptr, const float* rowscale = nullptr, int rps = 0,
                const act_t* residual = nullptr, float* colstats = nullptr, const act_t* dact_pre = nullptr, int dact = 0) {
    // (launch-bound sizes stay on the f32 GEMM, whose 64 x 64 tiles and split-K fill the chip where a 256- / 128-row split tile would leave most CUs idle:
    // the split form needs at least half the CUs' worth of tiles)
    const int64_t split_tiles = ((M + (K >= 384 ? 255 : 127)) / (K >= 384 ? 256 : 128)) * ((N + 127) / 128);
    // (dev: GG_SPLIT_MIN_TILES lowers the threshold so that the oracle gate can take every split route at a batch the CPU oracle finishes in seconds)
    static const char* min_tiles_env = gg_dev_env("GG_SPLIT_MIN_TILES");
    static const int64_t min_tiles = min_tiles_env ? atoll(min_tiles_env) : 128;
    if (e.m->split && split_tiles >= min_tiles && (K & 7) == 0 && (lda & 3) == 0 && (!colstats || !(bias || act || preact || rowscale || residual || dact_pre))) {
        // fp32_split mode: a Linear whose weight operand has cached planes runs as a split product (A = the f32 activation itself, split in the kernel's loader)
        const int64_t off = reinterpret_cast<const char*>(Bm) - e.wc;
        for (const Model::PlaneOf& po : e.m->plane_of) {
            if (po.w != off) continue;
            if (po.ld != ldb || po.rows < N) break;
            GgSplit3Args g;
            memset(&g, 0, sizeof(g));
            g.b_planes = e.wc + po.planes; g.ldb = ldb; g.M = (int)M; g.N = N; g.K = K; g.C = (float*)C; g.ldc = ldc;
            g.bias = bias; g.act = act; g.preact = (float*)preact; g.rowscale = rowscale; g.rows_per_scale = rps; g.residual = (const float*)residual; g.ldr = ldc;
            g.dact_preact = (const float*)dact_pre; g.dact = dact;
            return gg_gemm_nt_split3_af32_stats(&g, (const float*)A, lda, (int64_t)po.rows * po.ld, colstats, e.st);
        }
    }
    GgGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K;
    g.bias = bias; g.act = act; g.preact = preact; g.rowscale = rowscale; g.rows_per_scale = rps;
    g.residual = residual; g.ldr = ldc; g.colstats = colstats; g.dact_preact = dact_pre; g.dact = dact;
    return e.f32 ? gg_gemm_nt_f32(&g, e.st) : gg_gemm_nt(&g, e.st);
}

// BatchNorm statistics for a ConvNorm whose producer wrote `nparts` partial rows into statpart
static int bn_stats(const Exec& e, const BNP& bn, const Act& a, int nparts, int64_t count) {
    if (e.training) {
        GG_TRY(gg_bn_finalize(e.F(e.L->statpart), nparts, bn.C, count, e.m->cfg.bn_eps, e.m->cfg.bn_momentum, e.F(a.stat),
                              e.buffers + bn.rm, e.buffers + bn.rv, e.st));
    } else {
        GG_TRY(gg_bn_eval_stat(e.buffers + bn.rm, e.buffers + bn.rv, bn.C, e.m->cfg.bn_eps, e.F(a.stat), e.st));
    }
    return 0;
}
// dense ConvNorm: y = A . Wn^T (+ partial stats) ; stat
static int conv_dense_fwd(const Exec& e, const ConvBNDense& c, const Act& a, const act_t* A, int64_t lda, int64_t M) {
    float* part = e.training ? e.F(e.L->statpart) : nullptr;
    GG_TRY(gemm(e, A, lda, e.Wn(c.w), c.w.Kp, e.A(a.y), c.w.N, M, c.w.N, c.w.Kp, nullptr, 0, nullptr, nullptr, 0, nullptr, part));
    return bn_stats(e, c.bn, a, gg_gemm_colstats_rows((int)M), M);
}
// dense ConvNorm whose input is act(BN(prev.y)) of the preceding ConvNorm, formed while the GEMM stages its A tile
static int conv_dense_fwd_pro(const Exec& e, const ConvBNDense& c, const Act& a, const BNP& prev_bn, const Act& prev, int in_act, int64_t M) {
    // fp32_split mode: the same fusion on the split kernels (the transform rides on the loader, in front of the split; 256-row tiles: K >= 384)
    static const char* min_tiles_env = gg_dev_env("GG_SPLIT_MIN_TILES");
    const int64_t min_tiles = min_tiles_env ? atoll(min_tiles_env) : 128;
    static const char* nopro_env = gg_dev_env("GG_SPLIT3_NO_PRO");          // dev: the f32-MFMA prologue GEMM in the split mode too
    if (e.m->split && !nopro_env && c.w.Kp >= 384 && c.w.Kp <= 1024 && (c.w.Kp & 7) == 0 && ((M + 255) / 256) * ((c.w.N + 127) / 128) >= min_tiles) {
        const int64_t off = reinterpret_cast<const char*>(e.Wn(c.w)) - e.wc;
        for (const Model::PlaneOf& po : e.m->plane_of) {
            if (po.w != off) continue;
            if (po.ld != c.w.Kp || po.rows < c.w.N) break;
            GgSplit3Args g;
            memset(&g, 0, sizeof(g));
            g.b_planes = e.wc + po.planes; g.ldb = c.w.Kp; g.M = (int)M; g.N = c.w.N; g.K = c.w.Kp; g.C = (float*)e.A(a.y); g.ldc = c.w.N;
            GG_TRY(gg_gemm_nt_split3_af32_pro(&g, (const float*)e.A(prev.y), c.w.Kp, (int64_t)po.rows * po.ld, e.F(prev.stat), e.P(prev_bn.t_g), e.P(prev_bn.t_b), in_act,
                                              e.training ? e.F(e.L->statpart) : nullptr, e.st));
            return bn_stats(e, c.bn, a, gg_gemm_colstats_rows((int)M), M);
        }
    }
    GgGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = e.A(prev.y); g.lda = c.w.Kp; g.B = e.Wn(c.w); g.ldb = c.w.Kp; g.C = e.A(a.y); g.ldc = c.w.N;
    g.M = (int)M; g.N = c.w.N; g.K = c.w.Kp;
    g.colstats = e.training ? e.F(e.L->statpart) : nullptr;
    g.a_bn_stat = e.F(prev.stat); g.a_bn_gamma = e.P(prev_bn.t_g); g.a_bn_beta = e.P(prev_bn.t_b); g.a_bn_act = in_act;
    GG_TRY(e.f32 ? gg_gemm_nt_f32(&g, e.st) : gg_gemm_nt(&g, e.st));
    return bn_stats(e, c.bn, a, gg_gemm_colstats_rows((int)M), M);
}
static int conv_dw_fwd(const Exec& e, const ConvBNDw& c, const Act& a, const act_t* x, int B, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    float* part = e.training ? e.F(e.L->statpart) : nullptr;
    if (e.f32) {
        GG_TRY(gg_dwconv3x3_fwd_f32((const float*)x, e.Taps(c.w), (float*)e.A(a.y), B, H, W, c.w.C, stride, part, e.st));
        return bn_stats(e, c.bn, a, gg_dwconv_f32_stat_rows(B, Ho, Wo, c.w.C, stride), (int64_t)B * Ho * Wo);
    }
    GG_TRY(gg_dwconv3x3_fwd(x, e.Taps(c.w), e.A(a.y), B, H, W, c.w.C, stride, part, e.st));
    return bn_stats(e, c.bn, a, gg_dwconv_stat_rows(B, Ho, Wo, c.w.C, stride), (int64_t)B * Ho * Wo);
}
// depthwise ConvNorm whose input is act(BN(prev.y)) of the preceding ConvNorm, formed on the fly while staging
static int conv_dw_fwd_fused(const Exec& e, const ConvBNDw& c, const Act& a, const BNP& prev_bn, const Act& prev, int in_act, int B, int H,
                             int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    float* part = e.training ? e.F(e.L->statpart) : nullptr;
    if (e.f32) {
        GG_TRY(gg_dwconv3x3_fwd_fused_f32((const float*)e.A(prev.y), e.F(prev.stat), e.P(prev_bn.t_g), e.P(prev_bn.t_b), in_act, e.Taps(c.w),
                                          (float*)e.A(a.y), B, H, W, c.w.C, stride, part, e.st));
        return bn_stats(e, c.bn, a, gg_dwconv_f32_stat_rows(B, Ho, Wo, c.w.C, stride), (int64_t)B * Ho * Wo);
    }
    GG_TRY(gg_dwconv3x3_fwd_fused(e.A(prev.y), e.F(prev.stat), e.P(prev_bn.t_g), e.P(prev_bn.t_b), in_act, e.Taps(c.w), e.A(a.y), B, H, W,
                                  c.w.C, stride, part, e.st));
    return bn_stats(e, c.bn, a, gg_dwconv_fwd_fused_stat_rows(B, H, W, c.w.C, stride), (int64_t)B * Ho * Wo);
}
static int bn_apply(const Exec& e, const BNP& bn, const Act& a, int64_t M, int act, act_t* out, const act_t* residual = nullptr,
                    const float* rowscale = nullptr, int rps = 0) {
    if (e.f32) return gg_bn_apply_f32((const float*)e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, (const float*)residual,
                                      rowscale, rps, (float*)out, e.st);
    return gg_bn_apply(e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, residual, rowscale, rps, out, e.st);
}
// window attention arguments of one TinyVitBlock (forward fields; the caller adds the backward ones)
static void attn_args(const Exec& e, const StageL& st, const BlockL& l, const BlockAct& a, int B, GgAttnArgs& at) {
    const int C = st.C;
    memset(&at, 0, sizeof(at));
    at.qkv = e.A(a.qkv); at.ld = 3 * C; at.q_off = 0; at.k_off = 32; at.v_off = 64; at.head_stride = 96; at.head_dim = 32;
    at.num_heads = st.heads; at.tokens_per_window = st.ws * st.ws;
    at.num_windows = B * (st.res / st.ws) * (st.res / st.ws);
    at.window_size = st.ws; at.map_h = st.res; at.map_w = st.res;
    at.bias = l.bias_full >= 0 ? e.wc + l.bias_full : nullptr;      // expanded bf16 table (register-resident kernels)
    at.bias_table = e.P(l.t_ab);                                     // compact f32 parameter (online-softmax kernels)
    at.scale = kAttnScale;
    at.out = e.A(a.o); at.ldo = C; at.lse = e.F(a.lse);
}

// ------------------------------------------------------------------------------------------- forward
static int forward_impl(Exec& e, const float* x, float* out) {
    const Model& m = *e.m; const Layout& L = *e.L; const GgTinyVitCfg& c = m.cfg;
    const int B = e.B;
    const int* d = c.embed_dims;
    const int H = c.img_size, H1 = H / 2, H0 = m.res0;
    const int64_t M1 = (int64_t)B * H1 * H1, M0 = (int64_t)B * H0 * H0;
    // (num_batches_tracked counters are bumped by the host shim: they are int64 bookkeeping, not arithmetic)

    // PatchEmbed: conv3x3 s2 + BN + GELU, conv3x3 s2 + BN
    if (e.f32) GG_TRY(gg_im2col_nchw3_f32_f32(x, (float*)e.A(L.col1), B, H, H, 2, e.st));
    else GG_TRY(gg_im2col_nchw3_f32(x, e.A(L.col1), B, H, H, 2, e.st));
    GG_TRY(conv_dense_fwd(e, m.pe1, L.pe1, e.A(L.col1), 32, M1));
    GG_CHECK(m.pe2.w.Kp == m.pe2.w.K, "tinyvit: patch_embed.conv2 K=%d must be a multiple of 8", m.pe2.w.K);
    // BN1 + GELU ride on conv2's im2col gather: the activation tensor (M1 x 48, the largest of the model) is never written
    if (e.f32) GG_TRY(gg_im2col_nhwc_f32((const float*)e.A(L.pe1.y), e.F(L.pe1.stat), e.P(m.pe1.bn.t_g), e.P(m.pe1.bn.t_b), GG_ACT_GELU,
                                         (float*)e.A(L.col2), B, H1, H1, d[0] / 2, 2, e.st));
    else GG_TRY(gg_im2col_nhwc_bn_bf16(e.A(L.pe1.y), e.F(L.pe1.stat), e.P(m.pe1.bn.t_g), e.P(m.pe1.bn.t_b), GG_ACT_GELU, e.A(L.col2), B, H1, H1,
                                       d[0] / 2, 2, e.st));
    GG_TRY(conv_dense_fwd(e, m.pe2, L.pe2, e.A(L.col2), m.pe2.w.Kp, M0));
    GG_TRY(bn_apply(e, m.pe2.bn, L.pe2, M0, GG_ACT_NONE, e.A(L.x_pe)));

    // stage 0: MBConv blocks
    const int mid = (int)(d[0] * c.mbconv_expand_ratio);
    int slot = 0;
    const int rps0 = H0 * H0;
    for (size_t i = 0; i < m.mb.size(); ++i) {
        const MBConvL& l = m.mb[i]; const MBAct& a = L.mb[i];
        GG_TRY(conv_dense_fwd(e, l.c1, a.c1, e.A(a.x), d[0], M0));
        if (e.fuse_dw || e.fuse_dw_s1) {
            GG_TRY(conv_dw_fwd_fused(e, l.c2, a.c2, l.c1.bn, a.c1, GG_ACT_GELU, B, H0, H0, 1));   // act1 is never materialised
        } else {
            GG_TRY(bn_apply(e, l.c1.bn, a.c1, M0, GG_ACT_GELU, e.A(a.a1)));
            GG_TRY(conv_dw_fwd(e, l.c2, a.c2, e.A(a.a1), B, H0, H0, 1));
        }
        // conv3 reads BN2+GELU of conv2's output through its A prologue unless its weight gradient needs that tensor
        if (e.fuse_pro && !(e.training && e.tr(l.c3.w.t_w)) && l.c3.w.Kp == mid && mid <= 1024 && l.c3.w.N <= 128) {
            GG_TRY(conv_dense_fwd_pro(e, l.c3, a.c3, l.c2.bn, a.c2, GG_ACT_GELU, M0));
        } else {
            GG_TRY(bn_apply(e, l.c2.bn, a.c2, M0, GG_ACT_GELU, e.A(a.a2)));
            GG_TRY(conv_dense_fwd(e, l.c3, a.c3, e.A(a.a2), mid, M0));
        }
        GG_TRY(bn_apply(e, l.c3.bn, a.c3, M0, GG_ACT_GELU, e.A(a.out), e.A(a.x), e.training ? e.dropv(slot) : nullptr, rps0));
        slot++;
    }
    int64_t prev = L.mb.empty() ? L.x_pe : L.mb.back().out;
    int res = H0, Cprev = d[0];
    int64_t Mprev = M0;
    for (int s = 0; s < 3; ++s) {
        const StageL& st = m.stages[s];
        const int C = st.C;
        const int64_t M = (int64_t)B * st.res * st.res;
        const MergeAct& ma = L.merge[s];
        // PatchMerging
        GG_TRY(conv_dense_fwd(e, st.merge.c1, ma.c1, e.A(prev), Cprev, Mprev));
        if (e.fuse_dw || e.fuse_dw_s2) {
            GG_TRY(conv_dw_fwd_fused(e, st.merge.c2, ma.c2, st.merge.c1.bn, ma.c1, GG_ACT_GELU, B, res, res, 2));
        } else {
            GG_TRY(bn_apply(e, st.merge.c1.bn, ma.c1, Mprev, GG_ACT_GELU, e.A(ma.a1)));
            GG_TRY(conv_dw_fwd(e, st.merge.c2, ma.c2, e.A(ma.a1), B, res, res, 2));
        }
        GG_TRY(bn_apply(e, st.merge.c2.bn, ma.c2, M, GG_ACT_GELU, e.A(ma.a2)));
        GG_TRY(conv_dense_fwd(e, st.merge.c3, ma.c3, e.A(ma.a2), C, M));
        GG_TRY(bn_apply(e, st.merge.c3.bn, ma.c3, M, GG_ACT_NONE, e.A(ma.out)));
        const int hid = (int)(C * c.mlp_ratio);
        const int rps = st.res * st.res;
        for (size_t i = 0; i < st.blocks.size(); ++i) {
            const BlockL& l = st.blocks[i]; const BlockAct& a = L.blocks[s][i];
            const float* s1 = e.training ? e.dropv(slot) : nullptr;
            const float* s2 = e.training ? e.dropv(slot + 1) : nullptr;
            slot += 2;
            GG_TRY(gg_layernorm_fwd(e.A(a.x0), e.f32, e.P(l.ln1.t_g), e.P(l.ln1.t_b), M, C, c.ln_eps, e.A(a.a), e.f32, e.F(a.mean1), e.F(a.rstd1), e.st));
            GG_TRY(gemm(e, e.A(a.a), C, e.Wn(l.qkv), l.qkv.Kp, e.A(a.qkv), 3 * C, M, 3 * C, l.qkv.Kp, e.P(l.qkv.t_b)));
            GgAttnArgs at;
            attn_args(e, st, l, a, B, at);
            GG_TRY(e.f32 ? gg_attention_flash_fwd(&at, 1, e.st) : gg_attention_fwd(&at, e.st));
            GG_TRY(gemm(e, e.A(a.o), C, e.Wn(l.proj), l.proj.Kp, e.A(a.x1), C, M, C, l.proj.Kp, e.P(l.proj.t_b), 0, nullptr, s1, rps, e.A(a.x0)));
            GG_TRY(conv_dw_fwd(e, l.local, a.local, e.A(a.x1), B, st.res, st.res, 1));
            if (C <= 640 && e.f32 && e.fuse_lnbn) {      // BatchNorm apply of local_conv rides on norm2's load (x2 = the residual stream is written there)
                GG_TRY(gg_layernorm_fwd_bn_f32((const float*)e.A(a.local.y), e.F(a.local.stat), e.P(l.local.bn.t_g), e.P(l.local.bn.t_b), (float*)e.A(a.x2),
                                               e.P(l.ln2.t_g), e.P(l.ln2.t_b), M, C, c.ln_eps, (float*)e.A(a.b), e.F(a.mean2), e.F(a.rstd2), e.st));
            } else if (C <= 640 && !e.f32) {
                GG_TRY(gg_layernorm_fwd_bn(e.A(a.local.y), e.F(a.local.stat), e.P(l.local.bn.t_g), e.P(l.local.bn.t_b), e.A(a.x2), e.P(l.ln2.t_g),
                                           e.P(l.ln2.t_b), M, C, c.ln_eps, e.A(a.b), e.F(a.mean2), e.F(a.rstd2), e.st));
            } else {
                GG_TRY(bn_apply(e, l.local.bn, a.local, M, GG_ACT_NONE, e.A(a.x2)));
                GG_TRY(gg_layernorm_fwd(e.A(a.x2), e.f32, e.P(l.ln2.t_g), e.P(l.ln2.t_b), M, C, c.ln_eps, e.A(a.b), e.f32, e.F(a.mean2), e.F(a.rstd2), e.st));
            }
            GG_TRY(gemm(e, e.A(a.b), C, e.Wn(l.fc1), l.fc1.Kp, e.A(a.h), hid, M, hid, l.fc1.Kp, e.P(l.fc1.t_b), GG_ACT_GELU,
                        e.training ? (void*)e.A(a.hpre) : nullptr));
            GG_TRY(gemm(e, e.A(a.h), hid, e.Wn(l.fc2), l.fc2.Kp, e.A(a.x3), C, M, C, l.fc2.Kp, e.P(l.fc2.t_b), 0, nullptr, s2, rps, e.A(a.x2)));
            prev = a.x3;
        }
        if (st.blocks.empty()) prev = ma.out;
        res = st.res; Cprev = C; Mprev = M;
    }
    // head: global average pool -> LayerNorm
    const int T = res * res, C3 = d[3];
    if (e.f32) GG_TRY(gg_token_mean_fwd_f32((const float*)e.A(prev), e.F(L.pooled), B, T, C3, e.st));
    else GG_TRY(gg_token_mean_fwd(e.A(prev), e.F(L.pooled), B, T, C3, e.st));
    if (c.features_only) {     // models/tinyvit.py:139-143: the pooled last feature map, no head.norm
        GG_HIP(hipMemcpyAsync(out, e.F(L.pooled), (size_t)B * C3 * sizeof(float), hipMemcpyDeviceToDevice, e.st));
        return 0;
    }
    GG_TRY(gg_layernorm_fwd(e.F(L.pooled), 1, e.P(m.head.t_g), e.P(m.head.t_b), B, C3, c.ln_eps, out, 1, e.F(L.mean_h), e.F(L.rstd_h), e.st));
    return 0;
}

// BatchNorm-backward pieces on the shared scratch: partial rows at the start of `bnscratch`, coef [3][C] right behind them
static float* bn_coef(const Exec& e, int64_t M, int C) { return e.F(e.L->bnscratch) + ((int64_t)gg_bn_bwd_rows(M, C) + 64) * 2 * C; }
static int bn_bwd_reduce_fin(const Exec& e, const BNP& bn, const Act& a, int64_t M, int act, const act_t* dout, act_t* dz) {
    const bool tr = e.tr(bn.t_g);
    if (e.f32) GG_TRY(gg_bn_bwd_reduce_f32((const float*)dout, (const float*)e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, nullptr,
                                           nullptr, 0, (float*)dz, e.F(e.L->bnscratch), e.st));
    else GG_TRY(gg_bn_bwd_reduce(dout, e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, nullptr, nullptr, 0, dz,
                            e.F(e.L->bnscratch), e.st));
    return gg_bn_bwd_finalize(e.F(e.L->bnscratch), gg_bn_bwd_rows(M, bn.C), bn.C, M, e.F(a.stat), e.P(bn.t_g), bn_coef(e, M, bn.C),
                              tr ? e.Gd(bn.t_g) : nullptr, tr ? e.Gd(bn.t_b) : nullptr, 1, e.st);
}

// finalize BatchNorm-backward sums that a GEMM epilogue left in statpart
static int bn_bwd_fin_gemm(const Exec& e, const BNP& bn, const Act& a, int64_t M) {
    const bool tr = e.tr(bn.t_g);
    return gg_bn_bwd_finalize(e.F(e.L->statpart), gg_gemm_colstats_rows((int)M), bn.C, M, e.F(a.stat), e.P(bn.t_g), bn_coef(e, M, bn.C),
                              tr ? e.Gd(bn.t_g) : nullptr, tr ? e.Gd(bn.t_b) : nullptr, 1, e.st);
}

__global__ void conv_wgrad_scatter_kernel(const float* __restrict__ src, int N, int Kp, int cin, int taps, float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over N * cin * taps (grad layout (co, ci, tap))
    if (i >= N * cin * taps) return;
    const int tap = i % taps, ci = (i / taps) % cin, co = i / (taps * cin);
    grad[i] += src[(int64_t)co * Kp + tap * cin + ci];
}
static int conv_wgrad_scatter(const float* src, int N, int Kp, int cin, int taps, float* grad, hipStream_t st) {
    const int n = N * cin * taps;
    hipLaunchKernelGGL(conv_wgrad_scatter_kernel, dim3((unsigned)gg_cdiv(n, 256)), dim3(256), 0, st, src, N, Kp, cin, taps, grad);
    GG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------- backward helpers
// wgrad of a dense weight: dW[N,K] (+)= dY^T[N,M] . X[M,K]   (transposes into scratch, split-K over M)
static int dense_wgrad(const Exec& e, const DenseW& w, const act_t* X, int64_t ldx, const act_t* dY, int64_t ldy, int64_t M,
                       const float* rowscale, int rps, act_t* T0, act_t* T1, bool conv_reorder) {
    (void)T0; (void)T1;
    const int K = conv_reorder ? w.Kp : w.K;   // im2col'd operand has Kp columns
    // fp32_split mode: the weight gradient of a block Linear (the tensors that have planes) as split products too, same slab protocol
    static const char* tn_min_env = gg_dev_env("GG_SPLIT_TN_MIN_M");      // (dev: the same for the weight gradients' row threshold)
    static const int64_t tn_min_m = tn_min_env ? atoll(tn_min_env) : 1024;
    const bool sp = e.m->split && w.wn3 >= 0 && !conv_reorder && M >= tn_min_m;
    const int split = sp ? gg_gemm_tn_split3_splits((int)M, w.N, K) : e.f32 ? gg_gemm_tn_f32_splits((int)M, w.N, K) : gg_gemm_tn_splits((int)M, w.N, K);
    if (sp) GG_TRY(gg_gemm_tn_split3((const float*)dY, ldy, (const float*)X, ldx, (int)M, w.N, K, rowscale, rps, e.F(e.L->splitk), split, e.st));
    else if (e.f32) GG_TRY(gg_gemm_tn_f32(dY, ldy, X, ldx, (int)M, w.N, K, rowscale, rps, e.F(e.L->splitk), split, e.st));
    else GG_TRY(gg_gemm_tn(dY, ldy, X, ldx, (int)M, w.N, K, rowscale, rps, e.F(e.L->splitk), split, e.st));
    float* gw = e.Gd(w.t_w);
    if (!conv_reorder) {
        GG_TRY(gg_splitk_reduce(e.F(e.L->splitk), gw, (int64_t)w.N * K, split, 1, 1.0f, e.st));
    } else {
        // reduce in place, then scatter (co,(ky,kx,ci)) -> (co,ci,ky,kx)
        GG_TRY(gg_splitk_reduce(e.F(e.L->splitk), e.F(e.L->splitk), (int64_t)w.N * K, split, 0, 1.0f, e.st));
        GG_TRY(conv_wgrad_scatter(e.F(e.L->splitk), w.N, K, w.cin, w.taps, gw, e.st));
    }
    return 0;
}
// weight gradient of a ConvNorm whose dy feeds nothing else (the first conv of the network): BatchNorm backward stops after
// reduce + finalize, and the TN GEMM forms dy = c0*dz + c1*y + c2 from (dz, y) while loading -- no apply pass, no dy tensor
// (dcol != null: dout does not exist yet -- the reduce rides on the col2im that would have produced it from dcol [B, H, W] stride 2)
static int convnorm_wgrad_from_dz(const Exec& e, const ConvBNDense& c, const Act& a, int64_t M, int act, const act_t* dout, act_t* dz,
                                  const act_t* X, int64_t ldx, const act_t* dcol = nullptr, int B = 0, int H = 0, int W = 0) {
    const BNP& bn = c.bn;
    const bool tr = e.tr(bn.t_g);
    float* part = e.F(e.L->bnscratch);
    const int nb = std::min(gg_bn_bwd_rows(M, bn.C), 65535);
    float* coef = part + ((int64_t)gg_bn_bwd_rows(M, bn.C) + GG_REDUCE_SLICES) * 2 * bn.C;
    if (dcol && e.f32) GG_TRY(gg_col2im_nhwc_bnbwd_f32((const float*)dcol, (const float*)e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), act, (float*)dz, part, nb, B, H, W, bn.C, e.st));
    else if (dcol) GG_TRY(gg_col2im_nhwc_bnbwd_bf16(dcol, e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), act, dz, part, nb, B, H, W, bn.C, e.st));
    else if (e.f32) GG_TRY(gg_bn_bwd_reduce_f32((const float*)dout, (const float*)e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, nullptr, nullptr, 0, (float*)dz, part, e.st));
    else GG_TRY(gg_bn_bwd_reduce(dout, e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, nullptr, nullptr, 0, dz, part, e.st));
    GG_TRY(gg_bn_bwd_finalize(part, nb, bn.C, M, e.F(a.stat), e.P(bn.t_g), coef, tr ? e.Gd(bn.t_g) : nullptr, tr ? e.Gd(bn.t_b) : nullptr, 1, e.st));
    if (!e.tr(c.w.t_w)) return 0;
    const DenseW& w = c.w;
    const int K = w.Kp;
    const int split = e.f32 ? gg_gemm_tn_f32_splits((int)M, w.N, K) : gg_gemm_tn_splits((int)M, w.N, K);
    const act_t* dzs = dz ? dz : dout;      // no activation, no residual: dz == dout and the reduce writes nothing
    if (e.f32) GG_TRY(gg_gemm_tn_bn_f32(dzs, e.A(a.y), bn.C, coef, X, ldx, (int)M, w.N, K, e.F(e.L->splitk), split, e.st));
    else GG_TRY(gg_gemm_tn_bn(dzs, e.A(a.y), bn.C, coef, X, ldx, (int)M, w.N, K, e.F(e.L->splitk), split, e.st));
    GG_TRY(gg_splitk_reduce(e.F(e.L->splitk), e.F(e.L->splitk), (int64_t)w.N * K, split, 0, 1.0f, e.st));
    return conv_wgrad_scatter(e.F(e.L->splitk), w.N, K, w.cin, w.taps, e.Gd(w.t_w), e.st);
}
static int bias_grad(const Exec& e, int t_b, const act_t* dY, int64_t ld, int64_t M, int N, const float* rowscale, int rps) {
    if (e.f32) return gg_colsum_f32((const float*)dY, ld, (int)M, N, rowscale, rps, e.F(e.L->colsum), e.Gd(t_b), 1, e.st);
    return gg_colsum_bf16(dY, ld, (int)M, N, rowscale, rps, e.F(e.L->colsum), e.Gd(t_b), 1, e.st);
}
// BatchNorm backward of one ConvNorm: dout (grad wrt post-activation output) -> dy (grad wrt the conv output)
static int bn_bwd(const Exec& e, const BNP& bn, const Act& a, int64_t M, int act, const act_t* dout, act_t* dz, act_t* dy,
                  const act_t* residual = nullptr, const float* rowscale = nullptr, int rps = 0) {
    const bool tr = e.tr(bn.t_g);
    if (e.f32) return gg_bn_bwd_f32((const float*)dout, (const float*)e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act,
                                    (const float*)residual, rowscale, rps, (float*)dz, (float*)dy, e.F(e.L->bnscratch),
                                    tr ? e.Gd(bn.t_g) : nullptr, tr ? e.Gd(bn.t_b) : nullptr, 1, e.st);
    return gg_bn_bwd(dout, e.A(a.y), e.F(a.stat), e.P(bn.t_g), e.P(bn.t_b), M, bn.C, act, residual, rowscale, rps, dz, dy,
                     e.F(e.L->bnscratch), tr ? e.Gd(bn.t_g) : nullptr, tr ? e.Gd(bn.t_b) : nullptr, 1, e.st);
}

static int bn_bwd_apply_only(const Exec& e, const act_t* dz, const act_t* y, const float* coef, int64_t M, int C, act_t* dy) {
    if (e.f32) return gg_bn_bwd_apply_f32((const float*)dz, (const float*)y, coef, M, C, nullptr, 0, (float*)dy, e.st);
    return gg_bn_bwd_apply(dz, y, coef, M, C, nullptr, 0, dy, e.st);
}
// stride-1 depthwise data gradient with the BatchNorm-backward fusions on its input (apply) and output (reduce) sides
static int dw_bwd_data_fused(const Exec& e, const act_t* dz_in, const act_t* y_in, const float* in_coef, const DwW& w, act_t* out, int B, int H, int W,
                             const act_t* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta, int ep_act, float* ep_part) {
    if (e.f32) return gg_dwconv3x3_bwd_data_fused_f32((const float*)dz_in, (const float*)y_in, in_coef, e.Taps(w), (float*)out, B, H, W, w.C,
                                                      (const float*)ep_y, ep_stat, ep_gamma, ep_beta, ep_act, ep_part, e.st);
    return gg_dwconv3x3_bwd_data_fused(dz_in, y_in, in_coef, e.Taps(w), out, B, H, W, w.C, ep_y, ep_stat, ep_gamma, ep_beta, ep_act, ep_part, e.st);
}
static int dw_fused_rows(const Exec& e, int B, int H, int W, int C, int with_input_fusion) {
    return e.f32 ? gg_dwconv_f32_stat_rows(B, H, W, C, 1) : gg_dwconv_fused_stat_rows(B, H, W, C, with_input_fusion);
}
static int dw_bwd_data(const Exec& e, const DwW& w, const act_t* dy, act_t* dx, int B, int H, int W, int stride) {
    if (e.f32) return gg_dwconv3x3_bwd_data_f32((const float*)dy, e.Taps(w), (float*)dx, B, H, W, w.C, stride, e.st);
    return gg_dwconv3x3_bwd_data(dy, e.Taps(w), dx, B, H, W, w.C, stride, e.st);
}
static int dw_bwd_weight(const Exec& e, const DwW& w, const act_t* x, const act_t* dy, int B, int H, int W, int stride) {
    if (e.f32) return gg_dwconv3x3_bwd_weight_f32((const float*)x, (const float*)dy, B, H, W, w.C, stride, e.F(e.L->bnscratch), e.Gd(w.t_w), 1, e.st);
    return gg_dwconv3x3_bwd_weight(x, dy, B, H, W, w.C, stride, e.F(e.L->bnscratch), e.Gd(w.t_w), 1, e.st);
}

static int backward_impl(Exec& e, const float* d_out) {
    const Model& m = *e.m; const Layout& L = *e.L; const GgTinyVitCfg& c = m.cfg;
    const int B = e.B;
    const int* d = c.embed_dims;
    const int H = c.img_size, H1 = H / 2, H0 = m.res0;
    const int64_t M1 = (int64_t)B * H1 * H1, M0 = (int64_t)B * H0 * H0;
    act_t* G0 = e.A(L.G[0]); act_t* G1 = e.A(L.G[1]); act_t* G2 = e.A(L.G[2]); act_t* G3 = e.A(L.G[3]); act_t* G4 = e.A(L.G[4]);

    // drop-path slot bookkeeping mirrors forward
    int nslots = (int)m.mb.size();
    for (int s = 0; s < 3; ++s) nslots += 2 * (int)m.stages[s].blocks.size();
    int slot = nslots;

    // ---- head: LayerNorm (f32) + average pool ----
    const int res3 = m.stages[2].res, T = res3 * res3, C3 = d[3];
    {
        const bool tr = e.tr(m.head.t_g);
        float* dpool = reinterpret_cast<float*>(G1);
        if (c.features_only) GG_HIP(hipMemcpyAsync(dpool, d_out, (size_t)B * C3 * sizeof(float), hipMemcpyDeviceToDevice, e.st));
        else GG_TRY(gg_layernorm_bwd(d_out, e.F(L.pooled), 1, e.F(L.mean_h), e.F(L.rstd_h), e.P(m.head.t_g), B, C3, nullptr, dpool,
                                     e.F(L.lnscratch), tr ? e.Gd(m.head.t_g) : nullptr, tr ? e.Gd(m.head.t_b) : nullptr, 1, e.st));
        if (e.f32) GG_TRY(gg_token_mean_bwd_f32(dpool, (float*)G0, B, T, C3, e.st));
        else GG_TRY(gg_token_mean_bwd(dpool, G0, B, T, C3, e.st));
    }
    act_t* dx = G0;      // gradient w.r.t. the current activation (block output), bf16 [M, C]
    // free buffers for the block-level temporaries
    for (int s = 2; s >= 0; --s) {
        if (s < 2) e.done(s + 2);          // model stage s+2 (blocks + PatchMerging) is enqueued: its parameter gradients are final
        const StageL& st = m.stages[s];
        const int C = st.C;
        const int64_t M = (int64_t)B * st.res * st.res;
        const int hid = (int)(C * c.mlp_ratio);
        const int rps = st.res * st.res;
        const int rin = s == 0 ? H0 : m.stages[s - 1].res;
        const int Cin = s == 0 ? d[0] : m.stages[s - 1].C;
        const int64_t Min = (int64_t)B * rin * rin;
        for (int i = (int)st.blocks.size() - 1; i >= 0; --i) {
            const BlockL& l = st.blocks[i]; const BlockAct& a = L.blocks[s][i];
            slot -= 2;
            const float* s1 = e.dropv(slot);
            const float* s2 = e.dropv(slot + 1);
            // dx == d(x3).  MLP branch: x3 = x2 + s2*(fc2(gelu(fc1(ln2(x2)))))
            act_t* t_a = (dx == G0) ? G1 : G0;    // scratch distinct from dx
            act_t* t_b = G2; act_t* t_c = G3; act_t* t_d = G4;
            // dh = (s2*dx) . W2  * gelu'(hpre)                      [M, hid]
            GG_TRY(gemm(e, dx, C, e.Wt(l.fc2), l.fc2.Np, t_b, hid, M, hid, C, nullptr, 0, nullptr, s2, rps, nullptr, nullptr, e.A(a.hpre), GG_ACT_GELU));
            if (e.tr(l.fc2.t_w)) {
                GG_TRY(dense_wgrad(e, l.fc2, e.A(a.h), hid, dx, C, M, s2, rps, t_c, t_d, false));
                GG_TRY(bias_grad(e, l.fc2.t_b, dx, C, M, C, s2, rps));
            }
            // db = dh . W1                                           [M, C]
            GG_TRY(gemm(e, t_b, hid, e.Wt(l.fc1), l.fc1.Np, t_a, C, M, C, hid));
            if (e.tr(l.fc1.t_w)) {
                GG_TRY(dense_wgrad(e, l.fc1, e.A(a.b), C, t_b, hid, M, nullptr, 0, t_c, t_d, false));
                GG_TRY(bias_grad(e, l.fc1.t_b, t_b, hid, M, hid, nullptr, 0));
            }
            // dx2 = LN2bwd(db) + dx                                   -> t_b
            // frozen block: the same kernel also leaves (sum dx2*x2, sum dx2) per column, all that local_conv's BatchNorm backward needs
            const bool lncol = e.fuse_lncol && e.fuse_bnbwd && C <= 640 && !e.tr(l.local.w.t_w) && !e.tr(l.local.bn.t_g) && !e.tr(l.local.bn.t_b) &&
                               !e.tr(l.ln2.t_g) && !e.tr(l.ln2.t_b);       // (the fused form produces no LayerNorm / BatchNorm parameter gradients: every one of them must be frozen)
            if (lncol) {
                GG_TRY(gg_layernorm_bwd_colsum(t_a, e.A(a.x2), e.f32, e.F(a.mean2), e.F(a.rstd2), e.P(l.ln2.t_g), M, C, dx, t_b, e.F(L.lnscratch), e.st));
            } else {
                const bool tr = e.tr(l.ln2.t_g);
                GG_TRY(gg_layernorm_bwd(t_a, e.A(a.x2), e.f32, e.F(a.mean2), e.F(a.rstd2), e.P(l.ln2.t_g), M, C, dx, t_b, e.F(L.lnscratch),
                                        tr ? e.Gd(l.ln2.t_g) : nullptr, tr ? e.Gd(l.ln2.t_b) : nullptr, 1, e.st));
            }
            // local_conv: x2 = BN(dw(x1)).  dy -> t_a (dz scratch t_c), dx1 = dwT(dy) -> t_c
            if (e.tr(l.local.w.t_w) || !e.fuse_bnbwd) {
                GG_TRY(bn_bwd(e, l.local.bn, a.local, M, GG_ACT_NONE, t_b, t_c, t_a));
                if (e.tr(l.local.w.t_w))
                    GG_TRY(dw_bwd_weight(e, l.local.w, e.A(a.x1), t_a, B, st.res, st.res, 1));
                GG_TRY(dw_bwd_data(e, l.local.w, t_a, t_c, B, st.res, st.res, 1));
            } else {
                // frozen taps: BN-backward apply is folded into the conv's staging (no dz / dy tensors at all)
                if (lncol) GG_TRY(gg_bn_bwd_coef_from_x(e.F(L.lnscratch), gg_layernorm_bwd_colsum_rows(M), C, M, e.F(a.local.stat), e.P(l.local.bn.t_g),
                                                        e.P(l.local.bn.t_b), bn_coef(e, M, C), e.st));
                else GG_TRY(bn_bwd_reduce_fin(e, l.local.bn, a.local, M, GG_ACT_NONE, t_b, nullptr));
                GG_TRY(dw_bwd_data_fused(e, t_b, e.A(a.local.y), bn_coef(e, M, C), l.local.w, t_c, B, st.res, st.res,
                                                   nullptr, nullptr, nullptr, nullptr, 0, nullptr));
            }
            act_t* dx1 = t_c;
            // attention branch: x1 = x0 + s1*(proj(o)+b)
            // do = (s1*dx1) . Wproj                                   -> t_a  [M, C]
            GG_TRY(gemm(e, dx1, C, e.Wt(l.proj), l.proj.Np, t_a, C, M, C, C, nullptr, 0, nullptr, s1, rps));
            if (e.tr(l.proj.t_w)) {
                GG_TRY(dense_wgrad(e, l.proj, e.A(a.o), C, dx1, C, M, s1, rps, t_b, t_d, false));
                GG_TRY(bias_grad(e, l.proj.t_b, dx1, C, M, C, s1, rps));
            }
            // dqkv                                                     -> t_b  [M, 3C]
            GgAttnArgs at;
            attn_args(e, st, l, a, B, at);
            at.dout = t_a; at.lddo = C; at.dqkv = t_b;
            at.dbias = e.tr(l.t_ab) ? e.Gd(l.t_ab) : nullptr;
            const bool flash = e.f32 || at.tokens_per_window > 256 || st.ws > 16;
            const int64_t prow = flash ? gg_attention_flash_dbias_rows(at.num_windows, at.tokens_per_window) : (int64_t)at.num_windows + 64;
            if (at.dbias && prow * st.heads * st.ws * st.ws * 4 <= ((int64_t)64 << 20))
                at.dbias_scratch = e.F(L.splitk);      // per-workgroup partials -> deterministic second stage
            static const bool ds_off = gg_dev_env("GG_ATTN_NO_DS_SCRATCH") != nullptr;
            if (e.f32 && L.attn_ds >= 0 && !ds_off) at.ds_scratch = e.F(L.attn_ds);
            GG_TRY(e.f32 ? gg_attention_flash_bwd(&at, 1, e.st) : gg_attention_bwd(&at, e.st));
            // da = dqkv . Wqkv                                         -> t_a  [M, C]
            GG_TRY(gemm(e, t_b, 3 * C, e.Wt(l.qkv), l.qkv.Np, t_a, C, M, C, 3 * C));
            if (e.tr(l.qkv.t_w)) {
                act_t* t_e = dx;   // the old block-output gradient is dead by now
                GG_TRY(dense_wgrad(e, l.qkv, e.A(a.a), C, t_b, 3 * C, M, nullptr, 0, t_e, t_d, false));
                GG_TRY(bias_grad(e, l.qkv.t_b, t_b, 3 * C, M, 3 * C, nullptr, 0));
            }
            // dx0 = LN1bwd(da) + dx1                                   -> old dx buffer
            {
                const bool tr = e.tr(l.ln1.t_g);
                GG_TRY(gg_layernorm_bwd(t_a, e.A(a.x0), e.f32, e.F(a.mean1), e.F(a.rstd1), e.P(l.ln1.t_g), M, C, dx1, dx, e.F(L.lnscratch),
                                        tr ? e.Gd(l.ln1.t_g) : nullptr, tr ? e.Gd(l.ln1.t_b) : nullptr, 1, e.st));
            }
            (void)hid;
        }
        // ---- PatchMerging backward: out = BN3(conv3(a2)); a2 = gelu(BN2(dw s2(a1))); a1 = gelu(BN1(conv1(x))) ----
        const MergeAct& ma = L.merge[s];
        act_t* t_a = (dx == G0) ? G1 : G0;
        act_t* t_b = G2; act_t* t_c = G3; act_t* t_d = G4;
        const int64_t xin = s == 0 ? (L.mb.empty() ? L.x_pe : L.mb.back().out)
                                   : (L.blocks[s - 1].empty() ? L.merge[s - 1].out : L.blocks[s - 1].back().x3);
        GG_TRY(bn_bwd(e, st.merge.c3.bn, ma.c3, M, GG_ACT_NONE, dx, t_b, t_a));                         // dy3 -> t_a
        if (e.tr(st.merge.c3.w.t_w)) GG_TRY(dense_wgrad(e, st.merge.c3.w, e.A(ma.a2), C, t_a, C, M, nullptr, 0, t_b, t_c, false));
        if (e.fuse_bngemm && !e.tr(st.merge.c1.w.t_w) && !e.tr(st.merge.c2.w.t_w) && C % 64 == 0) {
            GG_TRY(gemm_bnbwd(e, t_a, C, e.Wt(st.merge.c3.w), st.merge.c3.w.Np, t_d, M, C, C, st.merge.c2.bn, ma.c2, GG_ACT_GELU));   // dz2 -> t_d
            GG_TRY(bn_bwd_fin_gemm(e, st.merge.c2.bn, ma.c2, M));
            if (e.fuse_bnbwd && e.fuse_bnbwd_epi) {
                // the stride-2 data gradient forms dy2 from (dz2, y2) at its taps and emits dz1 = da1*GELU'(BN1(y1)) + BN1's sums
                if (e.f32) GG_TRY(gg_dwconv3x3_s2_bwd_data_fused_f32((const float*)t_d, (const float*)e.A(ma.c2.y), bn_coef(e, M, C), e.Taps(st.merge.c2.w),
                                                                     (float*)t_b, B, rin, rin, C, (const float*)e.A(ma.c1.y), e.F(ma.c1.stat),
                                                                     e.P(st.merge.c1.bn.t_g), e.P(st.merge.c1.bn.t_b), GG_ACT_GELU, e.F(L.statpart), e.st));
                else GG_TRY(gg_dwconv3x3_s2_bwd_data_fused(t_d, e.A(ma.c2.y), bn_coef(e, M, C), e.Taps(st.merge.c2.w), t_b, B, rin, rin, C,
                                                      e.A(ma.c1.y), e.F(ma.c1.stat), e.P(st.merge.c1.bn.t_g), e.P(st.merge.c1.bn.t_b),
                                                      GG_ACT_GELU, e.F(L.statpart), e.st));                    // dz1 -> t_b
                const bool tr1 = e.tr(st.merge.c1.bn.t_g);
                GG_TRY(gg_bn_bwd_finalize(e.F(L.statpart), e.f32 ? gg_dwconv_f32_s2_fused_stat_rows(B, rin, rin, C) : gg_dwconv_s2_fused_stat_rows(B, rin, rin, C), C, Min, e.F(ma.c1.stat),
                                          e.P(st.merge.c1.bn.t_g), bn_coef(e, Min, C), tr1 ? e.Gd(st.merge.c1.bn.t_g) : nullptr,
                                          tr1 ? e.Gd(st.merge.c1.bn.t_b) : nullptr, 1, e.st));
                GG_TRY(gemm_folded_dgrad(e, st.merge.c1.w, t_b, e.A(ma.c1.y), bn_coef(e, Min, C), e.F(ma.c1.stat), dx, Min, nullptr));
                continue;
            }
            GG_TRY(bn_bwd_apply_only(e, t_d, e.A(ma.c2.y), bn_coef(e, M, C), M, C, t_a));       // dy2 -> t_a
            GG_TRY(dw_bwd_data(e, st.merge.c2.w, t_a, t_b, B, rin, rin, 2));         // da1 -> t_b [Min, C]
            GG_TRY(bn_bwd_reduce_fin(e, st.merge.c1.bn, ma.c1, Min, GG_ACT_GELU, t_b, t_d));                 // dz1 -> t_d
            GG_TRY(gemm_folded_dgrad(e, st.merge.c1.w, t_d, e.A(ma.c1.y), bn_coef(e, Min, C), e.F(ma.c1.stat), dx, Min, nullptr));
            continue;
        }
        GG_TRY(gemm(e, t_a, C, e.Wt(st.merge.c3.w), st.merge.c3.w.Np, t_b, C, M, C, C));                 // da2 -> t_b
        GG_TRY(bn_bwd(e, st.merge.c2.bn, ma.c2, M, GG_ACT_GELU, t_b, t_c, t_a));                          // dy2 -> t_a
        if (e.tr(st.merge.c2.w.t_w)) {
            if (e.fuse_dw || e.fuse_dw_s2) GG_TRY(bn_apply(e, st.merge.c1.bn, ma.c1, Min, GG_ACT_GELU, e.A(ma.a1)));      // act1 was fused away in forward
            GG_TRY(dw_bwd_weight(e, st.merge.c2.w, e.A(ma.a1), t_a, B, rin, rin, 2));
        }
        GG_TRY(dw_bwd_data(e, st.merge.c2.w, t_a, t_b, B, rin, rin, 2));        // da1 -> t_b [Min, C]
        GG_TRY(bn_bwd(e, st.merge.c1.bn, ma.c1, Min, GG_ACT_GELU, t_b, t_c, t_a));                        // dy1 -> t_a
        if (e.tr(st.merge.c1.w.t_w)) GG_TRY(dense_wgrad(e, st.merge.c1.w, e.A(xin), Cin, t_a, C, Min, nullptr, 0, t_b, t_c, false));
        GG_TRY(gemm(e, t_a, C, e.Wt(st.merge.c1.w), st.merge.c1.w.Np, dx, Cin, Min, Cin, C));             // dx_in -> dx
    }
    e.done(1);
    // ---- stage 0: MBConv backward ----
    const int mid = (int)(d[0] * c.mbconv_expand_ratio);
    const int rps0 = H0 * H0;
    for (int i = (int)m.mb.size() - 1; i >= 0; --i) {
        const MBConvL& l = m.mb[i]; const MBAct& a = L.mb[i];
        slot -= 1;
        const float* s0 = e.dropv(slot);
        act_t* t_a = (dx == G0) ? G1 : G0;
        act_t* t_b = G2; act_t* t_c = G3; act_t* t_d = G4;
        // out = gelu(x + s*BN3(y3)):  dz(=dpre, also the skip gradient) -> t_b, dy3 -> t_a
        GG_TRY(bn_bwd(e, l.c3.bn, a.c3, M0, GG_ACT_GELU, dx, t_b, t_a, e.A(a.x), s0, rps0));
        // (a2 exists: the forward only skips it when its `trainable` mask freezes conv3 -- the two calls must get the same mask)
        if (e.tr(l.c3.w.t_w)) GG_TRY(dense_wgrad(e, l.c3.w, e.A(a.a2), mid, t_a, d[0], M0, nullptr, 0, t_c, t_d, false));
        if (e.fuse_bngemm && !e.tr(l.c1.w.t_w) && !e.tr(l.c2.w.t_w) && mid % 64 == 0) {
            GG_TRY(gemm_bnbwd(e, t_a, d[0], e.Wt(l.c3.w), l.c3.w.Np, t_d, M0, mid, d[0], l.c2.bn, a.c2, GG_ACT_GELU));                   // dz2 -> t_d
            GG_TRY(bn_bwd_fin_gemm(e, l.c2.bn, a.c2, M0));
            act_t* dz1;
            if (e.fuse_bnbwd && e.fuse_bnbwd_epi) {
                GG_TRY(dw_bwd_data_fused(e, t_d, e.A(a.c2.y), bn_coef(e, M0, mid), l.c2.w, t_c, B, H0, H0, e.A(a.c1.y),
                                                   e.F(a.c1.stat), e.P(l.c1.bn.t_g), e.P(l.c1.bn.t_b), GG_ACT_GELU, e.F(L.statpart)));   // dz1 -> t_c
                const bool tr1 = e.tr(l.c1.bn.t_g);
                GG_TRY(gg_bn_bwd_finalize(e.F(L.statpart), dw_fused_rows(e, B, H0, H0, mid, 1), mid, M0, e.F(a.c1.stat), e.P(l.c1.bn.t_g),
                                          bn_coef(e, M0, mid), tr1 ? e.Gd(l.c1.bn.t_g) : nullptr, tr1 ? e.Gd(l.c1.bn.t_b) : nullptr, 1, e.st));
                dz1 = t_c;
            } else {
                if (e.fuse_bnbwd) {      // dy2 is formed from (dz2, y2) inside the conv; da1 -> t_c
                    GG_TRY(dw_bwd_data_fused(e, t_d, e.A(a.c2.y), bn_coef(e, M0, mid), l.c2.w, t_c, B, H0, H0, nullptr,
                                                       nullptr, nullptr, nullptr, 0, nullptr));
                } else {
                    GG_TRY(bn_bwd_apply_only(e, t_d, e.A(a.c2.y), bn_coef(e, M0, mid), M0, mid, t_a));   // dy2 -> t_a
                    GG_TRY(dw_bwd_data(e, l.c2.w, t_a, t_c, B, H0, H0, 1));             // da1 -> t_c
                }
                GG_TRY(bn_bwd_reduce_fin(e, l.c1.bn, a.c1, M0, GG_ACT_GELU, t_c, t_d));                        // dz1 -> t_d
                dz1 = t_d;
            }
            GG_TRY(gemm_folded_dgrad(e, l.c1.w, dz1, e.A(a.c1.y), bn_coef(e, M0, mid), e.F(a.c1.stat), dx, M0, t_b));   // + dpre
            continue;
        }
        if (e.fuse_bngemm && e.fuse_bnbwd && e.fuse_bnbwd_epi && mid % 64 == 0) {
            // trainable weights: dy2 / dy1 are materialised for the weight gradients, but both BatchNorm-backward REDUCE passes still
            // ride on the kernels that produce their inputs (conv3's dgrad epilogue, the depthwise data gradient's epilogue)
            GG_TRY(gemm_bnbwd(e, t_a, d[0], e.Wt(l.c3.w), l.c3.w.Np, t_d, M0, mid, d[0], l.c2.bn, a.c2, GG_ACT_GELU));   // dz2 -> t_d
            GG_TRY(bn_bwd_fin_gemm(e, l.c2.bn, a.c2, M0));
            GG_TRY(bn_bwd_apply_only(e, t_d, e.A(a.c2.y), bn_coef(e, M0, mid), M0, mid, t_a));                // dy2 -> t_a
            if (e.tr(l.c2.w.t_w)) {
                if (e.fuse_dw || e.fuse_dw_s1) GG_TRY(bn_apply(e, l.c1.bn, a.c1, M0, GG_ACT_GELU, e.A(a.a1)));
                GG_TRY(dw_bwd_weight(e, l.c2.w, e.A(a.a1), t_a, B, H0, H0, 1));
            }
            GG_TRY(dw_bwd_data_fused(e, t_a, nullptr, nullptr, l.c2.w, t_c, B, H0, H0, e.A(a.c1.y), e.F(a.c1.stat),
                                               e.P(l.c1.bn.t_g), e.P(l.c1.bn.t_b), GG_ACT_GELU, e.F(L.statpart)));   // dz1 -> t_c
            const bool tr1 = e.tr(l.c1.bn.t_g);
            GG_TRY(gg_bn_bwd_finalize(e.F(L.statpart), dw_fused_rows(e, B, H0, H0, mid, 0), mid, M0, e.F(a.c1.stat), e.P(l.c1.bn.t_g),
                                      bn_coef(e, M0, mid), tr1 ? e.Gd(l.c1.bn.t_g) : nullptr, tr1 ? e.Gd(l.c1.bn.t_b) : nullptr, 1, e.st));
            GG_TRY(bn_bwd_apply_only(e, t_c, e.A(a.c1.y), bn_coef(e, M0, mid), M0, mid, t_a));                // dy1 -> t_a
            if (e.tr(l.c1.w.t_w)) GG_TRY(dense_wgrad(e, l.c1.w, e.A(a.x), d[0], t_a, mid, M0, nullptr, 0, t_c, t_d, false));
            GG_TRY(gemm(e, t_a, mid, e.Wt(l.c1.w), l.c1.w.Np, dx, d[0], M0, d[0], mid, nullptr, 0, nullptr, nullptr, 0, t_b));
            continue;
        }
        GG_TRY(gemm(e, t_a, d[0], e.Wt(l.c3.w), l.c3.w.Np, t_c, mid, M0, mid, d[0]));                     // da2 -> t_c [M0, mid]
        if (e.tr(l.c2.w.t_w) || !e.fuse_bnbwd || !e.fuse_bnbwd_epi) {
            GG_TRY(bn_bwd(e, l.c2.bn, a.c2, M0, GG_ACT_GELU, t_c, t_d, t_a));                              // dy2 -> t_a
            if (e.tr(l.c2.w.t_w)) {
                if (e.fuse_dw || e.fuse_dw_s1) GG_TRY(bn_apply(e, l.c1.bn, a.c1, M0, GG_ACT_GELU, e.A(a.a1)));             // act1 was fused away in forward
                GG_TRY(dw_bwd_weight(e, l.c2.w, e.A(a.a1), t_a, B, H0, H0, 1));
            }
            GG_TRY(dw_bwd_data(e, l.c2.w, t_a, t_c, B, H0, H0, 1));           // da1 -> t_c
            GG_TRY(bn_bwd(e, l.c1.bn, a.c1, M0, GG_ACT_GELU, t_c, t_d, t_a));                              // dy1 -> t_a
        } else {
            // frozen taps: 3 streaming passes instead of 5.  reduce(c2) -> dz2; the depthwise data gradient forms dy2 from
            // (dz2, y2) while staging and emits dz1 = da1*GELU'(BN1(y1)) + BN1's backward statistics; apply(c1) -> dy1.
            GG_TRY(bn_bwd_reduce_fin(e, l.c2.bn, a.c2, M0, GG_ACT_GELU, t_c, t_d));                        // dz2 -> t_d
            GG_TRY(dw_bwd_data_fused(e, t_d, e.A(a.c2.y), bn_coef(e, M0, mid), l.c2.w, t_c, B, H0, H0, e.A(a.c1.y),
                                               e.F(a.c1.stat), e.P(l.c1.bn.t_g), e.P(l.c1.bn.t_b), GG_ACT_GELU, e.F(L.statpart)));   // dz1 -> t_c
            const bool tr1 = e.tr(l.c1.bn.t_g);
            GG_TRY(gg_bn_bwd_finalize(e.F(L.statpart), dw_fused_rows(e, B, H0, H0, mid, 1), mid, M0, e.F(a.c1.stat), e.P(l.c1.bn.t_g),
                                      bn_coef(e, M0, mid), tr1 ? e.Gd(l.c1.bn.t_g) : nullptr, tr1 ? e.Gd(l.c1.bn.t_b) : nullptr, 1, e.st));
            GG_TRY(bn_bwd_apply_only(e, t_c, e.A(a.c1.y), bn_coef(e, M0, mid), M0, mid, t_a));                              // dy1 -> t_a
        }
        if (e.tr(l.c1.w.t_w)) GG_TRY(dense_wgrad(e, l.c1.w, e.A(a.x), d[0], t_a, mid, M0, nullptr, 0, t_c, t_d, false));
        // dx_in = dy1 . W1 + dpre
        GG_TRY(gemm(e, t_a, mid, e.Wt(l.c1.w), l.c1.w.Np, dx, d[0], M0, d[0], mid, nullptr, 0, nullptr, nullptr, 0, t_b));
    }
    e.done(0);
    // ---- PatchEmbed backward (dgrad only to conv1's output; the image needs no gradient) ----
    {
        act_t* t_a = (dx == G0) ? G1 : G0;
        act_t* t_b = G2; act_t* t_c = G3; act_t* t_d = G4;
        const bool need1 = e.tr(m.pe1.w.t_w) || e.tr(m.pe1.bn.t_g);
        const bool need2 = e.tr(m.pe2.w.t_w) || e.tr(m.pe2.bn.t_g) || need1;
        // (conv2 keeps the three-pass BatchNorm backward: forming dy2 from (dx, y2) inside both of its GEMMs was measured at +1.7 ms of GEMM time
        // against the 0.75 ms apply pass it removes -- the two-source prologue kernel at K = 96 runs at 62 TFLOP/s, the plain one at 86)
        if (need2) {
            GG_TRY(bn_bwd(e, m.pe2.bn, L.pe2, M0, GG_ACT_NONE, dx, t_b, t_a));                             // dy2 -> t_a [M0, C0]
            if (e.tr(m.pe2.w.t_w)) GG_TRY(dense_wgrad(e, m.pe2.w, e.A(L.col2), m.pe2.w.Kp, t_a, d[0], M0, nullptr, 0, t_b, t_c, true));
        }
        if (need1) {
            GG_TRY(gemm(e, t_a, d[0], e.Wt(m.pe2.w), m.pe2.w.Np, t_b, m.pe2.w.Kp, M0, m.pe2.w.Kp, d[0])); // dcol2 -> t_b
            if (e.fuse_bnbwd && m.pe1.w.N == m.pe1.bn.C && (m.pe1.bn.C & (e.f32 ? 3 : 7)) == 0) {
                // col2im + BN1-backward reduce in one pass (dz1 -> t_d; da1 and dy1 are never formed), weight gradient from (dz1, y1, coef)
                GG_TRY(convnorm_wgrad_from_dz(e, m.pe1, L.pe1, M1, GG_ACT_GELU, nullptr, t_d, e.A(L.col1), 32, t_b, B, H1, H1));
            } else {
                if (e.f32) GG_TRY(gg_col2im_nhwc_f32((const float*)t_b, (float*)t_c, B, H1, H1, d[0] / 2, 2, e.st));
                else GG_TRY(gg_col2im_nhwc_bf16(t_b, t_c, B, H1, H1, d[0] / 2, 2, e.st));                  // da1 -> t_c [M1, C0/2]
                GG_TRY(bn_bwd(e, m.pe1.bn, L.pe1, M1, GG_ACT_GELU, t_c, t_d, t_a));                         // dy1 -> t_a
                if (e.tr(m.pe1.w.t_w)) GG_TRY(dense_wgrad(e, m.pe1.w, e.A(L.col1), 32, t_a, d[0] / 2, M1, nullptr, 0, t_b, t_c, true));
            }
        }
    }
    e.done(-1);
    return 0;
}

template <typename T>
__global__ void repack_weight_kernel(const float* __restrict__ src, int N, int cin, int taps, T* __restrict__ Wn, int ldn,
                                     T* __restrict__ Wt, int ldt) {
    const int K = cin * taps;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    const int co = i / K, k = i % K;
    const int tap = k / cin, ci = k % cin;
    const T v = (T)src[((int64_t)co * cin + ci) * taps + tap];
    Wn[(int64_t)co * ldn + k] = v;
    Wt[(int64_t)k * ldt + co] = v;
}
__global__ void repack_taps_kernel(const float* __restrict__ src, int C, float* __restrict__ taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * C) return;
    const int t = i / C, c = i % C;
    taps[i] = src[c * 9 + t];
}
static int repack_dense(const DenseW& w, const float* params, const Model& m, char* wc, hipStream_t st) {
    const int n = w.N * w.K;
    if (m.f32)
        hipLaunchKernelGGL(repack_weight_kernel<float>, dim3((unsigned)gg_cdiv(n, 256)), dim3(256), 0, st, params + m.tensors[w.t_w].offset, w.N,
                           w.cin, w.taps, reinterpret_cast<float*>(wc + w.wn), w.Kp, reinterpret_cast<float*>(wc + w.wt), w.Np);
    else
        hipLaunchKernelGGL(repack_weight_kernel<bf16>, dim3((unsigned)gg_cdiv(n, 256)), dim3(256), 0, st, params + m.tensors[w.t_w].offset, w.N,
                           w.cin, w.taps, reinterpret_cast<bf16*>(wc + w.wn), w.Kp, reinterpret_cast<bf16*>(wc + w.wt), w.Np);
    GG_LAUNCH_CHECK();
    return 0;
}
static int repack_dw(const DwW& w, const float* params, const Model& m, char* wc, hipStream_t st) {
    hipLaunchKernelGGL(repack_taps_kernel, dim3((unsigned)gg_cdiv(9 * w.C, 256)), dim3(256), 0, st, params + m.tensors[w.t_w].offset, w.C,
                       reinterpret_cast<float*>(wc + w.taps));
    GG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------- C ABI
extern "C" int gg_tinyvit_num_tensors(const GgTinyVitCfg* cfg) {
    Model m;
    if (build_model(cfg, m)) return -1;
    return (int)m.tensors.size();
}
extern "C" int gg_tinyvit_tensor_info(const GgTinyVitCfg* cfg, int i, char* name, int name_cap, int64_t* offset, int64_t* numel,
                                      int* ndim, int64_t* shape4, int* kind) {
    Model m;
    GG_TRY(build_model(cfg, m));
    GG_CHECK(i >= 0 && i < (int)m.tensors.size(), "gg_tinyvit_tensor_info: index %d out of range", i);
    const TensorInfo& t = m.tensors[i];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", t.name.c_str());
    if (offset) *offset = t.offset;
    if (numel) *numel = t.numel;
    if (ndim) *ndim = t.ndim;
    if (shape4) for (int j = 0; j < 4; ++j) shape4[j] = t.shape[j];
    if (kind) *kind = t.kind;
    return 0;
}
extern "C" int64_t gg_tinyvit_param_floats(const GgTinyVitCfg* cfg) { Model m; return build_model(cfg, m) ? -1 : m.param_floats; }
extern "C" int64_t gg_tinyvit_buffer_floats(const GgTinyVitCfg* cfg) { Model m; return build_model(cfg, m) ? -1 : m.buffer_floats; }
extern "C" int gg_tinyvit_num_counters(const GgTinyVitCfg* cfg) { Model m; return build_model(cfg, m) ? -1 : m.num_counters; }
extern "C" int gg_tinyvit_num_drop_slots(const GgTinyVitCfg* cfg) {
    Model m;
    if (build_model(cfg, m)) return -1;
    int n = (int)m.mb.size();
    for (int s = 0; s < 3; ++s) n += 2 * (int)m.stages[s].blocks.size();
    return n;
}
extern "C" int64_t gg_tinyvit_wcache_bytes(const GgTinyVitCfg* cfg) { Model m; return build_model(cfg, m) ? -1 : m.wcache_bytes; }
extern "C" int64_t gg_tinyvit_workspace_bytes_masked(const GgTinyVitCfg* cfg, int batch, int training, const uint8_t* trainable) {
    Model m;
    if (build_model(cfg, m)) return -1;
    if (batch <= 0) { gg_set_error("gg_tinyvit_workspace_bytes: batch must be > 0"); return -1; }
    Plan p; Layout L;
    plan_make(m, batch, training != 0, p, L, trainable);
    return p.total;
}
extern "C" int64_t gg_tinyvit_workspace_bytes(const GgTinyVitCfg* cfg, int batch, int training) {
    return gg_tinyvit_workspace_bytes_masked(cfg, batch, training, nullptr);
}
extern "C" int gg_tinyvit_activation_info_masked(const GgTinyVitCfg* cfg, int batch, const char* name, const uint8_t* trainable, int64_t* offset,
                                                 int64_t* bytes) {
    Model m;
    GG_TRY(build_model(cfg, m));
    Plan p; Layout L;
    plan_make(m, batch, true, p, L, trainable);
    auto it = p.index.find(name);
    GG_CHECK(it != p.index.end(), "gg_tinyvit_activation_info: no activation named '%s'", name);
    GG_CHECK(!p.temps.count(name), "gg_tinyvit_activation_info: '%s' is not retained under this trainable mask (a temporary between its producer and its one consumer)", name);
    if (offset) *offset = p.regs[it->second].offset;
    if (bytes) *bytes = p.regs[it->second].bytes;
    return 0;
}
extern "C" int gg_tinyvit_activation_info(const GgTinyVitCfg* cfg, int batch, const char* name, int64_t* offset, int64_t* bytes) {
    return gg_tinyvit_activation_info_masked(cfg, batch, name, nullptr, offset, bytes);
}
// `only` (host, one byte per tensor, or NULL = every tensor): the tensors whose cached forms are rebuilt.  After an optimizer step only the
// trainable tensors changed -- under the reference freeze policy 14 of the 52 cached matrices -- so the per-step refresh skips the frozen ones.
static int refresh_weights(const GgTinyVitCfg* cfg, const float* params, void* wcache, const uint8_t* only, void* stream);
extern "C" int gg_tinyvit_refresh_weights(const GgTinyVitCfg* cfg, const float* params, void* wcache, void* stream) {
    return refresh_weights(cfg, params, wcache, nullptr, stream);
}
extern "C" int gg_tinyvit_refresh_weights_masked(const GgTinyVitCfg* cfg, const float* params, void* wcache, const uint8_t* only, void* stream) {
    return refresh_weights(cfg, params, wcache, only, stream);
}
static int refresh_weights(const GgTinyVitCfg* cfg, const float* params, void* wcache, const uint8_t* only, void* stream) {
    Model m;
    GG_TRY(build_model(cfg, m));
    GG_CHECK(params && wcache, "gg_tinyvit_refresh_weights: null pointer");
    char* wc = (char*)wcache;
    hipStream_t st = (hipStream_t)stream;
    auto repack_dense = [&](const DenseW& w, const float* pp, const Model& mm, char* c, hipStream_t s) -> int {
        if (only && !only[w.t_w]) return 0;
        GG_TRY(::repack_dense(w, pp, mm, c, s));
        if (w.wn3 >= 0) GG_TRY(gg_split3_bf16(reinterpret_cast<const float*>(c + w.wn), w.N, w.Kp, w.Kp, c + w.wn3, s));
        if (w.wt3 >= 0) GG_TRY(gg_split3_bf16(reinterpret_cast<const float*>(c + w.wt), w.Kp, w.Np, w.Np, c + w.wt3, s));
        return 0;
    };
    auto repack_dw = [&](const DwW& w, const float* pp, const Model& mm, char* c, hipStream_t s) -> int {
        return (only && !only[w.t_w]) ? 0 : ::repack_dw(w, pp, mm, c, s);
    };
    GG_TRY(repack_dense(m.pe1.w, params, m, wc, st));
    GG_TRY(repack_dense(m.pe2.w, params, m, wc, st));
    for (auto& l : m.mb) {
        GG_TRY(repack_dense(l.c1.w, params, m, wc, st));
        GG_TRY(repack_dw(l.c2.w, params, m, wc, st));
        GG_TRY(repack_dense(l.c3.w, params, m, wc, st));
    }
    for (int s = 0; s < 3; ++s) {
        GG_TRY(repack_dense(m.stages[s].merge.c1.w, params, m, wc, st));
        GG_TRY(repack_dw(m.stages[s].merge.c2.w, params, m, wc, st));
        GG_TRY(repack_dense(m.stages[s].merge.c3.w, params, m, wc, st));
        for (auto& b : m.stages[s].blocks) {
            GG_TRY(repack_dense(b.qkv, params, m, wc, st));
            GG_TRY(repack_dense(b.proj, params, m, wc, st));
            GG_TRY(repack_dense(b.fc1, params, m, wc, st));
            GG_TRY(repack_dense(b.fc2, params, m, wc, st));
            GG_TRY(repack_dw(b.local.w, params, m, wc, st));
            if (b.bias_full >= 0 && !(only && !only[b.t_ab]))
                GG_TRY(gg_attention_expand_bias(params + m.tensors[b.t_ab].offset, m.stages[s].heads, m.stages[s].ws, kAttnScale,
                                                wc + b.bias_full, stream));
        }
    }
    return 0;
}
extern "C" int gg_tinyvit_forward(const GgTinyVitCfg* cfg, int batch, int training, const float* params, float* buffers,
                                  int64_t* counters, const void* wcache, const float* x, const float* drop_scales, void* workspace,
                                  float* out, const uint8_t* trainable, void* stream) {
    Model m;
    GG_TRY(build_model(cfg, m));
    GG_CHECK(batch > 0 && params && buffers && wcache && x && workspace && out, "gg_tinyvit_forward: null pointer / bad batch");
    GG_CHECK(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wcache & 255) == 0, "gg_tinyvit_forward: workspace/wcache must be 256-byte aligned");
    Plan p; Layout L;
    plan_make(m, batch, training != 0, p, L, trainable);
    auto body = [&](hipStream_t st) -> int {
        Exec e;
        e.m = &m; e.L = &L; e.B = batch; e.training = training != 0; e.params = params; e.buffers = buffers; e.counters = counters;
        e.wc = (const char*)wcache; e.ws = (char*)workspace; e.st = st; e.drop = drop_scales; e.grads = nullptr;
        e.trainable = trainable;       // NULL: keep every activation a weight gradient could need
        e.exec_init();
        return forward_impl(e, x, out);
    };
    // launch-bound sizes (a serving panorama, small training batches: a few hundred launches of microseconds each) replay a captured graph
    if (!gg_graph_wanted((int64_t)batch * cfg->img_size * cfg->img_size <= (int64_t)64 * 224 * 224)) return body((hipStream_t)stream);
    GgGraphKey key;
    key.add('F').add_bytes(cfg, sizeof(*cfg)).add(batch).add(training).add(params).add(buffers).add(counters).add(wcache).add(x).add(drop_scales).add(workspace).add(out)
       .add_bytes(trainable, trainable ? m.tensors.size() : 0);
    return gg_graph_run(key, (hipStream_t)stream, body);
}
extern "C" int gg_tinyvit_backward(const GgTinyVitCfg* cfg, int batch, const float* params, const void* wcache, const float* drop_scales,
                                   void* workspace, const float* d_out, float* grads, const uint8_t* trainable, void* stream,
                                   GgStageDoneFn stage_done, void* stage_user) {
    Model m;
    GG_TRY(build_model(cfg, m));
    GG_CHECK(batch > 0 && params && wcache && workspace && d_out && grads, "gg_tinyvit_backward: null pointer / bad batch");
    Plan p; Layout L;
    plan_make(m, batch, true, p, L, trainable);          // the SAME mask the training forward was called with: it decides the workspace layout
    Exec e;
    e.m = &m; e.L = &L; e.B = batch; e.training = true; e.params = params; e.buffers = nullptr; e.counters = nullptr;
    e.wc = (const char*)wcache; e.ws = (char*)workspace; e.st = (hipStream_t)stream; e.drop = drop_scales; e.grads = grads;
    e.trainable = trainable; e.stage_done = stage_done; e.stage_user = stage_user;
    if (trainable) {
        // The schedule forms the two gradients of a (weight, bias) / (gamma, beta) pair together (BatchNorm / LayerNorm finalize kernels, the fused
        // frozen-chain forms): a mask that trains one tensor of a pair and freezes the other has no schedule -- refuse it by name instead of silently
        // leaving a gradient at zero.  (Every policy of the reference freezes whole modules: models/tinyvit.py:90-111.)
        std::map<std::string, int> by_name;
        for (size_t i = 0; i < m.tensors.size(); ++i) if (m.tensors[i].kind == GG_KIND_PARAM) by_name[m.tensors[i].name] = (int)i;
        for (const auto& kv : by_name) {
            const std::string& n = kv.first;
            if (n.size() < 7 || n.compare(n.size() - 7, 7, ".weight") != 0) continue;
            const auto it = by_name.find(n.substr(0, n.size() - 7) + ".bias");
            if (it == by_name.end()) continue;
            GG_CHECK((trainable[kv.second] != 0) == (trainable[it->second] != 0),
                     "gg_tinyvit_backward: %s and %s must be trainable or frozen together (requires_grad differs within the pair)", n.c_str(), it->first.c_str());
        }
    }
    if (stage_done || !gg_graph_wanted((int64_t)batch * cfg->img_size * cfg->img_size <= (int64_t)64 * 224 * 224)) {      // a host callback per stage (N > 1): eager
        e.exec_init();
        return backward_impl(e, d_out);
    }
    auto body = [&](hipStream_t st) -> int {
        Exec g = e;
        g.st = st;
        g.exec_init();
        return backward_impl(g, d_out);
    };
    GgGraphKey key;
    key.add('B').add_bytes(cfg, sizeof(*cfg)).add(batch).add(params).add(wcache).add(drop_scales).add(workspace).add(d_out).add(grads)
       .add_bytes(trainable, trainable ? m.tensors.size() : 0);
    return gg_graph_run(key, (hipStream_t)stream, body);
}
