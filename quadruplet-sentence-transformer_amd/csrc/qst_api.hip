// qst_api.hip -- C-ABI entry points of libqst.so: arena layout, encoder handle, forward/backward
// orchestration (which kernels run, in which order, on which buffers). See include/qst.h.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include <unordered_map>
#include <algorithm>

#include "qst_common.h"
#include "qst_kernels.h"

extern "C" int qst_adamw_launch(float*, float*, float*, float*, const uint8_t*, int64_t, float, float, float, float,
                                float, float, float, int64_t, float*, float*, hipStream_t);

static thread_local int g_last_hip_error = 0;
extern "C" int qst_set_hip_error(int code) { g_last_hip_error = code; return code; }
extern "C" int qst_last_hip_error(void) { return g_last_hip_error; }
extern "C" int qst_version(void) { return 100; }

extern "C" const char* qst_strerror(int s) {
    switch (s) {
        case QST_OK: return "ok";
        case QST_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or bad enum)";
        case QST_ERR_UNSUPPORTED: return "unsupported dimensions for the gfx950 kernels";
        case QST_ERR_WORKSPACE: return "workspace or saved-activation arena too small";
        case QST_ERR_HIP: return "HIP runtime error (see qst_last_hip_error)";
        case QST_ERR_NO_DEVICE: return "no HIP device";
        case QST_ERR_COMM: return "RCCL error (see qst_comm_last_error)";
        case QST_ERR_NO_FORWARD: return "no matching training forward: `saved` is not an activation arena that a training forward of this "
                                        "precision and shape (nseq, L) has filled in this process";
        default: return "unknown qst status";
    }
}

// ------------------------------------------------------------------ layout
namespace {

constexpr int64_t kAlign = 256;

struct Seg {
    std::string name;
    int64_t off, numel;
    int rows, cols;     // 2-D view (cols = 1 rows = numel for vectors)
    int decay, gemm;
    int64_t shadow_off; // bf16 shadow arena offset of W (W^T follows at shadow_off + align(numel)); -1 if none
};

struct Layout {
    std::vector<Seg> segs;
    int64_t total = 0, shadow_total = 0;
    // indices
    int word = -1, pos = -1, type = -1, eg = -1, eb = -1, rel = -1;
    std::vector<int> layer0;   // index of w_qkv per layer; following 11 entries are fixed order
};

bool cfg_ok(const qst_config* c) {
    if (!c) return false;
    if (c->arch != QST_ARCH_BERT && c->arch != QST_ARCH_MPNET) return false;
    if (c->hidden_size <= 0 || c->num_layers <= 0 || c->num_heads <= 0 || c->intermediate_size <= 0) return false;
    if (c->vocab_size <= 0 || c->max_position <= 0 || c->type_vocab_size < 0) return false;
    return true;
}

Layout build_layout(const qst_config* c) {
    Layout L;
    const int H = c->hidden_size, I = c->intermediate_size;
    int64_t off = 0, soff = 0;
    auto add = [&](const std::string& nm, int rows, int cols, int decay, int gemm) {
        Seg s;
        s.name = nm; s.off = off; s.rows = rows; s.cols = cols; s.numel = (int64_t)rows * cols;
        s.decay = decay; s.gemm = gemm; s.shadow_off = -1;
        if (gemm) { s.shadow_off = soff; soff += 2 * qst_align_up(s.numel, kAlign); }
        off = qst_align_up(off + s.numel, kAlign);
        L.segs.push_back(s);
        return (int)L.segs.size() - 1;
    };
    L.word = add("word_emb", c->vocab_size, H, 1, 0);
    L.pos = add("pos_emb", c->max_position, H, 1, 0);
    if (c->type_vocab_size > 0) L.type = add("type_emb", c->type_vocab_size, H, 1, 0);
    L.eg = add("emb_ln_g", H, 1, 0, 0);
    L.eb = add("emb_ln_b", H, 1, 0, 0);
    if (c->arch == QST_ARCH_MPNET) L.rel = add("rel_bias", c->rel_buckets, c->num_heads, 1, 0);
    for (int l = 0; l < c->num_layers; ++l) {
        const std::string p = "layer." + std::to_string(l) + ".";
        L.layer0.push_back(add(p + "w_qkv", 3 * H, H, 1, 1));
        add(p + "b_qkv", 3 * H, 1, 0, 0);
        add(p + "w_o", H, H, 1, 1);
        add(p + "b_o", H, 1, 0, 0);
        add(p + "ln1_g", H, 1, 0, 0);
        add(p + "ln1_b", H, 1, 0, 0);
        add(p + "w_1", I, H, 1, 1);
        add(p + "b_1", I, 1, 0, 0);
        add(p + "w_2", H, I, 1, 1);
        add(p + "b_2", H, 1, 0, 0);
        add(p + "ln2_g", H, 1, 0, 0);
        add(p + "ln2_b", H, 1, 0, 0);
    }
    L.total = off;
    L.shadow_total = soff;
    return L;
}
enum { W_QKV = 0, B_QKV, W_O, B_O, LN1_G, LN1_B, W_1, B_1, W_2, B_2, LN2_G, LN2_B };

}  // namespace

// What the training forward that FILLED an activation arena did with dropout, and which kind of arena it left: a backward
// regenerates the masks of that forward -- with ITS thresholds, whatever qst_encoder_set_dropout has been told since (fit()
// and bench.py switch dropout on live encoders; several forwards may be alive before their backwards run) and whichever
// handle runs the backward. Process-wide, keyed by the arena's device address; a backward over an arena no training forward
// of this process has filled, or over the other kind of arena, is refused (QST_ERR_BAD_ARG) instead of guessing the rates.
// (Rounds 2-3 kept a 16-entry ring per handle and fell back on the handle's current rates when it had no entry: wrong
// gradients without an error once the ring wrapped or a backward ran on another handle -- ADVICE r03.)
namespace {
enum { ARENA_BF16 = 0, ARENA_X3 = 1, ARENA_F16 = 2 };   // the bf16 activation arena (bf16 and fp8 forwards) / the fp32 one of bf16x3 / the f16 one
// A record also holds the shape of its forward: an arena whose address was reused by a later allocation of another shape (or a
// backward called with another nseq / L than its forward) is refused instead of read with the wrong plan.
struct FwdRec { uint32_t hidden = 0, attn = 0; int kind = ARENA_BF16; uint64_t seq = 0; int nseq = 0, L = 0; };
std::mutex g_rec_mu;
std::unordered_map<const void*, FwdRec> g_recs;
uint64_t g_rec_seq = 0;
void rec_put(const void* saved, uint32_t hidden, uint32_t attn, int kind, int nseq, int L) {
    std::lock_guard<std::mutex> lk(g_rec_mu);
    g_recs[saved] = FwdRec{hidden, attn, kind, ++g_rec_seq, nseq, L};
    if (g_recs.size() > 4096) {
        // arenas that were freed long ago: keep the 2048 most recently WRITTEN records (by count -- a cut at "sequence number -
        // 2048" dropped live arenas once hot ones had been re-put often enough)
        std::vector<uint64_t> seqs;
        seqs.reserve(g_recs.size());
        for (const auto& kv : g_recs) seqs.push_back(kv.second.seq);
        std::nth_element(seqs.begin(), seqs.end() - 2048, seqs.end());
        const uint64_t cut = *(seqs.end() - 2048);
        for (auto it = g_recs.begin(); it != g_recs.end();) it = it->second.seq < cut ? g_recs.erase(it) : std::next(it);
    }
}
bool rec_get(const void* saved, int kind, int nseq, int L, FwdRec* out) {
    std::lock_guard<std::mutex> lk(g_rec_mu);
    const auto it = g_recs.find(saved);
    if (it == g_recs.end() || it->second.kind != kind || it->second.nseq != nseq || it->second.L != L) return false;
    *out = it->second;
    return true;
}
}  // namespace

struct qst_encoder {
    qst_config cfg;
    Layout lay;
    uint8_t* chunk_decay = nullptr;   // device: decay flag per 256-element chunk
    int32_t* rel_lut = nullptr;       // device: MPNet bucket of (j - i), index (j - i) + 511
    int64_t* shadow_tab = nullptr;    // device: one row per GEMM weight for qst_shadow_all
    int shadow_nseg = 0, shadow_blocks = 0;
    // dropout (qst_encoder_set_dropout): 16-bit thresholds of the hidden-state and attention-probability masks, and the
    // caller's device counter {seed lo, seed hi, step, 0} that every training forward advances
    uint32_t drop_hidden = 0, drop_attn = 0;
    uint32_t* drop_state = nullptr;
    // Where the one-kernel feed-forward block (csrc/ffn.hip) is used (qst_encoder_set_ffn_chain): bit 0 = inference forward
    // (default: the [M, I] tensor never leaves the chip, 121 vs 144 us per layer at M = 32768), bit 1 = training forward,
    // bit 2 = backward. The training variants are correct and tested but measured SLOWER than the two-kernel path (178 vs
    // 144 us and 157 vs 125 us): their side outputs (gelu'(u), h / du: 100-200 MB per call) are stored by the same waves
    // that wait on the LDS-DMA stream, and stores and DMAs retire through one in-order counter.
    int ffn_chain = 1;
    int ln_fusion = 0;      // qst_encoder_set_ln_fusion: 0 = by size, 1 = wherever a fused kernel exists, 2 = never
};

extern "C" int64_t qst_arena_elems(const qst_config* cfg) { return cfg_ok(cfg) ? build_layout(cfg).total : QST_ERR_BAD_ARG; }
// (QST_PREC_F16W: a second arena of the same layout behind the first holds the low halves of the split weights)
extern "C" int64_t qst_shadow_elems(const qst_config* cfg) {
    if (!cfg_ok(cfg)) return QST_ERR_BAD_ARG;
    const int64_t n = build_layout(cfg).shadow_total;
    return cfg->precision == QST_PREC_F16W ? 2 * n : n;
}
// fp8 shadow: the weight bytes of segment s sit at byte offset s.shadow_off (inside the first half of what is the [W | W^T]
// region in bf16-element units), its fp32 row scales at byte offset s.shadow_off + align(numel): same total, in bytes.
extern "C" int64_t qst_shadow8_bytes(const qst_config* cfg) { return cfg_ok(cfg) ? build_layout(cfg).shadow_total : QST_ERR_BAD_ARG; }
extern "C" int qst_arena_num_segments(const qst_config* cfg) { return cfg_ok(cfg) ? (int)build_layout(cfg).segs.size() : QST_ERR_BAD_ARG; }
extern "C" int qst_arena_segment(const qst_config* cfg, int idx, const char** name_out, int64_t* offset_out,
                                 int64_t* numel_out, int32_t* decay_out, int32_t* gemm_out) {
    if (!cfg_ok(cfg)) return QST_ERR_BAD_ARG;
    static thread_local Layout L;      // keeps name storage alive for the caller
    L = build_layout(cfg);
    if (idx < 0 || idx >= (int)L.segs.size()) return QST_ERR_BAD_ARG;
    const Seg& s = L.segs[idx];
    if (name_out) *name_out = s.name.c_str();
    if (offset_out) *offset_out = s.off;
    if (numel_out) *numel_out = s.numel;
    if (decay_out) *decay_out = s.decay;
    if (gemm_out) *gemm_out = s.gemm;
    return QST_OK;
}

// MPNet relative_position_bucket (modeling_mpnet.py:329-348) evaluated on the host in the same float32
// steps torch takes: log(n.float()/max_exact) / math.log(max_distance/max_exact) * (nb - max_exact) -> trunc.
extern "C" int qst_rel_bucket_host(int rel /* j - i */, int num_buckets, int max_distance) {
    int n = -rel;
    const int nb = num_buckets / 2;
    int ret = (n < 0) ? nb : 0;
    n = n < 0 ? -n : n;
    const int max_exact = nb / 2;
    if (n < max_exact) return ret + n;
    const float ratio = (float)n / (float)max_exact;
    const float denom = (float)log((double)max_distance / (double)max_exact);
    float v = logf(ratio) / denom;
    v = v * (float)(nb - max_exact);
    int large = max_exact + (int)v;
    if (large > nb - 1) large = nb - 1;
    return ret + large;
}

extern "C" void qst_encoder_destroy(qst_encoder* e);
extern "C" int qst_encoder_create(const qst_config* cfg, qst_encoder** out) {
    if (!cfg_ok(cfg) || !out) return QST_ERR_BAD_ARG;
    const int d = cfg->hidden_size / cfg->num_heads;
    if (cfg->hidden_size % cfg->num_heads != 0 || (d != 32 && d != 64)) return QST_ERR_UNSUPPORTED;
    if (cfg->hidden_size % 64 != 0 || cfg->intermediate_size % 64 != 0 || cfg->hidden_size > 1024) return QST_ERR_UNSUPPORTED;
    if (cfg->type_vocab_size > 2) return QST_ERR_UNSUPPORTED;
    if (cfg->precision != QST_PREC_BF16 && cfg->precision != QST_PREC_BF16X3 && cfg->precision != QST_PREC_FP8 &&
        cfg->precision != QST_PREC_F16 && cfg->precision != QST_PREC_F16W)
        return QST_ERR_UNSUPPORTED;
    if (cfg->precision == QST_PREC_FP8 && (cfg->hidden_size % 128 != 0 || cfg->intermediate_size % 128 != 0))
        return QST_ERR_UNSUPPORTED;                      // the fp8 K loop takes 128-deep stages
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QST_ERR_NO_DEVICE;
    qst_encoder* e = new qst_encoder();
    e->cfg = *cfg;
    e->lay = build_layout(cfg);
    const int64_t nchunks = e->lay.total / kAlign;
    std::vector<uint8_t> flags((size_t)nchunks, 0);
    for (const Seg& s : e->lay.segs)
        if (s.decay)
            for (int64_t ch = s.off / kAlign; ch < qst_align_up(s.off + s.numel, kAlign) / kAlign; ++ch) flags[(size_t)ch] = 1;
    if (hipMalloc((void**)&e->chunk_decay, (size_t)nchunks) != hipSuccess) { delete e; return QST_ERR_HIP; }
    if (hipMemcpy(e->chunk_decay, flags.data(), (size_t)nchunks, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(e->chunk_decay); delete e; return QST_ERR_HIP;
    }
    if (cfg->arch == QST_ARCH_MPNET) {
        std::vector<int32_t> lut(1023);
        for (int r = -511; r <= 511; ++r) lut[(size_t)(r + 511)] = qst_rel_bucket_host(r, cfg->rel_buckets, cfg->rel_max_distance);
        if (hipMalloc((void**)&e->rel_lut, 1023 * sizeof(int32_t)) != hipSuccess ||
            hipMemcpy(e->rel_lut, lut.data(), 1023 * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(e->chunk_decay); if (e->rel_lut) (void)hipFree(e->rel_lut); delete e; return QST_ERR_HIP;
        }
    }
    {
        std::vector<int64_t> tab;
        int64_t blocks = 0;
        for (const Seg& s : e->lay.segs) {
            if (!s.gemm) continue;
            tab.insert(tab.end(), {s.off, (int64_t)s.rows, (int64_t)s.cols, s.shadow_off,
                                   s.shadow_off + qst_align_up(s.numel, kAlign), blocks});
            blocks += (int64_t)((s.cols + 31) / 32) * ((s.rows + 31) / 32);
        }
        e->shadow_nseg = (int)(tab.size() / 6);
        e->shadow_blocks = (int)blocks;
        if (hipMalloc((void**)&e->shadow_tab, tab.size() * sizeof(int64_t)) != hipSuccess ||
            hipMemcpy(e->shadow_tab, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) {
            qst_encoder_destroy(e);
            return QST_ERR_HIP;
        }
    }
    *out = e;
    return QST_OK;
}

extern "C" void qst_encoder_destroy(qst_encoder* e) {
    if (!e) return;
    if (e->chunk_decay) (void)hipFree(e->chunk_decay);
    if (e->rel_lut) (void)hipFree(e->rel_lut);
    if (e->shadow_tab) (void)hipFree(e->shadow_tab);
    delete e;
}

// ------------------------------------------------------------------ activation arena
namespace {

struct LayerAct {
    size_t qkv, lse, ctx, y1, y1b, xh1, rs1, u, hact, x, xb, xh2, rs2;   // x/xb/xh2/rs2 = layer OUTPUT
};
struct ActPlan {
    size_t pos_ids, x0, x0b, xh0, rs0, s_scratch, pooled, rel;
    size_t dropst;          // uint32[4]: the dropout counter as THIS forward used it (backward rebuilds the masks from it)
    std::vector<LayerAct> layers;
    size_t total;
};

ActPlan plan_acts(const qst_config& c, int nseq, int L, bool training) {
    ActPlan p;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.pos_ids = take(M * 4);
    p.dropst = take(16);
    p.x0 = take(M * H * 4); p.x0b = take(M * H * 2); p.xh0 = take(M * H * 2); p.rs0 = take(M * 4);
    p.s_scratch = take(M * H * 4);
    p.pooled = take((size_t)nseq * H * 4);
    p.rel = (c.arch == QST_ARCH_MPNET) ? take(A * (size_t)2 * L * 4) : 0;      // relative-position vectors [A][2L]
    p.layers.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
        LayerAct& a = p.layers[l];
        if (training || l < 2) {
            a.qkv = take(M * 3 * H * 2); a.lse = take((size_t)nseq * A * L * 4); a.ctx = take(M * H * 2);
            a.y1 = take(M * H * 4); a.y1b = take(M * H * 2); a.xh1 = take(M * H * 2); a.rs1 = take(M * 4);
            a.u = take(M * I * 2); a.hact = take(M * I * 2);
            a.x = take(M * H * 4); a.xb = take(M * H * 2); a.xh2 = take(M * H * 2); a.rs2 = take(M * 4);
        } else {
            a = p.layers[l - 2];      // inference: ping-pong two layer slots (a layer reads the previous slot's x)
        }
    }
    p.total = off;
    return p;
}

// QST_PREC_BF16X3 forward: fp32 activations, inference only (two residual slots ping-pong)
struct X3Plan { size_t pos_ids, x[2], qkv, ctx, s, y1, h, pooled, rel, total; };
X3Plan plan_x3(const qst_config& c, int nseq, int L) {
    X3Plan p;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.pos_ids = take(M * 4);
    p.x[0] = take(M * H * 4); p.x[1] = take(M * H * 4);
    p.qkv = take(M * 3 * H * 4); p.ctx = take(M * H * 4); p.s = take(M * H * 4); p.y1 = take(M * H * 4);
    p.h = take(M * I * 4);
    p.pooled = take((size_t)nseq * H * 4);
    p.rel = (c.arch == QST_ARCH_MPNET) ? take(A * (size_t)L * L * 4) : 0;
    p.total = off;
    return p;
}

// QST_PREC_BF16X3 TRAINING: every tensor the fp32-class backward needs, kept in fp32 (the parity path: sized for parity
// runs, 4x the bf16 path's activation bytes). s0 / s1 / s2 are the PRE-norm inputs of the three LayerNorms (the backward
// recomputes mean and rstd from them), u the pre-GELU tensor.
struct X3Layer { size_t qkv, ctx, s1, y1, u, h, s2, x; };
struct X3TrainPlan { size_t pos_ids, dropst, s0, x0, pooled, rel, total; std::vector<X3Layer> layers; };
X3TrainPlan plan_x3_train(const qst_config& c, int nseq, int L) {
    X3TrainPlan p;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.pos_ids = take(M * 4);
    p.dropst = take(16);                               // the dropout state {seed, step} this forward ran under
    p.s0 = take(M * H * 4); p.x0 = take(M * H * 4);
    p.pooled = take((size_t)nseq * H * 4);
    p.rel = (c.arch == QST_ARCH_MPNET) ? take(A * (size_t)L * L * 4) : 0;
    p.layers.resize(c.num_layers);
    for (auto& a : p.layers) {
        a.qkv = take(M * 3 * H * 4); a.ctx = take(M * H * 4); a.s1 = take(M * H * 4); a.y1 = take(M * H * 4);
        a.u = take(M * I * 4); a.h = take(M * I * 4); a.s2 = take(M * H * 4); a.x = take(M * H * 4);
    }
    p.total = off;
    return p;
}
// its backward's scratch: gradient activations, one transposed weight (the dgrad's B operand) and the attention backward's
// per-query statistics
struct X3BwdPlan { size_t dx, dy, ds, dbig, dctx, dqkv, wT, drel, astats, total; };
X3BwdPlan plan_x3_bwd(const qst_config& c, int nseq, int L) {
    X3BwdPlan p;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.dx = take(M * H * 4); p.dy = take(M * H * 4); p.ds = take(M * H * 4); p.dbig = take(M * I * 4);
    p.dctx = take(M * H * 4); p.dqkv = take(M * 3 * H * 4);
    p.wT = take((H * I > 3 * H * H ? H * I : 3 * H * H) * 4);
    p.drel = (c.arch == QST_ARCH_MPNET) ? take(A * (size_t)L * L * 4) : 0;
    p.astats = take(qst_attention_bwd_x3_scratch_bytes(nseq, L, (int)A));
    p.total = off;
    return p;
}

// QST_PREC_FP8 forward (inference): MXFP8 operands for every Linear, bf16 attention, fp32 residual stream / LayerNorm
struct MxPlan { size_t pos_ids, x[2], xb, xq, xs, qkv, ctx, cq, cs, s, y1, y1b, yq, ys, hq, hs, pooled, rel, total; };
MxPlan plan_mx(const qst_config& c, int nseq, int L) {
    MxPlan p;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.pos_ids = take(M * 4);
    p.x[0] = take(M * H * 4); p.x[1] = take(M * H * 4);
    p.xb = take(M * H * 2); p.xq = take(M * H); p.xs = take(M * H / 32);
    p.qkv = take(M * 3 * H * 2); p.ctx = take(M * H * 2); p.cq = take(M * H); p.cs = take(M * H / 32);
    p.s = take(M * H * 4); p.y1 = take(M * H * 4); p.y1b = take(M * H * 2); p.yq = take(M * H); p.ys = take(M * H / 32);
    p.hq = take(M * I); p.hs = take(M * I / 32);
    p.pooled = take((size_t)nseq * H * 4);
    p.rel = (c.arch == QST_ARCH_MPNET) ? take(A * (size_t)2 * L * 4) : 0;
    p.total = off;
    return p;
}

struct BwdPlan { size_t dxa, dxb, ds, dsb, dsb1, du, dctx, dqkv, drel, lnred, lnred_stride, delta, total; };
BwdPlan plan_bwd(const qst_config& c, int nseq, int L) {
    BwdPlan p;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.dxa = take(M * H * 4); p.dxb = take(M * H * 4); p.ds = take(M * H * 4); p.dsb = take(M * H * 2);
    p.dsb1 = take(M * H * 2);
    p.du = take(M * I * 2); p.dctx = take(M * H * 2); p.dqkv = take(M * 3 * H * 2);
    p.drel = (c.arch == QST_ARCH_MPNET) ? take(A * (size_t)2 * L * 4) : 0;
    p.lnred_stride = (qst_ln_bwd_scratch_bytes((int)M, (int)H) + 255) / 256 * 256;
    p.lnred = take(p.lnred_stride * (size_t)(2 * c.num_layers + 1));
    p.delta = take((size_t)nseq * A * L * 4);
    p.total = off;
    return p;
}

int shape_ok(const qst_encoder* e, int nseq, int L) {
    if (!e || nseq <= 0 || L <= 0) return QST_ERR_BAD_ARG;
    if (L % 32 != 0 || L > 512) return QST_ERR_UNSUPPORTED;
    if ((int64_t)nseq * L * e->cfg.intermediate_size * 2 >= ((int64_t)1 << 32)) return QST_ERR_UNSUPPORTED;  // 32-bit buffer offsets
    if (e->cfg.arch == QST_ARCH_BERT && L > e->cfg.max_position) return QST_ERR_UNSUPPORTED;
    if (e->cfg.arch == QST_ARCH_MPNET && L + e->cfg.pad_token_id + 1 > e->cfg.max_position) return QST_ERR_UNSUPPORTED;
    return QST_OK;
}

int nt3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, const float* resid, int ldr,
        int M, int N, int K, int epi, hipStream_t st) {
    QstGemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    return qst_gemm_nt_x3(&g, epi, st);
}
// The GEMM helpers below take the dropout mask of their next call from a one-shot description (drop_next;
// QstGemmArgs.drop / drop_where), set right before the call.
struct DropNext { QstDrop d; int where; };
static thread_local DropNext t_drop = {{nullptr, 0u, 0u}, 0};
// `state` = the counter copy inside the activation arena of the forward at hand: several training forwards may be live
// before their backwards run (fit() encodes the four columns one after the other), each with its own step value
// thresholds in force for one pass: the handle's current ones (forward) or the recorded ones of the forward a backward undoes
struct DropThr { uint32_t hidden, attn; };
static QstDrop drop_of(const DropThr& t, const void* state, bool attn, uint32_t site) {
    QstDrop d = {nullptr, site, 0u};
    const uint32_t thr = attn ? t.attn : t.hidden;
    if (state && thr) { d.state = (const uint32_t*)state; d.thr16 = thr; }
    return d;
}
static void drop_next(const DropThr& t, const void* state, bool on, uint32_t site, int where) {
    t_drop = DropNext{{nullptr, 0u, 0u}, 0};
    if (on && t.hidden) t_drop = DropNext{drop_of(t, state, false, site), where};
}
static void take_drop(QstGemmArgs& g) {
    g.drop = t_drop.d; g.drop_where = t_drop.where;
    t_drop = DropNext{{nullptr, 0u, 0u}, 0};
}
extern "C" int qst_encoder_set_ffn_chain(qst_encoder* e, int mask) {
    if (!e || mask < 0 || mask > 7) return QST_ERR_BAD_ARG;
    e->ffn_chain = mask;
    return QST_OK;
}
extern "C" int qst_encoder_set_ln_fusion(qst_encoder* e, int mode) {
    if (!e || mode < 0 || mode > 2) return QST_ERR_BAD_ARG;
    e->ln_fusion = mode;
    return QST_OK;
}
extern "C" int qst_encoder_set_dropout(qst_encoder* e, float p_hidden, float p_attn, uint32_t* state_dev) {
    if (!e || !(p_hidden >= 0.f && p_hidden < 1.f) || !(p_attn >= 0.f && p_attn < 1.f)) return QST_ERR_BAD_ARG;
    if ((p_hidden > 0.f || p_attn > 0.f) && !state_dev) return QST_ERR_BAD_ARG;
    auto thr = [](float p) { const long t = lroundf(p * 65536.f); return (uint32_t)(t > 65535 ? 65535 : t); };
    e->drop_hidden = thr(p_hidden); e->drop_attn = thr(p_attn);
    e->drop_state = (e->drop_hidden || e->drop_attn) ? state_dev : nullptr;
    return QST_OK;
}

// The kernels whose operands / 16-bit outputs are bf16 (QST_PREC_BF16, and the backward of QST_PREC_FP8) or IEEE half
// (QST_PREC_F16): the same sources compiled on either type (qst_common.h: op16), entry points qst_* and qst_*_f16. The bf16
// forward / backward below are written once against this table.
struct OpKernels {
    decltype(&qst_gemm_nt) gemm_nt;
    decltype(&qst_gemm_nt_ln) gemm_nt_ln;
    decltype(&qst_ffn_chain) ffn_chain;
    decltype(&qst_gemm_tn_group) gemm_tn_group;
    decltype(&qst_embed_ln_fwd_drop) embed_ln_fwd_drop;
    decltype(&qst_ln_fwd) ln_fwd;
    decltype(&qst_ln_bwd_drop) ln_bwd_drop;
    decltype(&qst_attention_fwd_ex) attention_fwd_ex;
    decltype(&qst_attention_bwd_ex) attention_bwd_ex;
    decltype(&qst_shadow_all) shadow_all;
    int arena_kind;            // what a training forward through these kernels leaves behind (FwdRec.kind)
};
const OpKernels kOpBf16 = {qst_gemm_nt, qst_gemm_nt_ln, qst_ffn_chain, qst_gemm_tn_group, qst_embed_ln_fwd_drop, qst_ln_fwd,
                           qst_ln_bwd_drop, qst_attention_fwd_ex, qst_attention_bwd_ex, qst_shadow_all, ARENA_BF16};
const OpKernels kOpF16 = {qst_gemm_nt_f16, qst_gemm_nt_ln_f16, qst_ffn_chain_f16, qst_gemm_tn_group_f16,
                          qst_embed_ln_fwd_drop_f16, qst_ln_fwd_f16, qst_ln_bwd_drop_f16, qst_attention_fwd_ex_f16,
                          qst_attention_bwd_ex_f16, qst_shadow_all_f16, ARENA_F16};
const OpKernels& op_kernels(const qst_config& c) {
    return (c.precision == QST_PREC_F16 || c.precision == QST_PREC_F16W) ? kOpF16 : kOpBf16;
}

// sat: forward launches of the f16 build saturate their 16-bit outputs (QstGemmArgs.sat16); the bf16 build ignores it
// B2: the low halves of split weights (QST_PREC_F16W forward), or null
int nt(const OpKernels& K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* C2, const void* aux,
       const float* bias, const float* resid, int ldr, int M, int N, int K_, int epi, bool sat, hipStream_t st,
       const void* B2 = nullptr, int b2_n0 = 0) {
    QstGemmArgs g{};
    g.A = A; g.B = B; g.B2 = B2; g.b2_n0 = b2_n0; g.C = C; g.C2 = C2; g.aux = aux; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K_; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
    g.sat16 = sat ? 1 : 0;
    take_drop(g);
    return K.gemm_nt(&g, epi, st);
}
// GEMM with the following LayerNorm (mode 0) / LayerNorm backward (mode 1) fused into its epilogue (N = H = 384)
int nt_ln(const OpKernels& K, const void* A, int lda, const void* B, int ldb, float* C, void* C2, const float* bias,
          const float* resid, int M, int H, int K_, int mode, const float* gamma, const float* beta, float eps, void* xhat,
          float* rstd, float* partials, hipStream_t st, const void* B2 = nullptr) {
    QstGemmArgs g{};
    g.A = A; g.B = B; g.B2 = B2; g.C = C; g.C2 = C2; g.bias = bias; g.resid = resid;
    g.M = M; g.N = H; g.K = K_; g.lda = lda; g.ldb = ldb; g.ldc = H; g.ldr = H;
    take_drop(g);
    QstLnEpi e{};
    e.gamma = gamma; e.beta = beta; e.eps = eps; e.xhat = xhat; e.rstd = rstd; e.partials = partials;
    return K.gemm_nt_ln(&g, &e, mode, st);
}

// the feed-forward block as one kernel (csrc/ffn.hip): mode 0 forward, mode 1 backward
int ffn_chain(const OpKernels& K, const void* A, const void* B1, const void* B2, const float* bias1, const float* bias2, const float* resid,
              const void* aux, void* save_gp, void* save_h, float* C, void* C2, int M, int H, int I, int mode,
              const float* gamma, const float* beta, float eps, void* xhat, float* rstd, float* partials, hipStream_t st) {
    QstFfnArgs g{};
    g.A = A; g.B1 = B1; g.B2 = B2; g.bias1 = bias1; g.bias2 = bias2; g.resid = resid; g.aux = aux;
    g.save_gp = save_gp; g.save_h = save_h; g.C = C; g.C2 = C2; g.M = M; g.H = H; g.I = I;
    QstLnEpi e{};
    e.gamma = gamma; e.beta = beta; e.eps = eps; e.xhat = xhat; e.rstd = rstd; e.partials = partials;
    return K.ffn_chain(&g, &e, mode, st);
}

constexpr int kFuseLnMinRows = 16384;      // token rows from which the fused GEMM+LayerNorm kernels win (see forward)
// ... and for H = 512 / 768 / 1024, where a row spans several 256-column tiles whose workgroups exchange the row statistics
// (gemm8.hip): from two tiles per CU on (256 x 256, or 128 x 384 where only that gives two). Measured at H = 768 (tools/ln8_bench.py, fused against the GEMM + row-kernel pair):
// M = 49,152: forward -6 ... -10%, backward -8 ... -9%; M = 196,608: forward -6 ... -10%, backward -13 ... -19%.
static bool fuse_ln_rows(int H, int M, int mode) {
    if (!qst_gemm_nt_ln_supported(H) || mode == 2) return false;
    if (mode == 1) return true;
    if (qst_gemm_nt_ln_block_rows(H) == 128) return M >= kFuseLnMinRows;
    return (int64_t)((M + 255) / 256) * (H / 256) >= 512 || (H % 384 == 0 && (int64_t)((M + 127) / 128) * (H / 384) >= 512);
}

#define QST_TRY(expr) do { int _rc = (expr); if (_rc != QST_OK) return _rc; } while (0)

}  // namespace

// (behind the bf16 activation arena of an fp8 TRAINING forward: the MXFP8 operand copies; see forward_mx_train)
struct MxTrainTmp { size_t xq, xs, cq, cs, yq, ys, hq, hs, total; };
static MxTrainTmp plan_mx_train_tmp(const qst_config& c, int nseq, int L, size_t base) {
    MxTrainTmp t;
    const size_t M = (size_t)nseq * L, H = c.hidden_size, I = c.intermediate_size;
    size_t off = base;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    t.xq = take(M * H); t.xs = take(M * H / 32 + 1024);
    t.cq = take(M * H); t.cs = take(M * H / 32 + 1024);
    t.yq = take(M * H); t.ys = take(M * H / 32 + 1024);
    t.hq = take(M * I); t.hs = take(M * I / 32 + 1024);
    t.total = off;
    return t;
}
extern "C" size_t qst_encoder_saved_bytes(const qst_encoder* e, int nseq, int L, int training) {
    if (shape_ok(e, nseq, L) != QST_OK) return 0;
    if (e->cfg.precision == QST_PREC_BF16X3) return training ? plan_x3_train(e->cfg, nseq, L).total : plan_x3(e->cfg, nseq, L).total;
    if (e->cfg.precision == QST_PREC_FP8)
        return training ? plan_mx_train_tmp(e->cfg, nseq, L, plan_acts(e->cfg, nseq, L, true).total).total : plan_mx(e->cfg, nseq, L).total;
    return plan_acts(e->cfg, nseq, L, training != 0).total;
}
extern "C" size_t qst_encoder_bwd_workspace_bytes(const qst_encoder* e, int nseq, int L) {
    if (shape_ok(e, nseq, L) != QST_OK) return 0;
    if (e->cfg.precision == QST_PREC_BF16X3) return plan_x3_bwd(e->cfg, nseq, L).total;
    return plan_bwd(e->cfg, nseq, L).total;
}

extern "C" int qst_refresh_shadow(const qst_encoder* e, const float* params, void* shadow, void* stream) {
    if (!e || !params || !shadow) return QST_ERR_BAD_ARG;
    // (a QST_PREC_F16 / F16W handle fills the same layout with IEEE half; every other handle with bf16)
    if (e->cfg.precision == QST_PREC_F16W)
        return qst_shadow_all_split_f16(params, shadow, (uint16_t*)shadow + e->lay.shadow_total, e->shadow_tab, e->shadow_nseg,
                                        e->shadow_blocks, stream);
    return op_kernels(e->cfg).shadow_all(params, shadow, e->shadow_tab, e->shadow_nseg, e->shadow_blocks, stream);
}

// QST_PREC_FP8: every GEMM weight as MXFP8 -- e4m3 bytes at the segment's shadow offset, E8M0 block scales (one per 32
// input features of an output row) behind them at shadow_off + align(numel); the buffer is qst_shadow8_bytes() long.
extern "C" int qst_refresh_shadow_mx(const qst_encoder* e, const float* params, void* shadow_mx, void* stream) {
    if (!e || !params || !shadow_mx) return QST_ERR_BAD_ARG;
    for (const Seg& s : e->lay.segs) {
        if (!s.gemm) continue;
        if (s.cols % 32 != 0) return QST_ERR_UNSUPPORTED;
        uint8_t* q = (uint8_t*)shadow_mx + s.shadow_off;
        uint8_t* sc = q + qst_align_up(s.numel, kAlign);
        int rc = qst_quant_mx(params + s.off, 0, s.rows, s.cols, q, sc, stream);
        if (rc != QST_OK) return rc;
    }
    return QST_OK;
}

// fp8 matrix-core forward (QST_PREC_FP8; BASELINE configs[4]): the operator sequence of the bf16 forward with every
// Linear on MXFP8 operands (qst_gemm_nt_f8). Activations are quantised where they are produced when the producer is a
// GEMM (gelu(u) never exists in another format) or a LayerNorm (qst_ln_fwd_mx), and by qst_quant_mx from the bf16 tensor
// the attention kernel writes.
// fp8 GEMM + LayerNorm in one launch (csrc/gemm8.hip gemm_nt8_ln_kernel<0, *, true>): y fp32, y 16-bit / xhat / rstd (nullable)
// and y as MXFP8 for the next fp8 GEMM
static int f8_ln(const void* Aq, const void* As, int K, const void* Bq, const void* Bs, const float* bias, const float* resid,
                 const float* gamma, const float* beta, float eps, int M, int N, float* y, void* yb, void* xh, float* rs,
                 void* yq, void* ys, const QstDrop* drop, hipStream_t st) {
    QstGemmArgs g{};
    g.A = Aq; g.aux = As; g.B = Bq; g.bscale = (const float*)Bs; g.C = y; g.C2 = yb; g.C3 = yq; g.C4 = ys; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N;
    if (drop) { g.drop = *drop; g.drop_where = 1; }
    QstLnEpi e{};
    e.gamma = gamma; e.beta = beta; e.eps = eps; e.xhat = xh; e.rstd = rs;
    return qst_gemm_nt8_f8_ln(&g, &e, st);
}

static int forward_mx(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids, int nseq, int L,
                      const float* params, const void* shadow, float* out_emb, float* out_tok, void* saved, size_t saved_bytes,
                      hipStream_t st) {
    const qst_config& c = e->cfg;
    const MxPlan p = plan_mx(c, nseq, L);
    if (saved_bytes < p.total) return QST_ERR_WORKSPACE;
    char* sv = (char*)saved;
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    auto WQ = [&](int seg) { return (const uint8_t*)shadow + lay.segs[seg].shadow_off; };
    auto WS = [&](int seg) { return (const uint8_t*)shadow + lay.segs[seg].shadow_off + qst_align_up(lay.segs[seg].numel, kAlign); };
    auto gemm = [&](const void* Aq, const void* As, int K, int wseg, void* Cout, void* C2, int N, int bseg, const float* resid, int epi) {
        QstGemmArgs g{};
        g.A = Aq; g.aux = As; g.B = WQ(wseg); g.bscale = (const float*)WS(wseg); g.C = Cout; g.C2 = C2; g.bias = P(bseg); g.resid = resid;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N;
        take_drop(g);
        return qst_gemm_nt_f8(&g, epi, st);
    };
    int32_t* pos_ids = (int32_t*)(sv + p.pos_ids);
    QST_TRY(qst_position_ids(ids, nseq, L, c.arch, c.pad_token_id, pos_ids, st));
    float* x = (float*)(sv + p.x[0]);
    QST_TRY(qst_embed_ln_fwd_mx(ids, type_ids, pos_ids, P(lay.word), P(lay.pos), lay.type >= 0 ? P(lay.type) : nullptr,
                                P(lay.eg), P(lay.eb), c.layer_norm_eps, M, H, x, nullptr, sv + p.xq, sv + p.xs, st));
    const float* rel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        QST_TRY(qst_rel_pos_fwd(P(lay.rel), e->rel_lut, A, L, (float*)(sv + p.rel), st));
        rel = (const float*)(sv + p.rel);
    }
    float* s = (float*)(sv + p.s);
    float* y1 = (float*)(sv + p.y1);
    // H = 512 / 768 / 1024, two tiles per CU or more: projection + LayerNorm (+ MX emission) as one launch (gemm8.hip)
    const bool fuse_ln = qst_gemm_nt8_ln_supported(H) != 0 && fuse_ln_rows(H, M, e->ln_fusion) && H % 128 == 0 && I % 128 == 0;
    for (int l = 0; l < c.num_layers; ++l) {
        const int b = lay.layer0[l];
        float* xn = (float*)(sv + p.x[(l + 1) & 1]);
        QST_TRY(gemm(sv + p.xq, sv + p.xs, H, b + W_QKV, sv + p.qkv, nullptr, 3 * H, b + B_QKV, nullptr, QST_EPI_BF16));
        {
            QstAttnDesc q{};
            q.qkv = sv + p.qkv; q.mask = mask; q.rel_pos = rel; q.nseq = nseq; q.L = L; q.A = A; q.d = d;
            q.ctx = sv + p.ctx;
            QST_TRY(qst_attention_fwd_ex(&q, st));
        }
        QST_TRY(qst_quant_mx(sv + p.ctx, 1, M, H, sv + p.cq, sv + p.cs, st));
        if (fuse_ln) {
            QST_TRY(f8_ln(sv + p.cq, sv + p.cs, H, WQ(b + W_O), WS(b + W_O), P(b + B_O), x, P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps,
                          M, H, y1, nullptr, nullptr, nullptr, sv + p.yq, sv + p.ys, nullptr, st));
        } else {
            QST_TRY(gemm(sv + p.cq, sv + p.cs, H, b + W_O, s, nullptr, H, b + B_O, x, QST_EPI_F32_RESID));
            QST_TRY(qst_ln_fwd_mx(s, P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps, M, H, y1, nullptr, sv + p.yq, sv + p.ys, st));
        }
        QST_TRY(gemm(sv + p.yq, sv + p.ys, H, b + W_1, sv + p.hq, sv + p.hs, I, b + B_1, nullptr, QST_EPI_GELU_MX));
        if (fuse_ln) {
            QST_TRY(f8_ln(sv + p.hq, sv + p.hs, I, WQ(b + W_2), WS(b + W_2), P(b + B_2), y1, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps,
                          M, H, xn, nullptr, nullptr, nullptr, sv + p.xq, sv + p.xs, nullptr, st));
        } else {
            QST_TRY(gemm(sv + p.hq, sv + p.hs, I, b + W_2, s, nullptr, H, b + B_2, y1, QST_EPI_F32_RESID));
            QST_TRY(qst_ln_fwd_mx(s, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, M, H, xn, nullptr, sv + p.xq, sv + p.xs, st));
        }
        x = xn;
    }
    QST_TRY(qst_pool_norm_fwd(x, mask, nseq, L, H, c.normalize, out_emb, (float*)(sv + p.pooled), st));
    if (out_tok) QST_HIP_CHECK(hipMemcpyAsync(out_tok, x, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    return QST_OK;
}

// QST_PREC_FP8 TRAINING forward (BASELINE configs[4] as a fine-tuning configuration): every Linear of the forward runs on
// the fp8 matrix cores (MXFP8 weights and activations, as forward_mx) and leaves, in the bf16 path's activation arena
// (ActPlan), exactly what the bf16 backward reads -- bf16 copies of every GEMM input, xhat / rstd of every LayerNorm, the
// attention statistics, gelu'(u) and h. The backward is then qst_encoder_backward_stage as it is, on the bf16 shadows:
// fp8 forward GEMMs, bf16 dgrad / wgrad from fp32 master weights. The MXFP8 operand copies live behind the arena.
static int forward_mx_train(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids, int nseq, int L,
                            const float* params, const void* shadow, float* out_emb, float* out_tok, void* saved,
                            size_t saved_bytes, hipStream_t st) {
    const qst_config& c = e->cfg;
    const ActPlan p = plan_acts(c, nseq, L, true);
    const MxTrainTmp t = plan_mx_train_tmp(c, nseq, L, p.total);
    if (saved_bytes < t.total) return QST_ERR_WORKSPACE;
    char* sv = (char*)saved;
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    auto WQ = [&](int seg) { return (const uint8_t*)shadow + lay.segs[seg].shadow_off; };
    auto WS = [&](int seg) { return (const uint8_t*)shadow + lay.segs[seg].shadow_off + qst_align_up(lay.segs[seg].numel, kAlign); };
    // dropout exactly as the bf16 training forward has it (same sites, same counter-based masks, the same snapshot beside
    // the activations): the bf16 backward over this arena recomputes the masks from that record
    const bool dropping = e->drop_state != nullptr;
    const void* dst8 = sv + p.dropst;
    const DropThr thr = dropping ? DropThr{e->drop_hidden, e->drop_attn} : DropThr{0u, 0u};
    auto gemm = [&](const void* Aq, const void* As, int K, int wseg, void* Cout, void* C2, int N, int bseg, const float* resid, int epi,
                    int64_t drop_site = -1) {             // (site 0 is a real site: layer 0's attention output)
        QstGemmArgs g{};
        g.A = Aq; g.aux = As; g.B = WQ(wseg); g.bscale = (const float*)WS(wseg); g.C = Cout; g.C2 = C2; g.bias = P(bseg); g.resid = resid;
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N;
        if (dropping && thr.hidden && drop_site >= 0) { g.drop = drop_of(thr, dst8, false, (uint32_t)drop_site); g.drop_where = 1; }
        return qst_gemm_nt_f8(&g, epi, st);
    };
    {
        rec_put(saved, thr.hidden, thr.attn, ARENA_BF16, nseq, L);
    }
    int32_t* pos_ids = (int32_t*)(sv + p.pos_ids);
    QST_TRY(qst_forward_prologue(ids, nseq, L, c.arch, c.pad_token_id, pos_ids, dropping ? e->drop_state : nullptr,
                                 dropping ? (uint32_t*)(sv + p.dropst) : nullptr, st));
    {
        const QstDrop de = drop_of(thr, dst8, false, QST_DROP_SITE_EMBED);
        QST_TRY(qst_embed_ln_fwd_mx_train(ids, type_ids, pos_ids, P(lay.word), P(lay.pos), lay.type >= 0 ? P(lay.type) : nullptr,
                                          P(lay.eg), P(lay.eb), c.layer_norm_eps, M, H, (float*)(sv + p.x0), sv + p.x0b, sv + p.xh0,
                                          (float*)(sv + p.rs0), sv + t.xq, sv + t.xs, dropping ? &de : nullptr, st));
    }
    const float* rel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        QST_TRY(qst_rel_pos_fwd(P(lay.rel), e->rel_lut, A, L, (float*)(sv + p.rel), st));
        rel = (const float*)(sv + p.rel);
    }
    const float* x = (const float*)(sv + p.x0);
    float* s = (float*)(sv + p.s_scratch);
    const bool fuse_ln = qst_gemm_nt8_ln_supported(H) != 0 && fuse_ln_rows(H, M, e->ln_fusion) && H % 128 == 0 && I % 128 == 0;
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerAct& a = p.layers[l];
        const int b = lay.layer0[l];
        QST_TRY(gemm(sv + t.xq, sv + t.xs, H, b + W_QKV, sv + a.qkv, nullptr, 3 * H, b + B_QKV, nullptr, QST_EPI_BF16));
        {
            QstAttnDesc q{};
            q.qkv = sv + a.qkv; q.mask = mask; q.rel_pos = rel; q.nseq = nseq; q.L = L; q.A = A; q.d = d;
            q.ctx = sv + a.ctx; q.lse = (float*)(sv + a.lse);
            if (dropping) q.drop = drop_of(thr, dst8, true, QST_DROP_SITE_PROBS(l));
            QST_TRY(qst_attention_fwd_ex(&q, st));
        }
        QST_TRY(qst_quant_mx(sv + a.ctx, 1, M, H, sv + t.cq, sv + t.cs, st));
        QstDrop dd{};
        auto hd = [&](uint32_t site) -> const QstDrop* {
            if (!dropping || !thr.hidden) return nullptr;
            dd = drop_of(thr, dst8, false, site);
            return &dd;
        };
        if (fuse_ln) {
            QST_TRY(f8_ln(sv + t.cq, sv + t.cs, H, WQ(b + W_O), WS(b + W_O), P(b + B_O), x, P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps,
                          M, H, (float*)(sv + a.y1), sv + a.y1b, sv + a.xh1, (float*)(sv + a.rs1), sv + t.yq, sv + t.ys,
                          hd(QST_DROP_SITE_ATTN_OUT(l)), st));
        } else {
            QST_TRY(gemm(sv + t.cq, sv + t.cs, H, b + W_O, s, nullptr, H, b + B_O, x, QST_EPI_F32_RESID, QST_DROP_SITE_ATTN_OUT(l)));
            QST_TRY(qst_ln_fwd_mx_train(s, P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps, M, H, (float*)(sv + a.y1), sv + a.y1b,
                                        sv + a.xh1, (float*)(sv + a.rs1), sv + t.yq, sv + t.ys, st));
        }
        // FFN-1: gelu'(u) and h leave as bf16 (the backward's operands) and, from the same epilogue, the bf16-rounded h as
        // MXFP8 for FFN-2
        {
            QstGemmArgs g{};
            g.A = sv + t.yq; g.aux = sv + t.ys; g.B = WQ(b + W_1); g.bscale = (const float*)WS(b + W_1);
            g.C = sv + a.u; g.C2 = sv + a.hact; g.C3 = sv + t.hq; g.C4 = sv + t.hs; g.bias = P(b + B_1);
            g.M = M; g.N = I; g.K = H; g.lda = H; g.ldb = H; g.ldc = I; g.ldr = I;
            QST_TRY(qst_gemm_nt_f8(&g, QST_EPI_GELU_MX_TRAIN, st));
        }
        if (fuse_ln) {
            QST_TRY(f8_ln(sv + t.hq, sv + t.hs, I, WQ(b + W_2), WS(b + W_2), P(b + B_2), (const float*)(sv + a.y1), P(b + LN2_G),
                          P(b + LN2_B), c.layer_norm_eps, M, H, (float*)(sv + a.x), sv + a.xb, sv + a.xh2, (float*)(sv + a.rs2),
                          sv + t.xq, sv + t.xs, hd(QST_DROP_SITE_FFN_OUT(l)), st));
        } else {
            QST_TRY(gemm(sv + t.hq, sv + t.hs, I, b + W_2, s, nullptr, H, b + B_2, (const float*)(sv + a.y1), QST_EPI_F32_RESID,
                         QST_DROP_SITE_FFN_OUT(l)));
            QST_TRY(qst_ln_fwd_mx_train(s, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, M, H, (float*)(sv + a.x), sv + a.xb,
                                        sv + a.xh2, (float*)(sv + a.rs2), sv + t.xq, sv + t.xs, st));
        }
        x = (const float*)(sv + a.x);
    }
    QST_TRY(qst_pool_norm_fwd(x, mask, nseq, L, H, c.normalize, out_emb, (float*)(sv + p.pooled), st));
    if (out_tok) QST_HIP_CHECK(hipMemcpyAsync(out_tok, x, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    return QST_OK;
}

// Parity-precision forward (QST_PREC_BF16X3): same operator sequence as below on fp32 activations and the fp32
// master weights, contractions through split-bf16 x3 MFMA kernels (csrc/x3.hip).
static int forward_x3(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids, int nseq, int L,
                      const float* params, float* out_emb, float* out_tok, void* saved, size_t saved_bytes, hipStream_t st) {
    const qst_config& c = e->cfg;
    const X3Plan p = plan_x3(c, nseq, L);
    if (saved_bytes < p.total) return QST_ERR_WORKSPACE;
    char* sv = (char*)saved;
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    int32_t* pos_ids = (int32_t*)(sv + p.pos_ids);
    QST_TRY(qst_position_ids(ids, nseq, L, c.arch, c.pad_token_id, pos_ids, st));
    float* x = (float*)(sv + p.x[0]);
    QST_TRY(qst_embed_ln_fwd(ids, type_ids, pos_ids, P(lay.word), P(lay.pos), lay.type >= 0 ? P(lay.type) : nullptr,
                             P(lay.eg), P(lay.eb), c.layer_norm_eps, M, H, x, nullptr, nullptr, nullptr, st));
    const float* rel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        QST_TRY(qst_rel_bias_fwd(P(lay.rel), e->rel_lut, A, L, (float*)(sv + p.rel), st));
        rel = (const float*)(sv + p.rel);
    }
    float* qkv = (float*)(sv + p.qkv); float* ctx = (float*)(sv + p.ctx); float* s = (float*)(sv + p.s);
    float* y1 = (float*)(sv + p.y1); float* h = (float*)(sv + p.h);
    for (int l = 0; l < c.num_layers; ++l) {
        const int b = lay.layer0[l];
        float* xn = (float*)(sv + p.x[(l + 1) & 1]);
        QST_TRY(nt3(x, H, P(b + W_QKV), H, qkv, 3 * H, P(b + B_QKV), nullptr, 0, M, 3 * H, H, 0, st));
        QST_TRY(qst_attention_fwd_x3(qkv, mask, rel, nseq, L, A, d, ctx, st));
        QST_TRY(nt3(ctx, H, P(b + W_O), H, s, H, P(b + B_O), x, H, M, H, H, 1, st));
        QST_TRY(qst_ln_fwd(s, P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps, M, H, y1, nullptr, nullptr, nullptr, st));
        QST_TRY(nt3(y1, H, P(b + W_1), H, h, I, P(b + B_1), nullptr, 0, M, I, H, 2, st));
        QST_TRY(nt3(h, I, P(b + W_2), I, s, H, P(b + B_2), y1, H, M, H, I, 1, st));
        QST_TRY(qst_ln_fwd(s, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, M, H, xn, nullptr, nullptr, nullptr, st));
        x = xn;
    }
    QST_TRY(qst_pool_norm_fwd(x, mask, nseq, L, H, c.normalize, out_emb, (float*)(sv + p.pooled), st));
    if (out_tok) QST_HIP_CHECK(hipMemcpyAsync(out_tok, x, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    return QST_OK;
}

// QST_PREC_BF16X3 training forward: the same arithmetic as forward_x3, every intermediate kept (X3TrainPlan)
static int forward_x3_train(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids, int nseq, int L,
                            const float* params, float* out_emb, float* out_tok, void* saved, size_t saved_bytes, hipStream_t st) {
    const qst_config& c = e->cfg;
    const X3TrainPlan p = plan_x3_train(c, nseq, L);
    if (saved_bytes < p.total) return QST_ERR_WORKSPACE;
    char* sv = (char*)saved;
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    auto F = [&](size_t o) { return (float*)(sv + o); };
    int32_t* pos_ids = (int32_t*)(sv + p.pos_ids);
    // dropout as the bf16 training forward has it (same sites, same counter-based masks, the snapshot beside the activations);
    // the hidden-state masks are applied by a pass of their own (qst_dropout_apply_f32): speed is not what this path is for
    const bool dropping = e->drop_state != nullptr;
    const void* dst8 = sv + p.dropst;
    const DropThr thr = dropping ? DropThr{e->drop_hidden, e->drop_attn} : DropThr{0u, 0u};
    QST_TRY(qst_forward_prologue(ids, nseq, L, c.arch, c.pad_token_id, pos_ids, dropping ? e->drop_state : nullptr,
                                 dropping ? (uint32_t*)(sv + p.dropst) : nullptr, st));
    {
        rec_put(saved, thr.hidden, thr.attn, ARENA_X3, nseq, L);
    }
    const bool hdrop = dropping && thr.hidden != 0;
    // out = (A . W^T + bias) * mask(site) + resid   (BertSelfOutput / BertOutput: LayerNorm(dropout(dense(x)) + input))
    auto proj = [&](const float* Ain, int K, int wseg, int bseg, const float* resid, float* out, uint32_t site) -> int {
        if (!hdrop) return nt3(Ain, K, P(wseg), K, out, H, P(bseg), resid, H, M, H, K, 1, st);
        QST_TRY(nt3(Ain, K, P(wseg), K, out, H, P(bseg), nullptr, 0, M, H, K, 0, st));
        const QstDrop dd = drop_of(thr, dst8, false, site);
        return qst_dropout_apply_f32(&dd, out, resid, (int64_t)M * H, out, st);
    };
    QST_TRY(qst_embed_sum_f32(ids, type_ids, pos_ids, P(lay.word), P(lay.pos), lay.type >= 0 ? P(lay.type) : nullptr, M, H,
                              F(p.s0), st));
    QST_TRY(qst_ln_fwd(F(p.s0), P(lay.eg), P(lay.eb), c.layer_norm_eps, M, H, F(p.x0), nullptr, nullptr, nullptr, st));
    if (hdrop) {
        const QstDrop de = drop_of(thr, dst8, false, QST_DROP_SITE_EMBED);
        QST_TRY(qst_dropout_apply_f32(&de, F(p.x0), nullptr, (int64_t)M * H, F(p.x0), st));
    }
    const float* rel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        QST_TRY(qst_rel_bias_fwd(P(lay.rel), e->rel_lut, A, L, F(p.rel), st));
        rel = F(p.rel);
    }
    const float* x = F(p.x0);
    for (int l = 0; l < c.num_layers; ++l) {
        const int b = lay.layer0[l];
        const X3Layer& a = p.layers[l];
        QST_TRY(nt3(x, H, P(b + W_QKV), H, F(a.qkv), 3 * H, P(b + B_QKV), nullptr, 0, M, 3 * H, H, 0, st));
        {
            const QstDrop dp = drop_of(thr, dst8, true, QST_DROP_SITE_PROBS(l));
            QST_TRY(qst_attention_fwd_x3_drop(F(a.qkv), mask, rel, nseq, L, A, d, F(a.ctx), dropping ? &dp : nullptr, st));
        }
        QST_TRY(proj(F(a.ctx), H, b + W_O, b + B_O, x, F(a.s1), QST_DROP_SITE_ATTN_OUT(l)));
        QST_TRY(qst_ln_fwd(F(a.s1), P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps, M, H, F(a.y1), nullptr, nullptr, nullptr, st));
        {                                                      // u = y1 W1^T + b1 and h = gelu(u) from one launch
            QstGemmArgs g{};
            g.A = F(a.y1); g.B = P(b + W_1); g.C = F(a.u); g.C2 = F(a.h); g.bias = P(b + B_1);
            g.M = M; g.N = I; g.K = H; g.lda = H; g.ldb = H; g.ldc = I;
            QST_TRY(qst_gemm_nt_x3(&g, 4, st));
        }
        QST_TRY(proj(F(a.h), I, b + W_2, b + B_2, F(a.y1), F(a.s2), QST_DROP_SITE_FFN_OUT(l)));
        QST_TRY(qst_ln_fwd(F(a.s2), P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, M, H, F(a.x), nullptr, nullptr, nullptr, st));
        x = F(a.x);
    }
    QST_TRY(qst_pool_norm_fwd(x, mask, nseq, L, H, c.normalize, out_emb, F(p.pooled), st));
    if (out_tok) QST_HIP_CHECK(hipMemcpyAsync(out_tok, x, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    return QST_OK;
}

// ... and its backward: fp32-class gradients ACCUMULATED into `grads`. Every contraction runs on split-bf16 x3 products: a
// dgrad is gemm_nt_x3 against the transposed weight, a wgrad (+ bias gradient) one gemm_tn_x3 launch over the token rows.
// Stages as the bf16 backward has them (head -> layers [layer_lo, layer_hi) top-down -> embeddings; the running d(loss)/d(x)
// lives in the workspace between calls), so that a data-parallel step can hand a finished layer's gradients to the all-reduce
// while the layers below are still running (round 5; rounds 3-4 ran it as one call and reduced afterwards).
static int backward_x3(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids, int nseq, int L,
                       const float* params, const float* grad_emb, float* grads, void* saved, size_t saved_bytes,
                       void* workspace, size_t workspace_bytes, bool do_head, int layer_hi, int layer_lo, bool do_embed,
                       hipStream_t st) {
    const qst_config& c = e->cfg;
    const X3TrainPlan p = plan_x3_train(c, nseq, L);
    const X3BwdPlan w = plan_x3_bwd(c, nseq, L);
    if (saved_bytes < p.total || workspace_bytes < w.total) return QST_ERR_WORKSPACE;
    char* sv = (char*)saved;
    char* ws = (char*)workspace;
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    auto G = [&](int seg) { return grads + lay.segs[seg].off; };
    auto F = [&](size_t o) { return (float*)(sv + o); };
    auto Wk = [&](size_t o) { return (float*)(ws + o); };
    float* dx = Wk(w.dx); float* dy = Wk(w.dy); float* ds = Wk(w.ds); float* dbig = Wk(w.dbig);
    float* dctx = Wk(w.dctx); float* dqkv = Wk(w.dqkv); float* wT = Wk(w.wT);
    // dX[M, in] = dY[M, out] . W[out, in] (+ resid)
    auto dgrad = [&](const float* dY, int out, int wseg, int in, float* dX, const float* resid) -> int {
        QST_TRY(qst_transpose_f32(P(wseg), out, in, in, wT, out, st));
        return nt3(dY, out, wT, out, dX, in, nullptr, resid, in, M, in, out, resid ? 1 : 0, st);
    };
    // dW[out, in] += dY^T . X ; db[out] += column sums of dY: one launch, straight from the row-major activations
    auto wgrad = [&](const float* dY, int out, const float* X, int in, int wseg, int bseg) -> int {
        QstGemmArgs g{};
        g.A = dY; g.B = X; g.C = G(wseg); g.colsum = G(bseg);
        g.M = M; g.N = out; g.K = in; g.lda = out; g.ldb = in; g.ldc = in;
        return qst_gemm_tn_x3(&g, st);
    };
    const float* rel = nullptr;
    float* drel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        rel = F(p.rel);
        drel = Wk(w.drel);
        if (do_head) QST_HIP_CHECK(hipMemsetAsync(drel, 0, (size_t)A * L * L * 4, st));
    }
    // dropout: the masks of the forward that filled `saved` (its thresholds from the handle's record, its (seed, step) from the
    // snapshot in the arena). ds = d(loss)/d(LayerNorm input) continues down the residual path as it is; the projection
    // that was dropped sees ds * mask (dsm, in a buffer that is free at that point).
    FwdRec fr;
    if (!rec_get(saved, ARENA_X3, nseq, L, &fr)) return QST_ERR_NO_FORWARD;      // not an arena a bf16x3 training forward of this shape has filled
    const DropThr thr = {fr.hidden, fr.attn};
    const void* dst8 = sv + p.dropst;
    const bool hdrop = thr.hidden != 0, adrop = thr.attn != 0;
    auto masked = [&](const float* g, uint32_t site, float* tmp, const float** out) -> int {
        *out = g;
        if (!hdrop) return QST_OK;
        const QstDrop dd = drop_of(thr, dst8, false, site);
        *out = tmp;
        return qst_dropout_apply_f32(&dd, g, nullptr, (int64_t)M * H, tmp, st);
    };
    if (do_head) QST_TRY(qst_pool_norm_bwd(grad_emb, F(p.pooled), mask, nseq, L, H, c.normalize, dx, st));
    for (int l = layer_hi - 1; l >= layer_lo; --l) {
        const int b = lay.layer0[l];
        const X3Layer& a = p.layers[l];
        const float* xin = l == 0 ? F(p.x0) : F(p.layers[l - 1].x);
        QST_TRY(qst_ln_bwd_f32(dx, F(a.s2), P(b + LN2_G), c.layer_norm_eps, M, H, ds, G(b + LN2_G), G(b + LN2_B), st));
        const float* dsm = ds;
        QST_TRY(masked(ds, QST_DROP_SITE_FFN_OUT(l), dctx, &dsm));                   // (dctx is free until the attention part)
        {                                                                            // du = (dsm . W2) * gelu'(u)
            QST_TRY(qst_transpose_f32(P(b + W_2), H, I, I, wT, H, st));
            QstGemmArgs g{};
            g.A = dsm; g.B = wT; g.C = dbig; g.aux = F(a.u);
            g.M = M; g.N = I; g.K = H; g.lda = H; g.ldb = H; g.ldc = I;
            QST_TRY(qst_gemm_nt_x3(&g, 5, st));
        }
        QST_TRY(wgrad(dsm, H, F(a.h), I, b + W_2, b + B_2));
        QST_TRY(dgrad(dbig, I, b + W_1, H, dy, ds));                                 // dy1 = du . W1 + ds2
        QST_TRY(wgrad(dbig, I, F(a.y1), H, b + W_1, b + B_1));
        QST_TRY(qst_ln_bwd_f32(dy, F(a.s1), P(b + LN1_G), c.layer_norm_eps, M, H, ds, G(b + LN1_G), G(b + LN1_B), st));
        QST_TRY(masked(ds, QST_DROP_SITE_ATTN_OUT(l), dy, &dsm));                    // (dy has been consumed)
        QST_TRY(dgrad(dsm, H, b + W_O, H, dctx, nullptr));
        QST_TRY(wgrad(dsm, H, F(a.ctx), H, b + W_O, b + B_O));
        {
            const QstDrop dp = drop_of(thr, dst8, true, QST_DROP_SITE_PROBS(l));
            QST_TRY(qst_attention_bwd_x3(F(a.qkv), F(a.ctx), dctx, mask, rel, nseq, L, A, d, dqkv, drel, ws + w.astats, adrop ? &dp : nullptr, st));
        }
        QST_TRY(dgrad(dqkv, 3 * H, b + W_QKV, H, dx, ds));                           // dx_in = dqkv . Wqkv + ds1
        QST_TRY(wgrad(dqkv, 3 * H, xin, H, b + W_QKV, b + B_QKV));
    }
    if (!do_embed) return QST_OK;
    if (hdrop) {                                       // the embedding dropout sits AFTER its LayerNorm
        const QstDrop de = drop_of(thr, dst8, false, QST_DROP_SITE_EMBED);
        QST_TRY(qst_dropout_apply_f32(&de, dx, nullptr, (int64_t)M * H, dx, st));
    }
    QST_TRY(qst_ln_bwd_f32(dx, F(p.s0), P(lay.eg), c.layer_norm_eps, M, H, ds, G(lay.eg), G(lay.eb), st));
    QST_TRY(qst_embed_bwd(ds, ids, type_ids, (const int32_t*)(sv + p.pos_ids), nseq, L, H, c.type_vocab_size,
                          G(lay.word), G(lay.pos), lay.type >= 0 ? G(lay.type) : nullptr, st));
    if (c.arch == QST_ARCH_MPNET) QST_TRY(qst_rel_bias_bwd(drel, e->rel_lut, c.rel_buckets, A, L, G(lay.rel), st));
    return QST_OK;
}

extern "C" int qst_encoder_forward(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids,
                                   int nseq, int L, const float* params, const void* shadow, float* out_emb,
                                   float* out_tok, void* saved, size_t saved_bytes, int training, void* stream) {
    if (!e || !ids || !mask || !params || !out_emb || !saved) return QST_ERR_BAD_ARG;
    QST_TRY(shape_ok(e, nseq, L));
    const qst_config& c = e->cfg;
    if (c.precision == QST_PREC_BF16X3)
        return training ? forward_x3_train(e, ids, mask, type_ids, nseq, L, params, out_emb, out_tok, saved, saved_bytes, (hipStream_t)stream)
                        : forward_x3(e, ids, mask, type_ids, nseq, L, params, out_emb, out_tok, saved, saved_bytes, (hipStream_t)stream);
    if (!shadow) return QST_ERR_BAD_ARG;
    if (c.precision == QST_PREC_FP8)
        return training ? forward_mx_train(e, ids, mask, type_ids, nseq, L, params, shadow, out_emb, out_tok, saved, saved_bytes, (hipStream_t)stream)
                        : forward_mx(e, ids, mask, type_ids, nseq, L, params, shadow, out_emb, out_tok, saved, saved_bytes, (hipStream_t)stream);
    const ActPlan p = plan_acts(c, nseq, L, training != 0);
    if (saved_bytes < p.total) return QST_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* sv = (char*)saved;
    const uint16_t* sh = (const uint16_t*)shadow;          // bf16, or IEEE half on a QST_PREC_F16 handle
    const OpKernels& K = op_kernels(c);
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    auto W = [&](int seg) { return sh + lay.segs[seg].shadow_off; };
    // QST_PREC_F16W: every forward Linear multiplies by hi + lo of the weight (a second pass over K: QstGemmArgs.B2)
    const bool splitw = c.precision == QST_PREC_F16W;
    auto WL = [&](int seg) -> const void* { return splitw ? sh + lay.shadow_total + lay.segs[seg].shadow_off : nullptr; };
    auto linear = [&](const void* Ain, int Kd, int wseg, void* Cout, int N, void* C2, int bseg, const float* resid, int epi,
                      int b2_n0 = 0) {
        return nt(K, Ain, Kd, W(wseg), Kd, Cout, N, C2, nullptr, P(bseg), resid, N, M, N, Kd, epi, true, st, WL(wseg), b2_n0);
    };

    int32_t* pos_ids = (int32_t*)(sv + p.pos_ids);
    // dropout: training forwards of a handle that has it on; the step counter moves first (in the prologue launch, which
    // also leaves its snapshot beside the activations), backward reuses its value
    const bool dropping = training && e->drop_state != nullptr;
    const void* dst8 = sv + p.dropst;
    const DropThr thr = dropping ? DropThr{e->drop_hidden, e->drop_attn} : DropThr{0u, 0u};
    QST_TRY(qst_forward_prologue(ids, nseq, L, c.arch, c.pad_token_id, pos_ids, dropping ? e->drop_state : nullptr,
                                 dropping ? (uint32_t*)(sv + p.dropst) : nullptr, st));
    if (training) {                                   // remember what this forward did, for the backward over the same arena
        rec_put(saved, thr.hidden, thr.attn, K.arena_kind, nseq, L);
    }
    {
        const QstDrop de = drop_of(thr, dst8, false, QST_DROP_SITE_EMBED);
        QST_TRY(K.embed_ln_fwd_drop(ids, type_ids, pos_ids, P(lay.word), P(lay.pos), lay.type >= 0 ? P(lay.type) : nullptr,
                                      P(lay.eg), P(lay.eb), c.layer_norm_eps, M, H, (float*)(sv + p.x0), sv + p.x0b,
                                      sv + p.xh0, (float*)(sv + p.rs0), dropping ? &de : nullptr, st));
    }
    const float* rel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        QST_TRY(qst_rel_pos_fwd(P(lay.rel), e->rel_lut, A, L, (float*)(sv + p.rel), st));
        rel = (const float*)(sv + p.rel);
    }
    const float* x = (const float*)(sv + p.x0);
    const void* xb = sv + p.x0b;
    float* s = (float*)(sv + p.s_scratch);
    // H = 384: the LayerNorm after each projection runs inside that GEMM's epilogue (full-row tiles) -- from M = 16384
    // token rows on: one 128-row tile per workgroup gives a small batch too few workgroups (measured: the unfused pair
    // is 5-25% faster up to M = 8192, equal at 16384, 25% slower at 32768)
    const bool fuse_ln = fuse_ln_rows(H, M, e->ln_fusion) && !(splitw && qst_gemm_nt_ln_block_rows(H) != 128);   // (no split weights on the 8-phase loop)
    // ... and the whole feed-forward block (FFN-1, GELU, FFN-2, LayerNorm) is ONE kernel: h never returns from HBM, and
    // an inference forward does not write it at all
    const bool fuse_ffn = fuse_ln && !dropping && !splitw && (e->ffn_chain & (training ? 2 : 1)) && qst_ffn_chain_supported(H, I) != 0;
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerAct& a = p.layers[l];
        const int b = lay.layer0[l];
        // (split weights: the value third of the fused QKV product only -- the rounding of the query / key weights perturbs
        //  logits that the softmax and the pooling average out: all-split and v-only measure the same, DESIGN.md finding 34)
        QST_TRY(linear(xb, H, b + W_QKV, sv + a.qkv, 3 * H, nullptr, b + B_QKV, nullptr, QST_EPI_BF16, 2 * H));
        {
            QstAttnDesc q{};
            q.qkv = sv + a.qkv; q.mask = mask; q.rel_pos = rel; q.nseq = nseq; q.L = L; q.A = A; q.d = d;
            q.ctx = sv + a.ctx; q.lse = (float*)(sv + a.lse);
            if (dropping) q.drop = drop_of(thr, dst8, true, QST_DROP_SITE_PROBS(l));
            QST_TRY(K.attention_fwd_ex(&q, st));
        }
        drop_next(thr, dst8, dropping, QST_DROP_SITE_ATTN_OUT(l), 1);
        if (fuse_ln) {
            QST_TRY(nt_ln(K, sv + a.ctx, H, W(b + W_O), H, (float*)(sv + a.y1), sv + a.y1b, P(b + B_O), x, M, H, H, 0,
                          P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps, sv + a.xh1, (float*)(sv + a.rs1), nullptr, st, WL(b + W_O)));
        } else {
            QST_TRY(linear(sv + a.ctx, H, b + W_O, s, H, nullptr, b + B_O, x, QST_EPI_F32_RESID));
            QST_TRY(K.ln_fwd(s, P(b + LN1_G), P(b + LN1_B), c.layer_norm_eps, M, H, (float*)(sv + a.y1), sv + a.y1b,
                               sv + a.xh1, (float*)(sv + a.rs1), st));
        }
        if (fuse_ffn) {
            QST_TRY(ffn_chain(K, sv + a.y1b, W(b + W_1), W(b + W_2), P(b + B_1), P(b + B_2), (const float*)(sv + a.y1), nullptr,
                              training ? sv + a.u : nullptr, training ? sv + a.hact : nullptr, (float*)(sv + a.x), sv + a.xb,
                              M, H, I, 0, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, sv + a.xh2, (float*)(sv + a.rs2),
                              nullptr, st));
            x = (const float*)(sv + a.x);
            xb = sv + a.xb;
            continue;
        }
        QST_TRY(linear(sv + a.y1b, H, b + W_1, sv + a.u, I, sv + a.hact, b + B_1, nullptr, QST_EPI_GELU));
        drop_next(thr, dst8, dropping, QST_DROP_SITE_FFN_OUT(l), 1);
        if (fuse_ln) {
            QST_TRY(nt_ln(K, sv + a.hact, I, W(b + W_2), I, (float*)(sv + a.x), sv + a.xb, P(b + B_2),
                          (const float*)(sv + a.y1), M, H, I, 0, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, sv + a.xh2,
                          (float*)(sv + a.rs2), nullptr, st, WL(b + W_2)));
        } else {
            QST_TRY(linear(sv + a.hact, I, b + W_2, s, H, nullptr, b + B_2, (const float*)(sv + a.y1), QST_EPI_F32_RESID));
            QST_TRY(K.ln_fwd(s, P(b + LN2_G), P(b + LN2_B), c.layer_norm_eps, M, H, (float*)(sv + a.x), sv + a.xb,
                               sv + a.xh2, (float*)(sv + a.rs2), st));
        }
        x = (const float*)(sv + a.x);
        xb = sv + a.xb;
    }
    QST_TRY(qst_pool_norm_fwd(x, mask, nseq, L, H, c.normalize, out_emb, (float*)(sv + p.pooled), st));
    if (out_tok) QST_HIP_CHECK(hipMemcpyAsync(out_tok, x, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    return QST_OK;
}

// Stages of one backward pass, top to bottom: head (pool/normalise), layers N-1..0, embeddings. A caller may run
// them in several calls (layer_hi > layer_lo) to launch the gradient all-reduce of finished layers in between;
// the running d(loss)/d(x) lives in the workspace between calls.
//
// qst_encoder_backward_stage adds two flags for the LAST layer range of a data-parallel step: with
// QST_BWD_SKIP_WGRAD the range's weight-gradient launch is left out (its operands stay in the workspace), so the
// caller can finish the embedding stage first, start the all-reduce of the embedding gradients -- half of a MiniLM
// arena, and otherwise the one bucket with nothing left to hide behind -- and then run the postponed launch with
// QST_BWD_WGRAD_ONLY underneath it.
extern "C" int qst_encoder_backward_stage(qst_encoder* e, const int64_t* ids, const int64_t* mask,
                                          const int64_t* type_ids, int nseq, int L, const float* params,
                                          const void* shadow, const float* grad_emb, float* grads, void* saved,
                                          size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                          int flags, int layer_hi, int layer_lo, void* stream) {
    const int do_head = (flags & QST_BWD_HEAD) != 0, do_embed = (flags & QST_BWD_EMBED) != 0;
    const bool skip_wgrad = (flags & QST_BWD_SKIP_WGRAD) != 0, wgrad_only = (flags & QST_BWD_WGRAD_ONLY) != 0;
    if (wgrad_only && (skip_wgrad || do_head || do_embed)) return QST_ERR_BAD_ARG;
    // postponing a layer's weight gradients is sound for layer 0 only: the stage of layer l - 1 overwrites the gradients
    // (ds / dsb / du / dqkv in the workspace) that layer l's launch reads
    if ((skip_wgrad || wgrad_only) && (layer_lo != 0 || layer_hi != 1)) return QST_ERR_BAD_ARG;
    if (!e || !ids || !mask || !params || !grads || !saved || !workspace) return QST_ERR_BAD_ARG;
    if (e->cfg.precision == QST_PREC_BF16X3) {
        // the parity path (no shadow): staged like the bf16 one, without the postponed weight-gradient launch of layer 0
        if (skip_wgrad || wgrad_only) return QST_ERR_UNSUPPORTED;
        if (do_head && !grad_emb) return QST_ERR_BAD_ARG;
        if (layer_lo < 0 || layer_hi > e->cfg.num_layers || layer_lo > layer_hi) return QST_ERR_BAD_ARG;
        if (int rc = shape_ok(e, nseq, L)) return rc;
        return backward_x3(e, ids, mask, type_ids, nseq, L, params, grad_emb, grads, saved, saved_bytes, workspace, workspace_bytes,
                           do_head != 0, layer_hi, layer_lo, do_embed != 0, (hipStream_t)stream);
    }
    if (!shadow) return QST_ERR_BAD_ARG;
    // (a QST_PREC_FP8 handle: the bf16 backward over the arena its training forward filled; `shadow` = the bf16 shadows)
    if (do_head && !grad_emb) return QST_ERR_BAD_ARG;
    if (layer_lo < 0 || layer_hi > e->cfg.num_layers || layer_lo > layer_hi) return QST_ERR_BAD_ARG;
    QST_TRY(shape_ok(e, nseq, L));
    const qst_config& c = e->cfg;
    const ActPlan p = plan_acts(c, nseq, L, true);
    const BwdPlan w = plan_bwd(c, nseq, L);
    if (saved_bytes < p.total || workspace_bytes < w.total) return QST_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* sv = (char*)saved;
    char* ws = (char*)workspace;
    const uint16_t* sh = (const uint16_t*)shadow;
    const OpKernels& K = op_kernels(c);
    const int M = nseq * L, H = c.hidden_size, I = c.intermediate_size, A = c.num_heads, d = H / A;
    const Layout& lay = e->lay;
    auto P = [&](int seg) { return params + lay.segs[seg].off; };
    auto G = [&](int seg) { return grads + lay.segs[seg].off; };
    auto WT = [&](int seg) { return sh + lay.segs[seg].shadow_off + qst_align_up(lay.segs[seg].numel, kAlign); };

    float* dxa = (float*)(ws + w.dxa);
    float* dxb = (float*)(ws + w.dxb);
    float* ds = (float*)(ws + w.ds);
    void* dsb = ws + w.dsb;
    void* dsb1 = ws + w.dsb1;
    void* du = ws + w.du;
    void* dctx = ws + w.dctx;
    void* dqkv = ws + w.dqkv;
    // LayerNorm gamma/beta gradients: every ln_bwd of this call writes per-block partials into its own slot; one
    // batched launch at the end reduces them all (13 small launches per step -> 1-7)
    QstLnReduceBatch lnb{};
    lnb.H = H;
    lnb.nblocks = (int)(qst_ln_bwd_scratch_bytes(M, H) / ((size_t)2 * H * sizeof(float)));
    // H = 384: every LayerNorm backward except the top one (whose input comes from the pooling head, not from a GEMM)
    // runs inside the epilogue of the dgrad GEMM that produces its input; those write one partial row per 128-row tile
    const bool fuse_ln = fuse_ln_rows(H, M, e->ln_fusion);
    // dropout: the masks of the forward that filled `saved` are recomputed from its (seed, step) snapshot in the arena and
    // ITS thresholds (recorded by that forward, process-wide: any handle of the same model may run the backward)
    FwdRec fr;
    if (!rec_get(saved, K.arena_kind, nseq, L, &fr)) return QST_ERR_NO_FORWARD;  // not an arena a training forward of this operand type and shape has filled
    const DropThr thr = {fr.hidden, fr.attn};
    const bool dropping = thr.hidden != 0 || thr.attn != 0;
    const void* dst8 = sv + p.dropst;
    const bool fuse_ffn = fuse_ln && !dropping && (e->ffn_chain & 4) && qst_ffn_chain_supported(H, I) != 0;
    const int fused_rows = (M + qst_gemm_nt_ln_block_rows_m(H, M) - 1) / qst_gemm_nt_ln_block_rows_m(H, M);
    auto hdrop = [&](uint32_t site, QstDrop& d) -> const QstDrop* {          // hidden-state mask of `site`, or none
        if (!dropping || !thr.hidden) return nullptr;
        d = drop_of(thr, dst8, false, site);
        return &d;
    };
    auto ln_slot = [&](int slot, float* dg, float* db, int nrows = 0) {
        float* sp = (float*)(ws + w.lnred + (size_t)slot * w.lnred_stride);
        lnb.partials[lnb.count] = sp; lnb.dgamma[lnb.count] = dg; lnb.dbeta[lnb.count] = db;
        lnb.nblocks_each[lnb.count] = nrows;
        ++lnb.count;
        return sp;
    };
    float* drel = nullptr;
    const float* rel = nullptr;
    if (c.arch == QST_ARCH_MPNET) {
        drel = (float*)(ws + w.drel);
        rel = (const float*)(sv + p.rel);
        if (do_head) QST_HIP_CHECK(hipMemsetAsync(drel, 0, (size_t)A * 2 * L * 4, st));
    }
    if (do_head)
        QST_TRY(qst_pool_norm_bwd(grad_emb, (const float*)(sv + p.pooled), mask, nseq, L, H, c.normalize, dxa, st));
    // all four weight gradients (+ bias gradients) of a layer in one grouped launch
    auto wgrad = [&](int l) -> int {
        const LayerAct& a = p.layers[l];
        const int b = lay.layer0[l];
        const void* xin_b = (l == 0) ? (const void*)(sv + p.x0b) : (const void*)(sv + p.layers[l - 1].xb);
        QstTnGroup grp{};
        grp.nprob = 4;
        grp.splits = 0;
        auto set = [&](int i, const void* dY, int N, const void* X, int K, int wseg, int bseg) {
            QstGemmArgs& q = grp.prob[i];
            q.A = dY; q.B = X; q.C = G(wseg); q.colsum = G(bseg); q.M = M; q.N = N; q.K = K;
            q.lda = N; q.ldb = K; q.ldc = K;
        };
        set(0, dsb, H, sv + a.hact, I, b + W_2, b + B_2);          // dW2 [H, I]
        set(1, du, I, sv + a.y1b, H, b + W_1, b + B_1);            // dW1 [I, H]
        set(2, dsb1, H, sv + a.ctx, H, b + W_O, b + B_O);          // dWo [H, H]
        set(3, dqkv, 3 * H, xin_b, H, b + W_QKV, b + B_QKV);       // dWqkv [3H, H]
        return K.gemm_tn_group(&grp, st);
    };
    if (wgrad_only) {
        // the dY tensors of exactly one layer live in the workspace: the one whose stage ran with QST_BWD_SKIP_WGRAD
        if (layer_hi - layer_lo != 1) return QST_ERR_BAD_ARG;
        return wgrad(layer_lo);
    }
    if (skip_wgrad && layer_hi - layer_lo != 1) return QST_ERR_BAD_ARG;
    for (int l = layer_hi - 1; l >= layer_lo; --l) {
        const LayerAct& a = p.layers[l];
        const int b = lay.layer0[l];
        // LN2 -> ds2 (fp32 for the residual path, bf16 for the GEMMs). Fused mode: only the top layer runs it as a
        // row kernel; below, (ds, dsb) were written by the QKV dgrad of layer l+1.
        if (!fuse_ln || l == c.num_layers - 1) {
            QstDrop dd;
            QST_TRY(K.ln_bwd_drop(dxa, sv + a.xh2, (const float*)(sv + a.rs2), P(b + LN2_G), M, H, ds, dsb, nullptr, nullptr,
                                    ln_slot(2 * l + 1, G(b + LN2_G), G(b + LN2_B)), nullptr, hdrop(QST_DROP_SITE_FFN_OUT(l), dd), st));
        }
        // FFN2 dgrad through GELU: du = (ds2 . W2) * gelu'(u)   (a.u holds gelu'(u), written by the forward epilogue)
        if (!fuse_ffn)
            QST_TRY(nt(K, dsb, H, WT(b + W_2), H, du, I, nullptr, sv + a.u, nullptr, nullptr, 0, M, I, H, QST_EPI_GELU_BWD, false, st));
        // FFN1 dgrad + residual: dy1 = du . W1 + ds2 ; LN1 backward -> ds1 (fp32 in `ds1`, bf16 in dsb1)
        const float* ds1 = ds;
        if (fuse_ffn) {
            // both dgrads of the feed-forward block and the LayerNorm-1 backward in one kernel; du is written once
            // (the weight gradients need it) and never read back by this chain
            QST_TRY(ffn_chain(K, dsb, WT(b + W_2), WT(b + W_1), nullptr, nullptr, ds, sv + a.u, nullptr, du, dxb, dsb1, M, H, I, 1,
                              P(b + LN1_G), nullptr, 0.f, sv + a.xh1, (float*)(sv + a.rs1),
                              ln_slot(2 * l, G(b + LN1_G), G(b + LN1_B), fused_rows), st));
            ds1 = dxb;
        } else if (fuse_ln) {
            drop_next(thr, dst8, dropping, QST_DROP_SITE_ATTN_OUT(l), 2);
            QST_TRY(nt_ln(K, du, I, WT(b + W_1), I, dxb, dsb1, nullptr, ds, M, H, I, 1, P(b + LN1_G), nullptr, 0.f, sv + a.xh1,
                          (float*)(sv + a.rs1), ln_slot(2 * l, G(b + LN1_G), G(b + LN1_B), fused_rows), st));
            ds1 = dxb;
        } else {
            QstDrop dd;
            QST_TRY(nt(K, du, I, WT(b + W_1), I, dxb, H, nullptr, nullptr, nullptr, ds, H, M, H, I, QST_EPI_F32_RESID, false, st));
            QST_TRY(K.ln_bwd_drop(dxb, sv + a.xh1, (const float*)(sv + a.rs1), P(b + LN1_G), M, H, ds, dsb1, nullptr, nullptr,
                                    ln_slot(2 * l, G(b + LN1_G), G(b + LN1_B)), nullptr, hdrop(QST_DROP_SITE_ATTN_OUT(l), dd), st));
        }
        // attention output projection dgrad, attention core
        QST_TRY(nt(K, dsb1, H, WT(b + W_O), H, dctx, H, nullptr, nullptr, nullptr, nullptr, 0, M, H, H, QST_EPI_BF16, false, st));
        {
            QstAttnDesc q{};
            q.qkv = sv + a.qkv; q.mask = mask; q.rel_pos = rel; q.nseq = nseq; q.L = L; q.A = A; q.d = d;
            q.ctx = sv + a.ctx; q.lse = (float*)(sv + a.lse); q.dctx = dctx; q.dqkv = dqkv; q.drel = drel;
            q.delta_scratch = (float*)(ws + w.delta);
            if (dropping) q.drop = drop_of(thr, dst8, true, QST_DROP_SITE_PROBS(l));
            QST_TRY(K.attention_bwd_ex(&q, st));
        }
        if (!skip_wgrad) QST_TRY(wgrad(l));
        // QKV projection dgrad + residual: dx_in = dqkv . Wqkv + ds1. Fused mode: followed in the same kernel by the
        // backward of the LayerNorm that produced this layer's input (LN2 of layer l-1, or the embedding LayerNorm)
        if (fuse_ln && l > 0) drop_next(thr, dst8, dropping, QST_DROP_SITE_FFN_OUT(l - 1), 2);   // dsb = d(FFN-2 output of layer l-1)
        else if (fuse_ln) drop_next(thr, dst8, dropping, QST_DROP_SITE_EMBED, 3);                // embedding dropout follows its LN
        if (fuse_ln && l > 0) {
            const LayerAct& lo = p.layers[l - 1];
            const int bl = lay.layer0[l - 1];
            QST_TRY(nt_ln(K, dqkv, 3 * H, WT(b + W_QKV), 3 * H, ds, dsb, nullptr, ds1, M, H, 3 * H, 1, P(bl + LN2_G), nullptr, 0.f,
                          sv + lo.xh2, (float*)(sv + lo.rs2),
                          ln_slot(2 * (l - 1) + 1, G(bl + LN2_G), G(bl + LN2_B), fused_rows), st));
        } else if (fuse_ln) {
            QST_TRY(nt_ln(K, dqkv, 3 * H, WT(b + W_QKV), 3 * H, ds, nullptr, nullptr, ds1, M, H, 3 * H, 1, P(lay.eg), nullptr, 0.f,
                          sv + p.xh0, (float*)(sv + p.rs0), ln_slot(2 * c.num_layers, G(lay.eg), G(lay.eb), fused_rows), st));
        } else {
            QST_TRY(nt(K, dqkv, 3 * H, WT(b + W_QKV), 3 * H, dxa, H, nullptr, nullptr, nullptr, ds1, H, M, H, 3 * H,
                       QST_EPI_F32_RESID, false, st));
        }
    }
    if (do_embed && !fuse_ln) {
        QstDrop dd;
        QST_TRY(K.ln_bwd_drop(dxa, sv + p.xh0, (const float*)(sv + p.rs0), P(lay.eg), M, H, ds, nullptr, nullptr, nullptr,
                                ln_slot(2 * c.num_layers, G(lay.eg), G(lay.eb)), hdrop(QST_DROP_SITE_EMBED, dd), nullptr, st));
    }
    if (lnb.count > 0) {
        if (lnb.count > QST_LN_BATCH_MAX) return QST_ERR_UNSUPPORTED;
        QST_TRY(qst_ln_bwd_reduce_batch(&lnb, st));
    }
    if (!do_embed) return QST_OK;
    // embeddings
    QST_TRY(qst_embed_bwd(ds, ids, type_ids, (const int32_t*)(sv + p.pos_ids), nseq, L, H, c.type_vocab_size,
                          G(lay.word), G(lay.pos), lay.type >= 0 ? G(lay.type) : nullptr, st));
    if (c.arch == QST_ARCH_MPNET) {
        QST_TRY(qst_rel_pos_bwd(drel, e->rel_lut, c.rel_buckets, A, L, G(lay.rel), st));
    }
    return QST_OK;
}

extern "C" int qst_encoder_backward_partial(qst_encoder* e, const int64_t* ids, const int64_t* mask,
                                            const int64_t* type_ids, int nseq, int L, const float* params,
                                            const void* shadow, const float* grad_emb, float* grads, void* saved,
                                            size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                            int do_head, int layer_hi, int layer_lo, int do_embed, void* stream) {
    return qst_encoder_backward_stage(e, ids, mask, type_ids, nseq, L, params, shadow, grad_emb, grads, saved, saved_bytes,
                                      workspace, workspace_bytes, (do_head ? QST_BWD_HEAD : 0) | (do_embed ? QST_BWD_EMBED : 0),
                                      layer_hi, layer_lo, stream);
}

extern "C" int qst_encoder_backward(qst_encoder* e, const int64_t* ids, const int64_t* mask, const int64_t* type_ids,
                                    int nseq, int L, const float* params, const void* shadow, const float* grad_emb,
                                    float* grads, void* saved, size_t saved_bytes, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!e) return QST_ERR_BAD_ARG;
    return qst_encoder_backward_partial(e, ids, mask, type_ids, nseq, L, params, shadow, grad_emb, grads, saved,
                                        saved_bytes, workspace, workspace_bytes, 1, e->cfg.num_layers, 0, 1, stream);
}

extern "C" int qst_adamw_launch_sched(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                      const uint8_t* chunk_decay, int64_t n, float base_lr, float beta1, float beta2,
                                      float eps, float weight_decay, float max_grad_norm, float grad_scale,
                                      int64_t warmup_steps, int64_t total_steps, int64_t* step_dev, float* norm_out,
                                      float* scratch, hipStream_t st);

extern "C" int qst_clip_adamw_step_sched(const qst_encoder* e, float* params, float* grads, float* exp_avg,
                                         float* exp_avg_sq, float base_lr, float beta1, float beta2, float eps,
                                         float weight_decay, float max_grad_norm, float grad_scale,
                                         int64_t warmup_steps, int64_t total_steps, int64_t* step_dev,
                                         float* norm_out, float* scratch, void* stream) {
    if (!e) return QST_ERR_BAD_ARG;
    return qst_adamw_launch_sched(params, grads, exp_avg, exp_avg_sq, e->chunk_decay, e->lay.total, base_lr, beta1, beta2,
                                  eps, weight_decay, max_grad_norm, grad_scale, warmup_steps, total_steps, step_dev,
                                  norm_out, scratch, (hipStream_t)stream);
}

extern "C" int qst_adamw_launch_amp(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                    const uint8_t* chunk_decay, int64_t n, float base_lr, float beta1, float beta2,
                                    float eps, float weight_decay, float max_grad_norm, float grad_scale,
                                    int64_t warmup_steps, int64_t total_steps, int64_t* step_dev, float* scaler_dev,
                                    float growth, float backoff, int growth_interval, float* norm_out, float* scratch,
                                    hipStream_t st);

extern "C" int qst_clip_adamw_step_amp(const qst_encoder* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                       float base_lr, float beta1, float beta2, float eps, float weight_decay,
                                       float max_grad_norm, float grad_scale, int64_t warmup_steps, int64_t total_steps,
                                       int64_t* step_dev, float* scaler_dev, float growth_factor, float backoff_factor,
                                       int32_t growth_interval, float* norm_out, float* scratch, void* stream) {
    if (!e) return QST_ERR_BAD_ARG;
    return qst_adamw_launch_amp(params, grads, exp_avg, exp_avg_sq, e->chunk_decay, e->lay.total, base_lr, beta1, beta2, eps,
                                weight_decay, max_grad_norm, grad_scale, warmup_steps, total_steps, step_dev, scaler_dev,
                                growth_factor, backoff_factor, growth_interval, norm_out, scratch, (hipStream_t)stream);
}

extern "C" int qst_clip_adamw_step(const qst_encoder* e, float* params, float* grads, float* exp_avg,
                                   float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float max_grad_norm, float grad_scale, int64_t step,
                                   float* norm_out, float* scratch, void* stream) {
    if (!e) return QST_ERR_BAD_ARG;
    return qst_adamw_launch(params, grads, exp_avg, exp_avg_sq, e->chunk_decay, e->lay.total, lr, beta1, beta2, eps,
                            weight_decay, max_grad_norm, grad_scale, step, norm_out, scratch, (hipStream_t)stream);
}

extern "C" int64_t qst_abi_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(QstGemmArgs);
        case 1: return sizeof(QstLnEpi);
        case 2: return sizeof(QstFfnArgs);
        case 3: return sizeof(QstTnGroup);
        case 4: return sizeof(QstLnReduceBatch);
        case 5: return sizeof(QstDrop);
        case 6: return sizeof(QstAttnDesc);
        default: return QST_ERR_BAD_ARG;
    }
}
