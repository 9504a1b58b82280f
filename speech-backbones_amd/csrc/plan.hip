// plan.hip -- host side of libgradtts_gfx950: the plan (state_dict layout, packed-weight layout, op program,
// workspace allocation) and the extern "C" entry points declared in include/gradtts_abi.h.
//
// The plan is pure host metadata.  Every device byte belongs to the caller (packed blob, workspace, tensors);
// every call just enqueues kernels on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

using namespace gtts;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
namespace gtts {
int set_error(int code, const char *msg) { g_err = msg; return code; }     // for the other translation units
}
#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                              \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
struct gtts_plan;
static int plan_nsplit(const gtts_plan *p);      // 2: hi/lo operand planes (BF16X3, and everything F16F8 leaves on it), 1: plain bf16
#ifndef GTTS_FUSE_TAIL_CTX
#define GTTS_FUSE_TAIL_CTX 1      // 0: tail_identity + attention context as two launches everywhere; 1: fused for C = 64; 2: for every C % 32 == 0
                                  // (measured, same box: 2 loses -- the wider attentions stage their x tile in two workgroups, the tail's Mish
                                  // runs twice in a VALU-bound kernel: tail + context 100 + 104 -> 231 us at level 1, 51 + 59 -> 135 at level 2)
#endif

// ------------------------------------------------------------------------------------------------ plan data
enum TensorKind { TK_ACT, TK_PERB, TK_PART, TK_APART, TK_BYTES_PERB, TK_ROWS };
struct Tensor {
    std::string name;
    int kind;
    int C;          // TK_ACT: channels; TK_PERB: floats per sample; TK_BYTES_PERB: bytes per sample (in `bytes`)
    int lvl;        // resolution level of a TK_ACT / TK_PART / TK_APART tensor
    size_t bytes;   // TK_BYTES_PERB
    int mode, cout; // TK_PART: conv geometry that produces it
    bool ws;        // TK_PART: written by the wave-specialised kernel (conv_ws.hip): one slot per pixel tile of ITS tiling
    bool external;  // not in the workspace (inputs / outputs of the call)
    bool tref;      // frame axis is the reference mel's (T_ref) instead of T  (DiffVC RefBlock tensors)
    int first, last;
};

enum OpKind { OP_CONV, OP_GNFIN, OP_TAILID, OP_ACTX, OP_AMERGE, OP_AFOLD, OP_INSTATS, OP_REFPOOL, OP_VCCOND, OP_PREPVC };
struct Op {
    int kind;
    // conv
    int mode, pro, epi;
    int src0, src1, c0, c1, cout, lvl_in, lvl_out;
    int sc, sh, tb_off;            // PRO_GN inputs (tensor ids) and column inside the tb row
    size_t w_off, b_off;           // blob offsets (shared weights) ...
    int w_t, bias_t;               // ... or per-sample tensors (attention pass 2), -1 if unused
    int out, part, eh, esc, esh, eres;
    // gn finalize
    int C;
    size_t gamma_off, beta_off;
    // attention
    size_t wkv_off, wq_off, wout_off, bout_off, g_off;
    int apart, ctxn;
    int gn_op;                     // EPI_STATS conv: index of the OP_GNFIN it can absorb (-1: none)
    int use_ref;                   // conv / stats run on the reference mel geometry (ref_mask, T_ref)
    int has_tb;                    // PRO_IGLU: a time bias column is added (tb_off valid)
    int fused;                     // OP_TAILID: done by the following attention context pass (no launch); OP_ACTX: carries that tail
                                   // (eh / esc / esh = the tail's GroupNorm inputs, eres = the block's input, src0 = the tail's output)
    std::string label;
};

struct ParamDesc {
    std::string name;
    int rank;
    int dims[4];
    size_t off;      // blob byte offset of the packed / copied form
    int pack;        // 0 copy fp32, 1 conv C3, 2 conv DN, 3 conv UP, 4 conv P1, 5 to_qkv (kv packed + q copy), 6 conv C3 in the f16 + fp8 format,
                     // 7 conv UP in the f16 + fp8 format
    size_t off2;     // to_qkv: fp32 copy of the q rows
    int cin, cout;
};

struct gtts_plan {
    gtts_unet_cfg cfg;
    int cin0;
    int nlev;
    int dims[4];
    std::vector<ParamDesc> params;
    std::map<std::string, int> pidx;
    size_t blob_bytes;
    size_t freq_off;
    size_t status_off;         // 256 bytes: range record of gtts_pack_weights {count, max |w| bits, 1 + parameter index} (GTTS_PREC_F16F8)
    TimeMlpDesc tmlp;
    size_t spk_w0, spk_b0, spk_w2, spk_b2;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    int t_x0, t_s, t_tb, t_final_raw, t_final_sc, t_final_sh;
    int t_xtref = -1, t_cond = -1;   // DiffVC: diffused reference mel [B,1,F,T_ref], condition vector [B,dim_cond]
    int t_ticket = -1;               // per-sample tickets of the fused GroupNorm finalize (zeroed once per call)
    int cache_Tr = -1;
    size_t fw_off, fb_off;     // final_conv weight / bias (fp32)
    // extra (non-program) launches, profiled as ops n_ops .. n_ops+3: prep_input, time_mlp, final_euler, mul_mask
    // profiling
    int prof_on = 0;           // 0 off, 1 per-op durations (unsplit), 2 timeline (sub-batch streams stay on; gtts_profile_timeline)
    struct ProfRec { int op; hipEvent_t a, b; int stream; };
    std::vector<ProfRec> prof;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    // Caller-owned side streams for the sampler's sub-batch split (gtts_plan_set_streams); the fork / join events
    // are host objects of the plan, created when the streams are registered.
    int nsub = 0;
    hipStream_t sub[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // hipGraph replay of whole sampler calls (gtts_plan_set_graph): one instantiated graph per distinct argument tuple
    bool graph_on = false;
    struct GraphRec { std::vector<unsigned long long> key; hipGraph_t graph; hipGraphExec_t exec; unsigned long long used; };
    std::vector<GraphRec> graphs;
    unsigned long long graph_clock = 0;
    // Execution state below (layout cache, profiling record) is mutated by the enqueueing calls: they take this
    // mutex for the duration of the (host-side, asynchronous) enqueue, so one plan may be shared by several host
    // threads / streams as long as each call brings its own workspace.
    std::mutex mu;
    // workspace layout cache of the enqueueing calls (guarded by mu)
    int cache_B = -1, cache_T = -1;
    std::vector<size_t> offsets;
    size_t ws_bytes = 0;
};
struct Layout { std::vector<size_t> offsets; size_t ws_bytes = 0; };

// ------------------------------------------------------------------------------------------------ builders
static int add_param(gtts_plan *p, const std::string &name, std::vector<int> dims, int pack, int cin = 0, int cout = 0) {
    ParamDesc d;
    d.name = name;
    d.rank = (int)dims.size();
    for (int i = 0; i < 4; ++i) d.dims[i] = i < (int)dims.size() ? dims[i] : 1;
    d.pack = pack;
    d.cin = cin;
    d.cout = cout;
    d.off = p->blob_bytes;
    d.off2 = 0;
    size_t bytes = 0;
    size_t n = 1;
    for (int v : dims) n *= (size_t)v;
    switch (pack) {
        case 0: bytes = n * 4; break;
        case 1: case 6: bytes = conv_packed_bytes(CONV_C3, cin, cout); break;      // (6: cin % 32 == 0, the same bytes in 32-channel chunks)
        case 2: bytes = conv_packed_bytes(CONV_DN, cin, cout); break;
        case 3: case 7: bytes = conv_packed_bytes(CONV_UP, cin, cout); break;      // (7: the same bytes in 32-channel chunks, f16 + fp8 format)
        case 4: bytes = conv_packed_bytes(CONV_P1, cin, cout); break;
        case 5: bytes = attn_kv_packed_bytes(cin); break;
    }
    p->blob_bytes = align_up(p->blob_bytes + bytes, 256);
    if (pack == 5) {
        d.off2 = p->blob_bytes;
        p->blob_bytes = align_up(p->blob_bytes + (size_t)128 * cin * 4, 256);
    }
    p->pidx[name] = (int)p->params.size();
    p->params.push_back(d);
    return (int)p->params.size() - 1;
}
static size_t poff(const gtts_plan *p, const std::string &name) { return p->params[p->pidx.at(name)].off; }

static int add_tensor(gtts_plan *p, const std::string &name, int kind, int C, int lvl, bool external = false) {
    Tensor t;
    t.name = name; t.kind = kind; t.C = C; t.lvl = lvl; t.bytes = 0; t.mode = 0; t.cout = 0;
    t.ws = false; t.external = external; t.tref = false; t.first = 1 << 30; t.last = -1;
    p->tensors.push_back(t);
    return (int)p->tensors.size() - 1;
}

static Op blank_op(int kind, const std::string &label) {
    Op o;
    o.kind = kind; o.mode = 0; o.pro = 0; o.epi = 0; o.src0 = o.src1 = -1; o.c0 = o.c1 = 0; o.cout = 0;
    o.lvl_in = o.lvl_out = 0; o.sc = o.sh = -1; o.tb_off = 0; o.w_off = o.b_off = 0; o.w_t = o.bias_t = -1;
    o.out = o.part = o.eh = o.esc = o.esh = o.eres = -1; o.C = 0; o.gamma_off = o.beta_off = 0;
    o.wkv_off = o.wq_off = o.wout_off = o.bout_off = o.g_off = 0; o.apart = o.ctxn = -1; o.use_ref = 0; o.has_tb = 0;
    o.gn_op = -1;
    o.fused = 0;
    o.label = label;
    return o;
}

// Block: conv3x3 (+stats) and the GroupNorm finalize.  Returns raw / sc / sh tensor ids.
static void add_block(gtts_plan *p, const std::string &pre, const std::string &tname, int src0, int c0, int src1,
                      int c1, int cout, int lvl, int pro, int psc, int psh, int tb_off, int *raw, int *sc, int *sh) {
    const int cin = c0 + c1;
    // GTTS_PREC_F16F8: eligible layers keep their weights in the f16 + fp8 format (pack kind 6; same size as kind 1)
    const bool f8 = p->cfg.precision == GTTS_PREC_F16F8 && conv_f16f8_ok(CONV_C3, c0, c1, cout, pro, EPI_STATS, p->cfg.conv_ws);
    add_param(p, pre + "block.0.weight", {cout, cin, 3, 3}, f8 ? 6 : 1, cin, cout);
    add_param(p, pre + "block.0.bias", {cout}, 0);
    add_param(p, pre + "block.1.weight", {cout}, 0);
    add_param(p, pre + "block.1.bias", {cout}, 0);
    *raw = add_tensor(p, tname + ".raw", TK_ACT, cout, lvl);
    int part = add_tensor(p, tname + ".part", TK_PART, p->cfg.groups, lvl);
    p->tensors[part].mode = CONV_C3;
    p->tensors[part].cout = cout;
    p->tensors[part].ws = p->cfg.conv_ws && conv_ws_eligible(CONV_C3, c0, c1, cout, pro, EPI_STATS, plan_nsplit(p), p->cfg.precision == GTTS_PREC_F16F8);
    *sc = add_tensor(p, tname + ".sc", TK_PERB, cout, 0);
    *sh = add_tensor(p, tname + ".sh", TK_PERB, cout, 0);
    Op c = blank_op(OP_CONV, tname + ".conv");
    c.mode = CONV_C3; c.pro = pro; c.epi = EPI_STATS;
    c.src0 = src0; c.src1 = src1; c.c0 = c0; c.c1 = c1; c.cout = cout; c.lvl_in = c.lvl_out = lvl;
    c.sc = psc; c.sh = psh; c.tb_off = tb_off;
    c.w_off = poff(p, pre + "block.0.weight");
    c.b_off = poff(p, pre + "block.0.bias");
    c.out = *raw; c.part = part;
    c.gn_op = (int)p->ops.size() + 1;
    p->ops.push_back(c);
    Op g = blank_op(OP_GNFIN, tname + ".gn");
    g.part = part; g.C = cout; g.lvl_in = lvl;
    g.gamma_off = poff(p, pre + "block.1.weight");
    g.beta_off = poff(p, pre + "block.1.bias");
    g.sc = *sc; g.sh = *sh;
    p->ops.push_back(g);
}

// ResnetBlock (diffusion.py:61-79).  Input = cat(src0[c0], src1[c1]); returns the output tensor id.
static int add_resnet(gtts_plan *p, const std::string &name, int src0, int c0, int src1, int c1, int cout, int lvl) {
    const int cin = c0 + c1, dim = p->cfg.dim;
    const std::string pre = name + ".";
    // registration order inside ResnetBlock: mlp, block1, block2, res_conv
    add_param(p, pre + "mlp.1.weight", {cout, dim}, 0);
    add_param(p, pre + "mlp.1.bias", {cout}, 0);
    TimeMlpDesc &tm = p->tmlp;
    const int r = tm.n++;
    tm.cout[r] = cout;
    tm.w[r] = poff(p, pre + "mlp.1.weight");
    tm.b[r] = poff(p, pre + "mlp.1.bias");
    tm.off[r] = r == 0 ? 0 : tm.off[r - 1] + tm.cout[r - 1];
    int raw1, sc1, sh1, raw2, sc2, sh2;
    add_block(p, pre + "block1.", name + ".b1", src0, c0, src1, c1, cout, lvl, PRO_MASK, -1, -1, 0, &raw1, &sc1, &sh1);
    add_block(p, pre + "block2.", name + ".b2", raw1, cout, -1, 0, cout, lvl, PRO_GN, sc1, sh1, tm.off[r], &raw2, &sc2,
              &sh2);
    const int out = add_tensor(p, name + ".out", TK_ACT, cout, lvl);
    if (cin != cout) {
        add_param(p, pre + "res_conv.weight", {cout, cin, 1, 1}, 4, cin, cout);
        add_param(p, pre + "res_conv.bias", {cout}, 0);
        Op c = blank_op(OP_CONV, name + ".res_tail");
        c.mode = CONV_P1; c.pro = PRO_MASK; c.epi = EPI_TAIL;
        c.src0 = src0; c.src1 = src1; c.c0 = c0; c.c1 = c1; c.cout = cout; c.lvl_in = c.lvl_out = lvl;
        c.w_off = poff(p, pre + "res_conv.weight");
        c.b_off = poff(p, pre + "res_conv.bias");
        c.out = out; c.eh = raw2; c.esc = sc2; c.esh = sh2;
        p->ops.push_back(c);
    } else {
        Op t = blank_op(OP_TAILID, name + ".tail");
        t.src0 = src0; t.eh = raw2; t.esc = sc2; t.esh = sh2; t.out = out; t.C = cout; t.lvl_in = lvl;
        p->ops.push_back(t);
    }
    return out;
}

// Residual(Rezero(LinearAttention)) (diffusion.py:82-110)
static int add_attn(gtts_plan *p, const std::string &name, int src, int C, int lvl) {
    const std::string pre = name + ".";
    add_param(p, pre + "fn.g", {1}, 0);
    add_param(p, pre + "fn.fn.to_qkv.weight", {384, C, 1, 1}, 5, C, 384);
    add_param(p, pre + "fn.fn.to_out.weight", {C, 128, 1, 1}, 0);
    add_param(p, pre + "fn.fn.to_out.bias", {C}, 0);
    const int apart = add_tensor(p, name + ".apart", TK_APART, C, lvl);
    const int ctxn = add_tensor(p, name + ".ctx", TK_PERB, 4096, 0);
    const int wpk = add_tensor(p, name + ".wfold", TK_BYTES_PERB, 0, 0);
    p->tensors[wpk].bytes = align_up(conv_packed_bytes(CONV_P1, C, C), 256);
    const int biasb = add_tensor(p, name + ".bfold", TK_PERB, C, 0);
    const int out = add_tensor(p, name + ".out", TK_ACT, C, lvl);
    const ParamDesc &qkv = p->params[p->pidx.at(pre + "fn.fn.to_qkv.weight")];
    Op a = blank_op(OP_ACTX, name + ".ctx");
    a.src0 = src; a.C = C; a.lvl_in = lvl; a.wkv_off = qkv.off; a.apart = apart;
    p->ops.push_back(a);
    Op m = blank_op(OP_AMERGE, name + ".merge");
    m.apart = apart; m.ctxn = ctxn; m.lvl_in = lvl; m.C = C;
    p->ops.push_back(m);
    Op f = blank_op(OP_AFOLD, name + ".fold");
    f.ctxn = ctxn; f.C = C; f.wq_off = qkv.off2; f.wout_off = poff(p, pre + "fn.fn.to_out.weight");
    f.bout_off = poff(p, pre + "fn.fn.to_out.bias"); f.g_off = poff(p, pre + "fn.g");
    f.w_t = wpk; f.bias_t = biasb;
    p->ops.push_back(f);
    Op c = blank_op(OP_CONV, name + ".apply");
    c.mode = CONV_P1; c.pro = PRO_PLAIN; c.epi = EPI_ATTN;
    c.src0 = src; c.c0 = C; c.cout = C; c.lvl_in = c.lvl_out = lvl;
    c.w_t = wpk; c.bias_t = biasb; c.out = out; c.eres = src;
    p->ops.push_back(c);
    return out;
}

static int add_resample(gtts_plan *p, const std::string &name, int src, int C, int lvl, bool down) {
    const std::string pre = name + ".";
    if (down) add_param(p, pre + "conv.weight", {C, C, 3, 3}, 2, C, C);
    else add_param(p, pre + "conv.weight", {C, C, 4, 4}, (p->cfg.precision == GTTS_PREC_F16F8 && conv_up4_f16f8_ok(C, C)) ? 7 : 3, C, C);
    add_param(p, pre + "conv.bias", {C}, 0);
    const int lvl_out = down ? lvl + 1 : lvl - 1;
    const int out = add_tensor(p, name + ".out", TK_ACT, C, lvl_out);
    Op c = blank_op(OP_CONV, name);
    c.mode = down ? CONV_DN : CONV_UP; c.pro = PRO_MASK; c.epi = EPI_PLAIN;
    c.src0 = src; c.c0 = C; c.cout = C; c.lvl_in = lvl; c.lvl_out = lvl_out;
    c.w_off = poff(p, pre + "conv.weight"); c.b_off = poff(p, pre + "conv.bias"); c.out = out;
    p->ops.push_back(c);
    return out;
}

static void compute_liveness(gtts_plan *p) {
    auto touch = [&](int t, int i) {
        if (t < 0) return;
        p->tensors[t].first = std::min(p->tensors[t].first, i);
        p->tensors[t].last = std::max(p->tensors[t].last, i);
    };
    for (int i = 0; i < (int)p->ops.size(); ++i) {
        const Op &o = p->ops[i];
        const int ids[] = {o.src0, o.src1, o.sc, o.sh, o.w_t, o.bias_t, o.out, o.part, o.eh, o.esc, o.esh, o.eres,
                           o.apart, o.ctxn};
        for (int t : ids) touch(t, i);
        // A Block convolution with the GroupNorm finalize fused in writes the finalize's scale / shift DURING this op (the
        // sample that finishes first publishes while other workgroups still read this op's inputs): they are born here, not
        // at the (skipped) finalize op -- otherwise first-fit may place them inside the region of an input that dies here.
        if (o.kind == OP_CONV && o.gn_op >= 0) {
            touch(p->ops[o.gn_op].sc, i);
            touch(p->ops[o.gn_op].sh, i);
        }
    }
    const int n = (int)p->ops.size();
    // tensors used outside the op program: alive for the whole call
    for (int t : {p->t_x0, p->t_s, p->t_tb, p->t_final_raw, p->t_final_sc, p->t_final_sh, p->t_xtref, p->t_cond, p->t_ticket}) {
        if (t < 0) continue;
        p->tensors[t].first = -1;
        p->tensors[t].last = n + 1;
    }
}

// ------------------------------------------------------------------------------------------------ C ABI: plan
extern "C" int gtts_abi_version(void) { return GTTS_ABI_VERSION; }
extern "C" const char *gtts_last_error(void) { return g_err.c_str(); }

extern "C" int gtts_plan_create(const gtts_unet_cfg *cfg, gtts_plan **out) {
    if (!cfg || !out) return fail(GTTS_E_NULL, "gtts_plan_create: null argument");
    if (cfg->dim <= 0 || cfg->dim % 32 != 0) return fail(GTTS_E_CONFIG, "dim must be a positive multiple of 32 (got %d)", cfg->dim);
    if (cfg->n_feats <= 0 || cfg->n_feats % 4 != 0) return fail(GTTS_E_CONFIG, "n_feats must be a multiple of 4 (got %d)", cfg->n_feats);
    if (cfg->groups != 8) return fail(GTTS_E_CONFIG, "only groups == 8 is supported (got %d)", cfg->groups);
    if (cfg->precision != GTTS_PREC_BF16X3 && cfg->precision != GTTS_PREC_BF16 && cfg->precision != GTTS_PREC_BF16_STORE &&
        cfg->precision != GTTS_PREC_F16F8)
        return fail(GTTS_E_CONFIG, "unknown precision %d", cfg->precision);
    if (cfg->precision == GTTS_PREC_BF16_STORE && cfg->arch != 0)
        return fail(GTTS_E_CONFIG, "bf16 activation storage is implemented for the Grad-TTS decoder (arch 0) only");
    if (cfg->n_spks < 1) return fail(GTTS_E_CONFIG, "n_spks must be >= 1");
    gtts_plan *p = new gtts_plan();
    p->cfg = *cfg;
    if (p->cfg.precision != GTTS_PREC_BF16X3 && p->cfg.precision != GTTS_PREC_F16F8) p->cfg.conv_ws = 0;      // the persistent kernel exists for the fp32-grade splits only
    p->blob_bytes = 0;
    p->nlev = 3;
    const int dim = cfg->dim;
    const bool vc = cfg->arch == 1;
    if (cfg->arch != 0 && cfg->arch != 1) { delete p; return fail(GTTS_E_CONFIG, "unknown arch %d", cfg->arch); }
    if (vc && (cfg->dim_cond <= 0 || cfg->dim_cond % 32 != 0 || cfg->c_dim <= 0 || cfg->n_spks != 1)) {
        delete p;
        return fail(GTTS_E_CONFIG, "DiffVC plan needs dim_cond %% 32 == 0, c_dim > 0, n_spks == 1");
    }
    const bool multi = cfg->n_spks > 1;
    p->cin0 = vc ? 2 + cfg->dim_cond : 2 + (multi ? 1 : 0);
    p->dims[0] = p->cin0; p->dims[1] = dim; p->dims[2] = 2 * dim; p->dims[3] = 4 * dim;
    memset(&p->tmlp, 0, sizeof(p->tmlp));
    p->tmlp.dim = dim;

    // blob: frequencies first
    p->freq_off = p->blob_bytes;
    p->blob_bytes = align_up(p->blob_bytes + (size_t)(dim / 2) * 4, 256);
    p->status_off = p->blob_bytes;
    p->blob_bytes += 256;
    // ---- parameters in the reference's registration order (diffusion.py:139-172)
    if (multi) {
        const int E = cfg->spk_emb_dim;
        add_param(p, "spk_mlp.0.weight", {4 * E, E}, 0);
        add_param(p, "spk_mlp.0.bias", {4 * E}, 0);
        add_param(p, "spk_mlp.2.weight", {cfg->n_feats, 4 * E}, 0);
        add_param(p, "spk_mlp.2.bias", {cfg->n_feats}, 0);
        p->spk_w0 = poff(p, "spk_mlp.0.weight"); p->spk_b0 = poff(p, "spk_mlp.0.bias");
        p->spk_w2 = poff(p, "spk_mlp.2.weight"); p->spk_b2 = poff(p, "spk_mlp.2.bias");
    }
    add_param(p, "mlp.0.weight", {4 * dim, dim}, 0);
    add_param(p, "mlp.0.bias", {4 * dim}, 0);
    add_param(p, "mlp.2.weight", {dim, 4 * dim}, 0);
    add_param(p, "mlp.2.bias", {dim}, 0);
    p->tmlp.w0 = poff(p, "mlp.0.weight"); p->tmlp.b0 = poff(p, "mlp.0.bias");
    p->tmlp.w2 = poff(p, "mlp.2.weight"); p->tmlp.b2 = poff(p, "mlp.2.bias");

    p->t_x0 = add_tensor(p, "x0", TK_ACT, p->cin0, 0);
    p->t_s = multi ? add_tensor(p, "spk_s", TK_PERB, cfg->n_feats, 0) : -1;
    p->t_tb = add_tensor(p, "tb", TK_ROWS, 0, 0);
    p->t_ticket = add_tensor(p, "gn_ticket", TK_PERB, 1, 0);
    p->tmlp.semb_off = -1;

    if (vc) {
        // ---- DiffVC pre-trunk: RefBlock (modules.py:128-166), cond_block, 130-channel input (diffusion.py:62-76)
        const int dc = cfg->dim_cond, base = dc / 4;
        int S = -1;
        if (cfg->use_ref_t) {
            TimeMlpDesc &tm = p->tmlp;
            int tboff[2];
            const char *mn[2] = {"ref_block.mlp1.1", "ref_block.mlp2.1"};
            const int mc[2] = {base, 2 * base};
            for (int k = 0; k < 2; ++k) {
                add_param(p, std::string(mn[k]) + ".weight", {mc[k], dim}, 0);
                add_param(p, std::string(mn[k]) + ".bias", {mc[k]}, 0);
                const int r = tm.n++;
                tm.cout[r] = mc[k];
                tm.w[r] = poff(p, std::string(mn[k]) + ".weight");
                tm.b[r] = poff(p, std::string(mn[k]) + ".bias");
                tm.off[r] = r == 0 ? 0 : tm.off[r - 1] + tm.cout[r - 1];
                tboff[k] = tm.off[r];
            }
            p->t_xtref = add_tensor(p, "xt_ref", TK_ACT, 1, 0);
            p->tensors[p->t_xtref].tref = true;
            struct RB { const char *name; int cin, cout, tbk; };
            const RB rb[6] = {{"block11", 1, 2 * base, -1}, {"block12", base, 2 * base, -1}, {"block21", base, 4 * base, 0},
                              {"block22", 2 * base, 4 * base, -1}, {"block31", 2 * base, 8 * base, 1}, {"block32", 4 * base, 8 * base, -1}};
            int prev_raw = p->t_xtref, prev_sc = -1, prev_sh = -1;
            for (int k = 0; k < 6; ++k) {
                const std::string pre = std::string("ref_block.") + rb[k].name + ".";
                add_param(p, pre + "0.weight", {rb[k].cout, rb[k].cin, 3, 3}, 1, rb[k].cin, rb[k].cout);
                add_param(p, pre + "0.bias", {rb[k].cout}, 0);
                add_param(p, pre + "1.weight", {rb[k].cout}, 0);
                add_param(p, pre + "1.bias", {rb[k].cout}, 0);
                const int raw = add_tensor(p, std::string("ref.") + rb[k].name + ".raw", TK_ACT, rb[k].cout, 0);
                p->tensors[raw].tref = true;
                const int sc = add_tensor(p, std::string("ref.") + rb[k].name + ".sc", TK_PERB, rb[k].cout, 0);
                const int sh = add_tensor(p, std::string("ref.") + rb[k].name + ".sh", TK_PERB, rb[k].cout, 0);
                Op c = blank_op(OP_CONV, std::string("ref.") + rb[k].name + ".conv");
                c.mode = CONV_C3; c.epi = EPI_PLAIN; c.use_ref = 1;
                c.pro = k == 0 ? PRO_MASK : PRO_IGLU;
                c.src0 = prev_raw; c.c0 = rb[k].cin; c.cout = rb[k].cout;
                c.sc = prev_sc; c.sh = prev_sh;
                if (rb[k].tbk >= 0) { c.has_tb = 1; c.tb_off = tboff[rb[k].tbk]; }
                c.w_off = poff(p, pre + "0.weight"); c.b_off = poff(p, pre + "0.bias"); c.out = raw;
                p->ops.push_back(c);
                Op st = blank_op(OP_INSTATS, std::string("ref.") + rb[k].name + ".in");
                st.src0 = raw; st.C = rb[k].cout; st.use_ref = 1; st.sc = sc; st.sh = sh;
                st.gamma_off = poff(p, pre + "1.weight"); st.beta_off = poff(p, pre + "1.bias");
                p->ops.push_back(st);
                prev_raw = raw; prev_sc = sc; prev_sh = sh;
            }
            add_param(p, "ref_block.final_conv.weight", {dc, 4 * base, 1, 1}, 0);
            add_param(p, "ref_block.final_conv.bias", {dc}, 0);
            S = add_tensor(p, "ref.pool", TK_PERB, 4 * base, 0);
            Op rp = blank_op(OP_REFPOOL, "ref.pool");
            rp.src0 = prev_raw; rp.sc = prev_sc; rp.sh = prev_sh; rp.C = 4 * base; rp.out = S; rp.use_ref = 1;
            p->ops.push_back(rp);
        }
        const int nin = dim + (cfg->use_ref_t ? dc : 0) + cfg->c_dim;
        add_param(p, "cond_block.0.weight", {4 * dc, nin}, 0);
        add_param(p, "cond_block.0.bias", {4 * dc}, 0);
        add_param(p, "cond_block.2.weight", {dc, 4 * dc}, 0);
        add_param(p, "cond_block.2.bias", {dc}, 0);
        p->t_cond = add_tensor(p, "cond", TK_PERB, dc, 0);
        Op cd = blank_op(OP_VCCOND, "cond_block");
        cd.src0 = S; cd.C = 4 * base; cd.out = p->t_cond;
        if (cfg->use_ref_t) { cd.w_off = poff(p, "ref_block.final_conv.weight"); cd.b_off = poff(p, "ref_block.final_conv.bias"); }
        cd.wq_off = poff(p, "cond_block.0.weight"); cd.wout_off = poff(p, "cond_block.0.bias");
        cd.bout_off = poff(p, "cond_block.2.weight"); cd.g_off = poff(p, "cond_block.2.bias");
        p->ops.push_back(cd);
        Op pv = blank_op(OP_PREPVC, "prep_input");
        pv.src0 = p->t_cond; pv.out = p->t_x0;
        p->ops.push_back(pv);
    }

    // The U-Net is *executed* downs -> mid -> ups -> final, but `ups` is registered before `mid_*`
    // (diffusion.py:148-149: the empty ModuleList is created first).  Parameter order below follows execution;
    // gtts_plan_param_info re-sorts into registration order.
    int x = p->t_x0, xc = p->cin0;
    std::vector<int> hid, hidc;
    for (int lv = 0; lv < p->nlev; ++lv) {
        const int co = p->dims[lv + 1];
        char nm[64];
        snprintf(nm, sizeof nm, "downs.%d.0", lv);
        x = add_resnet(p, nm, x, xc, -1, 0, co, lv);
        snprintf(nm, sizeof nm, "downs.%d.1", lv);
        x = add_resnet(p, nm, x, co, -1, 0, co, lv);
        snprintf(nm, sizeof nm, "downs.%d.2", lv);
        x = add_attn(p, nm, x, co, lv);
        hid.push_back(x);
        hidc.push_back(co);
        if (lv < p->nlev - 1) {
            snprintf(nm, sizeof nm, "downs.%d.3", lv);
            x = add_resample(p, nm, x, co, lv, true);
        }
        xc = co;
    }
    const int mid = p->dims[p->nlev], lmid = p->nlev - 1;
    x = add_resnet(p, "mid_block1", x, mid, -1, 0, mid, lmid);
    x = add_attn(p, "mid_attn", x, mid, lmid);
    x = add_resnet(p, "mid_block2", x, mid, -1, 0, mid, lmid);
    xc = mid;
    for (int u = 0; u < p->nlev - 1; ++u) {
        const int lv = p->nlev - 1 - u;          // level this stage runs at
        const int ci = p->dims[lv];              // dim_in of the stage (output channels)
        const int co = p->dims[lv + 1];          // dim_out (== current x channels == skip channels)
        char nm[64];
        const int skip = hid.back(); hid.pop_back();
        const int skc = hidc.back(); hidc.pop_back();
        (void)co;
        snprintf(nm, sizeof nm, "ups.%d.0", u);
        x = add_resnet(p, nm, x, xc, skip, skc, ci, lv);      // torch.cat((x, hiddens.pop()), 1)  :207
        snprintf(nm, sizeof nm, "ups.%d.1", u);
        x = add_resnet(p, nm, x, ci, -1, 0, ci, lv);
        snprintf(nm, sizeof nm, "ups.%d.2", u);
        x = add_attn(p, nm, x, ci, lv);
        snprintf(nm, sizeof nm, "ups.%d.3", u);
        x = add_resample(p, nm, x, ci, lv, false);
        xc = ci;
    }
    int fsc, fsh, fraw;
    add_block(p, "final_block.", "final_block", x, xc, -1, 0, dim, 0, PRO_MASK, -1, -1, 0, &fraw, &fsc, &fsh);
    p->t_final_raw = fraw; p->t_final_sc = fsc; p->t_final_sh = fsh;
    add_param(p, "final_conv.weight", {1, dim, 1, 1}, 0);
    add_param(p, "final_conv.bias", {1}, 0);
    p->fw_off = poff(p, "final_conv.weight");
    p->fb_off = poff(p, "final_conv.bias");

    TimeMlpDesc &tm = p->tmlp;
    tm.temb_off = tm.n ? tm.off[tm.n - 1] + tm.cout[tm.n - 1] : 0;
    tm.tb_stride = tm.temb_off + dim;
    if (vc) { tm.semb_off = tm.tb_stride; tm.tb_stride += dim; }
    if (tm.n > 32) { delete p; return fail(GTTS_E_CONFIG, "too many ResnetBlocks"); }
    for (const Op &o : p->ops)
        if (o.kind == OP_CONV && o.c1 > 0 && (o.c0 % 8) != 0) { delete p; return fail(GTTS_E_CONFIG, "concat split must be a multiple of 8 channels"); }
    // ResnetBlock identity tail -> attention: the 64-channel context kernel applies the tail while it stages (attn.hip), so the tensor
    // is written once and read once less.  fp32 storage only; wider attentions keep the separate tail (their x tile is staged by two
    // workgroups, the tail's Mish would run twice).
    if (p->cfg.precision != GTTS_PREC_BF16_STORE && GTTS_FUSE_TAIL_CTX) {
        for (size_t i = 0; i + 1 < p->ops.size(); ++i) {
            Op &t = p->ops[i], &c = p->ops[i + 1];
            if (t.kind != OP_TAILID || c.kind != OP_ACTX || c.src0 != t.out || t.C != c.C) continue;
            if (!attn_head_per_wave(c.C) && !(GTTS_FUSE_TAIL_CTX > 1 && c.C % 32 == 0)) continue;
            t.fused = 1;
            c.fused = 1;
            c.eh = t.eh; c.esc = t.esc; c.esh = t.esh; c.eres = t.src0;
        }
    }
    compute_liveness(p);
    *out = p;
    return GTTS_OK;
}

extern "C" void gtts_plan_destroy(gtts_plan *plan) {
    if (!plan) return;
    for (int h = 0; h < 4; ++h)
        if (plan->ev_join[h]) (void)hipEventDestroy(plan->ev_join[h]);      // the side streams belong to the caller
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    for (auto &g : plan->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    for (auto &e : plan->prof_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &r : plan->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    delete plan;
}

// Sub-batch streams.  The sampler can run sub-batches of the utterance batch side by side on CALLER-OWNED side
// streams registered with gtts_plan_set_streams (results are bit-identical to the unsplit run: no operation mixes
// batch entries).  Measured on MI355X at B=16, T=1024: 8.23 / 7.88 / 7.76 / 9.02 ms per U-Net call for 1 / 2 / 3 / 4
// sub-batches.  Without registered streams everything runs on the stream passed to the call.
constexpr int MAX_SUB = 4;
static int sampler_parts(const gtts_plan *p, int B) { return std::max(1, std::min(B, p->nsub)); }

extern "C" int gtts_plan_set_streams(gtts_plan *plan, const gtts_stream_t *streams, int n) {
    if (!plan) return fail(GTTS_E_NULL, "null plan");
    if (n < 0 || n > MAX_SUB) return fail(GTTS_E_SHAPE, "at most %d side streams (got %d)", MAX_SUB, n);
    if (n == 1) return fail(GTTS_E_SHAPE, "register 0 (no split) or 2..%d side streams", MAX_SUB);
    if (n > 0 && !streams) return fail(GTTS_E_NULL, "gtts_plan_set_streams: null stream array");
    std::lock_guard<std::mutex> lk(plan->mu);
    for (int h = 0; h < n; ++h) {
        if (!plan->ev_join[h]) HIPCHK(hipEventCreateWithFlags(&plan->ev_join[h], hipEventDisableTiming));
        plan->sub[h] = (hipStream_t)streams[h];
    }
    if (n > 0 && !plan->ev_fork) HIPCHK(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
    plan->nsub = n;
    return GTTS_OK;
}

#ifdef GTTS_DIAG
// Diagnostic builds only (-DGTTS_DIAG, never the product library).  GTTS_SKIP_OPS: bit 0 skip gn_finalize, bit 1 skip
// attn_merge + attn_fold (results WRONG); bit 2 launch gn_finalize twice, bit 3 launch attn_merge / attn_fold twice
// (idempotent, results valid: the time difference to the normal run is the in-situ cost of those launches).
static int skip_op_mask() {
    const char *e = getenv("GTTS_SKIP_OPS");
    return e ? atoi(e) : 0;
}
#else
static constexpr int skip_op_mask() { return 0; }
#endif

// registration order: spk_mlp, mlp, downs, ups, mid_block1, mid_attn, mid_block2, final_block, final_conv
static int reg_rank(const std::string &n) {
    if (n.rfind("spk_mlp", 0) == 0) return 0;
    if (n.rfind("mlp.", 0) == 0) return 1;
    if (n.rfind("ref_block", 0) == 0) return 1;     // DiffVC: mlp, ref_block, cond_block precede the trunk, in this order
    if (n.rfind("cond_block", 0) == 0) return 1;
    if (n.rfind("downs", 0) == 0) return 2;
    if (n.rfind("ups", 0) == 0) return 3;
    if (n.rfind("mid_block1", 0) == 0) return 4;
    if (n.rfind("mid_attn", 0) == 0) return 5;
    if (n.rfind("mid_block2", 0) == 0) return 6;
    if (n.rfind("final_block", 0) == 0) return 7;
    return 8;
}
static std::vector<int> reg_order(const gtts_plan *p) {
    std::vector<int> idx(p->params.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return reg_rank(p->params[a].name) < reg_rank(p->params[b].name); });
    return idx;
}

static int plan_nsplit(const gtts_plan *p) { return (p->cfg.precision == GTTS_PREC_BF16X3 || p->cfg.precision == GTTS_PREC_F16F8) ? 2 : 1; }

extern "C" int gtts_plan_num_params(const gtts_plan *plan) { return plan ? (int)plan->params.size() : 0; }

extern "C" int gtts_plan_param_info(const gtts_plan *plan, int i, const char **name, int *rank, int dims[4]) {
    if (!plan) return fail(GTTS_E_NULL, "null plan");
    if (i < 0 || i >= (int)plan->params.size()) return fail(GTTS_E_SHAPE, "parameter index %d out of range", i);
    const ParamDesc &d = plan->params[reg_order(plan)[i]];
    if (name) *name = d.name.c_str();
    if (rank) *rank = d.rank;
    if (dims) for (int k = 0; k < 4; ++k) dims[k] = d.dims[k];
    return GTTS_OK;
}

extern "C" size_t gtts_packed_weight_bytes(const gtts_plan *plan) { return plan ? plan->blob_bytes : 0; }

extern "C" int gtts_pack_weights(const gtts_plan *plan, const void *const *param_ptrs, int n_params, const float *freq,
                                 void *packed, gtts_stream_t stream) {
    if (!plan || !param_ptrs || !packed || !freq) return fail(GTTS_E_NULL, "gtts_pack_weights: null argument");
    if (n_params != (int)plan->params.size()) return fail(GTTS_E_PARAMS, "expected %d parameters, got %d", (int)plan->params.size(), n_params);
    hipStream_t st = (hipStream_t)stream;
    unsigned char *blob = (unsigned char *)packed;
    HIPCHK(hipMemsetAsync(blob, 0, plan->blob_bytes, st));
    HIPCHK(launch_copy_f32(freq, (float *)(blob + plan->freq_off), plan->cfg.dim / 2, st));
    std::vector<int> order = reg_order(plan);
    for (int i = 0; i < n_params; ++i) {
        const ParamDesc &d = plan->params[order[i]];
        const float *src = (const float *)param_ptrs[i];
        if (!src) return fail(GTTS_E_NULL, "parameter %s is null", d.name.c_str());
        size_t n = 1;
        for (int k = 0; k < d.rank; ++k) n *= (size_t)d.dims[k];
        switch (d.pack) {
            case 0: HIPCHK(launch_copy_f32(src, (float *)(blob + d.off), n, st)); break;
            case 1: HIPCHK(launch_pack_conv(CONV_C3, src, blob + d.off, d.cin, d.cout, st)); break;
            case 2: HIPCHK(launch_pack_conv(CONV_DN, src, blob + d.off, d.cin, d.cout, st)); break;
            case 3: HIPCHK(launch_pack_conv(CONV_UP, src, blob + d.off, d.cin, d.cout, st)); break;
            case 7: HIPCHK(launch_pack_conv(CONV_UP | 32, src, blob + d.off, d.cin, d.cout, st, (unsigned *)(blob + plan->status_off), (unsigned)i)); break;
            case 4: HIPCHK(launch_pack_conv(CONV_P1, src, blob + d.off, d.cin, d.cout, st)); break;
            case 6: HIPCHK(launch_pack_conv(CONV_C3 | 32, src, blob + d.off, d.cin, d.cout, st, (unsigned *)(blob + plan->status_off), (unsigned)i)); break;
            case 5:
                HIPCHK(launch_pack_attn_kv(src, blob + d.off, d.cin, st));
                HIPCHK(launch_copy_f32(src, (float *)(blob + d.off2), (size_t)128 * d.cin, st));   // q rows 0..127
                break;
        }
    }
    if (plan->cfg.precision == GTTS_PREC_F16F8) {
        // Range contract of the f16 + fp8 format (include/gradtts_abi.h): a Block-convolution weight with |w| 2^S beyond the fp16
        // range cannot be represented (it was packed saturated, never inf).  The packer counted such weights on the device; this
        // one call of the ABI synchronises its stream to read the count, so that a checkpoint outside the range is REFUSED here
        // instead of sampling silently at a lower grade.
        unsigned rec[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(rec, blob + plan->status_off, sizeof(rec), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (rec[0] != 0) {
            const float mx = __builtin_bit_cast(float, rec[1]);
            const int pi = (int)rec[2] - 1;
            const char *nm = (pi >= 0 && pi < n_params) ? plan->params[order[pi]].name.c_str() : "?";
            return fail(GTTS_E_RANGE, "GTTS_PREC_F16F8 needs |w| < %.2f on the 3x3 Block convolutions: %u weight(s) out of range, max |w| = %g "
                        "(e.g. in %s); pack this model with GTTS_PREC_BF16X3", 65504.0 / (double)(1 << F8_S), rec[0], (double)mx, nm);
        }
    }
    return GTTS_OK;
}

// ------------------------------------------------------------------------------------------------ workspace
// GroupNorm partial slots per sample of a TK_PART tensor (the count its producing kernel writes)
static int part_slots(const Tensor &t, int H, int W) {
    return t.ws ? conv_ws_nparts(t.cout, H, W) : conv_nparts(t.mode, t.cout, H, W);
}

static size_t tensor_bytes(const gtts_plan *p, const Tensor &t, int B, int T, int rows, int Tr) {
    const int F = p->cfg.n_feats;
    const size_t H = (size_t)F >> t.lvl, W = (size_t)(t.tref ? Tr : T) >> t.lvl;
    switch (t.kind) {
        case TK_ACT: return (size_t)B * t.C * H * W * (p->cfg.precision == GTTS_PREC_BF16_STORE ? 2 : 4);
        case TK_PERB: return (size_t)B * t.C * 4;
        case TK_PART: return (size_t)B * part_slots(t, (int)H, (int)W) * t.C * 2 * 4;
        case TK_APART: return (size_t)B * 4 * attn_geom((int)(H * W), t.C).nrec * ATTN_REC * 4;
        case TK_BYTES_PERB: return (size_t)B * t.bytes;
        case TK_ROWS: return ((size_t)rows * p->tmlp.tb_stride + 4096) * 4;   // + the sampler's step times
    }
    return 0;
}

// (B,T) -> offsets.  keep_intermediates: every tensor gets its own slot; otherwise first-fit reuse by liveness.
// Pure function of the plan (queries use it directly; the enqueueing calls cache its result under the plan mutex).
static Layout compute_layout(const gtts_plan *p, int B, int T, int rows, int Tr) {
    if (Tr <= 0) Tr = T;
    Layout L;
    const int n = (int)p->tensors.size();
    L.offsets.assign(n, 0);
    std::vector<size_t> sz(n);
    for (int i = 0; i < n; ++i) sz[i] = align_up(tensor_bytes(p, p->tensors[i], B, T, rows, Tr), 256);
    // 256 bytes of slack on both sides: the 16-byte halo loads of conv_ws.hip start one frame in front of a row and end up to
    // three frames behind it (those frames are masked out, but the addresses must be mapped)
    constexpr size_t PAD = 256;
    size_t top = PAD;
    if (p->cfg.keep_intermediates) {
        for (int i = 0; i < n; ++i) { L.offsets[i] = top; top += sz[i]; }
    } else {
        struct Blk { size_t off, size; int last; };
        std::vector<Blk> live;
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p->tensors[a].first < p->tensors[b].first; });
        for (int id : order) {
            const Tensor &t = p->tensors[id];
            if (t.last < 0) continue;     // never used
            // drop blocks whose tensor died before this one is first written
            live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk &b) { return b.last < t.first; }), live.end());
            std::sort(live.begin(), live.end(), [](const Blk &a, const Blk &b) { return a.off < b.off; });
            size_t pos = PAD;
            for (const Blk &b : live) {
                if (pos + sz[id] <= b.off) break;
                pos = std::max(pos, b.off + b.size);
            }
            L.offsets[id] = pos;
            live.push_back({pos, sz[id], t.last});
            top = std::max(top, pos + sz[id]);
        }
    }
    L.ws_bytes = top + PAD;
    return L;
}

// cached variant for the enqueueing calls (caller holds p->mu)
static void layout_workspace(gtts_plan *p, int B, int T, int rows, int Tr = 0) {
    if (Tr <= 0) Tr = T;
    if (p->cache_B == B && p->cache_T == T * 4096 + rows && p->cache_Tr == Tr) return;
    Layout L = compute_layout(p, B, T, rows, Tr);
    p->offsets.swap(L.offsets);
    p->ws_bytes = L.ws_bytes;
    p->cache_Tr = Tr;
    p->cache_B = B;
    p->cache_T = T * 4096 + rows;
}

static int check_shape(const gtts_plan *p, int B, int T) {
    if (!p) return fail(GTTS_E_NULL, "null plan");
    if (B <= 0 || T <= 0) return fail(GTTS_E_SHAPE, "B and T must be positive (B=%d, T=%d)", B, T);
    if (T % 4 != 0) return fail(GTTS_E_SHAPE, "T must be a multiple of 4 (fix_len_compatibility), got %d", T);
    return GTTS_OK;
}

extern "C" size_t gtts_workspace_bytes(const gtts_plan *plan, int B, int T) {
    if (check_shape(plan, B, T) != GTTS_OK) return 0;
    // sized for the sampler's worst case: one tb row per step is tiny, allow up to 4096 rows
    size_t whole = compute_layout(plan, B, T, std::max(B, 4096), T).ws_bytes;
    const int parts = sampler_parts(plan, B);     // the sampler runs sub-batches side by side, each in its own slice
    if (parts > 1) {
        const int Bh = (B + parts - 1) / parts;
        whole = std::max(whole, parts * align_up(compute_layout(plan, Bh, T, std::max(Bh, 4096), T).ws_bytes, 256));
    }
    return whole;
}

// The activation range record of the last estimator / sampler call that used this workspace (common.h, f8_range_note): the one
// query of the ABI that synchronises.
extern "C" int gtts_workspace_status(const void *workspace, unsigned *n_events, float *max_abs, gtts_stream_t stream) {
    if (!workspace) return fail(GTTS_E_NULL, "gtts_workspace_status: null workspace");
    unsigned rec[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(rec, workspace, sizeof(rec), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (n_events) *n_events = rec[0];
    if (max_abs) *max_abs = __builtin_bit_cast(float, rec[1]);
    return GTTS_OK;
}

extern "C" int gtts_plan_num_tensors(const gtts_plan *plan) { return plan ? (int)plan->tensors.size() : 0; }

extern "C" int gtts_vc_tensor_info(const gtts_plan *plan, int i, int B, int T, int T_ref, const char **name, size_t *offset,
                                   int dims[4]);
extern "C" int gtts_plan_tensor_info(const gtts_plan *plan, int i, int B, int T, const char **name, size_t *offset,
                                     int dims[4]) {
    return gtts_vc_tensor_info(plan, i, B, T, T, name, offset, dims);
}

extern "C" int gtts_vc_tensor_info(const gtts_plan *plan, int i, int B, int T, int T_ref, const char **name, size_t *offset,
                                   int dims[4]) {
    if (check_shape(plan, B, T) != GTTS_OK) return GTTS_E_SHAPE;
    if (i < 0 || i >= (int)plan->tensors.size()) return fail(GTTS_E_SHAPE, "tensor index out of range");
    const gtts_plan *p = plan;
    const Tensor &t = p->tensors[i];
    if (name) *name = t.name.c_str();
    if (offset) *offset = compute_layout(p, B, T, std::max(B, 4096), T_ref).offsets[i];
    if (dims) {
        dims[0] = B; dims[1] = t.C; dims[2] = p->cfg.n_feats >> t.lvl; dims[3] = (t.tref ? T_ref : T) >> t.lvl;
        if (t.kind != TK_ACT) { dims[2] = 1; dims[3] = 1; }
        if (t.kind == TK_ROWS) { dims[0] = B; dims[1] = p->tmlp.tb_stride; }   // first B rows (estimator call)
    }
    return GTTS_OK;
}

// ------------------------------------------------------------------------------------------------ execution
struct RunCtx {
    gtts_plan *p;
    const unsigned char *blob;
    unsigned char *ws;
    const float *mask;
    int B, T;
    const float *tb_row;   // tb rows of this call
    int tb_bstride;        // floats between samples' rows (0: one row shared by the batch)
    hipStream_t st;
    unsigned *sat = nullptr;   // activation range record of the call (first 16 bytes of the caller's workspace; gtts_workspace_status)
    // DiffVC extras
    const float *ref_mask = nullptr;
    int Tr = 0;
    const float *in_x = nullptr, *in_mean = nullptr, *in_c = nullptr;
};

enum { XOP_PREP = 0, XOP_TIME = 1, XOP_FINAL = 2, XOP_MULMASK = 3, XOP_SPK = 4, XOP_COUNT = 5 };

struct ProfScope {
    gtts_plan *p;
    hipStream_t st;
    int idx;
    ProfScope(gtts_plan *plan, hipStream_t s, int op) : p(plan), st(s), idx(-1) {
        if (!p->prof_on) return;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (!p->prof_pool.empty()) { ev = p->prof_pool.back(); p->prof_pool.pop_back(); }
        else { if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return; }
        int si = 0;                                            // 0: the call's stream, 1 + h: side stream h
        for (int h = 0; h < p->nsub; ++h) if (p->sub[h] == s) si = 1 + h;
        p->prof.push_back({op, ev.first, ev.second, si});
        idx = (int)p->prof.size() - 1;
        (void)hipEventRecord(ev.first, st);
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(p->prof[idx].b, st); }
};

static inline float *tptr(const RunCtx &c, int id) { return id < 0 ? nullptr : (float *)(c.ws + c.p->offsets[id]); }

static int run_ops(const RunCtx &c) {
    gtts_plan *p = c.p;
    const int F = p->cfg.n_feats, nsplit = plan_nsplit(p);
    const int abf = p->cfg.precision == GTTS_PREC_BF16_STORE ? 1 : 0;
    for (size_t oi = 0; oi < p->ops.size(); ++oi) {
        const Op &o = p->ops[oi];
        if (o.kind == OP_GNFIN && oi > 0 && p->ops[oi - 1].kind == OP_CONV && p->ops[oi - 1].gn_op == (int)oi && !p->ops[oi - 1].use_ref)
            continue;                               // done by the producing convolution's last workgroup (no launch)
        if (o.kind == OP_TAILID && o.fused) continue;   // done by the attention context pass that follows (no launch)
        ProfScope prof_scope(p, c.st, (int)oi);
        switch (o.kind) {
            case OP_CONV: {
                ConvArgs a;
                memset(&a, 0, sizeof(a));
                a.src0 = tptr(c, o.src0);
                a.src1 = o.src1 >= 0 ? tptr(c, o.src1) : a.src0;
                a.c0 = o.c0; a.c1 = o.c1; a.cin = o.c0 + o.c1; a.nchunk = 0;
                a.B = c.B;
                const int Tw = o.use_ref ? c.Tr : c.T;
                a.Hin = F >> o.lvl_in; a.Win = Tw >> o.lvl_in;
                a.Hout = F >> o.lvl_out; a.Wout = Tw >> o.lvl_out;
                a.mask = o.use_ref ? c.ref_mask : c.mask; a.T = Tw; a.lvl_in = o.lvl_in; a.lvl_out = o.lvl_out;
                a.pro = o.pro;
                a.sc = tptr(c, o.sc); a.sh = tptr(c, o.sh);
                a.tb = c.tb_row + o.tb_off; a.tb_stride = c.tb_bstride;
                if (o.pro == PRO_IGLU && !o.has_tb) a.tb = nullptr;
                if (o.w_t >= 0) {
                    a.w = (const unsigned char *)tptr(c, o.w_t);
                    a.w_bstride = p->tensors[o.w_t].bytes;
                    a.bias = tptr(c, o.bias_t);
                    a.bias_bstride = (size_t)o.cout;
                } else {
                    a.w = c.blob + o.w_off; a.w_bstride = 0;
                    a.bias = (const float *)(c.blob + o.b_off); a.bias_bstride = 0;
                }
                a.cout = o.cout; a.epi = o.epi;
                a.out = tptr(c, o.out);
                a.partials = tptr(c, o.part);
                a.nparts = conv_nparts(o.mode, o.cout, a.Hout, a.Wout);
                a.groups = p->cfg.groups;
                a.eh = tptr(c, o.eh); a.esc = tptr(c, o.esc); a.esh = tptr(c, o.esh); a.eres = tptr(c, o.eres);
                a.nsplit = nsplit;
                a.f16f8 = p->cfg.precision == GTTS_PREC_F16F8 ? 1 : 0;
                a.use_ws = p->cfg.conv_ws;
                a.act_bf16 = abf;
                a.sat = c.sat;
                if (o.epi == EPI_STATS && o.gn_op >= 0 && !o.use_ref) {          // GroupNorm finalize rides in the epilogue
                    const Op &gn = p->ops[o.gn_op];
                    a.ticket = (unsigned *)tptr(c, p->t_ticket);
                    a.gn_gamma = (const float *)(c.blob + gn.gamma_off);
                    a.gn_beta = (const float *)(c.blob + gn.beta_off);
                    a.gn_sc = tptr(c, gn.sc);
                    a.gn_sh = tptr(c, gn.sh);
                    a.gn_count = (float)((double)(o.cout / p->cfg.groups) * (double)a.Hout * (double)a.Wout);
                }
                hipError_t e = launch_conv(o.mode, a, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "conv %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
            case OP_GNFIN: {
                if (skip_op_mask() & 1) break;      // timing-only ablation (GTTS_SKIP_OPS), results are wrong
                const int H = F >> o.lvl_in, W = c.T >> o.lvl_in;
                const Tensor &pt = p->tensors[o.part];
                hipError_t e = hipSuccess;
                for (int rep = 0; rep < ((skip_op_mask() & 4) ? 2 : 1); ++rep)      // bit 2: launch twice (idempotent)
                e = launch_gn_finalize(tptr(c, o.part), part_slots(pt, H, W), p->cfg.groups, o.C,
                                                  H * W, (const float *)(c.blob + o.gamma_off),
                                                  (const float *)(c.blob + o.beta_off), tptr(c, o.sc), tptr(c, o.sh), c.B, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "gn_finalize %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
            case OP_TAILID: {
                if (o.fused) break;                 // applied by the attention context pass that follows (no launch)
                const int H = F >> o.lvl_in, W = c.T >> o.lvl_in;
                hipError_t e = launch_tail_identity(tptr(c, o.eh), tptr(c, o.src0), tptr(c, o.esc), tptr(c, o.esh), c.mask,
                                                    tptr(c, o.out), c.B, o.C, H, W, c.T, o.lvl_in, c.st, abf);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "tail %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
            case OP_ACTX: {
                const int HW = (F >> o.lvl_in) * (c.T >> o.lvl_in);
                AttnTail tl;
                if (o.fused) {
                    tl.h = tptr(c, o.eh); tl.xin = tptr(c, o.eres); tl.esc = tptr(c, o.esc); tl.esh = tptr(c, o.esh);
                    tl.mask = c.mask; tl.W = c.T >> o.lvl_in; tl.T = c.T; tl.lvl = o.lvl_in;
                }
                hipError_t e = launch_attn_ctx(tptr(c, o.src0), c.blob + o.wkv_off, tptr(c, o.apart), c.B, o.C, HW, nsplit, c.st, abf,
                                               o.fused ? &tl : nullptr);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "attn_ctx %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
            case OP_AMERGE: {
                if (skip_op_mask() & 2) break;
                const int HW = (F >> o.lvl_in) * (c.T >> o.lvl_in);
                hipError_t e = hipSuccess;
                for (int rep = 0; rep < ((skip_op_mask() & 8) ? 2 : 1); ++rep)      // bit 3: launch twice (idempotent)
                e = launch_attn_merge(tptr(c, o.apart), tptr(c, o.ctxn), c.B, attn_geom(HW, o.C).nrec, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "attn_merge %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
            case OP_INSTATS: {
                const int HW = (F >> o.lvl_in) * ((o.use_ref ? c.Tr : c.T) >> o.lvl_in);
                hipError_t e = launch_instnorm_stats(tptr(c, o.src0), (const float *)(c.blob + o.gamma_off),
                                                     (const float *)(c.blob + o.beta_off), tptr(c, o.sc), tptr(c, o.sh), c.B, o.C, HW, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "instnorm_stats %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
            case OP_REFPOOL: {
                hipError_t e = launch_ref_pool(tptr(c, o.src0), tptr(c, o.sc), tptr(c, o.sh), c.ref_mask, tptr(c, o.out), c.B, o.C, F, c.Tr, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "ref_pool: %s", hipGetErrorString(e));
                break;
            }
            case OP_VCCOND: {
                const gtts_unet_cfg &cf = p->cfg;
                hipError_t e = launch_vc_cond(c.tb_row, c.tb_bstride, p->tmlp.semb_off, cf.dim, tptr(c, o.src0), c.ref_mask,
                                              (const float *)(c.blob + o.w_off), (const float *)(c.blob + o.b_off), c.in_c,
                                              (const float *)(c.blob + o.wq_off), (const float *)(c.blob + o.wout_off),
                                              (const float *)(c.blob + o.bout_off), (const float *)(c.blob + o.g_off), tptr(c, o.out),
                                              c.B, o.C, cf.dim_cond, cf.c_dim, F, c.Tr, cf.use_ref_t, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "vc_cond: %s", hipGetErrorString(e));
                break;
            }
            case OP_PREPVC: {
                hipError_t e = launch_prep_vc(c.in_mean, c.in_x, tptr(c, o.src0), tptr(c, o.out), c.B, F, c.T, p->cfg.dim_cond, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "prep_vc: %s", hipGetErrorString(e));
                break;
            }
            case OP_AFOLD: {
                if (skip_op_mask() & 2) break;
                hipError_t e = hipSuccess;
                for (int rep = 0; rep < ((skip_op_mask() & 8) ? 2 : 1); ++rep)
                e = launch_attn_fold(tptr(c, o.ctxn), (const float *)(c.blob + o.wq_off),
                                                (const float *)(c.blob + o.wout_off), (const float *)(c.blob + o.bout_off),
                                                (const float *)(c.blob + o.g_off), (unsigned char *)tptr(c, o.w_t),
                                                p->tensors[o.w_t].bytes, tptr(c, o.bias_t), c.B, o.C, c.st);
                if (e != hipSuccess) return fail(GTTS_E_HIP, "attn_fold %s: %s", o.label.c_str(), hipGetErrorString(e));
                break;
            }
        }
    }
    return GTTS_OK;
}

// step times of the samplers, computed on the device in double like the reference's Python arithmetic:
// Grad-TTS t_i = float(1 - (i + 0.5) h) (midpoint, diffusion.py:259), DiffVC t_i = float(1 - i h) (DiffVC/model/diffusion.py:170)
__global__ void sampler_times_kernel(float *t, int n, double shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = (float)(1.0 - ((double)i + shift) * (1.0 / (double)n));
}

static int prepare(gtts_plan *p, int B, int T, int rows, size_t workspace_bytes) {
    layout_workspace(p, B, T, std::max(B, 4096));
    (void)rows;
    if (workspace_bytes < p->ws_bytes) return fail(GTTS_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p->ws_bytes, workspace_bytes);
    return GTTS_OK;
}

extern "C" int gtts_estimator_forward(gtts_plan *plan, const void *packed, const float *x, const float *mask,
                                      const float *mu, const float *t, const float *spk, float *out, void *workspace,
                                      size_t workspace_bytes, int B, int T, gtts_stream_t stream) {
    int rc = check_shape(plan, B, T);
    if (rc) return rc;
    if (!packed || !x || !mask || !mu || !t || !out || !workspace) return fail(GTTS_E_NULL, "gtts_estimator_forward: null argument");
    if (plan->cfg.arch != 0) return fail(GTTS_E_CONFIG, "gtts_estimator_forward needs a Grad-TTS plan (arch 0); use gtts_vc_estimator_forward");
    gtts_plan *p = plan;
    std::lock_guard<std::mutex> lk(p->mu);
    const bool multi = p->cfg.n_spks > 1;
    if (multi && !spk) return fail(GTTS_E_NULL, "multi-speaker plan needs spk");
    rc = prepare(p, B, T, B, workspace_bytes);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char *blob = (const unsigned char *)packed;
    RunCtx c{p, blob, (unsigned char *)workspace, mask, B, T, nullptr, 0, st};
    const int F = p->cfg.n_feats;
    c.sat = (unsigned *)workspace;
    HIPCHK(hipMemsetAsync(workspace, 0, 16, st));                             // activation range record of this call (gtts_workspace_status)
    HIPCHK(hipMemsetAsync(tptr(c, p->t_ticket), 0, (size_t)B * 4, st));      // tickets of the fused GroupNorm finalize
    float *s = nullptr;
    if (multi) {
        s = tptr(c, p->t_s);
        { ProfScope ps_(p, st, (int)p->ops.size() + XOP_SPK); HIPCHK(launch_spk_mlp(spk, (const float *)(blob + p->spk_w0), (const float *)(blob + p->spk_b0),
                              (const float *)(blob + p->spk_w2), (const float *)(blob + p->spk_b2), s, B, p->cfg.spk_emb_dim, F, st)); }
    }
    float *tb = tptr(c, p->t_tb);
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_TIME); HIPCHK(launch_time_mlp(t, (const float *)(blob + p->freq_off), p->cfg.pe_scale, blob, p->tmlp, tb, B, st)); }
    c.tb_row = tb;
    c.tb_bstride = p->tmlp.tb_stride;
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_PREP); HIPCHK(launch_prep_input(mu, x, s, tptr(c, p->t_x0), B, F, T, p->cin0, st, p->cfg.precision == GTTS_PREC_BF16_STORE)); }
    rc = run_ops(c);
    if (rc) return rc;
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_FINAL); HIPCHK(launch_final_euler(tptr(c, p->t_final_raw), tptr(c, p->t_final_sc), tptr(c, p->t_final_sh),
                              (const float *)(blob + p->fw_off), (const float *)(blob + p->fb_off), mask, B, p->cfg.dim, F, T,
                              out, nullptr, nullptr, nullptr, 0.f, 0.f, st, nullptr, p->cfg.precision == GTTS_PREC_BF16_STORE)); }
    return GTTS_OK;
}

extern "C" int gtts_euler_step(float *xt, const float *mu, const float *est, const float *mask, const float *noise,
                               float beta_t, float h, int B, int F, int T, gtts_stream_t stream) {
    if (!xt || !mu || !est || !mask) return fail(GTTS_E_NULL, "gtts_euler_step: null argument");
    if (B <= 0 || F <= 0 || T <= 0) return fail(GTTS_E_SHAPE, "gtts_euler_step: bad shape");
    HIPCHK(launch_euler_step(xt, mu, est, mask, noise, beta_t, h, B, F, T, (hipStream_t)stream));
    return GTTS_OK;
}

// enqueue one sampler call (caller holds p->mu and has validated the arguments)
static int enqueue_reverse_diffusion(gtts_plan *p, const void *packed, const float *z, const float *mask, const float *mu,
                                     const float *spk, const float *noise, float *out, void *workspace, size_t workspace_bytes,
                                     int B, int T, int n_timesteps, int step_begin, int step_end, hipStream_t st) {
    int rc = GTTS_OK;
    const bool multi = p->cfg.n_spks > 1;
    const unsigned char *blob = (const unsigned char *)packed;
    const int F = p->cfg.n_feats, N = n_timesteps, E = p->cfg.spk_emb_dim;

    // Utterances are independent, so the batch is split in two halves that run the whole N-step loop side by side
    // on two streams: the tail of one half's kernel overlaps the other half's next kernel, and HBM-bound kernels
    // overlap MFMA-bound ones.  Results are bit-identical to the unsplit run (no operation mixes batch entries).
    // per-op profiling (gtts_profile_enable) runs the batch unsplit: one launch per op owns the GPU, so an op's
    // HIP-event duration and its whole-batch algorithmic work describe the same thing
    const int nhalf = p->prof_on == 1 ? 1 : sampler_parts(p, B);
    const int Bh0 = (B + nhalf - 1) / nhalf;
    layout_workspace(p, Bh0, T, std::max(Bh0, 4096));
    const size_t ws_half = align_up(p->ws_bytes, 256);
    if (workspace_bytes < ws_half * nhalf)
        return fail(GTTS_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", ws_half * nhalf, workspace_bytes);

    RunCtx c0{p, blob, (unsigned char *)workspace, mask, B, T, nullptr, 0, st};
    // time embeddings of all N steps in one launch: t is batch-uniform inside the sampler (diffusion.py:259).
    // The step times live in the 4096 floats behind the tb rows.  Both halves read these rows.
    float *tb = tptr(c0, p->t_tb);
    float *tvals = tb + (size_t)std::max(Bh0, 4096) * p->tmlp.tb_stride;
    hipLaunchKernelGGL(sampler_times_kernel, dim3((N + 255) / 256), dim3(256), 0, st, tvals, N, 0.5);
    HIPCHK(hipGetLastError());
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_TIME); HIPCHK(launch_time_mlp(tvals, (const float *)(blob + p->freq_off), p->cfg.pe_scale, blob, p->tmlp, tb, N, st)); }
    if (step_begin == 0) { ProfScope ps_(p, st, (int)p->ops.size() + XOP_MULMASK); HIPCHK(launch_mul_mask(z, mask, out, B, F, T, st)); }   // xt = z * mask  (diffusion.py:257)
    if (step_begin == 0) HIPCHK(hipMemsetAsync(workspace, 0, 16, st));      // activation range record: sticky over the step ranges of one sampling run

    struct Half { RunCtx c; int b0; float *s; };
    Half hv[MAX_SUB];
    for (int h = 0; h < nhalf; ++h) {
        // balanced contiguous split (16 utterances on 3 streams: 6 + 5 + 5, not 6 + 6 + 4)
        const int base = B / nhalf, rem = B % nhalf;
        const int b0 = h * base + std::min(h, rem);
        const int bn = base + (h < rem ? 1 : 0);
        hipStream_t hs = nhalf > 1 ? p->sub[h] : st;
        hv[h].c = RunCtx{p, blob, (unsigned char *)workspace + (size_t)h * ws_half, mask + (size_t)b0 * T, bn, T, nullptr, 0, hs};
        hv[h].c.sat = (unsigned *)workspace;         // one record per call: every sub-batch adds to the first slice's
        hv[h].b0 = b0;
        hv[h].s = nullptr;
    }
    if (nhalf > 1) {
        HIPCHK(hipEventRecord(p->ev_fork, st));
        for (int h = 0; h < nhalf; ++h) HIPCHK(hipStreamWaitEvent(p->sub[h], p->ev_fork, 0));
    }
    const double hd = 1.0 / (double)N;
    const float h = (float)hd;
    const float bmin = p->cfg.beta_min, bdiff = (float)((double)p->cfg.beta_max - (double)p->cfg.beta_min);
    auto steps = [&]() -> int {
        for (int hh = 0; hh < nhalf; ++hh)
            if (hv[hh].c.B > 0) HIPCHK(hipMemsetAsync(tptr(hv[hh].c, p->t_ticket), 0, (size_t)hv[hh].c.B * 4, hv[hh].c.st));
        if (multi) {
            for (int hh = 0; hh < nhalf; ++hh) {
                Half &H = hv[hh];
                if (H.c.B <= 0) continue;
                H.s = tptr(H.c, p->t_s);
                ProfScope ps_(p, H.c.st, (int)p->ops.size() + XOP_SPK);
                HIPCHK(launch_spk_mlp(spk + (size_t)H.b0 * E, (const float *)(blob + p->spk_w0), (const float *)(blob + p->spk_b0),
                                      (const float *)(blob + p->spk_w2), (const float *)(blob + p->spk_b2), H.s, H.c.B, E, F, H.c.st));
            }
        }
        for (int i = step_begin; i < step_end; ++i) {
            const float t = (float)(1.0 - ((double)i + 0.5) * hd);
            const float beta = bmin + bdiff * t;                      // get_noise, fp32 like the reference tensor math
            for (int hh = 0; hh < nhalf; ++hh) {
                Half &H = hv[hh];
                if (H.c.B <= 0) continue;
                const size_t off = (size_t)H.b0 * F * T;
                H.c.tb_row = tb + (size_t)i * p->tmlp.tb_stride;
                H.c.tb_bstride = 0;
                { ProfScope ps_(p, H.c.st, (int)p->ops.size() + XOP_PREP); HIPCHK(launch_prep_input(mu + off, out + off, H.s, tptr(H.c, p->t_x0), H.c.B, F, T, p->cin0, H.c.st, p->cfg.precision == GTTS_PREC_BF16_STORE)); }
                const int rc2 = run_ops(H.c);
                if (rc2) return rc2;
                const float *nz = noise ? noise + (size_t)(i - step_begin) * B * F * T + off : nullptr;
                { ProfScope ps_(p, H.c.st, (int)p->ops.size() + XOP_FINAL); HIPCHK(launch_final_euler(tptr(H.c, p->t_final_raw), tptr(H.c, p->t_final_sc), tptr(H.c, p->t_final_sh),
                                          (const float *)(blob + p->fw_off), (const float *)(blob + p->fb_off), H.c.mask, H.c.B, p->cfg.dim, F, T,
                                          nullptr, out + off, mu + off, nz, beta, h, H.c.st, nullptr, p->cfg.precision == GTTS_PREC_BF16_STORE)); }
            }
        }
        return GTTS_OK;
    };
    rc = steps();
    // join the side streams on EVERY exit path after the fork: the caller may free z / mu / out / workspace as soon as
    // its stream has drained, so work already enqueued on the side streams must be ordered before that point
    if (nhalf > 1) {
        for (int hh = 0; hh < nhalf; ++hh) {
            if (hipEventRecord(p->ev_join[hh], p->sub[hh]) != hipSuccess || hipStreamWaitEvent(st, p->ev_join[hh], 0) != hipSuccess) {
                (void)hipStreamSynchronize(p->sub[hh]);       // last resort: never leave un-joined work behind
                if (rc == GTTS_OK) rc = fail(GTTS_E_HIP, "joining sub-batch stream %d failed", hh);
            }
        }
    }
    return rc;
}


extern "C" int gtts_plan_set_graph(gtts_plan *plan, int on) {
    if (!plan) return fail(GTTS_E_NULL, "null plan");
    std::lock_guard<std::mutex> lk(plan->mu);
    plan->graph_on = on != 0;
    if (!on) {
        for (auto &g : plan->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
        plan->graphs.clear();
    }
    return GTTS_OK;
}

extern "C" int gtts_reverse_diffusion(gtts_plan *plan, const void *packed, const float *z, const float *mask,
                                      const float *mu, const float *spk, const float *noise, float *out, void *workspace,
                                      size_t workspace_bytes, int B, int T, int n_timesteps, int step_begin, int step_end,
                                      gtts_stream_t stream) {
    int rc = check_shape(plan, B, T);
    if (rc) return rc;
    if (!packed || !z || !mask || !mu || !out || !workspace) return fail(GTTS_E_NULL, "gtts_reverse_diffusion: null argument");
    if (n_timesteps <= 0 || n_timesteps > 4096) return fail(GTTS_E_SHAPE, "n_timesteps must be in [1, 4096], got %d", n_timesteps);
    if (step_begin < 0 || step_end > n_timesteps || step_begin >= step_end)
        return fail(GTTS_E_SHAPE, "step range [%d, %d) is not inside [0, %d)", step_begin, step_end, n_timesteps);
    if (plan->cfg.arch != 0) return fail(GTTS_E_CONFIG, "gtts_reverse_diffusion needs a Grad-TTS plan (arch 0); use gtts_vc_reverse_diffusion");
    gtts_plan *p = plan;
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->cfg.n_spks > 1 && !spk) return fail(GTTS_E_NULL, "multi-speaker plan needs spk");
    hipStream_t st = (hipStream_t)stream;
    if (!p->graph_on || p->prof_on)
        return enqueue_reverse_diffusion(p, packed, z, mask, mu, spk, noise, out, workspace, workspace_bytes, B, T, n_timesteps,
                                         step_begin, step_end, st);

    // ---- hipGraph replay (small-batch, launch-bound regime): the ~95 launches per Euler step of a call are captured
    // once per distinct argument tuple (every device pointer is baked into the kernel nodes) and replayed with one
    // hipGraphLaunch afterwards.  The graph holds pointers to caller memory only; nothing is allocated on the device.
    std::vector<unsigned long long> key = {(unsigned long long)packed, (unsigned long long)z, (unsigned long long)mask,
                                           (unsigned long long)mu, (unsigned long long)spk, (unsigned long long)noise,
                                           (unsigned long long)out, (unsigned long long)workspace, (unsigned long long)workspace_bytes,
                                           (unsigned long long)B, (unsigned long long)T, (unsigned long long)n_timesteps,
                                           (unsigned long long)step_begin, (unsigned long long)step_end, (unsigned long long)p->nsub};
    for (int h = 0; h < p->nsub; ++h) key.push_back((unsigned long long)p->sub[h]);
    for (auto &g : p->graphs)
        if (g.key == key) {
            g.used = ++p->graph_clock;
            HIPCHK(hipGraphLaunch(g.exec, st));
            return GTTS_OK;
        }
    if (!st) return fail(GTTS_E_CONFIG, "hipGraph capture needs a non-default stream (got the null stream)");
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    rc = enqueue_reverse_diffusion(p, packed, z, mask, mu, spk, noise, out, workspace, workspace_bytes, B, T, n_timesteps,
                                   step_begin, step_end, st);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(st, &graph);
    if (rc != GTTS_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ec != hipSuccess || !graph) return fail(GTTS_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ec));
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) { (void)hipGraphDestroy(graph); return fail(GTTS_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei)); }
    if (p->graphs.size() >= 8) {      // bounded cache: drop the least recently used graph
        size_t lru = 0;
        for (size_t i = 1; i < p->graphs.size(); ++i) if (p->graphs[i].used < p->graphs[lru].used) lru = i;
        (void)hipGraphExecDestroy(p->graphs[lru].exec); (void)hipGraphDestroy(p->graphs[lru].graph);
        p->graphs.erase(p->graphs.begin() + lru);
    }
    p->graphs.push_back({key, graph, exec, ++p->graph_clock});
    HIPCHK(hipGraphLaunch(exec, st));
    return GTTS_OK;
}


// ------------------------------------------------------------------------------------------------ DiffVC
static int vc_check(const gtts_plan *plan, int B, int T, int Tr) {
    int rc = check_shape(plan, B, T);
    if (rc) return rc;
    if (plan->cfg.arch != 1) return fail(GTTS_E_CONFIG, "a DiffVC plan (arch 1) is required");
    if (Tr <= 0) return fail(GTTS_E_SHAPE, "T_ref must be positive (got %d)", Tr);
    return GTTS_OK;
}

extern "C" size_t gtts_vc_workspace_bytes(const gtts_plan *plan, int B, int T, int T_ref) {
    if (vc_check(plan, B, T, T_ref) != GTTS_OK) return 0;
    return compute_layout(plan, B, T, std::max(B, 4096), T_ref).ws_bytes;
}

extern "C" int gtts_vc_estimator_forward(gtts_plan *plan, const void *packed, const float *x, const float *x_mask,
                                         const float *mean, const float *xt_ref, const float *ref_mask, const float *c,
                                         const float *t, float *out, void *workspace, size_t workspace_bytes, int B, int T,
                                         int T_ref, gtts_stream_t stream) {
    int rc = vc_check(plan, B, T, T_ref);
    if (rc) return rc;
    if (!packed || !x || !x_mask || !mean || !c || !t || !out || !workspace) return fail(GTTS_E_NULL, "gtts_vc_estimator_forward: null argument");
    gtts_plan *p = plan;
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->cfg.use_ref_t && (!xt_ref || !ref_mask)) return fail(GTTS_E_NULL, "use_ref_t plan needs xt_ref and ref_mask");
    layout_workspace(p, B, T, std::max(B, 4096), T_ref);
    if (workspace_bytes < p->ws_bytes) return fail(GTTS_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p->ws_bytes, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const unsigned char *blob = (const unsigned char *)packed;
    RunCtx cx{p, blob, (unsigned char *)workspace, x_mask, B, T, nullptr, 0, st};
    cx.ref_mask = ref_mask; cx.Tr = T_ref; cx.in_x = x; cx.in_mean = mean; cx.in_c = c;
    const int F = p->cfg.n_feats;
    cx.sat = (unsigned *)workspace;
    HIPCHK(hipMemsetAsync(workspace, 0, 16, st));
    HIPCHK(hipMemsetAsync(tptr(cx, p->t_ticket), 0, (size_t)B * 4, st));
    float *tb = tptr(cx, p->t_tb);
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_TIME); HIPCHK(launch_time_mlp(t, (const float *)(blob + p->freq_off), p->cfg.pe_scale, blob, p->tmlp, tb, B, st)); }
    cx.tb_row = tb;
    cx.tb_bstride = p->tmlp.tb_stride;
    if (p->cfg.use_ref_t)
        HIPCHK(hipMemcpyAsync(tptr(cx, p->t_xtref), xt_ref, (size_t)B * F * T_ref * sizeof(float), hipMemcpyDeviceToDevice, st));
    rc = run_ops(cx);
    if (rc) return rc;
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_FINAL); HIPCHK(launch_final_euler(tptr(cx, p->t_final_raw), tptr(cx, p->t_final_sc), tptr(cx, p->t_final_sh),
                              (const float *)(blob + p->fw_off), (const float *)(blob + p->fb_off), x_mask, B, p->cfg.dim, F, T,
                              out, nullptr, nullptr, nullptr, 0.f, 0.f, st)); }
    return GTTS_OK;
}

// host scalars of the DiffVC schedule (DiffVC/model/diffusion.py:120-149), Python-double arithmetic
static double vc_gamma(const gtts_unet_cfg &cf, double s, double t, double pw = 1.0) {
    double bi = cf.vc_beta_min + 0.5 * (cf.vc_beta_max - cf.vc_beta_min) * (t + s);
    bi *= (t - s);
    return exp(-0.5 * pw * bi);
}

extern "C" int gtts_vc_reverse_diffusion(gtts_plan *plan, const void *packed, const float *z, const float *mask,
                                         const float *mean, const float *ref, const float *ref_mask, const float *mean_ref,
                                         const float *c, const float *noise, float *out, void *workspace,
                                         size_t workspace_bytes, int B, int T, int T_ref, int n_timesteps, int mode,
                                         int step_begin, int step_end, gtts_stream_t stream) {
    int rc = vc_check(plan, B, T, T_ref);
    if (rc) return rc;
    if (n_timesteps > 0 && (step_begin < 0 || step_end > n_timesteps || step_begin >= step_end))
        return fail(GTTS_E_SHAPE, "step range [%d, %d) is not inside [0, %d)", step_begin, step_end, n_timesteps);
    if (!packed || !z || !mask || !mean || !c || !out || !workspace) return fail(GTTS_E_NULL, "gtts_vc_reverse_diffusion: null argument");
    if (n_timesteps <= 0 || n_timesteps > 4096) return fail(GTTS_E_SHAPE, "n_timesteps must be in [1, 4096], got %d", n_timesteps);
    if (mode < 0 || mode > 2) return fail(GTTS_E_CONFIG, "mode must be 0 ('pf'), 1 ('em') or 2 ('ml')");
    if (mode != 0 && !noise) return fail(GTTS_E_NULL, "'em' / 'ml' sampling needs the pre-drawn noise tensor");
    gtts_plan *p = plan;
    std::lock_guard<std::mutex> lk(p->mu);
    const gtts_unet_cfg &cf = p->cfg;
    if (cf.use_ref_t && (!ref || !ref_mask || !mean_ref)) return fail(GTTS_E_NULL, "use_ref_t plan needs ref, ref_mask and mean_ref");
    layout_workspace(p, B, T, std::max(B, 4096), T_ref);
    if (workspace_bytes < p->ws_bytes) return fail(GTTS_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p->ws_bytes, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const unsigned char *blob = (const unsigned char *)packed;
    RunCtx cx{p, blob, (unsigned char *)workspace, mask, B, T, nullptr, 0, st};
    cx.ref_mask = ref_mask; cx.Tr = T_ref; cx.in_mean = mean; cx.in_c = c; cx.in_x = out;
    const int F = cf.n_feats, N = n_timesteps;
    cx.sat = (unsigned *)workspace;
    if (step_begin == 0) HIPCHK(hipMemsetAsync(workspace, 0, 16, st));
    HIPCHK(hipMemsetAsync(tptr(cx, p->t_ticket), 0, (size_t)B * 4, st));
    // step times t_i = 1 - i*h (left endpoint, diffusion.py:170) -> fp32 `time` tensor values, all rows in one launch
    float *tb = tptr(cx, p->t_tb);
    float *tvals = tb + (size_t)std::max(B, 4096) * p->tmlp.tb_stride;
    const double hd = 1.0 / (double)N;
    hipLaunchKernelGGL(sampler_times_kernel, dim3((N + 255) / 256), dim3(256), 0, st, tvals, N, 0.0);
    HIPCHK(hipGetLastError());
    { ProfScope ps_(p, st, (int)p->ops.size() + XOP_TIME); HIPCHK(launch_time_mlp(tvals, (const float *)(blob + p->freq_off), cf.pe_scale, blob, p->tmlp, tb, N, st)); }
    if (step_begin == 0) { ProfScope ps_(p, st, (int)p->ops.size() + XOP_MULMASK); HIPCHK(launch_mul_mask(z, mask, out, B, F, T, st)); }
    for (int i = step_begin; i < step_end; ++i) {
        const double t = 1.0 - (double)i * hd;
        const double beta_t = cf.vc_beta_min + (cf.vc_beta_max - cf.vc_beta_min) * t;
        double kappa = 0.0, omega = 0.0, sigma = 0.0;
        if (mode == 2) {
            kappa = vc_gamma(cf, 0, t - hd) * (1.0 - vc_gamma(cf, t - hd, t, 2.0));
            kappa /= (vc_gamma(cf, 0, t) * beta_t * hd);
            kappa -= 1.0;
            const double cden = 1.0 - vc_gamma(cf, 0, t, 2.0);
            const double nu = vc_gamma(cf, 0, t - hd) * (1.0 - vc_gamma(cf, t - hd, t, 2.0)) / cden;
            const double mu = vc_gamma(cf, t - hd, t) * (1.0 - vc_gamma(cf, 0, t - hd, 2.0)) / cden;
            omega = nu / vc_gamma(cf, 0, t);
            omega += mu;
            omega -= (0.5 * beta_t * hd + 1.0);
            sigma = sqrt((1.0 - vc_gamma(cf, 0, t - hd, 2.0)) * (1.0 - vc_gamma(cf, t - hd, t, 2.0)) / cden);
        } else if (mode == 1) {
            sigma = sqrt(beta_t * hd);
        }
        VcStep vs;
        vs.mode = mode == 0 ? 1 : 2;
        vs.cm = (float)(0.5 * beta_t * hd + omega);
        vs.k1 = (float)(1.0 + kappa);
        vs.bh = (float)(beta_t * hd);
        vs.sigma = (float)sigma;
        if (cf.use_ref_t) {
            const double g0 = vc_gamma(cf, 0, t);
            HIPCHK(launch_xt_ref(ref, mean_ref, ref_mask, tptr(cx, p->t_xtref), (float)g0, (float)(1.0 - g0), B, F, T_ref, st));
        }
        cx.tb_row = tb + (size_t)i * p->tmlp.tb_stride;
        cx.tb_bstride = 0;
        rc = run_ops(cx);
        if (rc) return rc;
        const float *nz = (mode != 0) ? noise + (size_t)(i - step_begin) * B * F * T : nullptr;
        { ProfScope ps_(p, st, (int)p->ops.size() + XOP_FINAL); HIPCHK(launch_final_euler(tptr(cx, p->t_final_raw), tptr(cx, p->t_final_sc), tptr(cx, p->t_final_sh),
                                  (const float *)(blob + p->fw_off), (const float *)(blob + p->fb_off), mask, B, cf.dim, F, T,
                                  nullptr, out, mean, nz, 0.f, 0.f, st, &vs)); }
    }
    return GTTS_OK;
}

extern "C" size_t gtts_mas_scratch_bytes(int b, int tx, int ty) {
    if (b <= 0 || tx <= 0 || ty <= 0) return 0;
    // one byte per cell (column-sweep kernel, t_x > 1024) or 64 x u16 per column (one-wave kernel): the larger of the two
    return std::max((size_t)b * tx * ty, (size_t)b * ty * 128);
}

extern "C" int gtts_mas_maximum_path(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                                     void *scratch, int b, int tx, int ty, gtts_stream_t stream) {
    if (!value || !t_x || !t_y || !path || !scratch) return fail(GTTS_E_NULL, "gtts_mas_maximum_path: null argument");
    if (b <= 0 || tx <= 0 || ty <= 0) return fail(GTTS_E_SHAPE, "gtts_mas_maximum_path: bad shape b=%d tx=%d ty=%d", b, tx, ty);
    if ((size_t)2 * tx * 4 > 160 * 1024) return fail(GTTS_E_SHAPE, "t_x too large for the LDS column buffer (%d)", tx);
    HIPCHK(launch_mas(value, mask, t_x, t_y, path, (unsigned char *)scratch, b, tx, ty, (hipStream_t)stream));
    return GTTS_OK;
}


// CPU twin: the reference's maximum_path accepts tensors on any device and always runs its Cython kernel on the host
// (monotonic_align/__init__.py:8-23).  Same arithmetic as the GPU kernel and core.pyx:9-45: one fp32 max and one fp32
// add per cell, strict '<' in the backtrack -> bit-identical paths.  value / mask / path are HOST pointers.
extern "C" int gtts_mas_maximum_path_cpu(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                                         int b, int tx, int ty) {
    if (!value || !t_x || !t_y || !path) return fail(GTTS_E_NULL, "gtts_mas_maximum_path_cpu: null argument");
    if (b <= 0 || tx <= 0 || ty <= 0) return fail(GTTS_E_SHAPE, "gtts_mas_maximum_path_cpu: bad shape b=%d tx=%d ty=%d", b, tx, ty);
    const float NEG = -1e9f;
    std::vector<float> v((size_t)tx * ty);
    for (int i = 0; i < b; ++i) {
        const size_t base = (size_t)i * tx * ty;
        int *pp = path + base;
        memset(pp, 0, (size_t)tx * ty * sizeof(int));
        const int X = t_x[i], Y = t_y[i];
        if (X <= 0 || Y <= 0 || X > tx || Y > ty) continue;
        for (size_t k = 0; k < (size_t)tx * ty; ++k) v[k] = mask ? value[base + k] * mask[base + k] : value[base + k];
        for (int y = 0; y < Y; ++y) {
            const int lo = std::max(0, X + y - Y), hi = std::min(X, y + 1);
            for (int x = lo; x < hi; ++x) {
                const float v_cur = (x == y) ? NEG : v[(size_t)x * ty + y - 1];
                float v_prev;
                if (x == 0) v_prev = (y == 0) ? 0.f : NEG;
                else v_prev = v[(size_t)(x - 1) * ty + y - 1];
                v[(size_t)x * ty + y] = (v_cur > v_prev ? v_cur : v_prev) + v[(size_t)x * ty + y];
            }
        }
        int index = X - 1;
        for (int y = Y - 1; y >= 0; --y) {
            pp[(size_t)index * ty + y] = 1;
            if (index != 0 && (index == y || (y > 0 && v[(size_t)index * ty + y - 1] < v[(size_t)(index - 1) * ty + y - 1]))) index -= 1;
        }
    }
    return GTTS_OK;
}

// The one collective of the multi-GPU path: broadcast of the packed weight blob from `root` over RCCL.
// `comm` is the caller's ncclComm_t.  RCCL is resolved at call time from the process (the copy PyTorch-ROCm already
// loaded) or, failing that, from librccl.so -- the library itself carries no link-time dependency on it.
extern "C" int gtts_bcast_weights(void *packed, size_t bytes, int root, void *comm, gtts_stream_t stream) {
    if (!packed || !comm) return fail(GTTS_E_NULL, "gtts_bcast_weights: null argument");
    typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static bcast_fn fn = nullptr;
    if (!fn) {
        void *sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");
        if (!sym) {
            void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h) sym = dlsym(h, "ncclBroadcast");
        }
        if (!sym) return fail(GTTS_E_HIP, "gtts_bcast_weights: RCCL (ncclBroadcast) is not available in this process");
        fn = (bcast_fn)sym;
    }
    const int rc = fn(packed, packed, bytes, /*ncclChar*/ 0, root, comm, (hipStream_t)stream);
    if (rc != 0) return fail(GTTS_E_HIP, "ncclBroadcast failed with code %d", rc);
    return GTTS_OK;
}

extern "C" int gtts_expand_alignment(const float *duration, const float *x_mask, const int *y_lengths, const float *mu_x,
                                     const float *noise, float temperature, float *attn, float *mu_y, float *z, int B, int F,
                                     int t_x, int T, gtts_stream_t stream) {
    if (!duration || !x_mask || !y_lengths || !mu_x || !attn || !mu_y) return fail(GTTS_E_NULL, "gtts_expand_alignment: null argument");
    if (B <= 0 || F <= 0 || t_x <= 0 || T <= 0) return fail(GTTS_E_SHAPE, "gtts_expand_alignment: bad shape B=%d F=%d t_x=%d T=%d", B, F, t_x, T);
    if (noise && !(temperature > 0.f)) return fail(GTTS_E_SHAPE, "gtts_expand_alignment: temperature must be positive");
    if (((size_t)t_x + T) * 4 > 160 * 1024) return fail(GTTS_E_SHAPE, "t_x + T too large for the LDS tables (%d + %d)", t_x, T);
    HIPCHK(launch_expand_alignment(duration, x_mask, y_lengths, mu_x, noise, temperature, attn, mu_y, z, B, F, t_x, T,
                                   (hipStream_t)stream));
    return GTTS_OK;
}


extern "C" int gtts_log_prior(const float *mu_x, const float *y, float *log_prior, int B, int F, int t_x, int T,
                              gtts_stream_t stream) {
    if (!mu_x || !y || !log_prior) return fail(GTTS_E_NULL, "gtts_log_prior: null argument");
    if (B <= 0 || F <= 0 || t_x <= 0 || T <= 0) return fail(GTTS_E_SHAPE, "gtts_log_prior: bad shape B=%d F=%d t_x=%d T=%d", B, F, t_x, T);
    HIPCHK(launch_log_prior(mu_x, y, log_prior, B, F, t_x, T, (hipStream_t)stream));
    return GTTS_OK;
}

// ------------------------------------------------------------------------------------------------ measurement
// the template instance launch_conv picks (conv_mfma.hip: launch_prec / launch_cfg), as rocprofv3 prints it
static std::string conv_kernel_name(int mode, int cin, int cout, int pro, int epi, int nsplit, bool abf, bool small, bool ws, int B, int Ho,
                                    int Wo, int groups, bool f8 = false, bool up_f8 = false) {
    const bool wide = cout > 64;
    if (ws) {      // conv_ws.hip (launch_ws_pro)
        char wb[128];
        const bool sm = conv_ws_small(cout, groups, Ho, Wo, B, f8 ? 1 : 0);
        const int wm = sm ? 1 : (cout % 128 == 0 ? 2 : 1), wn = sm ? 1 : 2;       // (the f16 + fp8 64-channel tile: <1, 2, 2>)
        snprintf(wb, sizeof wb, "gtts::conv3x3_ws_kernel<%d, %d, %d, 5, %d, %d, %s, %d>", wm, wn, sm ? 1 : 2, pro, f8 ? 3 : nsplit,
                 abf ? "__bf16" : "float", f8 ? 2 : 3);
        return wb;
    }
    if (mode == CONV_UP && nsplit == 2 && !abf && cin % 16 == 0 && pro == PRO_MASK && epi == EPI_PLAIN)      // conv_up.hip
        return (up_f8 && conv_up4_f16f8_ok(cin, cout)) ? (conv_up4_ws_ok(cin, cout) ? conv_up4_ws_name() : conv_up4_f8_name()) : "gtts::conv_up4_kernel";
    const int kch = conv_geom(mode, cin, cout, f8 ? 1 : 0).kch;
    if (f8) nsplit = 3;
    const bool fullc = cin % 16 == 0;
    int wm, wn, mf;
    if (mode == CONV_DN) { wm = 2; wn = 2; mf = wide ? 2 : 1; }
    else if (wide) { wm = 2; wn = 2; mf = 2; }
    else { wm = 1; wn = 4; mf = 2; }
    if (small && mode == CONV_C3 && fullc && !abf && nsplit > 1 && pro != PRO_IGLU) {      // half-height tiles (conv_small_tiles)
        mf = 1;
        if (wide) { wm = 4; wn = 1; } else { wm = 2; wn = 2; }
    }
    char buf[128];
    snprintf(buf, sizeof buf, "gtts::conv_mfma_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d, %s, 2, 0>", mode, wm, wn, mf, kch, pro, epi, nsplit,
             fullc ? 1 : 0, abf ? "__bf16" : "float");
    return buf;
}

extern "C" int gtts_plan_num_ops(const gtts_plan *plan) { return plan ? (int)plan->ops.size() + XOP_COUNT : 0; }

extern "C" int gtts_plan_op_info(const gtts_plan *plan, int i, int B, int T, const char **label, const char **kernel,
                                 double *flops, double *bytes) {
    int rc = check_shape(plan, B, T);
    if (rc) return rc;
    static thread_local std::string s_label, s_kernel;
    const int n = (int)plan->ops.size();
    if (i < 0 || i >= n + XOP_COUNT) return fail(GTTS_E_SHAPE, "op index out of range");
    const double F = plan->cfg.n_feats;
    const double ab = plan->cfg.precision == GTTS_PREC_BF16_STORE ? 2.0 : 4.0;      // bytes per stored activation
    const bool abf = plan->cfg.precision == GTTS_PREC_BF16_STORE;
    double fl = 0, by = 0;
    if (i >= n) {
        const double FT = F * T * B;
        switch (i - n) {
            case XOP_PREP: s_label = "prep_input"; s_kernel = abf ? "gtts::prep_input_kernel<__bf16>" : "gtts::prep_input_kernel<float>"; by = (4.0 + ab) * FT * plan->cin0; break;
            case XOP_TIME: s_label = "time_mlp"; s_kernel = "gtts::time_mlp_kernel"; break;
            case XOP_FINAL: s_label = "final_conv+euler"; s_kernel = abf ? (T % 4 == 0 ? "gtts::final_euler_kernel<__bf16, 4>" : "gtts::final_euler_kernel<__bf16, 1>") : "gtts::final_euler_kernel<float, 1>";
                fl = 2.0 * plan->cfg.dim * FT; by = FT * (ab * plan->cfg.dim + 16.0); break;
            case XOP_MULMASK: s_label = "xt=z*mask"; s_kernel = "gtts::mul_mask_kernel"; by = 8.0 * FT; break;
            case XOP_SPK: s_label = "spk_mlp"; s_kernel = "gtts::spk_mlp_kernel"; break;
        }
    } else {
        const Op &o = plan->ops[i];
        s_label = o.label;
        const double Hi = (int)F >> o.lvl_in, Wi = T >> o.lvl_in, Ho = (int)F >> o.lvl_out, Wo = T >> o.lvl_out;
        switch (o.kind) {
            case OP_CONV: {
                const double cin = o.c0 + o.c1;
                const double taps = o.mode == CONV_P1 ? 1 : (o.mode == CONV_UP ? 4 : 9);
                fl = 2.0 * B * o.cout * cin * taps * Ho * Wo;
                by = ab * B * (cin * Hi * Wi + o.cout * Ho * Wo);
                if (o.epi == EPI_TAIL || o.epi == EPI_ATTN) by += ab * B * o.cout * Ho * Wo;
                s_kernel = conv_kernel_name(o.mode, o.c0 + o.c1, o.cout, o.pro, o.epi, plan_nsplit(plan),
                                            plan->cfg.precision == GTTS_PREC_BF16_STORE, conv_small_tiles(o.mode, o.cout, Ho, Wo, B),
                                            plan->cfg.conv_ws && conv_ws_eligible(o.mode, o.c0, o.c1, o.cout, o.pro, o.epi, plan_nsplit(plan), plan->cfg.precision == GTTS_PREC_F16F8),
                                            B, (int)Ho, (int)Wo, plan->cfg.groups,
                                            plan->cfg.precision == GTTS_PREC_F16F8 && conv_f16f8_ok(o.mode, o.c0, o.c1, o.cout, o.pro, o.epi, plan->cfg.conv_ws),
                                            plan->cfg.precision == GTTS_PREC_F16F8);
                break;
            }
            case OP_GNFIN: s_kernel = "gtts::gn_finalize_kernel"; break;
            case OP_TAILID: {
                // the instance launch_tail_identity picks (misc.hip): vector width by storage type / width, 4 items per thread on big planes
                const int vec = (abf && (int)Wi % 8 == 0) ? 8 : ((int)Wi % 4 == 0 ? 4 : 1);
                const int items = (vec > 1 && ((int)(Hi * Wi) / vec + 255) / 256 >= 8) ? 4 : 1;
                char nb[96];
                snprintf(nb, sizeof nb, "gtts::tail_identity_kernel<%d, %s, %d>", vec, abf ? "__bf16" : "float", items);
                s_kernel = o.fused ? "(fused into the attention context pass)" : nb;
                by = o.fused ? 0.0 : ab * B * o.C * Hi * Wi * 3;
                break;
            }
            case OP_ACTX: {
                char nb[96];
                if (attn_head_per_wave(o.C) && o.fused)
                    snprintf(nb, sizeof nb, "gtts::attn_ctx64_kernel<%d, float, 1>", plan_nsplit(plan));
                else if (attn_head_per_wave(o.C))
                    snprintf(nb, sizeof nb, "gtts::attn_ctx64_kernel<%d, %s, 0>", plan_nsplit(plan), abf ? "__bf16" : "float");
                else
                    snprintf(nb, sizeof nb, "gtts::attn_ctx_kernel<%d, %d, %s, %d, %d>", plan_nsplit(plan), o.C % 32 == 0 ? 1 : 0,
                             abf ? "__bf16" : "float", GTTS_ATTN_HPW, o.fused ? 1 : 0);
                s_kernel = nb;
                fl = 2.0 * B * Hi * Wi * (256.0 * o.C + 128.0 * 32);
                by = ab * B * o.C * Hi * Wi * (o.fused ? 3 : 1); break;      // (fused tail: block input + raw convolution output in, block output out)
            }
            case OP_AMERGE: s_kernel = "gtts::attn_merge_kernel"; break;
            case OP_INSTATS: s_kernel = "gtts::instnorm_stats_kernel"; by = 4.0 * B * o.C * Hi * Wi; break;
            case OP_REFPOOL: s_kernel = "gtts::ref_pool_kernel"; by = 8.0 * B * o.C * Hi * Wi; break;
            case OP_VCCOND: s_kernel = "gtts::vc_cond_kernel"; break;
            case OP_PREPVC: s_kernel = "gtts::prep_vc_kernel"; by = 4.0 * B * plan->cin0 * Hi * Wi; break;
            case OP_AFOLD: s_kernel = "gtts::attn_fold_kernel"; fl = 2.0 * B * (o.C * 128.0 * 32 + (double)o.C * o.C * 128); break;
        }
    }
    if (label) *label = s_label.c_str();
    if (kernel) *kernel = s_kernel.c_str();
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return GTTS_OK;
}

extern "C" int gtts_profile_enable(gtts_plan *plan, int on) {
    if (!plan) return fail(GTTS_E_NULL, "null plan");
    plan->prof_on = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return GTTS_OK;
}

extern "C" int gtts_profile_timeline(gtts_plan *plan, int cap, int *op, int *stream, double *t0_ms, double *t1_ms, int *n) {
    if (!plan || !op || !stream || !t0_ms || !t1_ms || !n) return fail(GTTS_E_NULL, "gtts_profile_timeline: null argument");
    *n = 0;
    if (plan->prof.empty()) return GTTS_OK;
    const hipEvent_t base = plan->prof[0].a;
    for (auto &r : plan->prof) {
        HIPCHK(hipEventSynchronize(r.b));
        float a = 0.f, b = 0.f;
        HIPCHK(hipEventElapsedTime(&a, base, r.a));
        HIPCHK(hipEventElapsedTime(&b, base, r.b));
        if (*n < cap) { op[*n] = r.op; stream[*n] = r.stream; t0_ms[*n] = a; t1_ms[*n] = b; ++*n; }
        plan->prof_pool.push_back({r.a, r.b});
    }
    plan->prof.clear();
    return GTTS_OK;
}

extern "C" int gtts_profile_collect(gtts_plan *plan, double *ms_per_op, long long *launches_per_op) {
    if (!plan || !ms_per_op || !launches_per_op) return fail(GTTS_E_NULL, "gtts_profile_collect: null argument");
    const int n = (int)plan->ops.size() + XOP_COUNT;
    for (auto &r : plan->prof) {
        HIPCHK(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.op >= 0 && r.op < n) { ms_per_op[r.op] += ms; launches_per_op[r.op] += 1; }
        plan->prof_pool.push_back({r.a, r.b});
    }
    plan->prof.clear();
    return GTTS_OK;
}
