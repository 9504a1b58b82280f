// postnet.hip -- DiffVC PostNet (DiffVC/model/postnet.py:15-53), the last stage of the "average voice" encoder
// (DiffVC/model/vc.py:33-41) that produces the decoder's prior mean:
//   x [B,80,T] -> init_conv 1x1 (1 -> dim) on x*mask -> ResnetBlock: Block = Conv2d 7x7 (pad 3) on x*mask -> GroupNorm(8)
//   -> Mish -> *mask, twice; + res 1x1 (x*mask) -> final_conv 1x1 (dim -> 1) on x*mask.
// The two 7x7 convolutions (131 GFLOP each per 80x1024 utterance at dim 128 -- as much as one score-network call) run on the
// MFMA convolution of conv_mfma.hip in its CONV_C7 mode (7 weight stages of 7 taps, halo 3; split-bf16, GroupNorm partial
// sums and fused finalize in the epilogue, GroupNorm + Mish + mask applied on load by the second convolution); the residual
// 1x1 + tail is the decoder's EPI_TAIL kernel; the two single-channel 1x1 convolutions are bandwidth kernels.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

// out[b][c][f][t] = w[c] * (x[b][f][t] * mask[b][t]) + bias[c]          (init_conv, postnet.py:43,49-51)
__global__ void postnet_init_kernel(const float *__restrict__ x, const float *__restrict__ mask, const float *__restrict__ w,
                                    const float *__restrict__ bias, float *__restrict__ out, int C, int F, int T) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F * T) return;
    const float v = x[(size_t)b * F * T + i] * mask[(size_t)b * T + i % T];
    out[((size_t)b * C + c) * F * T + i] = fmaf(w[c], v, bias[c]);
}

// out[b][f][t] = sum_c w[c] * (x[b][c][f][t] * mask[b][t]) + bias      (final_conv, postnet.py:45,53)
__global__ void postnet_final_kernel(const float *__restrict__ x, const float *__restrict__ mask, const float *__restrict__ w,
                                     const float *__restrict__ bias, float *__restrict__ out, int C, int F, int T) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < C; i += 256) sw[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int FT = F * T;
    if (i >= FT) return;
    const float m = mask[(size_t)b * T + i % T];
    const float *p = x + (size_t)b * C * FT + i;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(sw[c], p[(size_t)c * FT] * m, acc);
    out[(size_t)b * FT + i] = acc + bias[0];
}

}  // namespace gtts

using namespace gtts;

static int pfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define PCHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return pfail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static size_t palign(size_t x) { return (x + 255) / 256 * 256; }

struct PnParam { const char *name; int rank; int dims[4]; int kind; size_t off; };   // kind 0 fp32, 1 conv 7x7, 2 conv 1x1 (MFMA)
struct gtts_postnet {
    int dim, n_feats, groups;
    std::vector<PnParam> params;
    size_t blob_bytes = 0;
    size_t off(const char *n) const {
        for (const PnParam &p : params) if (!strcmp(p.name, n)) return p.off;
        return 0;
    }
};

extern "C" int gtts_postnet_create(int dim, int n_feats, int groups, gtts_postnet **out) {
    if (!out) return pfail(GTTS_E_NULL, "gtts_postnet_create: null argument");
    if (dim <= 0 || dim % 64 != 0 || groups != 8 || n_feats <= 0) return pfail(GTTS_E_CONFIG, "PostNet needs dim %% 64 == 0 and 8 groups (dim %d, groups %d)", dim, groups);
    gtts_postnet *p = new gtts_postnet();
    p->dim = dim; p->n_feats = n_feats; p->groups = groups;
    auto add = [&](const char *name, std::vector<int> d, int kind) {
        PnParam q;
        q.name = name; q.rank = (int)d.size(); q.kind = kind;
        for (int i = 0; i < 4; ++i) q.dims[i] = i < (int)d.size() ? d[i] : 1;
        size_t n = 1;
        for (int v : d) n *= (size_t)v;
        q.off = p->blob_bytes;
        const size_t bytes = kind == 1 ? conv_packed_bytes(CONV_C7, dim, dim) : (kind == 2 ? conv_packed_bytes(CONV_P1, dim, dim) : n * 4);
        p->blob_bytes = palign(p->blob_bytes + bytes);
        p->params.push_back(q);
    };
    // registration order of PostNet (postnet.py:42-45): init_conv, res_block (block1, block2, res), final_conv
    add("init_conv.weight", {dim, 1, 1, 1}, 0);
    add("init_conv.bias", {dim}, 0);
    add("res_block.block1.block.0.weight", {dim, dim, 7, 7}, 1);
    add("res_block.block1.block.0.bias", {dim}, 0);
    add("res_block.block1.block.1.weight", {dim}, 0);
    add("res_block.block1.block.1.bias", {dim}, 0);
    add("res_block.block2.block.0.weight", {dim, dim, 7, 7}, 1);
    add("res_block.block2.block.0.bias", {dim}, 0);
    add("res_block.block2.block.1.weight", {dim}, 0);
    add("res_block.block2.block.1.bias", {dim}, 0);
    add("res_block.res.weight", {dim, dim, 1, 1}, 2);
    add("res_block.res.bias", {dim}, 0);
    add("final_conv.weight", {1, dim, 1, 1}, 0);
    add("final_conv.bias", {1}, 0);
    *out = p;
    return GTTS_OK;
}
extern "C" void gtts_postnet_destroy(gtts_postnet *p) { delete p; }
extern "C" int gtts_postnet_num_params(const gtts_postnet *p) { return p ? (int)p->params.size() : 0; }
extern "C" int gtts_postnet_param_info(const gtts_postnet *p, int i, const char **name, int *rank, int dims[4]) {
    if (!p) return pfail(GTTS_E_NULL, "null postnet");
    if (i < 0 || i >= (int)p->params.size()) return pfail(GTTS_E_SHAPE, "parameter index out of range");
    if (name) *name = p->params[i].name;
    if (rank) *rank = p->params[i].rank;
    if (dims) for (int k = 0; k < 4; ++k) dims[k] = p->params[i].dims[k];
    return GTTS_OK;
}
extern "C" size_t gtts_postnet_packed_bytes(const gtts_postnet *p) { return p ? p->blob_bytes : 0; }
extern "C" int gtts_postnet_pack(const gtts_postnet *p, const void *const *ptrs, int n, void *packed, gtts_stream_t stream) {
    if (!p || !ptrs || !packed) return pfail(GTTS_E_NULL, "gtts_postnet_pack: null argument");
    if (n != (int)p->params.size()) return pfail(GTTS_E_PARAMS, "expected %d parameters, got %d", (int)p->params.size(), n);
    hipStream_t st = (hipStream_t)stream;
    unsigned char *blob = (unsigned char *)packed;
    PCHK(hipMemsetAsync(blob, 0, p->blob_bytes, st));
    for (int i = 0; i < n; ++i) {
        const PnParam &q = p->params[i];
        if (!ptrs[i]) return pfail(GTTS_E_NULL, "parameter %s is null", q.name);
        if (q.kind == 1) PCHK(launch_pack_conv(CONV_C7, (const float *)ptrs[i], blob + q.off, p->dim, p->dim, st));
        else if (q.kind == 2) PCHK(launch_pack_conv(CONV_P1, (const float *)ptrs[i], blob + q.off, p->dim, p->dim, st));
        else {
            size_t cnt = 1;
            for (int k = 0; k < q.rank; ++k) cnt *= (size_t)q.dims[k];
            PCHK(hipMemcpyAsync(blob + q.off, ptrs[i], cnt * 4, hipMemcpyDeviceToDevice, st));
        }
    }
    return GTTS_OK;
}

// workspace: X0, raw1, raw2, R (each [B,dim,F,T] fp32), partials, scale/shift x2, tickets, a zero time-bias row
static size_t pn_ws(const gtts_postnet *p, int B, int T, size_t off[10]) {
    const size_t act = palign((size_t)B * p->dim * p->n_feats * T * 4);
    const size_t part = palign((size_t)B * conv_nparts(CONV_C7, p->dim, p->n_feats, T) * p->groups * 2 * 4);
    const size_t perb = palign((size_t)B * p->dim * 4);
    size_t o = 0;
    for (int i = 0; i < 4; ++i) { off[i] = o; o += act; }
    off[4] = o; o += part;
    for (int i = 5; i < 9; ++i) { off[i] = o; o += perb; }      // sc1, sh1, sc2, sh2
    off[9] = o; o += palign((size_t)B * 4) + perb;              // tickets, then zeros (time bias)
    return o;
}
extern "C" size_t gtts_postnet_workspace_bytes(const gtts_postnet *p, int B, int T) {
    if (!p || B <= 0 || T <= 0) return 0;
    size_t off[10];
    return pn_ws(p, B, T, off);
}

// PostNet.forward (postnet.py:47-53): x [B,n_feats,T], mask [B,T] -> out [B,n_feats,T]
extern "C" int gtts_postnet_forward(const gtts_postnet *p, const void *packed, const float *x, const float *mask, float *out,
                                    void *workspace, size_t workspace_bytes, int B, int T, gtts_stream_t stream) {
    if (!p || !packed || !x || !mask || !out || !workspace) return pfail(GTTS_E_NULL, "gtts_postnet_forward: null argument");
    if (B <= 0 || T <= 0) return pfail(GTTS_E_SHAPE, "gtts_postnet_forward: bad shape B=%d T=%d", B, T);
    size_t off[10];
    if (workspace_bytes < pn_ws(p, B, T, off)) return pfail(GTTS_E_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const unsigned char *blob = (const unsigned char *)packed;
    unsigned char *ws = (unsigned char *)workspace;
    const int C = p->dim, F = p->n_feats;
    float *X0 = (float *)(ws + off[0]), *raw1 = (float *)(ws + off[1]), *raw2 = (float *)(ws + off[2]), *R = (float *)(ws + off[3]);
    float *part = (float *)(ws + off[4]);
    float *sc1 = (float *)(ws + off[5]), *sh1 = (float *)(ws + off[6]), *sc2 = (float *)(ws + off[7]), *sh2 = (float *)(ws + off[8]);
    unsigned *ticket = (unsigned *)(ws + off[9]);
    float *zeros = (float *)(ws + off[9] + palign((size_t)B * 4));
    PCHK(hipMemsetAsync(ticket, 0, palign((size_t)B * 4) + palign((size_t)B * C * 4), st));
    hipLaunchKernelGGL(postnet_init_kernel, dim3((F * T + 255) / 256, C, B), dim3(256), 0, st, x, mask,
                       (const float *)(blob + p->off("init_conv.weight")), (const float *)(blob + p->off("init_conv.bias")), X0, C, F, T);
    PCHK(hipGetLastError());
    auto block = [&](const float *src, int pro, const float *psc, const float *psh, const char *wn, const char *bn, const char *gn,
                     const char *be, float *raw, float *sc, float *sh) -> hipError_t {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.src0 = src; a.src1 = src; a.c0 = C; a.c1 = 0; a.cin = C;
        a.B = B; a.Hin = a.Hout = F; a.Win = a.Wout = T;
        a.mask = mask; a.T = T; a.lvl_in = a.lvl_out = 0;
        a.pro = pro; a.sc = psc; a.sh = psh; a.tb = zeros; a.tb_stride = C;
        a.w = blob + p->off(wn); a.bias = (const float *)(blob + p->off(bn));
        a.cout = C; a.epi = EPI_STATS; a.out = raw; a.partials = part;
        a.nparts = conv_nparts(CONV_C7, C, F, T); a.groups = p->groups; a.nsplit = 2;
        a.ticket = ticket; a.gn_gamma = (const float *)(blob + p->off(gn)); a.gn_beta = (const float *)(blob + p->off(be));
        a.gn_sc = sc; a.gn_sh = sh; a.gn_count = (float)((double)(C / p->groups) * (double)F * (double)T);
        return launch_conv(CONV_C7, a, st);
    };
    // Block 1: Conv7x7(x * mask) -> GroupNorm statistics           (postnet.py:21-23)
    PCHK(block(X0, PRO_MASK, nullptr, nullptr, "res_block.block1.block.0.weight", "res_block.block1.block.0.bias",
               "res_block.block1.block.1.weight", "res_block.block1.block.1.bias", raw1, sc1, sh1));
    // Block 2 on Mish(GN(raw1)) * mask (applied on load; no time bias here)
    PCHK(block(raw1, PRO_GN, sc1, sh1, "res_block.block2.block.0.weight", "res_block.block2.block.0.bias",
               "res_block.block2.block.1.weight", "res_block.block2.block.1.bias", raw2, sc2, sh2));
    // res(x * mask) + Mish(GN(raw2)) * mask                         (postnet.py:33-37)
    {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.src0 = X0; a.src1 = X0; a.c0 = C; a.c1 = 0; a.cin = C;
        a.B = B; a.Hin = a.Hout = F; a.Win = a.Wout = T;
        a.mask = mask; a.T = T; a.lvl_in = a.lvl_out = 0;
        a.pro = PRO_MASK; a.epi = EPI_TAIL;
        a.w = blob + p->off("res_block.res.weight"); a.bias = (const float *)(blob + p->off("res_block.res.bias"));
        a.cout = C; a.out = R; a.eh = raw2; a.esc = sc2; a.esh = sh2; a.groups = p->groups; a.nsplit = 2;
        PCHK(launch_conv(CONV_P1, a, st));
    }
    hipLaunchKernelGGL(postnet_final_kernel, dim3((F * T + 255) / 256, B), dim3(256), (size_t)C * 4, st, R, mask,
                       (const float *)(blob + p->off("final_conv.weight")), (const float *)(blob + p->off("final_conv.bias")), out, C, F, T);
    PCHK(hipGetLastError());
    return GTTS_OK;
}
