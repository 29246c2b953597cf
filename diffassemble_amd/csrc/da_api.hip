// C-ABI entry points of libdiffassemble_hip.so (include/diffassemble_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

#include "da_common.h"
#include "da_internal.h"
#include "da_config.h"

namespace da {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ConvW {
    void *w = nullptr;     // [4*HC, Din] act dtype, rows = Q | K | V | skip
    float *b = nullptr;    // [4*HC] fp32
    // the same projection for the matrix-core attention kernels (da_attn_dense.hip / da_attn_dual.hip): the Q rows and
    // biases are PRE-SCALED by log2(e) / sqrt(C), so that the kernels' scores arrive in log2 units and exp2 applies to them
    // directly -- the softmax scale costs no instruction per score (scaled in fp32, before the rounding to the act dtype)
    void *wd = nullptr;
    float *bd = nullptr;
    void *wdp = nullptr;   // wd in the fragment order of the row-panel projection kernel (da_gemm_xpanel.hip), bf16 only
    void *wqs = nullptr;   // wd as per-head MFMA fragments (pack_w_qs): hidden layers whose attention kernel does the projection itself
    int din = 0, hc = 0, C = 0;
};

// p[r][c] *= s for r < rows, c < cols (row pitch ld): scales the Q block of a packed projection in fp32
__global__ void k_scale_block(float *p, int rows, int cols, int ld, float s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * cols) p[(i / cols) * ld + (i % cols)] *= s;
}
static int scale_block(float *p, int rows, int cols, int ld, float s, hipStream_t st) {
    const size_t n = (size_t)rows * cols;
    if (!n) return 0;
    k_scale_block<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, rows, cols, ld, s);
    DA_LAUNCH_CHECK();
    return 0;
}
// dst[r][c] = bias[c] in the activation dtype (the feature share of mlp.0 for zero features)
__global__ void k_fill_bias_rows(int prec, int rows, int cols, const float *bias, void *dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    const float v = bias[i % cols];
    if (prec == DA_PREC_BF16) ((bf16_t *)dst)[i] = f2bf(v); else ((float *)dst)[i] = v;
}
// classifier-free guidance, spatial_diffusion.py:585-589: out = (1 + w) cond - w unc, in place on `cond`
__global__ void k_cfg_combine(size_t n, float wgt, float *cond, const float *unc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cond[i] = (1.0f + wgt) * cond[i] - wgt * unc[i];
}
static bool q_prescale_on() { return !(cfg().disable_folds & DA_FOLD_QSCALE); }
static float q_scale_log2(int C) { return 1.4426950408889634f / sqrtf((float)C); }

struct LoopKey {
    da_graph g;
    da_schedule s;
    int mean_type, ratio, max_iters;
    const float *x_init;
    float *traj, *x_final;
    void *ws;
    size_t ws_bytes;
    da_loop_opts opts;        // zero for the plain DDIM loop
    size_t traj_stride;       // elements between consecutive iterations of `traj` (0 = n_real * c)
    size_t noise_stride;      // the same for opts.noise (two-branch loops read row ranges of one [n_iters, N, c] buffer)
    da_config cfg;            // the switches the graph was recorded under (da_config_set between two calls records a new graph)
};

}  // namespace da

using namespace da;

struct da_denoiser {
    int prec = 0, variant = 0, arch = 0, steps = 0, c_in = 0, c_out = 0, F = 0, D = 0, hidden = 0, heads = 0,
        n_layers = 0, V = 0, head_hidden = 0;
    float *time_emb = nullptr, *pos_w0 = nullptr, *pos_b0 = nullptr, *pos_w1 = nullptr, *pos_b1 = nullptr;
    void *mlp_w0 = nullptr, *mlp_w1 = nullptr;
    float *mlp_b0 = nullptr, *mlp_b1 = nullptr;
    ConvW conv[DA_MAX_LAYERS];
    void *virt_emb = nullptr;
    void *head_w0 = nullptr;
    float *head_b0 = nullptr;
    float *head_w1 = nullptr, *head_b1 = nullptr, *head_r_w1 = nullptr, *head_r_b1 = nullptr;
    // 2D transformer arch: mlp.2 has no activation, so it is composed into its two consumers once per
    // checkpoint (see da_denoiser_create): conv-0 projection over the 128-wide hidden layer, and the residual's
    // share of final_mlp.0 as a pre-activation addend -- `combined` [N, 1152] is never materialised
    bool fused_mlp2 = false;
    void *conv0c_w = nullptr;         // [4*HC0, hidden] act dtype = Wcat0 . W2
    float *conv0c_b = nullptr;        // [4*HC0] = Wcat0 . b2 + bcat0
    void *headc_w = nullptr;          // [32, hidden] act dtype = Wf0 . W2
    float *headc_b = nullptr;         // [32] = Wf0 . b2
    void *virt_qkvs = nullptr;        // exophormer: [V, 4*HC0] act dtype = virt_emb . Wcat0^T + bcat0 (constant per checkpoint)
    void *conv0c_wd = nullptr, *virt_qkvs_d = nullptr;      // their dense-path variants (Q pre-scaled, see ConvW)
    void *conv0c_wdp = nullptr, *convLf_wp = nullptr;       // conv0c_wd / convLf_w packed for the row-panel kernel (see ConvW::wdp)
    void *conv0c_wqs = nullptr;                             // conv0c_wd's per-head fragments (see ConvW::wqs)
    float *conv0c_bd = nullptr;
    bool q_prescaled = false;
    // ... and the LAST conv's value / skip projections are folded with final_mlp.0 (its consumer, linear up to
    // the GELU): softmax(QK^T)(V Wf_h^T) == (softmax(QK^T) V) Wf_h^T per head, so the last attention runs with
    // 32-wide value heads and the [N, 1152] tensor z is never formed either (dense path only)
    bool lastfold = false;
    bool dense_only = false;     // every layer can take the block-diagonal MFMA attention: complete graphs never touch the CSR arrays
    void *convLf_w = nullptr;         // [2*HC + H*32, 256] act dtype: Wq | Wk | (Wf_h Wv_h)_h
    float *convLf_b = nullptr;        // [2*HC + H*32]
    void *skipc_w = nullptr;          // [32, 256] act dtype = Wf0 . Ws_last
    float *skipc_b = nullptr;         // [32] = Wf0 . bs_last + bf0
    std::vector<void *> owned;
    // optional per-kernel-class timing with HIP events (da_profile_*)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;      // pairs (start, stop)
    std::vector<int> prof_cls;
    size_t prof_used = 0;
    // cached sampling-loop graphs (a few distinct loops, e.g. a full loop and a remainder)
    struct LoopEntry { LoopKey key; hipGraphExec_t exec; };
    std::vector<LoopEntry> loops;
    hipStream_t cap_stream = nullptr;     // private stream used only to RECORD graphs (the caller's
                                          // stream may be the legacy null stream, which cannot capture)
    // hybrid graphs: the virtual rows of a hidden layer (one 16-wave workgroup per row, latency-bound, 256 rows
    // for 32 puzzles) run on this side stream UNDER the masked MFMA attention of the real rows -- the two
    // kernels read the same Q / K / V and write disjoint rows.  Fork / join with events, so the pattern is
    // captured into the sampling-loop hipGraph as two parallel branches.
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // (round 6, measured: a SECOND side stream for the second branch of two Batches in flight makes configuration 3 slower -- x0.98 against
    //  x1.14 - 1.25 with this one shared; small Batches do not fork at all, see forward_impl)
    // da_sample_loop_pair: the second branch of the two-branch loop graph
    hipStream_t pair_stream = nullptr;
    hipEvent_t ev_pair_fork = nullptr, ev_pair_join = nullptr;
    struct PairEntry { LoopKey a, b; hipGraphExec_t exec; hipGraphExec_t exec_b; };      // exec_b: the second branch as a graph of its own (split mode), else null
    std::vector<PairEntry> pair_loops;
};

namespace da {

struct Workspace {
    char *comb_in, *h, *combined, *qkvs, *xa, *xb, *z, *hh;
    void *pz;                         // [H, n_real, 32] act dtype: per-head outputs of the folded last attention
    char *head_pre;                   // [n_real, 32] act dtype: residual share of final_mlp.0 (fused_mlp2)
    char *feat_proj;                  // [n_real, hidden] act dtype: mlp.0 over the piece-feature columns (+ bias), once per Batch
    char *feat_proj_unc;              // the same for ZERO features (= the bias, broadcast): the unconditional pass of classifier-free guidance
    float *model_out_unc;
    char *dq, *dk, *dvt, *dskip;      // dense path: head-major Q, K, V ([H][n_pad][C] each), row-major skip
    size_t dense_off, dense_bytes;    // [dq, dq + dense_bytes) is zero-filled once per Batch
    unsigned *virt_cnt;               // hybrid graphs with virtual rows: arrival counters of the rows' workgroups (inside the zero-filled region) ...
    float *virt_part;                 // ... and their partial softmax states (AttnDenseParams::v_cnt / v_part)
    float *model_out, *xbuf0, *xbuf1;
    size_t total;
};

static Workspace carve(const da_denoiser *d, const da_graph *g, void *base) {
    const size_t s = esize(d->prec);
    const size_t n = (size_t)g->n_nodes, nr = (size_t)g->n_real;
    const size_t nrp = nr + 64, np = n + 64;          // slack rows so tile kernels may over-read
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return q;
    };
    Workspace w;
    int hcmax = 0;
    for (int l = 0; l < d->n_layers; ++l) hcmax = d->conv[l].hc > hcmax ? d->conv[l].hc : hcmax;
    w.comb_in = take(nrp * d->D * s);
    w.h = take(nrp * d->hidden * s);
    w.feat_proj = take(nrp * d->hidden * s);
    w.feat_proj_unc = take(nrp * d->hidden * s);
    w.head_pre = take(nrp * 32 * s);
    w.pz = take(nrp * 32 * (size_t)d->heads * sizeof(float));
    w.combined = take(np * d->D * s);
    w.qkvs = take(np * 4 * (size_t)hcmax * s);
    w.xa = take(np * 256 * s);
    w.xb = take(np * 256 * s);
    w.z = take(np * d->D * s);
    w.hh = take(nrp * d->head_hidden * s);
    w.dq = w.dk = w.dvt = w.dskip = nullptr;
    w.virt_cnt = nullptr; w.virt_part = nullptr;
    w.dense_off = off;
    w.dense_bytes = 0;
    if ((g->dense || g->hybrid) && g->n_pad > 0) {
        const size_t hb = ((size_t)g->n_pad + 64) * hcmax * s;
        w.dq = take(hb);
        w.dk = take(hb);
        w.dvt = take(hb);
        if (g->hybrid && n > nr) w.virt_cnt = (unsigned *)take((n - nr) * sizeof(unsigned));
        w.dense_bytes = off - w.dense_off;
        if (g->hybrid && n > nr) w.virt_part = (float *)take((n - nr) * (size_t)DA_VIRT_SPLIT_MAX * 64 * 6 * sizeof(float));
        w.dskip = take(np * (size_t)hcmax * s);
    }
    const int cpose = d->variant == DA_VARIANT_3D ? 7 : d->c_out;
    w.model_out = (float *)take(nr * cpose * sizeof(float));
    w.model_out_unc = (float *)take(nr * cpose * sizeof(float));
    w.xbuf0 = (float *)take(nr * 8 * sizeof(float));
    w.xbuf1 = (float *)take(nr * 8 * sizeof(float));
    w.total = off;
    return w;
}

static bool mfma_disabled() { return cfg().disable_mfma != 0; }

int linear(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias, int act,
           const void *res, void *out, int ldo, hipStream_t st) {
    if (!mfma_disabled()) {
        const int rc = launch_gemm_mfma(prec, M, K, Nout, A, lda, W, bias, act, res, out, ldo, nullptr, st);
        if (rc >= 0) return rc;
    }
    return launch_gemm_simple(prec == DA_PREC_F32_BF16MMA ? DA_PREC_F32 : prec, M, K, Nout, A, lda, W, bias, act, res, out, ldo, st);
}

static bool dense_disabled() { return cfg().disable_dense != 0; }

// The large-Batch step rule (da_config.xpanel = tail_next = -1): row-panel projections + the next step's embedding inside the tail kernel for
// Batches whose largest graph has >= 512 pieces (round 5) OR that hold >= 16 384 pieces in all (round 6: tools/ab_config.py on configuration 2,
// 512 puzzles of 144 pieces -- -6.9 % at 20-step loops, -4.0 % at 100, twelve of twelve interleaved pairs each; round 5's process-level A/B had
// read that Batch as a loss).  Small Batches (the scripted 8-puzzle ones) keep the old path: nothing to gain from a panel of nine row tiles.
static bool step_rule_batch(const da_graph *g) { return g->max_graph_nodes >= 512 || g->n_real >= 16384; }

static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}

static bool dense_ok(const da_graph *g, int heads, int C) {
    const bool hyb = g->hybrid && g->mask && g->mask_ptr && g->irr_row_ptr;
    return !dense_disabled() && (g->dense || hyb) && g->n_pad > 0 && g->graph_ptr && g->pad_ptr && g->row_map && heads == 8 &&
           (C == 32 || C == 144) && !mfma_disabled();
}

static int check_graph(const da_denoiser *d, const da_graph *g) {
    DA_REQUIRE(g && g->n_real > 0 && g->n_nodes >= g->n_real, "da_graph: bad node counts");
    // complete graphs (dense != 0) on a denoiser whose every layer has an MFMA attention kernel never walk the
    // edge list; the host may then leave the CSR arrays out (da_denoiser_flags bit 2) unless alpha is wanted
    const bool hyb = g->hybrid && g->mask && g->mask_ptr && g->irr_row_ptr;
    if (!((g->dense || hyb) && d->dense_only)) DA_REQUIRE(g->row_ptr && (g->col_src || g->n_edges == 0), "da_graph: CSR arrays missing");
    DA_REQUIRE(d->V == 0 || g->n_nodes == g->n_real + d->V * g->n_graphs,
               "da_graph: exophormer expects n_nodes = n_real + V*G (%d vs %d + %d*%d)", g->n_nodes, g->n_real,
               d->V, g->n_graphs);
    return 0;
}

static DeviceSchedule to_dev(const da_schedule *s) {
    DeviceSchedule r;
    r.steps = s->steps;
    r.betas = s->betas;
    r.alphas_cumprod = s->alphas_cumprod;
    r.sqrt_recip_alphas = s->sqrt_recip_alphas;
    r.sqrt_recip_alphas_cumprod = s->sqrt_recip_alphas_cumprod;
    r.sqrt_recipm1_alphas_cumprod = s->sqrt_recipm1_alphas_cumprod;
    r.sqrt_one_minus_alphas_cumprod = s->sqrt_one_minus_alphas_cumprod;
    r.posterior_variance = s->posterior_variance;
    return r;
}

// Bracket one launch with a (start, stop) event pair on the launch stream when profiling.
template <typename F>
static int timed(da_denoiser *d, int cls, hipStream_t st, F &&launch) {
    if (!d->prof_on) return launch();
    if (d->prof_used + 2 > d->prof_ev.size()) {
        for (int k = 0; k < 2; ++k) {
            hipEvent_t e;
            DA_CHECK_HIP(hipEventCreate(&e));
            d->prof_ev.push_back(e);
        }
    }
    hipEvent_t a = d->prof_ev[d->prof_used], b = d->prof_ev[d->prof_used + 1];
    DA_CHECK_HIP(hipEventRecord(a, st));
    int rc = launch();
    DA_CHECK_HIP(hipEventRecord(b, st));
    d->prof_used += 2;
    d->prof_cls.push_back(cls);
    return rc;
}

static int forward_impl(da_denoiser *d, const da_graph *g, const float *x, const int64_t *t, int64_t t_scalar,
                        float *out, float *alpha, int alpha_all, float *pre_head, const Workspace &w,
                        hipStream_t st, DdimFuse *ddim = nullptr, bool uncond = false, bool h_ready = false) {
    const int prec = d->prec, nr = g->n_real, n = g->n_nodes, D = d->D;
    int rc;
    // (a-3) embedding: pose MLP + learned timestep lookup into the concat buffer, then mlp
    // (h_ready: the previous step's tail kernel already wrote this step's w.h -- DdimFuse::nx_*, sampling loops only)
    // exophormer, dense / hybrid path: the virtual rows' constant conv-0 projections are placed by extra workgroups of the embedding's launch (they
    // depend on nothing of this step; the layer-0 code below skips its launch_scatter_virtual then)
    VirtScatter vsc;
    bool virt_scattered = false;
    if (!h_ready && d->fused_mlp2 && n > nr && !alpha && w.dq && d->virt_qkvs_d && dense_ok(g, d->heads, d->conv[0].C) && DA_XENV("DA_VIRT_SCATTER_IN_EMBED", 1)) {
        vsc.rows = n - nr; vsc.V = d->V; vsc.H = d->heads; vsc.C = d->conv[0].C; vsc.n_real = nr; vsc.n_pad = g->n_pad;
        vsc.src = d->virt_qkvs_d; vsc.row_map = g->row_map; vsc.Q = w.dq; vsc.K = w.dk; vsc.Vt = w.dvt; vsc.S = w.dskip;
        virt_scattered = true;
    }
    if (!h_ready && (rc = timed(d, DA_PROF_EMBED, st, [&] {
             return launch_embed_pos_time(prec, nr, d->c_in, d->F, D, x, t, t_scalar, d->steps, d->time_emb,
                                          d->pos_w0, d->pos_b0, d->pos_w1, d->pos_b1, w.comb_in, st, virt_scattered ? &vsc : nullptr); }))) return rc;
    const int act1 = d->variant == DA_VARIANT_3D ? DA_ACT_LEAKY02 : DA_ACT_GELU;
    const int act2 = d->variant == DA_VARIANT_3D ? DA_ACT_LEAKY02 : DA_ACT_NONE;
    // mlp.0 over [features | pose | time]: the feature columns (F of the D inputs) do not change inside a
    // sampling loop, so their part of the product (+ bias) was computed once in da_denoiser_set_features;
    // per step only the 64 pose / timestep columns are multiplied and the cached part is added before the
    // activation (the reference recomputes the whole Linear(1152 -> 128) every step)
    if (!h_ready && (rc = timed(d, DA_PROF_LINEAR_MLP, st, [&] {
             const size_t es_ = esize(prec);
             int r2 = mfma_disabled() ? -1
                                      : launch_gemm_mfma(prec, nr, D - d->F, d->hidden, w.comb_in + (size_t)d->F * es_, D,
                                                         (const char *)d->mlp_w0 + (size_t)d->F * es_, nullptr, act1, nullptr,
                                                         w.h, d->hidden, nullptr, st, D, uncond ? w.feat_proj_unc : w.feat_proj);
             if (r2 >= 0) return r2;
             if (uncond) { set_error("unconditional pass: the hoisted mlp.0 path is not available for this shape"); return 1; }
             return linear(prec, nr, D, d->hidden, w.comb_in, D, d->mlp_w0, d->mlp_b0, act1, nullptr, w.h, d->hidden, st); }))) return rc;
    // mlp.2 (Linear(128 -> 1152), no activation in 2D) only feeds two linear consumers -- the conv-0 projection
    // and, through the residual, final_mlp.0 -- so for the transformer arch it is folded into their weights at
    // create time: the [N, 1152] `combined` tensor is never written, conv 0 reduces over 128 instead of 1152
    const bool fused = d->fused_mlp2;
    if (!fused && (rc = timed(d, DA_PROF_LINEAR_MLP, st, [&] {
             return linear(prec, nr, d->hidden, D, w.h, d->hidden, d->mlp_w1, d->mlp_b1, act2, nullptr, w.combined, D, st); }))) return rc;
    // (a-4..a-6) graph transformer: fused Q|K|V|skip projection + attention per layer
    const void *xin = fused ? w.h : w.combined;
    int ldx = fused ? d->hidden : D;
    for (int l = 0; l < d->n_layers; ++l) {
        ConvW c = d->conv[l];
        if (fused && l == 0) { c.w = d->conv0c_w; c.b = d->conv0c_b; c.din = d->hidden; }
        // fused conv 0 of the exophormer arch: only the real rows go through the GEMM, the virtual rows'
        // projections are constants that launch_scatter_virtual places behind them
        const int nproj = (fused && l == 0) ? nr : n;
        const bool virt0 = fused && l == 0 && n > nr;
        const bool last = l == d->n_layers - 1;
        void *dst = last ? (void *)w.z : (void *)((l & 1) ? w.xb : w.xa);
        const int act = (!last && d->arch == DA_ARCH_TRANSFORMER) ? DA_ACT_GELU : DA_ACT_NONE;
        float *al = !alpha ? nullptr
                           : (alpha_all ? alpha + (size_t)l * g->n_edges * d->heads : (last ? alpha : nullptr));
        // last layer: the residual `feats + combined_feats` (efficient_gat.py:144) is fused in the epilogue
        const void *resid = (last && !fused) ? w.combined : nullptr;
        if (last && fused && d->lastfold && !al && w.dq && (g->hybrid || d->V == 0) && dense_ok(g, d->heads, c.C)) {
            // folded last layer: Q | K | (V Wf_h^T) projection, 32-wide value heads, per-head outputs to pz
            QkvScatter qs;
            qs.HC = c.hc; qs.C = c.C; qs.Cv = 32; qs.n_pad = g->n_pad; qs.row_map = g->row_map;
            qs.Q = w.dq; qs.K = w.dk; qs.Vt = w.dvt; qs.S = nullptr;
            rc = timed(d, DA_PROF_LINEAR_QKVS, st, [&] {
                return launch_gemm_mfma(prec, n, c.din, 2 * c.hc + d->heads * 32, xin, ldx, d->convLf_w, d->convLf_b, DA_ACT_NONE,
                                        nullptr, nullptr, 0, &qs, st, 0, nullptr, d->convLf_wp); });
            if (rc > 0) return rc;
            if (rc == 0) {
                DenseLayout L;
                L.Q = w.dq; L.K = w.dk; L.Vt = w.dvt; L.S = nullptr; L.n_pad = g->n_pad; L.q_prescaled = d->q_prescaled;
                DenseFold fo;
                fo.cv = 32; fo.out = w.pz; fo.n_rows = nr;
                // hybrid graphs: adjacency-masked, remainder edges of the real rows folded in the epilogue; the
                // virtual rows' outputs of the LAST layer are dropped by the model (exophormer_gnn.py:209), so the
                // CSR-side kernels are not needed here
                const DenseMask mk = dense_mask_of(g);
                rc = timed(d, DA_PROF_ATTN_LAST, st, [&] {
                    return launch_attn_dense(prec, L, d->heads, c.C, g->n_graphs, g->max_graph_nodes, g->graph_ptr,
                                             g->pad_ptr, g->dense == 2, nullptr, DA_ACT_NONE, nullptr, st,
                                             g->hybrid ? &mk : nullptr, &fo); });
                if (rc > 0) return rc;
                if (rc == 0) {
                    // the whole tail in one kernel where it is covered (bf16, hidden 128, conv input 256) ...
                    {
                        const bool fuse_ddim = ddim && d->c_out == d->c_in;
                        int rt = -1;
                        if ((rc = timed(d, DA_PROF_HEAD, st, [&] {
                                 rt = launch_tail_fused(prec, nr, d->heads, d->c_out, d->hidden, c.din, w.h, xin, ldx, d->headc_w, d->headc_b,
                                                        d->skipc_w, d->skipc_b, w.pz, d->head_w1, d->head_b1, out, st, fuse_ddim ? ddim : nullptr);
                                 return rt > 0 ? rt : 0; }))) return rc;
                        if (rt == 0) { if (fuse_ddim) ddim->done = 1; return 0; }
                    }
                    // ... else: pre-activation of final_mlp.0 from the hidden layer (mlp.2 share) and from this conv's input (skip share)
                    if ((rc = timed(d, DA_PROF_HEAD, st, [&] {
                             return linear(prec, nr, d->hidden, 32, w.h, d->hidden, d->headc_w, d->headc_b, DA_ACT_NONE, nullptr, w.head_pre, 32, st); }))) return rc;
                    if ((rc = timed(d, DA_PROF_HEAD, st, [&] {
                             return linear(prec, nr, c.din, 32, xin, ldx, d->skipc_w, d->skipc_b, DA_ACT_NONE, w.head_pre, w.head_pre, 32, st); }))) return rc;
                    if (ddim && d->c_out == d->c_in) ddim->done = 1;        // the head kernel applies the DDIM update itself
                    return timed(d, DA_PROF_HEAD, st, [&] {
                        return launch_head_fold(prec, nr, d->heads, d->c_out, w.pz, w.head_pre, d->head_w1, d->head_b1, out, st,
                                                (ddim && ddim->done) ? ddim : nullptr); });
                }
            }
        }
#ifdef DA_EXPERIMENTS          // (DA_CONV_FUSED=1: the one-kernel hidden conv, da_conv_fused.hip -- a measured tie / loss, experiments build only)
        if (!al && !last && g->dense && !g->hybrid && n == nr && !resid && dense_ok(g, d->heads, c.C) &&
            conv_fused_applicable(prec, d->heads, c.C, c.din, g->max_graph_nodes, c.hc)) {
            // hidden conv on complete graphs: projection + attention of a (graph, head) in ONE kernel, K / V in LDS
            rc = timed(d, DA_PROF_CONV_FUSED, st, [&] {
                return launch_conv_fused(prec, d->heads, c.C, c.din, g->n_graphs, g->max_graph_nodes, g->graph_ptr,
                                         g->dense == 2, xin, ldx, c.w, c.b, act, dst, c.hc, st); });
            if (rc) return rc < 0 ? 1 : rc;
            xin = dst; ldx = c.hc;
            continue;
        }
#endif
        if (!al && w.dq && dense_ok(g, d->heads, c.C)) {
            // complete graphs: projection scattered into head-major Q / K / V, block-diagonal MFMA attention
            QkvScatter qs;
            qs.HC = c.hc; qs.C = c.C; qs.n_pad = g->n_pad; qs.row_map = g->row_map;
            qs.Q = w.dq; qs.K = w.dk; qs.Vt = w.dvt; qs.S = w.dskip;
            const void *wdense = (fused && l == 0) ? d->conv0c_wd : c.wd;
            const float *bdense = (fused && l == 0) ? d->conv0c_bd : c.bd;
            const int xpm = xpanel_mode();
            const void *wpanel = (xpm == 1 || (xpm == 2 && step_rule_batch(g))) ? ((fused && l == 0) ? d->conv0c_wdp : c.wdp) : nullptr;
            // Hidden layers of large complete graphs: the resident attention kernel (k_attn_res<.., 64>) does the layer's projection in its own
            // prologue -- no projection kernel, K | V never leave the CU
            const void *wqs = (fused && l == 0) ? d->conv0c_wqs : c.wqs;
            const bool qsf = !last && !g->hybrid && !resid && !virt0 && n == nr && wqs && ldx == c.din &&
                             attn_qsf_applicable(prec, d->heads, c.C, c.din, g->n_graphs, g->max_graph_nodes, g->n_pad, d->q_prescaled);
            if (qsf) rc = 0; else
            rc = timed(d, DA_PROF_LINEAR_QKVS, st, [&] {
                return launch_gemm_mfma(prec, nproj, c.din, 4 * c.hc, xin, ldx, wdense, bdense, DA_ACT_NONE, nullptr, nullptr, 0, &qs, st, 0,
                                        nullptr, wpanel); });
            if (rc > 0) return rc;
            if (rc == 0 && virt0 && !virt_scattered &&
                (rc = launch_scatter_virtual(prec, n - nr, d->V, d->heads, c.C, d->virt_qkvs_d, nr, g->row_map, g->n_pad, w.dq, w.dk,
                                             w.dvt, w.dskip, nullptr, st))) return rc;
            if (rc == 0) {
                DenseLayout L;
                L.Q = w.dq; L.K = w.dk; L.Vt = w.dvt; L.S = w.dskip; L.n_pad = g->n_pad; L.q_prescaled = d->q_prescaled;
                if (qsf) { L.x = xin; L.ldx = ldx; L.kin = c.din; L.wqs = wqs; L.bias = bdense; }
                if (g->hybrid) {
                    // sparse-but-heavy graphs: masked MFMA attention over the regular edges (partial softmax
                    // state), then the remaining edges + normalisation + skip / activation on the CSR side
                    DenseMask mk = dense_mask_of(g);
                    // SMALL Batches (below 8 192 pieces: the ones that do not fork a side stream below): the virtual rows inside the masked
                    // attention's own launch (bf16 hidden layers: k_attn_optt<32, false, true>; da_config disable_folds bit 5 keeps the two-kernel
                    // form) -- the scripted 8-puzzle Batch 0.152 -> 0.114 ms per step, 0.085 -> 0.066 / 0.052 -> 0.043 per Batch-step with two /
                    // four Batches in flight.  Large Batches keep the side stream: configuration 3 measured -2 % ... +4 % with the rows in the
                    // launch, depending on where in the grid they sit (profiles/r06/r06_virtual_rows_in_launch_ab*.log).
                    // (experiments build: DA_VIRT_IN_LAUNCH=0 two kernels everywhere, =2 in the launch for every Batch)
                    int virt_taken = 0;
                    [[maybe_unused]] const int vil = DA_XENV("DA_VIRT_IN_LAUNCH", 1);
                    if (n > nr && !resid && w.virt_cnt && w.virt_part && !(cfg().disable_folds & DA_FOLD_HYBRID_OVERLAP) && vil && (g->n_real < 8192 || vil == 2)) {
                        mk.v_rows = n - nr; mk.v_n_real = nr;
                        mk.v_row_ptr = g->agg_row_ptr ? g->agg_row_ptr : g->irr_row_ptr; mk.v_col_src = g->agg_row_ptr ? g->agg_col_src : g->irr_col_src;
                        mk.v_mult = g->agg_row_ptr ? g->agg_mult : nullptr;
                        mk.v_part = w.virt_part; mk.v_cnt = w.virt_cnt; mk.v_taken = &virt_taken;
                        rc = timed(d, last ? DA_PROF_ATTN_LAST : DA_PROF_ATTN_HIDDEN, st, [&] {
                            return launch_attn_dense(prec, L, d->heads, c.C, g->n_graphs, g->max_graph_nodes, g->graph_ptr, g->pad_ptr, 0, resid, act, dst, st, &mk); });
                        if (rc) return rc < 0 ? 1 : rc;
                        if (!virt_taken) {          // (a kernel without the extra workgroups took the layer: the rows' own kernel behind it)
                            rc = timed(d, last ? DA_PROF_ATTN_LAST : DA_PROF_ATTN_HIDDEN, st, [&] {
                                return launch_attn_csr_cont(prec, n, nr, mk.v_row_ptr, mk.v_col_src, g->row_map, d->heads, c.C, g->n_pad, L, resid, act, dst, st, mk.v_mult); });
                            if (rc) return rc < 0 ? 1 : rc;
                        }
                        xin = dst; ldx = c.hc;
                        continue;
                    }
                    // (small Batches keep the virtual rows on the caller's stream: below ~8 k pieces the fork / join costs more than the overlap
                    //  buys -- the scripted 8-puzzle Batch: 0.165 -> 0.152 ms per step, profiles/r06/r06_scripted_side_stream_variants.log)
                    if (d->side_stream && !d->prof_on && n > nr && g->n_real >= 8192) {
                        // fork: virtual rows on the side stream, real rows here, join before the next projection
                        DA_CHECK_HIP(hipEventRecord(d->ev_fork, st));
                        DA_CHECK_HIP(hipStreamWaitEvent(d->side_stream, d->ev_fork, 0));
                        rc = launch_attn_csr_cont(prec, n, nr, g->agg_row_ptr ? g->agg_row_ptr : g->irr_row_ptr, g->agg_row_ptr ? g->agg_col_src : g->irr_col_src,
                                                  g->row_map, d->heads, c.C, g->n_pad, L, resid, act, dst, d->side_stream, g->agg_row_ptr ? g->agg_mult : nullptr);
                        if (rc) return rc < 0 ? 1 : rc;
                        DA_CHECK_HIP(hipEventRecord(d->ev_join, d->side_stream));
                        rc = launch_attn_dense(prec, L, d->heads, c.C, g->n_graphs, g->max_graph_nodes, g->graph_ptr,
                                               g->pad_ptr, 0, resid, act, dst, st, &mk);
                        if (rc) return rc < 0 ? 1 : rc;
                        DA_CHECK_HIP(hipStreamWaitEvent(st, d->ev_join, 0));
                        xin = dst; ldx = c.hc;
                        continue;
                    }
                    rc = timed(d, last ? DA_PROF_ATTN_LAST : DA_PROF_ATTN_HIDDEN, st, [&] {
                        int r2 = launch_attn_dense(prec, L, d->heads, c.C, g->n_graphs, g->max_graph_nodes, g->graph_ptr,
                                                   g->pad_ptr, 0, resid, act, dst, st, &mk);
                        if (r2) return r2;
                        return launch_attn_csr_cont(prec, n, nr, g->agg_row_ptr ? g->agg_row_ptr : g->irr_row_ptr, g->agg_row_ptr ? g->agg_col_src : g->irr_col_src,
                                                    g->row_map, d->heads, c.C, g->n_pad, L, resid, act, dst, st, g->agg_row_ptr ? g->agg_mult : nullptr); });
                    if (rc) return rc < 0 ? 1 : rc;
                    xin = dst; ldx = c.hc;
                    continue;
                }
                rc = timed(d, last ? DA_PROF_ATTN_LAST : DA_PROF_ATTN_HIDDEN, st, [&] {
                    return launch_attn_dense(prec, L, d->heads, c.C, g->n_graphs, g->max_graph_nodes, g->graph_ptr,
                                             g->pad_ptr, g->dense == 2, resid, act, dst, st); });
                if (rc > 0) return rc;
                if (rc == 0) { xin = dst; ldx = c.hc; continue; }
            }
        }
        if ((rc = timed(d, DA_PROF_LINEAR_QKVS, st, [&] {
                 return linear(prec, nproj, c.din, 4 * c.hc, xin, ldx, c.w, c.b, DA_ACT_NONE, nullptr, w.qkvs, 4 * c.hc, st); }))) return rc;
        if (virt0 && (rc = launch_scatter_virtual(prec, n - nr, d->V, d->heads, c.C, d->virt_qkvs, nr, nullptr, 0, nullptr, nullptr,
                                                  nullptr, nullptr, w.qkvs, st))) return rc;
        DA_REQUIRE(g->row_ptr && (g->col_src || g->n_edges == 0), "da_graph: this call walks the edge list (alpha requested or no "
                   "MFMA attention for this layer) but the CSR arrays are missing");
        // tiny complete graphs at a head width without a matrix-core kernel (the 3D variant: 20-fragment objects, C = 104): one workgroup per graph
        // with the graph's K | V rows in LDS instead of one walk of the edge list per destination
        int tiny = -1;
        if (!al && g->dense && n == nr && g->graph_ptr && !dense_disabled()) {
            if ((rc = timed(d, last ? DA_PROF_ATTN_LAST : DA_PROF_ATTN_HIDDEN, st, [&] {
                     tiny = launch_attn_tiny(prec, g->n_graphs, g->max_graph_nodes, g->graph_ptr, g->dense == 2, d->heads, c.C, w.qkvs, resid, act, dst, st);
                     return tiny > 0 ? tiny : 0; }))) return rc;
        }
        if (tiny != 0 && (rc = timed(d, last ? DA_PROF_ATTN_LAST : DA_PROF_ATTN_HIDDEN, st, [&] {
                 return launch_attn_csr(prec, n, g->row_ptr, g->col_src, g->edge_id, d->heads, c.C, w.qkvs,
                                        resid, act, dst, al, nullptr, st); }))) return rc;
        xin = dst;
        ldx = c.hc;
    }
    // (a-7 / a-13) pose head
    if (fused) {
        // final_mlp.0(conv_out + combined) = Wf . conv_out + (Wf W2) . h + (Wf b2 + bf): the second term first
        if ((rc = timed(d, DA_PROF_HEAD, st, [&] {
                 return linear(prec, nr, d->hidden, 32, w.h, d->hidden, d->headc_w, d->headc_b, DA_ACT_NONE, nullptr, w.head_pre, 32, st); }))) return rc;
        rc = timed(d, DA_PROF_HEAD, st, [&] {
            return launch_gemm_mfma(prec, nr, D, d->head_hidden, w.z, D, d->head_w0, d->head_b0, DA_ACT_GELU, nullptr, w.hh,
                                    d->head_hidden, nullptr, st, D, w.head_pre); });
        if (rc > 0) return rc;
        DA_REQUIRE(rc == 0, "fused head GEMM: shape not supported");
    } else if ((rc = timed(d, DA_PROF_HEAD, st, [&] {
             return linear(prec, nr, D, d->head_hidden, w.z, D, d->head_w0, d->head_b0, DA_ACT_GELU, nullptr, w.hh,
                           d->head_hidden, st); }))) return rc;
    return timed(d, DA_PROF_HEAD, st, [&] {
        if (d->variant == DA_VARIANT_3D)
            return launch_head3d(prec, nr, w.hh, d->head_w1, d->head_b1, d->head_r_w1, d->head_r_b1, out, pre_head, st);
        return launch_head2d(prec, nr, d->c_out, w.hh, d->head_w1, d->head_b1, out, st);
    });
}

}  // namespace da

// one wave that waits `ticks` of the 100 MHz real-time counter (phase-offset experiment of the two-branch loop)
__global__ void k_pair_delay(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" {

int da_abi_version(void) { return DA_ABI_VERSION; }
const char *da_last_error(void) { return da::g_err; }

int da_denoiser_create(const da_weights *w, int precision, void *stream, da_denoiser **out) {
    DA_REQUIRE(w && out, "da_denoiser_create: null argument");
    DA_REQUIRE(precision == DA_PREC_F32 || precision == DA_PREC_BF16, "bad precision %d", precision);
    DA_REQUIRE(w->n_layers >= 2 && w->n_layers <= DA_MAX_LAYERS, "n_layers out of range");
    DA_REQUIRE(w->heads == 8, "heads must be 8");
    // k_embed_pos_time keeps the pose-MLP weight rows in a fixed register array (da_basic.hip); fail here, not on the first
    // forward or inside a hipGraph capture (the reference uses c_in = 2 / 4 in 2D and 7 in 3D)
    DA_REQUIRE(w->c_in >= 1 && w->c_in <= 8, "da_denoiser_create: c_in = %d outside [1, 8]", w->c_in);
    hipStream_t st = (hipStream_t)stream;
    da_denoiser *d = new da_denoiser();
    d->prec = precision; d->variant = w->variant; d->arch = w->arch; d->steps = w->steps; d->c_in = w->c_in;
    d->c_out = w->c_out; d->F = w->feat_dim; d->D = w->feat_dim + 64; d->hidden = w->hidden; d->heads = w->heads;
    d->n_layers = w->n_layers; d->V = w->arch == DA_ARCH_EXOPHORMER ? w->virt_nodes : 0;
    d->head_hidden = w->variant == DA_VARIANT_3D ? 512 : 32;
    const int D = d->D, H = d->heads;
    const size_t s = esize(precision);
    int rc = 0;
    auto fail = [&](int code) { da_denoiser_destroy(d); return code; };
    auto alloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (hipMalloc(&p, bytes < 256 ? 256 : bytes) != hipSuccess) { set_error("hipMalloc(%zu) failed", bytes); return nullptr; }
        d->owned.push_back(p);
        return p;
    };
    auto copy_f32 = [&](const float *src, size_t n) -> float * {
        float *p = (float *)alloc(n * 4);
        if (!p || hipMemcpyAsync(p, src, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = 2; return nullptr; }
        return p;
    };
    auto pack = [&](const float *src, size_t n) -> void * {
        void *p = alloc(n * s + 4096);       // slack: tile kernels may over-read a partial tile
        if (!p || launch_convert(precision, n, src, p, st)) { rc = 2; return nullptr; }
        return p;
    };
    // per-head Q / skip fragments of a hidden layer's dense-path projection (bf16, 32-wide heads, K = 128 / 256); nullptr otherwise
    auto pack_qs = [&](const void *wsrc, int K, int hc, int C) -> void * {
#ifndef DA_EXPERIMENTS
        return nullptr;          // (k_attn_res<.., 64> lost its A/B: experiments build only)
#endif
        if (precision != DA_PREC_BF16 || C != 32 || H != 8 || (K != 128 && K != 256) || !wsrc || mfma_disabled()) return nullptr;
        void *p = alloc(w_qs_bytes(H, K));
        if (!p || pack_w_qs(H, K, hc, wsrc, p, st)) { rc = 2; return nullptr; }
        return p;
    };
    // fragment-major copy of a projection weight for the row-panel kernel; nullptr when the shape has no packed form
    auto pack_panel = [&](const void *wsrc, int K, int Nout) -> void * {
        const size_t nb = xpanel_in_model() ? da_linear_packed_bytes(precision, K, Nout) : 0;
        if (!nb || !wsrc) return nullptr;
        void *p = alloc(nb);
        if (!p || pack_w_xpanel(K, Nout, wsrc, K, p, st)) { rc = 2; return nullptr; }
        return p;
    };
    d->time_emb = copy_f32(w->time_emb, (size_t)w->steps * 32);
    d->pos_w0 = copy_f32(w->pos_w0, 16 * (size_t)w->c_in); d->pos_b0 = copy_f32(w->pos_b0, 16);
    d->pos_w1 = copy_f32(w->pos_w1, 32 * 16); d->pos_b1 = copy_f32(w->pos_b1, 32);
    d->mlp_w0 = pack(w->mlp_w0, (size_t)d->hidden * D); d->mlp_b0 = copy_f32(w->mlp_b0, d->hidden);
    d->mlp_w1 = pack(w->mlp_w1, (size_t)D * d->hidden); d->mlp_b1 = copy_f32(w->mlp_b1, D);
    if (rc) return fail(rc);
    for (int l = 0; l < d->n_layers; ++l) {
        ConvW &c = d->conv[l];
        c.din = l == 0 ? D : 32 * H;
        c.C = l == d->n_layers - 1 ? D / H : 32;
        c.hc = c.C * H;
        const size_t blk = (size_t)c.hc * c.din;
        char *wp = (char *)alloc(4 * blk * s + 4096);
        c.b = (float *)alloc(4 * (size_t)c.hc * 4);
        if (!wp || !c.b) return fail(2);
        c.w = wp;
        const float *ws[4] = {w->conv_wq[l], w->conv_wk[l], w->conv_wv[l], w->conv_ws[l]};
        const float *bs[4] = {w->conv_bq[l], w->conv_bk[l], w->conv_bv[l], w->conv_bs[l]};
        for (int k = 0; k < 4; ++k) {
            if (!ws[k] || !bs[k]) { set_error("conv %d: missing weight pointer", l); return fail(1); }
            if (launch_convert(precision, blk, ws[k], wp + k * blk * s, st)) return fail(2);
            if (hipMemcpyAsync(c.b + (size_t)k * c.hc, bs[k], (size_t)c.hc * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
        }
        c.wd = c.w; c.bd = c.b;
        if (q_prescale_on()) {
            char *wd = (char *)alloc(4 * blk * s + 4096);
            float *tq = (float *)alloc(blk * 4);
            c.bd = (float *)alloc(4 * (size_t)c.hc * 4);
            if (!wd || !tq || !c.bd) return fail(2);
            c.wd = wd;
            if (hipMemcpyAsync(tq, ws[0], blk * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
            if (scale_block(tq, c.hc, c.din, c.din, q_scale_log2(c.C), st)) return fail(2);
            if (launch_convert(precision, blk, tq, wd, st)) return fail(2);
            if (hipMemcpyAsync(wd + blk * s, wp + blk * s, 3 * blk * s, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
            if (hipMemcpyAsync(c.bd, c.b, 4 * (size_t)c.hc * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
            if (scale_block(c.bd, 1, c.hc, c.hc, q_scale_log2(c.C), st)) return fail(2);
        }
        c.wdp = pack_panel(c.wd, c.din, 4 * c.hc);
        if (l < d->n_layers - 1) c.wqs = pack_qs(c.wd, c.din, c.hc, c.C);
        if (rc) return fail(rc);
    }
    d->q_prescaled = q_prescale_on();
    if (d->V > 0) {
        if (!w->virt_emb) { set_error("exophormer: virt_emb missing"); return fail(1); }
        d->virt_emb = pack(w->virt_emb, (size_t)d->V * D);
    }
    if (d->variant == DA_VARIANT_3D) {
        char *hp = (char *)alloc(512 * (size_t)D * s + 4096);
        d->head_b0 = (float *)alloc(512 * 4);
        if (!hp || !d->head_b0) return fail(2);
        d->head_w0 = hp;
        if (launch_convert(precision, 256 * (size_t)D, w->head_w0, hp, st)) return fail(2);
        if (launch_convert(precision, 256 * (size_t)D, w->head_r_w0, hp + 256 * (size_t)D * s, st)) return fail(2);
        if (hipMemcpyAsync(d->head_b0, w->head_b0, 256 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
        if (hipMemcpyAsync(d->head_b0 + 256, w->head_r_b0, 256 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
        d->head_w1 = copy_f32(w->head_w1, 3 * 256); d->head_b1 = copy_f32(w->head_b1, 3);
        d->head_r_w1 = copy_f32(w->head_r_w1, 3 * 256); d->head_r_b1 = copy_f32(w->head_r_b1, 3);
    } else {
        d->head_w0 = pack(w->head_w0, 32 * (size_t)D); d->head_b0 = copy_f32(w->head_b0, 32);
        d->head_w1 = copy_f32(w->head_w1, (size_t)w->c_out * 32); d->head_b1 = copy_f32(w->head_b1, w->c_out);
    }
    if (rc) return fail(rc);
    {
        const bool off = (cfg().disable_folds & DA_FOLD_MLP2) != 0;
        if (!off && !mfma_disabled() && d->variant == DA_VARIANT_2D && d->hidden % 32 == 0) {
            // compose in fp32 from the caller's fp32 weights, then pack
            const int hid = d->hidden, hc0 = d->conv[0].hc;
            float *w2t = (float *)alloc((size_t)hid * D * 4);                  // W2^T [hidden, D]
            float *cw = (float *)alloc((size_t)4 * hc0 * hid * 4);             // Wcat0 . W2
            float *hw = (float *)alloc((size_t)32 * hid * 4);                  // Wf0 . W2
            d->conv0c_b = (float *)alloc((size_t)4 * hc0 * 4);
            d->headc_b = (float *)alloc(32 * 4);
            if (!w2t || !cw || !hw || !d->conv0c_b || !d->headc_b) return fail(2);
            if (launch_transpose_f32(D, hid, w->mlp_w1, w2t, st)) return fail(2);
            const float *ws[4] = {w->conv_wq[0], w->conv_wk[0], w->conv_wv[0], w->conv_ws[0]};
            const float *bs[4] = {w->conv_bq[0], w->conv_bk[0], w->conv_bv[0], w->conv_bs[0]};
            for (int k = 0; k < 4; ++k) {
                // rows of block k: [hc0, D] . W2 [D, hid]  ==  A [hc0, D] @ (W2^T [hid, D])^T
                if (launch_gemm_simple(DA_PREC_F32, hc0, D, hid, ws[k], D, w2t, nullptr, DA_ACT_NONE, nullptr,
                                       cw + (size_t)k * hc0 * hid, hid, st)) return fail(2);
                // bias: b2 [1, D] @ (W_k [hc0, D])^T + b_k
                if (launch_gemm_simple(DA_PREC_F32, 1, D, hc0, w->mlp_b1, D, ws[k], bs[k], DA_ACT_NONE, nullptr,
                                       d->conv0c_b + (size_t)k * hc0, hc0, st)) return fail(2);
            }
            if (launch_gemm_simple(DA_PREC_F32, 32, D, hid, w->head_w0, D, w2t, nullptr, DA_ACT_NONE, nullptr, hw, hid, st)) return fail(2);
            if (launch_gemm_simple(DA_PREC_F32, 1, D, 32, w->mlp_b1, D, w->head_w0, nullptr, DA_ACT_NONE, nullptr, d->headc_b, 32, st)) return fail(2);
            d->conv0c_w = pack(cw, (size_t)4 * hc0 * hid);
            d->headc_w = pack(hw, (size_t)32 * hid);
            d->conv0c_wd = d->conv0c_w; d->conv0c_bd = d->conv0c_b;
            if (d->q_prescaled) {          // dense-path variant: Q rows scaled in fp32, then packed (cw is not used afterwards)
                d->conv0c_bd = (float *)alloc((size_t)4 * hc0 * 4);
                if (!d->conv0c_bd) return fail(2);
                if (hipMemcpyAsync(d->conv0c_bd, d->conv0c_b, (size_t)4 * hc0 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
                if (scale_block(d->conv0c_bd, 1, hc0, hc0, q_scale_log2(d->conv[0].C), st)) return fail(2);
                if (scale_block(cw, hc0, hid, hid, q_scale_log2(d->conv[0].C), st)) return fail(2);
                d->conv0c_wd = pack(cw, (size_t)4 * hc0 * hid);
            }
            d->conv0c_wdp = pack_panel(d->conv0c_wd, hid, 4 * hc0);
            if (d->n_layers > 1) d->conv0c_wqs = pack_qs(d->conv0c_wd, hid, hc0, d->conv[0].C);
            if (d->V > 0) {            // exophormer: conv-0 projections of the virtual rows (they bypass mlp)
                float *vq = (float *)alloc((size_t)d->V * 4 * hc0 * 4);
                if (!vq) return fail(2);
                for (int k = 0; k < 4; ++k)
                    if (launch_gemm_simple(DA_PREC_F32, d->V, D, hc0, w->virt_emb, D, ws[k], bs[k], DA_ACT_NONE, nullptr,
                                           vq + (size_t)k * hc0, 4 * hc0, st)) return fail(2);
                d->virt_qkvs = pack(vq, (size_t)d->V * 4 * hc0);
                d->virt_qkvs_d = d->virt_qkvs;
                if (d->q_prescaled) {
                    if (scale_block(vq, d->V, hc0, 4 * hc0, q_scale_log2(d->conv[0].C), st)) return fail(2);
                    d->virt_qkvs_d = pack(vq, (size_t)d->V * 4 * hc0);
                }
            }
            if (rc) return fail(rc);
            d->fused_mlp2 = true;
            // last conv: value heads and skip folded with final_mlp.0
            const int L = d->n_layers - 1, hcL = d->conv[L].hc, CL = d->conv[L].C, dinL = d->conv[L].din;
            const bool off2 = (cfg().disable_folds & DA_FOLD_LAST) != 0;
            if (!off2 && CL == 144 && hcL == D && dinL % 32 == 0 && d->c_out <= 8) {        // (k_head_fold: 8 outputs per row at most)
                const int nf = 2 * hcL + H * 32;
                float *lw = (float *)alloc((size_t)nf * dinL * 4);
                float *sw = (float *)alloc((size_t)32 * dinL * 4);
                d->convLf_b = (float *)alloc((size_t)nf * 4);
                d->skipc_b = (float *)alloc(32 * 4);
                float *tmpb = (float *)alloc(32 * 4);
                if (!lw || !sw || !d->convLf_b || !d->skipc_b || !tmpb) return fail(2);
                const size_t blk = (size_t)hcL * dinL;
                if (hipMemcpyAsync(lw, w->conv_wq[L], blk * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
                if (hipMemcpyAsync(lw + blk, w->conv_wk[L], blk * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
                if (hipMemcpyAsync(d->convLf_b, w->conv_bq[L], (size_t)hcL * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
                if (hipMemcpyAsync(d->convLf_b + hcL, w->conv_bk[L], (size_t)hcL * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(2);
                for (int h = 0; h < H; ++h) {
                    // (Wf[:, hC:(h+1)C] [32, C]) . (Wv[hC:(h+1)C, :] [C, din]) and the same block of the bias
                    if (launch_mm_nn_f32(32, dinL, CL, w->head_w0 + (size_t)h * CL, D, w->conv_wv[L] + (size_t)h * CL * dinL, dinL,
                                         nullptr, lw + 2 * blk + (size_t)h * 32 * dinL, dinL, st)) return fail(2);
                    if (launch_mm_nn_f32(32, 1, CL, w->head_w0 + (size_t)h * CL, D, w->conv_bv[L] + (size_t)h * CL, 1, nullptr,
                                         d->convLf_b + 2 * hcL + h * 32, 1, st)) return fail(2);
                }
                // skip: Wf [32, D] . Ws [D, din];  bias: Wf . bs + bf0
                if (launch_mm_nn_f32(32, dinL, D, w->head_w0, D, w->conv_ws[L], dinL, nullptr, sw, dinL, st)) return fail(2);
                if (launch_mm_nn_f32(32, 1, D, w->head_w0, D, w->conv_bs[L], 1, w->head_b0, d->skipc_b, 1, st)) return fail(2);
                if (d->q_prescaled) {          // this projection only ever feeds the matrix-core attention
                    if (scale_block(lw, hcL, dinL, dinL, q_scale_log2(CL), st)) return fail(2);
                    if (scale_block(d->convLf_b, 1, hcL, hcL, q_scale_log2(CL), st)) return fail(2);
                }
                d->convLf_w = pack(lw, (size_t)nf * dinL);
                d->convLf_wp = pack_panel(d->convLf_w, dinL, nf);
                d->skipc_w = pack(sw, (size_t)32 * dinL);
                if (rc) return fail(rc);
                d->lastfold = true;
            }
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("da_denoiser_create: sync failed"); return fail(2); }
    {   // side stream of the hybrid path (see the struct); da_config.disable_folds bit 32 keeps everything on one stream
        if (d->V > 0 && !(cfg().disable_folds & DA_FOLD_HYBRID_OVERLAP)) {
            if (hipStreamCreateWithFlags(&d->side_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming) != hipSuccess) {
                set_error("da_denoiser_create: side stream / events");
                return fail(2);
            }
        }
    }
    d->dense_only = d->heads == 8 && !dense_disabled() && !mfma_disabled();
    for (int l = 0; l < d->n_layers; ++l) d->dense_only = d->dense_only && (d->conv[l].C == 32 || d->conv[l].C == 144);
    *out = d;
    return 0;
}

// bit 3: the hoisted mlp.0 product (feature columns once per loop, pose / timestep columns per step through launch_gemm_mfma's `pre`
// operand) exists for THIS denoiser's shapes -- the conditions launch_gemm_mfma itself checks (reduction a multiple of its 128-byte
// K stage, 16-byte aligned column slice and rows), not merely "MFMA enabled": the captured classifier-free-guidance loop needs it
static bool hoisted_mlp0_ok(const da_denoiser *d) {
    if (da::mfma_disabled()) return false;
    const int es = (int)da::esize(d->prec), bk = 128 / es, kpose = d->D - d->F;
    return kpose > 0 && kpose % bk == 0 && ((size_t)d->F * es) % 16 == 0 && ((size_t)d->D * es) % 16 == 0 && d->hidden % (16 / es) == 0;
}
int da_denoiser_flags(const da_denoiser *d) {
    return d ? (d->fused_mlp2 ? 1 : 0) | (d->lastfold ? 2 : 0) | (d->dense_only ? 4 : 0) | (hoisted_mlp0_ok(d) ? 8 : 0) : 0;
}

void da_denoiser_destroy(da_denoiser *d) {
    if (!d) return;
    for (hipEvent_t e : d->prof_ev) (void)hipEventDestroy(e);
    for (auto &e : d->loops) (void)hipGraphExecDestroy(e.exec);
    if (d->cap_stream) (void)hipStreamDestroy(d->cap_stream);
    if (d->side_stream) (void)hipStreamDestroy(d->side_stream);
    if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
    if (d->ev_join) (void)hipEventDestroy(d->ev_join);
    for (auto &e : d->pair_loops) { (void)hipGraphExecDestroy(e.exec); if (e.exec_b) (void)hipGraphExecDestroy(e.exec_b); }
    if (d->pair_stream) (void)hipStreamDestroy(d->pair_stream);
    if (d->ev_pair_fork) (void)hipEventDestroy(d->ev_pair_fork);
    if (d->ev_pair_join) (void)hipEventDestroy(d->ev_pair_join);
    for (void *p : d->owned) (void)hipFree(p);
    delete d;
}

size_t da_denoiser_workspace_bytes(const da_denoiser *d, const da_graph *g) {
    if (!d || !g) return 0;
    return da::carve(d, g, nullptr).total;
}

int da_denoiser_set_features(da_denoiser *d, const da_graph *g, const float *feats, void *workspace,
                             size_t workspace_bytes, void *stream) {
    DA_REQUIRE(d && g && feats && workspace, "da_denoiser_set_features: null argument");
    int rc = check_graph(d, g);
    if (rc) return rc;
    Workspace w = carve(d, g, workspace);
    DA_REQUIRE(workspace_bytes >= w.total, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = launch_set_feats(d->prec, g->n_real, d->F, d->D, feats, w.comb_in, st))) return rc;
    if (!mfma_disabled()) {       // loop-invariant part of mlp.0 (see forward_impl); unsupported shapes fall back there
        rc = launch_gemm_mfma(d->prec, g->n_real, d->F, d->hidden, w.comb_in, d->D, d->mlp_w0, d->mlp_b0, DA_ACT_NONE, nullptr,
                              w.feat_proj, d->hidden, nullptr, st, d->D, nullptr);
        if (rc > 0) return rc;
        DA_REQUIRE(rc == 0, "da_denoiser_set_features: feature projection shape not supported (F=%d)", d->F);
        const size_t nb = (size_t)g->n_real * d->hidden;
        k_fill_bias_rows<<<(unsigned)((nb + 255) / 256), 256, 0, st>>>(d->prec, g->n_real, d->hidden, d->mlp_b0, w.feat_proj_unc);
        DA_LAUNCH_CHECK();
    }
    if (w.dense_bytes)      // padded rows / columns of the head-major buffers must be finite
        DA_CHECK_HIP(hipMemsetAsync((char *)workspace + w.dense_off, 0, w.dense_bytes, st));
    if (d->V > 0) {
        char *dst = w.combined + (size_t)g->n_real * d->D * esize(d->prec);
        if ((rc = launch_set_virtual_rows(d->prec, g->n_nodes - g->n_real, d->V, d->D, d->virt_emb, dst, st))) return rc;
    }
    return 0;
}

int da_denoiser_forward(da_denoiser *d, const da_graph *g, const float *x, const int64_t *t, int64_t t_scalar,
                        float *out, float *alpha, int alpha_all_layers, float *pre_head, void *workspace,
                        size_t workspace_bytes, void *stream) {
    DA_REQUIRE(d && g && x && out && workspace, "da_denoiser_forward: null argument");
    int rc = check_graph(d, g);
    if (rc) return rc;
    DA_REQUIRE(!alpha || g->edge_id, "alpha requested but graph has no edge_id");
    Workspace w = carve(d, g, workspace);
    DA_REQUIRE(workspace_bytes >= w.total, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    return forward_impl(d, g, x, t, t_scalar, out, alpha, alpha_all_layers, pre_head, w, (hipStream_t)stream);
}

int da_ddim_step(const da_schedule *s, int variant, int mean_type, int n, int c, const float *x,
                 const float *model_out, const int64_t *t, int64_t t_scalar, int inference_ratio,
                 int prev_all_nonneg, float eta, const float *noise, float *x_prev, void *stream) {
    DA_REQUIRE(s && x && model_out && x_prev, "da_ddim_step: null argument");
    DA_REQUIRE(eta == 0.f || noise, "da_ddim_step: eta > 0 needs noise");
    if (variant == DA_VARIANT_3D) {
        DA_REQUIRE(c == 7, "da_ddim_step(3D): c must be 7");
        DA_REQUIRE(eta == 0.f, "da_ddim_step(3D): eta must be 0 (the reference binds DDIM only)");
        return launch_ddim3d(to_dev(s), mean_type, n, x, model_out, t, t_scalar, inference_ratio, prev_all_nonneg,
                             x_prev, (hipStream_t)stream);
    }
    return launch_ddim2d(to_dev(s), mean_type, n, c, x, model_out, t, t_scalar, inference_ratio, prev_all_nonneg, eta,
                         noise, x_prev, (hipStream_t)stream);
}

int da_ddpm_step(const da_schedule *s, int n, int c, const float *x, const float *model_out, const int64_t *t,
                 int64_t t_scalar, const float *noise, float *x_prev, void *stream) {
    DA_REQUIRE(s && x && model_out && x_prev, "da_ddpm_step: null argument");
    return launch_ddpm2d(to_dev(s), n, c, x, model_out, t, t_scalar, noise, x_prev, (hipStream_t)stream);
}

static int enqueue_loop(da_denoiser *d, const da_graph *g, const da_schedule *s, int mean_type, int ratio,
                        int n_iters, const float *x_init, float *traj, float *x_final, const Workspace &w,
                        hipStream_t st, const da_loop_opts *o = nullptr, size_t traj_stride = 0, size_t noise_stride = 0) {
    const int nr = g->n_real;
    const int c = d->variant == DA_VARIANT_3D ? 7 : d->c_in;
    const size_t bytes = (size_t)nr * c * sizeof(float);
    const DeviceSchedule ds = to_dev(s);
    const float *cur = x_init;
    int it = 0, rc;
    const int first = ((s->steps - 1) / ratio) * ratio;            // reversed(range(0, steps, ratio))[0]
    // the variants of the reference's samplers that need more than forward + deterministic update (all inside the same
    // enqueue, so all inside the captured graph): classifier-free guidance = a second forward over zero features and a
    // linear combination (spatial_diffusion.py:568-589); eta > 0 / DDPM = a noise term read from a caller-filled
    // [n_iters, n_real, c] buffer (:485-510, :620-627 draw torch.randn_like per step; the host draws them in one call)
    const bool cfg = o && o->cfg, ddpm = o && o->sampler == 1;
    const float eta = o ? o->eta : 0.f;
    const bool plain = !cfg && !ddpm && eta == 0.f;
    bool h_ready = false;           // w.h already holds this step's value (written by the previous step's tail kernel)
    for (int i = first; i >= 0 && it < n_iters; i -= ratio, ++it) {
        float *nxt = traj ? traj + (size_t)it * (traj_stride ? traj_stride : (size_t)nr * c) : ((it & 1) ? w.xbuf1 : w.xbuf0);
        const int nonneg = (i - ratio) >= 0;
        DdimFuse df;
        df.s = ds; df.mean_type = mean_type; df.ratio = ratio; df.prev_all_nonneg = nonneg; df.t = i; df.x = cur; df.x_prev = nxt;
        df.done = 0;
        df.nx_on = 0; df.nx_done = 0;
        const bool fuse_off = (da::cfg().disable_folds & DA_FOLD_DDIM) != 0;
        const bool try_fuse = plain && !fuse_off && d->variant == DA_VARIANT_2D && !d->prof_on;
        // the tail kernel of this step may also produce the NEXT step's h (embedding + mlp.0 over the hoisted feature part): bf16, the 2D
        // transformer widths, a next step that exists
        // DA_TAIL_NEXT: 1 = every Batch, 0 = never, unset = Batches whose largest graph has >= 512 pieces (DA_STEP_AUTO; 144-piece Batches lose)
        const int nx_mode = da::cfg().tail_next;          // -1 = the large-Batch step rule (step_rule_batch), 0 never, 1 always
        const bool nx_want = nx_mode == 1 || (nx_mode < 0 && step_rule_batch(g));
        if (try_fuse && nx_want && nonneg && it + 1 < n_iters && d->prec == DA_PREC_BF16 && d->hidden == 128 && d->D - d->F == 64 && w.feat_proj &&
            !mfma_disabled()) {
            df.nx_on = 1; df.nx_t = i - ratio; df.nx_steps = d->steps; df.nx_cin = d->c_in; df.nx_ldw = d->D;
            df.nx_time_emb = d->time_emb; df.nx_w0 = d->pos_w0; df.nx_b0 = d->pos_b0; df.nx_w1 = d->pos_w1; df.nx_b1 = d->pos_b1;
            df.nx_wp = (const char *)d->mlp_w0 + (size_t)d->F * esize(d->prec); df.nx_feat_proj = w.feat_proj; df.nx_h = w.h;
        }
        if ((rc = forward_impl(d, g, cur, nullptr, i, w.model_out, nullptr, 0, nullptr, w, st, try_fuse ? &df : nullptr, false, h_ready))) return rc;
        h_ready = df.done && df.nx_done;
        if (df.done) { cur = nxt; continue; }
        if (cfg) {
            if ((rc = forward_impl(d, g, cur, nullptr, i, w.model_out_unc, nullptr, 0, nullptr, w, st, nullptr, true))) return rc;
            const size_t ne = (size_t)nr * c;
            k_cfg_combine<<<(unsigned)((ne + 255) / 256), 256, 0, st>>>(ne, o->cfg_w, w.model_out, w.model_out_unc);
            DA_LAUNCH_CHECK();
        }
        const float *nz = (o && o->noise) ? o->noise + (size_t)it * (noise_stride ? noise_stride : (size_t)nr * c) : nullptr;
        rc = timed(d, DA_PROF_UPDATE, st, [&] {
            if (d->variant == DA_VARIANT_3D)
                return launch_ddim3d(ds, mean_type, nr, cur, w.model_out, nullptr, i, ratio, nonneg, nxt, st);
            if (ddpm)       // t_index == 0 adds no noise (spatial_diffusion.py:504-506)
                return launch_ddpm2d(ds, nr, c, cur, w.model_out, nullptr, i, i == 0 ? nullptr : nz, nxt, st);
            return launch_ddim2d(ds, mean_type, nr, c, cur, w.model_out, nullptr, i, ratio, nonneg, eta, eta > 0.f ? nz : nullptr, nxt, st);
        });
        if (rc) return rc;
        cur = nxt;
    }
    if (x_final && cur != x_final) DA_CHECK_HIP(hipMemcpyAsync(x_final, cur, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

int da_sample_loop_ex(da_denoiser *d, const da_graph *g, const da_schedule *s, int mean_type, int inference_ratio,
                      int max_iters, const float *x_init, float *traj, float *x_final, void *workspace,
                      size_t workspace_bytes, int use_graph, const da_loop_opts *opts, void *stream) {
    DA_REQUIRE(d && g && s && x_init && workspace, "da_sample_loop: null argument");
    da_loop_opts o;
    memset(&o, 0, sizeof(o));
    if (opts) { o.sampler = opts->sampler; o.eta = opts->eta; o.cfg = opts->cfg; o.cfg_w = opts->cfg_w; o.noise = opts->noise; }
    DA_REQUIRE(o.sampler == 0 || o.sampler == 1, "da_sample_loop_ex: sampler must be 0 (DDIM) or 1 (DDPM)");
    DA_REQUIRE(!(o.sampler == 1 || o.eta > 0.f) || o.noise, "da_sample_loop_ex: eta > 0 / DDPM need the noise buffer");
    DA_REQUIRE(d->variant == DA_VARIANT_2D || (!o.cfg && o.sampler == 0 && o.eta == 0.f), "da_sample_loop_ex: the 3D loop has no guidance / stochastic variant");
    DA_REQUIRE(inference_ratio >= 1 && s->steps >= 1, "da_sample_loop: bad ratio/steps");
    DA_REQUIRE(d->variant == DA_VARIANT_3D || d->c_in == d->c_out, "da_sample_loop: c_in != c_out");
    int rc = check_graph(d, g);
    if (rc) return rc;
    Workspace w = carve(d, g, workspace);
    DA_REQUIRE(workspace_bytes >= w.total, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    int total = (s->steps + inference_ratio - 1) / inference_ratio;
    const int n_iters = (max_iters > 0 && max_iters < total) ? max_iters : total;
    if (d->prof_on) use_graph = 0;       // event bracketing is not capturable
    // the CALLER is capturing `stream` (e.g. a predict_step wrapped in torch.cuda.graph): the loop's kernels are enqueued on it directly -- they
    // become nodes of the caller's graph -- instead of launching a graph of the library's own from inside a capture
    if (stream_is_capturing(st)) use_graph = 0;
    if (!use_graph) return enqueue_loop(d, g, s, mean_type, inference_ratio, n_iters, x_init, traj, x_final, w, st, &o);

    LoopKey key;
    memset(&key, 0, sizeof(key));
    key.g = *g; key.s = *s; key.mean_type = mean_type; key.ratio = inference_ratio; key.max_iters = n_iters;
    key.x_init = x_init; key.traj = traj; key.x_final = x_final; key.ws = workspace; key.ws_bytes = workspace_bytes;
    key.opts = o; key.cfg = cfg();
    hipGraphExec_t exec = nullptr;
    for (auto &e : d->loops)
        if (memcmp(&key, &e.key, sizeof(key)) == 0) exec = e.exec;
    if (!exec) {
        if (d->loops.size() >= 8) {                       // small LRU-less cache: drop the oldest (eight: four Batches in flight x two loop lengths)
            (void)hipGraphExecDestroy(d->loops.front().exec);
            d->loops.erase(d->loops.begin());
        }
        hipGraph_t graph = nullptr;
        if (!d->cap_stream) DA_CHECK_HIP(hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking));
        hipStream_t cs = d->cap_stream;
        DA_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
        rc = enqueue_loop(d, g, s, mean_type, inference_ratio, n_iters, x_init, traj, x_final, w, cs, &o);
        hipError_t e = hipStreamEndCapture(cs, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return 2; }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return 2; }
        d->loops.push_back({key, exec});
    }
    DA_CHECK_HIP(hipGraphLaunch(exec, st));
    return 0;
}

int da_sample_loop(da_denoiser *d, const da_graph *g, const da_schedule *s, int mean_type, int inference_ratio,
                   int max_iters, const float *x_init, float *traj, float *x_final, void *workspace,
                   size_t workspace_bytes, int use_graph, void *stream) {
    return da_sample_loop_ex(d, g, s, mean_type, inference_ratio, max_iters, x_init, traj, x_final, workspace, workspace_bytes,
                             use_graph, nullptr, stream);
}

// Two independent halves of one Batch as two parallel branches of ONE hipGraph: each branch is the complete loop of
// da_sample_loop over its own graphs, poses and workspace; the branches share nothing but the weights, so the runtime
// overlaps one half's projections / tail kernels with the other half's attention.
int da_sample_loop_pair_ex(da_denoiser *d, const da_schedule *s, int mean_type, int inference_ratio, int max_iters,
                           const da_graph *g_a, const float *x_init_a, float *x_final_a, void *workspace_a, size_t workspace_a_bytes,
                           const da_graph *g_b, const float *x_init_b, float *x_final_b, void *workspace_b, size_t workspace_b_bytes,
                           float *traj_a, float *traj_b, size_t traj_stride, const da_loop_opts *opts, const float *noise_b,
                           size_t noise_stride, void *stream) {
    DA_REQUIRE(d && s && g_a && g_b && x_init_a && x_init_b && x_final_a && x_final_b && workspace_a && workspace_b,
               "da_sample_loop_pair: null argument");
    // the samplers of da_sample_loop_ex on both branches: one option set, half a reads opts->noise, half b noise_b (row ranges
    // of one [n_iters, N, c] draw: the poses then equal the one-branch loop's bit for bit)
    da_loop_opts oa, ob;
    memset(&oa, 0, sizeof(oa));
    if (opts) { oa.sampler = opts->sampler; oa.eta = opts->eta; oa.cfg = opts->cfg; oa.cfg_w = opts->cfg_w; oa.noise = opts->noise; }
    ob = oa;
    ob.noise = noise_b;
    DA_REQUIRE(oa.sampler == 0 || oa.sampler == 1, "da_sample_loop_pair_ex: sampler must be 0 (DDIM) or 1 (DDPM)");
    DA_REQUIRE(!(oa.sampler == 1 || oa.eta > 0.f) || (oa.noise && ob.noise), "da_sample_loop_pair_ex: eta > 0 / DDPM need both noise pointers");
    DA_REQUIRE(d->variant == DA_VARIANT_2D || (!oa.cfg && oa.sampler == 0 && oa.eta == 0.f), "da_sample_loop_pair_ex: the 3D loop has no guidance / stochastic variant");
    DA_REQUIRE(inference_ratio >= 1 && s->steps >= 1, "da_sample_loop_pair: bad ratio/steps");
    DA_REQUIRE(d->variant == DA_VARIANT_3D || d->c_in == d->c_out, "da_sample_loop_pair: c_in != c_out");
    // hybrid graphs fork the denoiser's side stream (virtual rows) inside every layer: fine when each branch is recorded as a graph of its
    // own (pair_split, the default), a cross-branch dependency through that one stream when both are recorded in one capture
    DA_REQUIRE((!g_a->hybrid && !g_b->hybrid) || da::cfg().pair_split, "da_sample_loop_pair: hybrid graphs need the two-graph form (da_config.pair_split = 1)");
    DA_REQUIRE(!d->prof_on, "da_sample_loop_pair: not available while profiling (event bracketing is not capturable)");
    DA_REQUIRE((traj_a == nullptr) == (traj_b == nullptr), "da_sample_loop_pair_traj: both trajectory pointers or none");
    DA_REQUIRE(!traj_a || traj_stride >= (size_t)(g_a->n_real + g_b->n_real) * (d->variant == DA_VARIANT_3D ? 7 : d->c_in),
               "da_sample_loop_pair_traj: traj_stride smaller than one iteration of both halves");
    int rc = check_graph(d, g_a);
    if (rc) return rc;
    if ((rc = check_graph(d, g_b))) return rc;
    const Workspace wa = carve(d, g_a, workspace_a), wb = carve(d, g_b, workspace_b);
    DA_REQUIRE(workspace_a_bytes >= wa.total && workspace_b_bytes >= wb.total, "da_sample_loop_pair: workspace too small");
    {
        const char *a0 = (const char *)workspace_a, *b0 = (const char *)workspace_b;
        DA_REQUIRE(a0 + wa.total <= b0 || b0 + wb.total <= a0, "da_sample_loop_pair: the two workspaces overlap");
    }
    const int total = (s->steps + inference_ratio - 1) / inference_ratio;
    const int n_iters = (max_iters > 0 && max_iters < total) ? max_iters : total;
    auto make_key = [&](const da_graph *g, const float *xi, float *xf, void *ws, size_t wsb, float *tr) {
        LoopKey k;
        memset(&k, 0, sizeof(k));
        k.g = *g; k.s = *s; k.mean_type = mean_type; k.ratio = inference_ratio; k.max_iters = n_iters;
        k.x_init = xi; k.traj = tr; k.x_final = xf; k.ws = ws; k.ws_bytes = wsb; k.traj_stride = tr ? traj_stride : 0;
        k.cfg = cfg();
        return k;
    };
    LoopKey ka = make_key(g_a, x_init_a, x_final_a, workspace_a, workspace_a_bytes, traj_a);
    LoopKey kb = make_key(g_b, x_init_b, x_final_b, workspace_b, workspace_b_bytes, traj_b);
    ka.opts = oa; kb.opts = ob;
    ka.noise_stride = oa.noise ? noise_stride : 0; kb.noise_stride = ob.noise ? noise_stride : 0;
    // DA_PAIR_SPLIT (default 1): the two branches as TWO graphs, launched on the caller's stream and on the library's pair stream between a
    // fork and a join event, instead of two parallel branches of one graph.  Measured (tools/multi_branch_probe.py, round 5): two independently
    // launched one-branch graphs of 32 puzzles ran a 64-puzzle step in 0.6734 ms where the one-graph pair loop needed 0.7002 on the same box --
    // the runtime does not run the two branches of one graph as independently as it runs two graphs on two streams.
    const int split_mode = cfg().pair_split;
    if (stream_is_capturing((hipStream_t)stream)) {
        // the caller is capturing `stream`: both branches go straight into the caller's capture -- branch A on `stream`, branch B on the library's
        // pair stream between a fork and a join event (two parallel branches of the CALLER's graph) -- no graph of the library's own is launched
        hipStream_t us = (hipStream_t)stream;
        if (!d->pair_stream) {
            DA_CHECK_HIP(hipStreamCreateWithFlags(&d->pair_stream, hipStreamNonBlocking));
            DA_CHECK_HIP(hipEventCreateWithFlags(&d->ev_pair_fork, hipEventDisableTiming));
            DA_CHECK_HIP(hipEventCreateWithFlags(&d->ev_pair_join, hipEventDisableTiming));
        }
        hipStream_t ps = d->pair_stream;
        DA_CHECK_HIP(hipEventRecord(d->ev_pair_fork, us));
        DA_CHECK_HIP(hipStreamWaitEvent(ps, d->ev_pair_fork, 0));
        const int rca = enqueue_loop(d, g_a, s, mean_type, inference_ratio, n_iters, x_init_a, traj_a, x_final_a, wa, us, &oa, traj_stride, noise_stride);
        const int rcb = enqueue_loop(d, g_b, s, mean_type, inference_ratio, n_iters, x_init_b, traj_b, x_final_b, wb, ps, &ob, traj_stride, noise_stride);
        DA_CHECK_HIP(hipEventRecord(d->ev_pair_join, ps));           // (joined even when a branch failed: the capture must not be left forked)
        DA_CHECK_HIP(hipStreamWaitEvent(us, d->ev_pair_join, 0));
        return rca ? rca : rcb;
    }
    hipGraphExec_t exec = nullptr, exec_b = nullptr;
    for (auto &e : d->pair_loops)
        if (memcmp(&ka, &e.a, sizeof(ka)) == 0 && memcmp(&kb, &e.b, sizeof(kb)) == 0 && (e.exec_b != nullptr) == (split_mode != 0)) { exec = e.exec; exec_b = e.exec_b; }
    if (!exec && split_mode) {
        if (d->pair_loops.size() >= 4) {
            (void)hipGraphExecDestroy(d->pair_loops.front().exec);
            if (d->pair_loops.front().exec_b) (void)hipGraphExecDestroy(d->pair_loops.front().exec_b);
            d->pair_loops.erase(d->pair_loops.begin());
        }
        if (!d->cap_stream) DA_CHECK_HIP(hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking));
        if (!d->pair_stream) {
            // DA_PAIR_PRIO (experiment, default 0): 1 = the pair stream at the device's highest priority, -1 = at its lowest -- a standing
            // asymmetry between the branches (one takes free slots first, the other fills its tails) instead of two equals in lockstep
            const int prio = DA_XENV("DA_PAIR_PRIO", 0);
            if (prio) {
                int lo = 0, hi = 0;                                                     // lo = numerically greatest = lowest priority
                DA_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
                DA_CHECK_HIP(hipStreamCreateWithPriority(&d->pair_stream, hipStreamNonBlocking, prio > 0 ? hi : lo));
            } else
                DA_CHECK_HIP(hipStreamCreateWithFlags(&d->pair_stream, hipStreamNonBlocking));
            DA_CHECK_HIP(hipEventCreateWithFlags(&d->ev_pair_fork, hipEventDisableTiming));
            DA_CHECK_HIP(hipEventCreateWithFlags(&d->ev_pair_join, hipEventDisableTiming));
        }
        hipGraphExec_t ex2[2] = {nullptr, nullptr};
        for (int br = 0; br < 2; ++br) {
            hipStream_t cs = d->cap_stream;
            hipGraph_t graph = nullptr;
            DA_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
            const int rcx = br == 0 ? enqueue_loop(d, g_a, s, mean_type, inference_ratio, n_iters, x_init_a, traj_a, x_final_a, wa, cs, &oa, traj_stride, noise_stride)
                                    : enqueue_loop(d, g_b, s, mean_type, inference_ratio, n_iters, x_init_b, traj_b, x_final_b, wb, cs, &ob, traj_stride, noise_stride);
            const hipError_t e2 = hipStreamEndCapture(cs, &graph);
            hipError_t e3 = hipSuccess;
            if (!rcx && e2 == hipSuccess && graph) e3 = hipGraphInstantiate(&ex2[br], graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (rcx || e2 != hipSuccess || e3 != hipSuccess || !ex2[br]) {
                if (ex2[0]) (void)hipGraphExecDestroy(ex2[0]);
                if (rcx) return rcx;
                set_error("da_sample_loop_pair: capture of branch %d failed: %s", br, hipGetErrorString(e2 != hipSuccess ? e2 : e3));
                return 2;
            }
        }
        exec = ex2[0]; exec_b = ex2[1];
        d->pair_loops.push_back({ka, kb, exec, exec_b});
    }
    if (exec && exec_b) {
        hipStream_t us = (hipStream_t)stream, ps = d->pair_stream;
        DA_CHECK_HIP(hipEventRecord(d->ev_pair_fork, us));
        DA_CHECK_HIP(hipStreamWaitEvent(ps, d->ev_pair_fork, 0));
        DA_CHECK_HIP(hipGraphLaunch(exec, us));
        DA_CHECK_HIP(hipGraphLaunch(exec_b, ps));
        DA_CHECK_HIP(hipEventRecord(d->ev_pair_join, ps));
        DA_CHECK_HIP(hipStreamWaitEvent(us, d->ev_pair_join, 0));
        return 0;
    }
    if (!exec) {
        if (d->pair_loops.size() >= 4) {
            (void)hipGraphExecDestroy(d->pair_loops.front().exec);
            if (d->pair_loops.front().exec_b) (void)hipGraphExecDestroy(d->pair_loops.front().exec_b);
            d->pair_loops.erase(d->pair_loops.begin());
        }
        if (!d->cap_stream) DA_CHECK_HIP(hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking));
        if (!d->pair_stream) {
            DA_CHECK_HIP(hipStreamCreateWithFlags(&d->pair_stream, hipStreamNonBlocking));
            DA_CHECK_HIP(hipEventCreateWithFlags(&d->ev_pair_fork, hipEventDisableTiming));
            DA_CHECK_HIP(hipEventCreateWithFlags(&d->ev_pair_join, hipEventDisableTiming));
        }
        hipStream_t cs = d->cap_stream, ps = d->pair_stream;
        hipGraph_t graph = nullptr;
        DA_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
        hipError_t e = hipEventRecord(d->ev_pair_fork, cs);
        if (e == hipSuccess) e = hipStreamWaitEvent(ps, d->ev_pair_fork, 0);                 // ps joins the capture
        int rca = 0, rcb = 0;
        if (e == hipSuccess) {
            // experiment switch (DA_PAIR_DELAY_US, default 0): branch B starts this many microseconds after branch A, so that the two
            // branches run out of phase for the whole loop (one branch's attention beside the other's projections)
            const int delay_us = DA_XENV("DA_PAIR_DELAY_US", 0);
            if (delay_us > 0) k_pair_delay<<<1, 64, 0, ps>>>((unsigned long long)delay_us * 100ull);
            rca = enqueue_loop(d, g_a, s, mean_type, inference_ratio, n_iters, x_init_a, traj_a, x_final_a, wa, cs, &oa, traj_stride, noise_stride);
            rcb = enqueue_loop(d, g_b, s, mean_type, inference_ratio, n_iters, x_init_b, traj_b, x_final_b, wb, ps, &ob, traj_stride, noise_stride);
            e = hipEventRecord(d->ev_pair_join, ps);
            if (e == hipSuccess) e = hipStreamWaitEvent(cs, d->ev_pair_join, 0);
        }
        const hipError_t e2 = hipStreamEndCapture(cs, &graph);
        if (rca || rcb) { if (graph) (void)hipGraphDestroy(graph); return rca ? rca : rcb; }
        if (e != hipSuccess || e2 != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            set_error("da_sample_loop_pair: capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
            return 2;
        }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return 2; }
        d->pair_loops.push_back({ka, kb, exec, nullptr});
    }
    DA_CHECK_HIP(hipGraphLaunch(exec, (hipStream_t)stream));
    return 0;
}

int da_sample_loop_pair_traj(da_denoiser *d, const da_schedule *s, int mean_type, int inference_ratio, int max_iters,
                             const da_graph *g_a, const float *x_init_a, float *x_final_a, void *workspace_a, size_t workspace_a_bytes,
                             const da_graph *g_b, const float *x_init_b, float *x_final_b, void *workspace_b, size_t workspace_b_bytes,
                             float *traj_a, float *traj_b, size_t traj_stride, void *stream) {
    return da_sample_loop_pair_ex(d, s, mean_type, inference_ratio, max_iters, g_a, x_init_a, x_final_a, workspace_a, workspace_a_bytes,
                                  g_b, x_init_b, x_final_b, workspace_b, workspace_b_bytes, traj_a, traj_b, traj_stride, nullptr, nullptr, 0,
                                  stream);
}

int da_sample_loop_pair(da_denoiser *d, const da_schedule *s, int mean_type, int inference_ratio, int max_iters,
                        const da_graph *g_a, const float *x_init_a, float *x_final_a, void *workspace_a, size_t workspace_a_bytes,
                        const da_graph *g_b, const float *x_init_b, float *x_final_b, void *workspace_b, size_t workspace_b_bytes,
                        void *stream) {
    return da_sample_loop_pair_traj(d, s, mean_type, inference_ratio, max_iters, g_a, x_init_a, x_final_a, workspace_a, workspace_a_bytes,
                                    g_b, x_init_b, x_final_b, workspace_b, workspace_b_bytes, nullptr, nullptr, 0, stream);
}

size_t da_attn_dense_scratch_bytes(int prec, const da_graph *g, int heads, int C) {
    if (!g) return 0;
    const size_t s = esize(prec), hc = (size_t)heads * C;
    return 3 * align_up(((size_t)g->n_pad + 64) * hc * s, 256) + align_up(((size_t)g->n_nodes + 64) * hc * s, 256);
}

int da_conv_dense(int prec, const da_graph *g, int heads, int C, int Din, const void *x, const void *w,
                  const float *b, const void *residual, int act, void *out, void *scratch, void *stream) {
    DA_REQUIRE(g && x && w && out && scratch, "da_conv_dense: null argument");
    DA_REQUIRE(dense_ok(g, heads, C), "da_conv_dense: graph is not dense or head width %d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    const size_t s = esize(prec), hc = (size_t)heads * C;
#ifdef DA_EXPERIMENTS
    if (!residual && !g->hybrid && conv_fused_applicable(prec, heads, C, Din, g->max_graph_nodes, (int)hc)) {
        int rf = launch_conv_fused(prec, heads, C, Din, g->n_graphs, g->max_graph_nodes, g->graph_ptr, g->dense == 2, x, Din, w,
                                   b, act, out, (int)hc, st);
        DA_REQUIRE(rf == 0, "da_conv_dense: fused conv launch failed (%d)", rf);
        return 0;
    }
#endif
    const size_t hb = align_up(((size_t)g->n_pad + 64) * hc * s, 256);
    char *base = (char *)scratch;
    QkvScatter qs;
    qs.HC = (int)hc; qs.C = C; qs.n_pad = g->n_pad; qs.row_map = g->row_map;
    qs.Q = base; qs.K = base + hb; qs.Vt = base + 2 * hb; qs.S = base + 3 * hb;
    int rc = launch_gemm_mfma(prec, g->n_nodes, Din, 4 * (int)hc, x, Din, w, b, DA_ACT_NONE, nullptr, nullptr, 0, &qs, st);
    DA_REQUIRE(rc <= 0, "da_conv_dense: projection launch failed");
    DA_REQUIRE(rc == 0, "da_conv_dense: projection shape (Din=%d, HC=%d) not supported by the MFMA kernel", Din, (int)hc);
    DenseLayout L;
    L.Q = qs.Q; L.K = qs.K; L.Vt = qs.Vt; L.S = qs.S; L.n_pad = g->n_pad;
    rc = launch_attn_dense(prec, L, heads, C, g->n_graphs, g->max_graph_nodes, g->graph_ptr, g->pad_ptr,
                           g->dense == 2, residual, act, out, st);
    DA_REQUIRE(rc == 0, "da_conv_dense: attention launch failed (%d)", rc);
    return 0;
}

int da_conv_dense_ex(int prec, const da_graph *g, int heads, int C, int Din, const void *x, const void *w, const float *b,
                     const void *residual, int act, void *out, void *scratch, int flags, void *stream) {
    DA_REQUIRE(g && x && w && out && scratch, "da_conv_dense_ex: null argument");
    DA_REQUIRE(dense_ok(g, heads, C), "da_conv_dense_ex: graph is not dense or head width %d unsupported", C);
    const bool folded = (flags & DA_CONV_FOLDED_V32) != 0;
    DA_REQUIRE(!folded || (C == 144 && !residual && act == DA_ACT_NONE), "da_conv_dense_ex: folded value heads need C = 144, no residual, no activation");
    hipStream_t st = (hipStream_t)stream;
    const size_t s = esize(prec), hc = (size_t)heads * C;
    const size_t hb = align_up(((size_t)g->n_pad + 64) * hc * s, 256);
    char *base = (char *)scratch;
    QkvScatter qs;
    qs.HC = (int)hc; qs.C = C; qs.n_pad = g->n_pad; qs.row_map = g->row_map;
    qs.Q = base; qs.K = base + hb; qs.Vt = base + 2 * hb; qs.S = folded ? nullptr : base + 3 * hb;
    qs.Cv = folded ? 32 : 0;
    const int nout = folded ? 2 * (int)hc + heads * 32 : 4 * (int)hc;
    // the layer form the sampling loop takes for hidden layers of large complete graphs (run_forward): the projection happens in the prologue of
    // the resident attention kernel; its per-head weight fragments are packed into the scratch's (then unused) K region
    const bool qsf = !folded && !residual && !g->hybrid && g->n_nodes == g->n_real && b && w_qs_bytes(heads, Din) <= hb &&
                     attn_qsf_applicable(prec, heads, C, Din, g->n_graphs, g->max_graph_nodes, g->n_pad, (flags & DA_CONV_Q_PRESCALED) ? 1 : 0);
    void *wq_img = base + hb;                        // (the K region of the scratch: K | V stay in the CU)
    int rc = 0;
    if (qsf) {
        if ((rc = pack_w_qs(heads, Din, (int)hc, w, wq_img, st))) return rc;
    } else
    rc = launch_gemm_mfma(prec, g->n_nodes, Din, nout, x, Din, w, b, DA_ACT_NONE, nullptr, nullptr, 0, &qs, st);
    DA_REQUIRE(rc == 0, "da_conv_dense_ex: projection shape (Din=%d, Nout=%d) not supported by the MFMA kernels", Din, nout);
    DenseLayout L;
    L.Q = qs.Q; L.K = qs.K; L.Vt = qs.Vt; L.S = qs.S; L.n_pad = g->n_pad; L.q_prescaled = (flags & DA_CONV_Q_PRESCALED) ? 1 : 0;
    if (qsf) { L.x = x; L.ldx = Din; L.kin = Din; L.wqs = wq_img; L.bias = b; }
    DenseFold fo;
    fo.cv = 32; fo.out = out; fo.n_rows = g->n_real;
    const DenseMask mk = dense_mask_of(g);
    rc = launch_attn_dense(prec, L, heads, C, g->n_graphs, g->max_graph_nodes, g->graph_ptr, g->pad_ptr, g->dense == 2, residual, act,
                           folded ? nullptr : out, st, g->hybrid ? &mk : nullptr, folded ? &fo : nullptr);
    DA_REQUIRE(rc == 0, "da_conv_dense_ex: attention launch failed (%d)", rc);
    return 0;
}

int da_debug_counters(int64_t *out, int n, int reset) {
    DA_REQUIRE(out && n >= DA_DBG_NCOUNTERS, "da_debug_counters: need room for %d counters", DA_DBG_NCOUNTERS);
    unsigned long long a[4] = {0, 0, 0, 0}, b2[2] = {0, 0};
    int rc;
    if ((rc = da::attn_dense_counters(a, reset))) return rc;
#ifdef DA_EXPERIMENTS
    if ((rc = da::attn_dual_counters(b2, reset))) return rc;          // (k_attn_dual: experiments build only; the counter reads 0 otherwise)
#endif
    for (int k = 0; k < n; ++k) out[k] = 0;
    out[DA_DBG_OPT_GEN_WORKGROUPS] = (int64_t)a[0];
    out[DA_DBG_DENSE_FAST_EXITS] = (int64_t)a[1];
    out[DA_DBG_DUAL_GEN_SLABS] = (int64_t)b2[0];
    out[DA_DBG_OPT_MASKED_GEN_WORKGROUPS] = (int64_t)a[2];
    out[DA_DBG_RES_LAUNCHES] = (int64_t)da::attn_res_launches(reset);
    out[DA_DBG_VIRT_IN_LAUNCH] = (int64_t)da::attn_virt_launches(reset);
    return 0;
}

int da_profile_enable(da_denoiser *d, int on) {
    DA_REQUIRE(d, "da_profile_enable: null denoiser");
    d->prof_on = on != 0;
    d->prof_used = 0;
    d->prof_cls.clear();
    return 0;
}

int da_profile_read(da_denoiser *d, float *ms, int32_t *counts) {
    DA_REQUIRE(d && ms && counts, "da_profile_read: null argument");
    for (int k = 0; k < DA_PROF_NCLASS; ++k) { ms[k] = 0.f; counts[k] = 0; }
    for (size_t k = 0; k < d->prof_cls.size(); ++k) {
        DA_CHECK_HIP(hipEventSynchronize(d->prof_ev[2 * k + 1]));
        float t = 0.f;
        DA_CHECK_HIP(hipEventElapsedTime(&t, d->prof_ev[2 * k], d->prof_ev[2 * k + 1]));
        ms[d->prof_cls[k]] += t;
        counts[d->prof_cls[k]] += 1;
    }
    d->prof_used = 0;
    d->prof_cls.clear();
    return 0;
}

int da_linear(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias, int act,
              const void *residual, void *out, int ldo, void *stream) {
    DA_REQUIRE(A && W && out, "da_linear: null argument");
    return da::linear(prec, M, K, Nout, A, lda, W, bias, act, residual, out, ldo, (hipStream_t)stream);
}

size_t da_linear_packed_bytes(int prec, int K, int Nout) {
    if (prec != DA_PREC_BF16 || (K != 128 && K != 256) || Nout < 512 || Nout > 4096 || (Nout & 31)) return 0;
    return da::xpanel_packed_bytes(K, Nout);
}

int da_linear_pack(int prec, int K, int Nout, const void *W, int ldw, void *packed, void *stream) {
    DA_REQUIRE(W && packed, "da_linear_pack: null argument");
    DA_REQUIRE(da_linear_packed_bytes(prec, K, Nout) > 0, "da_linear_pack: no packed form for prec %d, K %d, Nout %d", prec, K, Nout);
    DA_REQUIRE((((size_t)W) & 15) == 0 && (((size_t)packed) & 15) == 0 && ((ldw <= 0 ? K : ldw) & 7) == 0, "da_linear_pack: W / packed must be 16-byte aligned");
    return da::pack_w_xpanel(K, Nout, W, ldw, packed, (hipStream_t)stream);
}

int da_linear_packed(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const void *packed,
                     const float *bias, int act, const void *residual, void *out, int ldo, void *stream) {
    DA_REQUIRE(A && W && out, "da_linear_packed: null argument");
    if (!da::mfma_disabled()) {
        const int rc = da::launch_gemm_mfma(prec, M, K, Nout, A, lda, W, bias, act, residual, out, ldo, nullptr, (hipStream_t)stream, 0, nullptr,
                                            packed);
        if (rc >= 0) return rc;
    }
    return da::linear(prec, M, K, Nout, A, lda, W, bias, act, residual, out, ldo, (hipStream_t)stream);
}

int da_attn_csr(int prec, const da_graph *g, int heads, int C, const void *qkvs, const void *residual, int act,
                void *out, float *alpha, void *stream) {
    DA_REQUIRE(g && qkvs && out, "da_attn_csr: null argument");
    // (as in the forward: tiny complete graphs at C = 104 take k_attn_tiny when no attention weights are asked for)
    if (!alpha && g->dense && g->n_nodes == g->n_real && g->graph_ptr && !dense_disabled()) {
        const int rt = launch_attn_tiny(prec, g->n_graphs, g->max_graph_nodes, g->graph_ptr, g->dense == 2, heads, C, qkvs, residual, act, out, (hipStream_t)stream);
        if (rt >= 0) return rt;
    }
    return launch_attn_csr(prec, g->n_nodes, g->row_ptr, g->col_src, g->edge_id, heads, C, qkvs, residual, act, out,
                           alpha, nullptr, (hipStream_t)stream);
}

}  // extern "C"
