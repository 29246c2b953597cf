// Internal launcher declarations shared by the translation units of libdiffassemble_hip.so.
#pragma once
#include "da_common.h"
#include "da_config.h"

namespace da {

struct DeviceSchedule {
    int steps;
    const float *betas, *alphas_cumprod, *sqrt_recip_alphas, *sqrt_recip_alphas_cumprod,
        *sqrt_recipm1_alphas_cumprod, *sqrt_one_minus_alphas_cumprod, *posterior_variance;
};

// DDIM update of one element, p_sample_ddim spatial_diffusion.py:555-566,603-627 (fp32, op order kept); shared by
// k_ddim2d and the fused tail of the folded head (k_head_fold), so both produce the same bits.
__device__ __forceinline__ float ddim2d_value(const DeviceSchedule &s, int mean_type, long long ti, int ratio,
                                              int prev_all_nonneg, float eta, float xv, float m, float noise) {
    ti = ti < 0 ? 0 : (ti >= s.steps ? s.steps - 1 : ti);
    const long long tp = ti - ratio;
    const float ap = s.alphas_cumprod[ti];
    const float ap_prev = (prev_all_nonneg && tp >= 0) ? s.alphas_cumprod[tp] : 1.0f;
    const float beta = 1.0f - ap;
    const float x0 = mean_type == DA_MEAN_START_X ? m : (xv - sqrtf(beta) * m) / sqrtf(ap);
    const float eps = (s.sqrt_recip_alphas_cumprod[ti] * xv - x0) / s.sqrt_recipm1_alphas_cumprod[ti];
    const float var = ((1.0f - ap_prev) / (1.0f - ap)) * (1.0f - ap / ap_prev);
    const float std_eta = eta * sqrtf(var);
    float prev = sqrtf(ap_prev) * x0 + sqrtf(1.0f - ap_prev - std_eta * std_eta) * eps;
    if (eta > 0.f) prev += std_eta * noise;
    return prev;
}

// Sampling loop: the deterministic DDIM update (eta = 0, scalar timestep) applied by the head kernel itself --
// one launch less per step.  `done` is set by forward_impl when the folded head took it.
struct DdimFuse {
    DeviceSchedule s;
    int mean_type, ratio, prev_all_nonneg;
    long long t;
    const float *x;            // [n, c] current sample (c == c_out)
    float *x_prev;             // [n, c]
    int done;
    // the NEXT step's embedding + mlp.0 computed by the tail kernel itself (k_tail_fused, plain DDIM loops): the rows of x_prev never leave the
    // registers before they become the next step's h = GELU(mlp.0([features | pos_mlp(x_prev) | time_emb[t - ratio]])) -- per-row work that
    // used to be two more launches per step (k_embed_pos_time + the K = 64 GEMM over the hoisted feature part).  nx_on: requested by the
    // loop; nx_done: the kernel that ran did it (the next forward skips its embedding and mlp.0).
    int nx_on, nx_done, nx_steps, nx_cin, nx_ldw;
    long long nx_t;
    const float *nx_time_emb, *nx_w0, *nx_b0, *nx_w1, *nx_b1;
    const void *nx_wp;         // bf16 mlp.0 weight, columns F .. F + 63 (row stride nx_ldw elements)
    const void *nx_feat_proj;  // bf16 [n, 128]: mlp.0 over the feature columns + bias (hoisted)
    void *nx_h;                // bf16 [n, 128]: written in place (the rows this wave has already consumed)
};

// da_basic.hip
int launch_set_feats(int prec, int n, int F, int D, const float *feats, void *comb_in, hipStream_t st);
int launch_set_virtual_rows(int prec, int rows, int V, int D, const void *emb, void *dst, hipStream_t st);
// the constant conv-0 projections of the exophormer's virtual rows (launch_scatter_virtual's arguments): given to launch_embed_pos_time they are
// placed by extra workgroups of the embedding's launch -- the first kernel of a step -- instead of a launch of their own behind the projection
struct VirtScatter {
    int rows = 0, V = 0, H = 0, C = 0, n_real = 0, n_pad = 0;
    const void *src = nullptr;
    const int32_t *row_map = nullptr;
    void *Q = nullptr, *K = nullptr, *Vt = nullptr, *S = nullptr;
};
int launch_embed_pos_time(int prec, int n, int c_in, int F, int D, const float *x, const int64_t *t, int64_t t_scalar,
                          int steps, const float *time_emb, const float *w0, const float *b0, const float *w1,
                          const float *b1, void *comb_in, hipStream_t st, const VirtScatter *vs = nullptr);
int launch_gemm_simple(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                       int act, const void *res, void *out, int ldo, hipStream_t st);
int launch_head2d(int prec, int n, int c_out, const void *hh, const float *w2, const float *b2, float *out, hipStream_t st);
int launch_ddim2d(const DeviceSchedule &s, int mean_type, int n, int c, const float *x, const float *mo,
                  const int64_t *t, int64_t t_scalar, int ratio, int prev_all_nonneg, float eta, const float *noise,
                  float *x_prev, hipStream_t st);
int launch_ddpm2d(const DeviceSchedule &s, int n, int c, const float *x, const float *mo, const int64_t *t,
                  int64_t t_scalar, const float *noise, float *x_prev, hipStream_t st);
int launch_convert(int prec, size_t n, const float *src, void *dst, hipStream_t st);
int launch_transpose_f32(int rows, int cols, const float *src, float *dst, hipStream_t st);      // da_train.hip
// C (+)= A^T B over rows (A [M, N], B [M, K] row-major; split-row partials through `partial`, >= 16 M floats)
int launch_gemm_tn(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, float *partial,
                   hipStream_t st, bool bfc = false);      // bfc: operands rounded to bf16 inside the kernel (DA_TRAIN_MMA_BF16)
int launch_gemm_tn_bf16(int M, int N, int K, const bf16_t *A, int lda, const bf16_t *B, int ldb, float *C, int ldc, float *partial,
                        hipStream_t st);
int launch_gemm_mfma_splitk(int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias, const void *res,
                            void *out, int ldo, float *partial, size_t partial_floats, hipStream_t st, bool in16 = false,
                            const float *gelu_pre = nullptr, float *act_out = nullptr);   // da_gemm_mfma.hip; -1: shape not taken.  gelu_pre: out *= gelu'(pre); act_out = gelu(out)
int launch_gemm_mfma_mixed(bool in16, bool out16, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                           const void *res, void *out, int ldo, hipStream_t st);                              // fp32 x fp32 -> bf16 / bf16 x bf16 -> fp32
int launch_gemm_tn_db(int M, int N, int K, const void *A, int lda, const void *B, int ldb, float *C, int ldc, float *partial,
                      float *db, float *bscratch, hipStream_t st, bool a16 = false, bool b16 = false);                                            // dW + db, bf16-operand mode
int colsum_add(int M, int N, const float *A, int lda, float *out, float *scratch, hipStream_t st);   // out[c] += sum_m A[m][c]
// da_encoder.hip: implicit-GEMM convolution over zero-haloed NHWC maps; the fp32 stem
int launch_conv(int prec, int B, const void *X, int Cin, int Hi, const void *W, const float *bias, const void *res, void *Y,
                int Cout, int ksize, int stride, int relu, hipStream_t st);
int launch_enc_stem(int prec, int B, const float *patches, const float *w, const float *bias, void *Y, int relu, hipStream_t st);
int launch_mm_nn_f32(int M, int N, int K, const float *A, int lda, const float *B, int ldb, const float *bias, float *C,
                     int ldc, hipStream_t st);
int launch_scatter_virtual(int prec, int rows, int V, int H, int C, const void *src, int n_real, const int32_t *row_map,
                           int n_pad, void *Q, void *K, void *Vt, void *S, void *qkvs, hipStream_t st);
int launch_tail_fused(int prec, int n, int H, int c_out, int hidden, int din, const void *h, const void *xin, int ldx, const void *wh,
                      const float *bh, const void *wsk, const float *bsk, const void *pz, const float *w2, const float *b2, float *out,
                      hipStream_t st, DdimFuse *dfp);
int launch_head_fold(int prec, int n, int H, int c_out, const void *pz, const void *pre, const float *w2, const float *b2,
                     float *out, hipStream_t st, const DdimFuse *df = nullptr);

// da_attn_csr.hip
// tiny complete graphs (<= 32 pieces) at C = 104: K | V of a graph staged in LDS (da_attn_csr.hip); 0 = launched, -1 = shape not covered
int launch_attn_tiny(int prec, int n_graphs, int max_graph_nodes, const int32_t *graph_ptr, int nodiag, int heads, int C, const void *qkvs,
                     const void *residual, int act, void *out, hipStream_t st);
int launch_attn_csr(int prec, int n_nodes, const int32_t *row_ptr, const int32_t *col_src, const int32_t *edge_id,
                    int heads, int C, const void *qkvs, const void *residual, int act, void *out, float *alpha,
                    float *stats /* [n, H, 2] running max and 1/(sum + 1e-16), or NULL */, hipStream_t st);

// da_so3.hip (3D head + SO(3) DDIM)
int launch_head3d(int prec, int n, const void *hh, const float *wt, const float *bt, const float *wr, const float *br,
                  float *out7, float *pre_head, hipStream_t st);
int launch_ddim3d(const DeviceSchedule &s, int mean_type, int n, const float *x, const float *mo, const int64_t *t,
                  int64_t t_scalar, int ratio, int prev_all_nonneg, float *x_prev, hipStream_t st);

// da_gemm_mfma.hip / da_attn_dense.hip (dense block-diagonal path)
// Joins a library-owned side stream back into the caller's stream on EVERY exit of the scope that forked it (an early
// `return rc` between fork and join left side-stream work un-joined while the caller freed or rewrote the workspace).  The success
// path calls join() itself (and sees its errors); the destructor covers the error exits, best effort.
struct StreamJoin {
    hipStream_t side = nullptr, caller = nullptr;
    hipEvent_t ev = nullptr;
    bool armed = false;
    void arm(hipStream_t side_, hipStream_t caller_, hipEvent_t ev_) { side = side_; caller = caller_; ev = ev_; armed = true; }
    hipError_t join() {
        if (!armed) return hipSuccess;
        armed = false;
        hipError_t e = hipEventRecord(ev, side);
        return e == hipSuccess ? hipStreamWaitEvent(caller, ev, 0) : e;
    }
    ~StreamJoin() { (void)join(); }
};
struct QkvScatter {            // where the fused projection scatters its four column blocks
    int HC, C, n_pad;
    int Cv = 0;                // > 0: three blocks only (Q | K | V'), V' heads Cv wide (folded value heads), no skip
    int blocks = 0;            // > 0: only the first `blocks` column blocks exist (2: a K | V-only projection hands its blocks in through the Q and K slots)
    const int32_t *row_map;    // node -> padded row
    void *Q, *K, *Vt, *S;      // [H][n_pad][C] x 3 (Vt keeps its name; V is row-major since the tr_b16 rewrite), [M][H*C]
};
struct DenseLayout {
    const void *Q, *K, *Vt, *S;
    int n_pad;
    int q_prescaled = 0;       // Q rows already carry log2(e) / sqrt(C) (projection weights scaled at pack time, da_api.hip ConvW::wd)
    // the layer's projection inside the resident attention kernel (attn_qsf_applicable; no projection kernel runs: the attention kernel writes Q / S
    // itself and keeps K | V in LDS):
    const void *x = nullptr;   // the layer's input rows [N][ldx], bf16
    int ldx = 0, kin = 0;
    const void *wqs = nullptr; // pack_w_qs image of the layer's projection weights
    const float *bias = nullptr;   // [4 * H * 32]: Q | K | V | skip
};
// da_attn_dense.hip / da_attn_opt.hip: whether launch_attn_dense will take the resident kernel's projection-in-the-prologue form for this layer
bool attn_qsf_applicable(int prec, int heads, int C, int kin, int n_graphs, int max_graph_nodes, int n_pad, int q_prescaled);
size_t w_qs_bytes(int heads, int kin);
int pack_w_qs(int heads, int kin, int hc, const void *wd, void *packed, hipStream_t st);
// wpacked: optional fragment-major copy of W (pack_w_xpanel) for the row-panel kernel (da_gemm_xpanel.hip)
int launch_gemm_mfma(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias,
                     int act, const void *res, void *out, int ldo, const QkvScatter *qs, hipStream_t st, int ldw = 0,
                     const void *pre = nullptr, const void *wpacked = nullptr);
size_t xpanel_packed_bytes(int K, int Nout);
bool xpanel_in_model();          // DA_ENABLE_XPANEL=1 (off by default, see da_gemm_xpanel.hip)
void gemm_thin_set(int v);      // run-time form of DA_GEMM_THIN (da_gemm_mfma.hip): 1 = projections take the co-resident kernel of da_gemm_thin.hip
int xpanel_mode();               // 0 never, 1 always, 2 = Batches whose largest graph has >= 512 pieces (the default, DA_STEP_AUTO)
int pack_w_xpanel(int K, int Nout, const void *W, int ldw, void *packed, hipStream_t st);
struct DenseMask {             // hybrid mode: adjacency bits of the regular edges + the remainder CSR (da_attn_dense.hip)
    const uint8_t *mask;
    const int64_t *mask_ptr;
    const int32_t *irr_row_ptr, *irr_col_src, *row_map;
    const int32_t *slot_node = nullptr;            // banded layout: slot -> node (da_graph.slot_node)
    const uint8_t *blk_class = nullptr;            // block classes (da_graph.blk_class*)
    const int64_t *blk_class_ptr = nullptr;
    int blk_class_stride = 0;
    const int32_t *rm_meta = nullptr;              // per-slot remainder metadata (da_graph.rm_meta)
    // the virtual rows of the exophormer arch inside the masked attention's launch (AttnDenseParams::v_*): set v_rows > 0 to ask for it;
    // *v_taken is set to 1 by the launcher whose kernel took them (else the caller launches launch_attn_csr_cont itself)
    int v_rows = 0, v_n_real = 0;
    const int32_t *v_row_ptr = nullptr, *v_col_src = nullptr;
    const float *v_mult = nullptr;
    float *v_part = nullptr;
    unsigned *v_cnt = nullptr;
    int *v_taken = nullptr;
};
constexpr int DA_VIRT_SPLIT_MAX = 8;               // workgroups per virtual row at most (sizes the scratch)
inline DenseMask dense_mask_of(const da_graph *g) {
    DenseMask mk;
    mk.mask = g->mask; mk.mask_ptr = g->mask_ptr; mk.irr_row_ptr = g->irr_row_ptr; mk.irr_col_src = g->irr_col_src; mk.row_map = g->row_map;
    mk.slot_node = g->slot_node;
    mk.rm_meta = g->rm_meta;
    const bool cls = g->blk_class && g->blk_class_ptr && g->blk_class_stride > 0;
    mk.blk_class = cls ? g->blk_class : nullptr; mk.blk_class_ptr = cls ? g->blk_class_ptr : nullptr; mk.blk_class_stride = cls ? g->blk_class_stride : 0;
    return mk;
}
struct DenseFold {             // value heads folded with the next linear layer: V is [H][n_pad][cv], output per head
    int cv;
    void *out;                 // [H][n_rows][cv] in the activation dtype, normalised
    int n_rows;
};
int launch_attn_dense(int prec, const DenseLayout &L, int heads, int C, int n_graphs, int max_graph_nodes,
                      const int32_t *graph_ptr, const int32_t *pad_ptr, int nodiag, const void *res, int act,
                      void *out, hipStream_t st, const DenseMask *mk = nullptr, const DenseFold *fold = nullptr);
// da_attn_dual.hip: bf16, complete graphs, two query slabs per wave; Q must arrive pre-scaled by log2(e) / sqrt(C)
int launch_attn_dual(const DenseLayout &L, int heads, int C, int n_graphs, int max_graph_nodes, const int32_t *graph_ptr,
                     const int32_t *pad_ptr, int nodiag, int act, void *out, const DenseFold *fold, hipStream_t st);
// fallback counters of the shift-free softmax kernels (da_debug_counters)
int attn_dense_counters(unsigned long long *out4, int reset);
long long attn_virt_launches(int reset);     // da_attn_opt.hip: masked hidden-layer launches that carried the virtual rows
long long attn_res_launches(int reset);      // da_attn_opt.hip: launches of the K / V-resident kernel since the last reset
int attn_dual_counters(unsigned long long *out2, int reset);
// hybrid mode: the rows the masked kernel does not own (virtual nodes) over their remainder edges
int launch_attn_csr_cont(int prec, int n_nodes, int n_real, const int32_t *irr_row_ptr, const int32_t *irr_col_src,
                         const int32_t *row_map, int heads, int C, int n_pad, const DenseLayout &L,
                         const void *residual, int act, void *out, hipStream_t st,
                         const float *mult = nullptr /* multiplicity per (aggregated) remainder edge, or NULL = 1 */);

// da_conv_fused.hip: one hidden conv (projection + attention of a (graph, head)) as ONE kernel, K / V resident in LDS
bool conv_fused_applicable(int prec, int heads, int C, int kin, int max_graph_nodes, int ldo);
int launch_conv_fused(int prec, int heads, int C, int kin, int n_graphs, int max_graph_nodes, const int32_t *graph_ptr,
                      int nodiag, const void *x, int ldx, const void *W, const float *bias, int act, void *out, int ldo,
                      hipStream_t st);

// generic linear dispatch (MFMA when the shape allows, else simple)
int linear(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const float *bias, int act,
           const void *res, void *out, int ldo, hipStream_t st);

}  // namespace da
