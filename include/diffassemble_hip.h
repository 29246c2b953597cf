/*
 * diffassemble_hip.h -- C ABI of libdiffassemble_hip.so (gfx950 / MI355X).
 *
 * The reference (IIT-PAVIS/DiffAssemble) is 100 % Python and has NO FFI layer for this
 * path: its "operator interface" is Python-level.  This header is therefore
 * the boundary the build defines.  Each entry point cites the reference code it replaces
 * (paths relative to /root/reference/puzzle_diff/model/).  Conventions:
 *
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *   - the library BORROWS caller (PyTorch) storage for the duration of a call and never
 *     frees it; the only memory it owns are the packed weights of a da_denoiser;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises, nothing allocates inside a forward/step call (the caller passes a
 *     workspace sized by da_denoiser_workspace_bytes), so calls are stream-capturable;
 *   - every function returns 0 on success, nonzero on error (da_last_error() has the text);
 *   - re-entrant per (denoiser, workspace, stream).
 *
 * Tensor layouts are row-major.  "act dtype" is fp32 in DA_PREC_F32 (parity mode,
 * 1e-4 relative vs the fp32 oracle) and bf16 in DA_PREC_BF16 (perf mode, fp32 accumulate).
 */
#ifndef DIFFASSEMBLE_HIP_H
#define DIFFASSEMBLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DA_ABI_VERSION 19

enum { DA_PREC_F32 = 0, DA_PREC_BF16 = 1 };
enum { DA_VARIANT_2D = 0, DA_VARIANT_3D = 1 };          /* Eff_GAT / Eff_GAT_3d            */
enum { DA_ARCH_TRANSFORMER = 0, DA_ARCH_EXOPHORMER = 1 }; /* GELU between convs / none      */
enum { DA_MEAN_EPSILON = 0, DA_MEAN_START_X = 1 };       /* spatial_diffusion.py:63-66      */
enum { DA_ACT_NONE = 0, DA_ACT_GELU = 1, DA_ACT_LEAKY02 = 2 };
enum { DA_MAX_LAYERS = 8 };

int da_abi_version(void);
const char *da_last_error(void);

/* ---------------------------------------------------------------------------------------
 * The library's switches (ABI 19).  The reference has no equivalent (its knobs are Python
 * kwargs); these replace what used to be ~70 getenv calls spread over the kernels' launchers.
 * Every field is initialised ONCE per process from the environment variable named beside
 * it, can be read / changed at run time (da_config_get / da_config_set: in-process A/B
 * measurements, tests), and is consulted when a call is made -- a denoiser keeps the folds
 * it was created with, a captured sampling-loop graph keeps the kernels it recorded.
 * No other environment variable changes what the default build of the library does:
 * A/B variants that lost, ablations and probes are compiled only into the EXPERIMENTS build
 * (DA_EXPERIMENTS=1 python __graft_entry__.py; da_build_flags() bit 0), see INTEGRATION.md.
 * ------------------------------------------------------------------------------------- */
typedef struct da_config {
    int32_t struct_bytes;        /* sizeof(da_config) as the caller compiled it (checked by da_config_set)          */
    int32_t disable_mfma;        /* DA_DISABLE_MFMA=1: generic (non matrix-core) linear kernels                      */
    int32_t disable_dense;       /* DA_DISABLE_DENSE=1: every attention walks the edge list (CSR kernels)            */
    int32_t disable_folds;       /* DA_DISABLE_FOLDS=<bits>: 1 mlp.2 composed into its consumers, 2 folded last      */
                                 /* layer, 4 softmax scale inside Wq, 8 DDIM update inside the head kernel,          */
                                 /* 16 one-kernel tail, 32 virtual rows on a side stream / inside the masked         */
                                 /* attention's launch (hybrid graphs)                                               */
    int32_t attn_level;          /* DA_ATTN_LEVEL: 0 = the general dense kernel only, 1 = + the optimistic ring      */
                                 /* kernels, 2 = + the K/V-resident hidden-layer kernel (default)                    */
    int32_t xpanel;              /* DA_ENABLE_XPANEL: -1 = row-panel projections for large Batches (largest graph    */
                                 /* >= 512 pieces or >= 16 384 pieces in all; default), 0 never, 1 always            */
    int32_t tail_next;           /* DA_TAIL_NEXT: -1 = next step's embedding inside the tail kernel for the same     */
                                 /* large Batches (default), 0 never, 1 always                                       */
    int32_t pair_split;          /* DA_PAIR_SPLIT: 1 = the pair loop as two graphs on two streams (default), 0 = one */
    int32_t train_attn;          /* DA_TRAIN_ATTN: 0 = edge-list kernels only, 1 = grouped-GEMM attention with pair  */
                                 /* matrices, 2 = + flash-style hybrid kernels in the bf16-operand mode (default)    */
    int32_t train_side_streams;  /* DA_TRAIN_SIDE_STREAMS bits: 1 weight-gradient products, 2 hybrid virtual rows    */
} da_config;
int da_config_get(da_config *out /* host */);
int da_config_set(const da_config *in /* host */);
int da_build_flags(void);        /* bit 0: EXPERIMENTS build                                                         */

/* ---------------------------------------------------------------------------------------
 * Weights of one denoiser, fp32 device pointers in the reference's state-dict layout
 * ([out, in] like nn.Linear).  Replaces: the parameters of backbones/efficient_gat.py:57-102
 * (Eff_GAT) or backbones/efficient_gat_3d.py:99-152 (Eff_GAT_3d), incl. the PyG
 * TransformerConv lin_{query,key,value,skip} of backbones/Transformer_GNN.py:9-25 and the
 * virt_node_embedding of backbones/exophormer_gnn.py:158-159.
 * ------------------------------------------------------------------------------------- */
typedef struct da_weights {
    int32_t variant;      /* DA_VARIANT_*                                                  */
    int32_t arch;         /* DA_ARCH_*                                                     */
    int32_t steps;        /* rows of time_emb                                              */
    int32_t c_in;         /* pose channels in: 2 | 4 (2D), 7 (3D); 1 <= c_in <= 8, checked by da_denoiser_create */
    int32_t c_out;        /* pose channels out: 2 | 4 (2D); 3D heads are 3 + 3             */
    int32_t feat_dim;     /* piece-feature width F: 1088 (2D) / 768 (3D vn_dgcnn)          */
    int32_t hidden;       /* mlp hidden width: 128 (2D) / 256 (3D)                         */
    int32_t heads;        /* 8                                                             */
    int32_t n_layers;     /* 4                                                             */
    int32_t virt_nodes;   /* V (exophormer), else 0                                        */
    const float *time_emb;                  /* [steps, 32]                                  */
    const float *pos_w0, *pos_b0;           /* [16, c_in], [16]                             */
    const float *pos_w1, *pos_b1;           /* [32, 16], [32]                               */
    const float *mlp_w0, *mlp_b0;           /* [hidden, D], [hidden]      D = F + 64        */
    const float *mlp_w1, *mlp_b1;           /* [D, hidden], [D]                             */
    const float *conv_wq[DA_MAX_LAYERS], *conv_bq[DA_MAX_LAYERS];   /* [H*C_l, Din_l], [H*C_l] */
    const float *conv_wk[DA_MAX_LAYERS], *conv_bk[DA_MAX_LAYERS];
    const float *conv_wv[DA_MAX_LAYERS], *conv_bv[DA_MAX_LAYERS];
    const float *conv_ws[DA_MAX_LAYERS], *conv_bs[DA_MAX_LAYERS];
    const float *virt_emb;                  /* [V, D] or NULL                               */
    const float *head_w0, *head_b0;         /* 2D final_mlp.0 [32, D]; 3D mlp_t.0 [256, D]  */
    const float *head_w1, *head_b1;         /* 2D final_mlp.2 [c_out, 32]; 3D mlp_t.2 [3,256] */
    const float *head_r_w0, *head_r_b0;     /* 3D mlp_r.0 [256, D] (NULL in 2D)             */
    const float *head_r_w1, *head_r_b1;     /* 3D mlp_r.2 [3, 256]                          */
} da_weights;

/* ---------------------------------------------------------------------------------------
 * Piece graph of one Batch.  Replaces: the PyG `edge_index` / `batch` arguments of
 * Eff_GAT.forward_with_feats (efficient_gat.py:121-129).  Built once per Batch by the host
 * (diffassemble_amd/graph_plan.py) from edge_index[2,E] (row 0 = source j, row 1 = target i):
 *   CSR by destination: incoming edges of node i are col_src[row_ptr[i] .. row_ptr[i+1]),
 *   multi-edges kept (PyG softmax semantics); edge_id[] maps a CSR slot to its position in
 *   the caller's edge_index (for the alpha[E,H] output).  For the exophormer arch the
 *   host appends the V*G virtual rows and the quirky virtual edges of
 *   exophormer_gnn.py:167-200 before building the CSR: n_nodes counts them, n_real does not.
 *   dense != 0 declares that every graph is complete (self loops iff dense == 1, none iff
 *   dense == 2), which lets the library take the block-diagonal MFMA attention kernel;
 *   graph_ptr[G+1] are the node offsets of the graphs (required when dense != 0).
 * ------------------------------------------------------------------------------------- */
typedef struct da_graph {
    int32_t n_nodes;          /* rows processed by the convs (real + virtual)              */
    int32_t n_real;           /* real pieces (rows of x / feats / out)                     */
    int32_t n_graphs;
    int32_t dense;            /* 0 = CSR only, 1 = complete + self loops, 2 = complete w/o */
    int64_t n_edges;
    const int32_t *row_ptr;   /* [n_nodes + 1]                                             */
    const int32_t *col_src;   /* [n_edges]                                                 */
    const int32_t *edge_id;   /* [n_edges] or NULL                                         */
    const int32_t *graph_ptr; /* [n_graphs + 1] or NULL                                    */
    int32_t max_graph_nodes;  /* largest graph (dense mode tiling)                         */
    /* dense mode only: every graph gets a 64-aligned slot of rows in the head-major Q / K / V
     * buffers: pad_ptr[g] = first padded row of graph g, n_pad = pad_ptr[G], row_map[i] =
     * padded row of node i.                                                               */
    int32_t n_pad;
    const int32_t *pad_ptr;   /* [n_graphs + 1] or NULL                                    */
    const int32_t *row_map;   /* [n_nodes] or NULL                                         */
    /* training only (da_train_backward): the same edges grouped by SOURCE node, i.e. the
     * outgoing edges of node j are out_dst[out_ptr[j] .. out_ptr[j+1]); multi-edges kept.
     * Hybrid graphs (below) train on adjacency-masked grouped GEMMs over their regular edges and
     * walk only the REMAINDER edges: out_ptr / out_dst then hold the by-source orientation of
     * irr_row_ptr / irr_col_src (out_dst may be empty), and row_ptr / col_src may be NULL.   */
    const int32_t *out_ptr;   /* [n_nodes + 1] or NULL                                     */
    const int32_t *out_dst;   /* [n_edges] or NULL                                         */
    /* hybrid mode (inference): for graphs that are neither complete nor small -- the Exphander
     * expanders of puzzle_dataset.py:33-152 with the exophormer virtual nodes -- the host splits the
     * edge list: edges whose two ends are real nodes of one graph and that occur once become one bit
     * of a per-graph adjacency mask (bit j of row i of graph g = edge j -> i; rows start at byte
     * mask_ptr[g], row stride = padded slot count / 8 bytes) and run through the masked MFMA attention;
     * every other edge (virtual nodes, duplicates, cross-graph pairs) stays a CSR by destination that a
     * second kernel merges into the same softmax.  Needs pad_ptr / row_map (virtual rows of a graph get
     * the slots behind its real nodes) and graph_ptr.                                            */
    int32_t hybrid;           /* 0 / 1                                                     */
    int32_t reserved0;
    const uint8_t *mask;      /* adjacency bits or NULL                                    */
    const int64_t *mask_ptr;  /* [n_graphs + 1] byte offsets                               */
    const int32_t *irr_row_ptr; /* [n_nodes + 1] remainder CSR                             */
    const int32_t *irr_col_src; /* [remainder edges]                                       */
    /* hybrid mode, optional (NULL = not given).  slot_node: the graph's padded slots may hold its nodes in any order
     * (row_map[node] = slot); slot_node[slot] = node is the inverse (-1 for padding), and the adjacency rows / bits are
     * then indexed by SLOT.  An Exphander graph (puzzle_dataset.py:115-152) laid out by its generator's permutation is a
     * circulant band in slot space.  blk_class: per graph a table of one byte per (32-slot query slab, 32-slot key block):
     * 0 = no regular edge, 1 = some, 2 = all 1024 pairs; rows of blk_class_stride bytes (a multiple of 4), graph g's table
     * at byte blk_class_ptr[g] (graphs of one shape may share a table, and mask_ptr[] entries may coincide likewise).     */
    const int32_t *slot_node;
    const uint8_t *blk_class;
    const int64_t *blk_class_ptr;
    int32_t blk_class_stride;
    int32_t reserved1;
    /* hybrid mode, optional: per padded SLOT the remainder-edge metadata the masked attention's epilogue needs, gathered
     * once per Batch so that the kernel reads ONE 16-byte record per query instead of walking slot_node -> irr_row_ptr ->
     * irr_col_src -> row_map (four dependent loads in front of every workgroup):
     *   rm_meta[slot] = { irr_row_ptr[node], irr_row_ptr[node + 1], row_map[irr_col_src[irr_row_ptr[node]]] (the slot of the
     *   first remainder source; any valid slot when there is none), node }; node = -1 for padding / virtual slots.        */
    const int32_t *rm_meta;   /* [n_pad][4] or NULL                                        */
    /* hybrid mode, optional: the remainder edges of the rows the masked attention does NOT own (the exophormer's virtual
     * nodes, exophormer_gnn.py:183-200) with duplicated (source, target) pairs merged: CSR by destination over ALL nodes (real
     * rows empty), one entry per distinct source and its multiplicity -- PyG's softmax counts a duplicated edge k times, i.e.
     * weighs exp(score) by k in numerator and denominator.  The pairing quirk of the reference makes most virtual -> virtual
     * edges duplicates (at V = 8, n = 900, 32 puzzles: 232 448 edges into virtual nodes, 29 256 distinct pairs).          */
    const int32_t *agg_row_ptr;   /* [n_nodes + 1] or NULL                                 */
    const int32_t *agg_col_src;   /* [distinct pairs]                                      */
    const float *agg_mult;        /* [distinct pairs] multiplicities                       */
} da_graph;

typedef struct da_denoiser da_denoiser;

/* Pack the weights into the library's own device buffers (act dtype; Q|K|V|skip fused per
 * layer).  Allocates (hipMalloc) -- not capturable; call once per checkpoint load.          */
int da_denoiser_create(const da_weights *w, int precision, void *stream, da_denoiser **out);
void da_denoiser_destroy(da_denoiser *d);

/* Which algebraic folds the packed denoiser uses (2D transformer arch): bit 0 = mlp.2 composed into the
 * conv-0 projection and final_mlp.0; bit 1 = last conv's value / skip projections composed with final_mlp.0
 * (32-wide value heads in the last attention).  For reporting executed vs algorithmic FLOPs.
 * bit 2 = every layer has a block-diagonal MFMA attention kernel: for complete graphs (da_graph.dense != 0)
 * row_ptr / col_src / edge_id may be NULL unless alpha is requested (the host can skip sorting the edge list).
 * bit 3 = the per-step mlp.0 runs on its hoisted form (feature columns multiplied once per Batch), which the UNCONDITIONAL pass
 * of classifier-free guidance inside the captured loops needs (da_loop_opts.cfg); clear (DA_DISABLE_MFMA=1): run guidance per step. */
int da_denoiser_flags(const da_denoiser *d);

/* Bytes of caller-provided workspace needed for a graph of this size.                      */
size_t da_denoiser_workspace_bytes(const da_denoiser *d, const da_graph *g);

/* Once per Batch (features are loop-invariant: spatial_diffusion.py:653 computes them once
 * per p_sample_loop): converts piece features [n_real, F] fp32 into the workspace and
 * writes the virtual-node rows.  Must precede da_denoiser_forward on that workspace.       */
int da_denoiser_set_features(da_denoiser *d, const da_graph *g, const float *feats,
                             void *workspace, size_t workspace_bytes, void *stream);

/* One denoiser forward == Eff_GAT.forward_with_feats (efficient_gat.py:121-146) or
 * Eff_GAT_3d.forward_with_feats (efficient_gat_3d.py:173-220), encoder bypassed.
 *   x      [n_real, c_in] fp32 noisy poses
 *   t      [n_real] int64 per-node timestep (reference contract), or NULL => t_scalar
 *   out    [n_real, c_out] fp32 (3D: [n_real, 7] = unit quaternion wxyz | translation)
 *   alpha  NULL, or fp32 attention weights in the caller's edge order (needs g->edge_id):
 *          alpha_all_layers == 0: [n_edges, heads] of the LAST conv (what Exophormer_GNN
 *          returns, exophormer_gnn.py:205-215); != 0: [n_layers, n_edges, heads], every conv
 *          (what Transformer_GNN returns, Transformer_GNN.py:32-41)
 *   pre_head    NULL or [n_real, 6] fp32 (3D only): raw (r_pred, t_pred) before exp/quat   */
int da_denoiser_forward(da_denoiser *d, const da_graph *g, const float *x, const int64_t *t,
                        int64_t t_scalar, float *out, float *alpha, int alpha_all_layers,
                        float *pre_head, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Diffusion schedule tables (fp32 [T] device pointers) == the registered buffers of
 * GNN_Diffusion.__init__, spatial_diffusion.py:282-321.
 * ------------------------------------------------------------------------------------- */
typedef struct da_schedule {
    int32_t steps;
    const float *betas;
    const float *alphas_cumprod;
    const float *sqrt_recip_alphas;
    const float *sqrt_recip_alphas_cumprod;
    const float *sqrt_recipm1_alphas_cumprod;
    const float *sqrt_one_minus_alphas_cumprod;
    const float *posterior_variance;
} da_schedule;

/* DDIM update after the model call == p_sample_ddim, spatial_diffusion.py:555-566,603-627
 * (2D) / spatial_diffusion_3d_test_double_diffusion.py:595-685 (3D: variant = DA_VARIANT_3D,
 * c = 7, SO(3) algebra of utils_3d.py:1018-1061 for the quaternion part).
 *   x, model_out, x_prev: [n, c] fp32.  t per node or NULL => t_scalar.
 *   prev_all_nonneg: the host-evaluated `(t - ratio >= 0).all()` branch of :560.
 *   eta: 0 for DDIM; > 0 adds eta*sqrt(var)*noise ([n, c] fp32, may be NULL when eta == 0). */
int da_ddim_step(const da_schedule *s, int variant, int mean_type, int n, int c, const float *x,
                 const float *model_out, const int64_t *t, int64_t t_scalar, int inference_ratio,
                 int prev_all_nonneg, float eta, const float *noise, float *x_prev, void *stream);

/* DDPM update == p_sample_ddpm, spatial_diffusion.py:485-510 (noise NULL <=> t_index == 0). */
int da_ddpm_step(const da_schedule *s, int n, int c, const float *x, const float *model_out,
                 const int64_t *t, int64_t t_scalar, const float *noise, float *x_prev,
                 void *stream);

/* The whole sampling loop == p_sample_loop, spatial_diffusion.py:635-676 /
 * ...double_diffusion.py:688-731: for i in reversed(range(0, steps, ratio)): forward + DDIM
 * update.  x_init [n_real, c] fp32; traj (nullable) [n_iters, n_real, c] receives every
 * step's poses (the reference keeps them all), x_final (nullable) the last.  max_iters <= 0
 * runs all iterations.  With use_graph != 0 the iterations are recorded once into a hipGraph
 * (cached inside the denoiser, keyed on the arguments) and replayed with one launch.       */
int da_sample_loop(da_denoiser *d, const da_graph *g, const da_schedule *s, int mean_type,
                   int inference_ratio, int max_iters, const float *x_init, float *traj,
                   float *x_final, void *workspace, size_t workspace_bytes, int use_graph,
                   void *stream);

/* The loop's other samplers, inside the same (capturable) enqueue -- spatial_diffusion.py:485-510 (p_sample_ddpm),
 * :568-589 (classifier-free guidance: a second denoiser pass over ZERO piece features, out = (1 + w) cond - w unc) and
 * :620-627 (eta > 0: + eta sqrt(var) noise).  `noise` [n_iters, n_real, c] fp32 holds one standard-normal draw per
 * iteration (the reference calls torch.randn_like once per step; the caller fills the buffer once per loop); it is read
 * by the update kernels of the replayed graph, so refill it in place before every launch.  2D only.  opts == NULL is
 * da_sample_loop. */
typedef struct da_loop_opts {
    int32_t sampler;      /* 0 = DDIM (p_sample_ddim), 1 = DDPM (p_sample_ddpm)                       */
    float eta;            /* DDIM only; 0 = deterministic                                             */
    int32_t cfg;          /* != 0: classifier-free guidance                                           */
    float cfg_w;          /* classifier_free_w                                                        */
    const float *noise;   /* [n_iters, n_real, c] or NULL (required when sampler == 1 or eta > 0)     */
} da_loop_opts;
int da_sample_loop_ex(da_denoiser *d, const da_graph *g, const da_schedule *s, int mean_type,
                      int inference_ratio, int max_iters, const float *x_init, float *traj,
                      float *x_final, void *workspace, size_t workspace_bytes, int use_graph,
                      const da_loop_opts *opts, void *stream);

/* The same loop over TWO disjoint sets of puzzles of one Batch (the puzzles of a Batch never
 * interact: spatial_diffusion.py:635-676 runs them through one block-diagonal graph), recorded
 * as two parallel branches of one hipGraph -- each with its own da_graph, poses and workspace
 * (features staged into each with da_denoiser_set_features) -- and replayed with one launch:
 * one half's projections overlap the other half's attention.  Complete graphs only (hybrid
 * graphs already fork a side stream inside the forward); no trajectory; always a hipGraph.   */
int da_sample_loop_pair(da_denoiser *d, const da_schedule *s, int mean_type, int inference_ratio,
                        int max_iters,
                        const da_graph *g_a, const float *x_init_a, float *x_final_a,
                        void *workspace_a, size_t workspace_a_bytes,
                        const da_graph *g_b, const float *x_init_b, float *x_final_b,
                        void *workspace_b, size_t workspace_b_bytes, void *stream);
/* The same two-branch loop keeping the trajectory the reference's p_sample_loop returns (spatial_diffusion.py:635-676 appends
 * every x_{t-1}): iteration i of half a / b is written to traj_a / traj_b + i * traj_stride (floats; the two halves normally are
 * row ranges of one [n_iters, N, c] buffer: traj_b = traj_a + n_real_a * c, traj_stride = N * c).  Both NULL = da_sample_loop_pair. */
int da_sample_loop_pair_traj(da_denoiser *d, const da_schedule *s, int mean_type, int inference_ratio,
                             int max_iters,
                             const da_graph *g_a, const float *x_init_a, float *x_final_a,
                             void *workspace_a, size_t workspace_a_bytes,
                             const da_graph *g_b, const float *x_init_b, float *x_final_b,
                             void *workspace_b, size_t workspace_b_bytes,
                             float *traj_a, float *traj_b, size_t traj_stride, void *stream);
/* ... and with the samplers of da_sample_loop_ex on both branches (ABI 17; spatial_diffusion.py:485-510,568-589,620-627):
 * `opts` as there, with opts->noise = half a's noise rows and noise_b = half b's, iteration i read at + i * noise_stride floats
 * (normally row ranges of ONE [n_iters, N, c] draw: noise_b = opts->noise + n_real_a * c, noise_stride = N * c -- the poses then
 * equal the one-branch loop's bit for bit).  opts == NULL is da_sample_loop_pair_traj. */
int da_sample_loop_pair_ex(da_denoiser *d, const da_schedule *s, int mean_type, int inference_ratio,
                           int max_iters,
                           const da_graph *g_a, const float *x_init_a, float *x_final_a,
                           void *workspace_a, size_t workspace_a_bytes,
                           const da_graph *g_b, const float *x_init_b, float *x_final_b,
                           void *workspace_b, size_t workspace_b_bytes,
                           float *traj_a, float *traj_b, size_t traj_stride,
                           const da_loop_opts *opts, const float *noise_b, size_t noise_stride, void *stream);

/* ---------------------------------------------------------------------------------------
 * Measurement aid (no reference counterpart: the reference has no profiler hooks).
 * While enabled, every kernel launch of da_denoiser_forward / da_sample_loop is bracketed by
 * a hipEvent pair on the launch stream (the loop then runs eagerly, not as a hipGraph).
 * da_profile_read synchronises, returns the summed milliseconds and launch counts per class
 * and resets the pool.
 * ------------------------------------------------------------------------------------- */
enum {
    DA_PROF_EMBED = 0,        /* pose MLP + timestep lookup                                  */
    DA_PROF_LINEAR_MLP = 1,   /* mlp.0 / mlp.2                                               */
    DA_PROF_LINEAR_QKVS = 2,  /* fused Q|K|V|skip projections                                */
    DA_PROF_ATTN_HIDDEN = 3,  /* attention of convs 0..n-2 (C = 32)                          */
    DA_PROF_ATTN_LAST = 4,    /* attention of the last conv (C = D/8)                        */
    DA_PROF_HEAD = 5,         /* pose head                                                   */
    DA_PROF_UPDATE = 6,       /* DDIM / DDPM update                                          */
    DA_PROF_CONV_FUSED = 7,   /* hidden conv as ONE kernel: projection + attention (C = 32)  */
    DA_PROF_NCLASS = 8
};
int da_profile_enable(da_denoiser *d, int on);
int da_profile_read(da_denoiser *d, float *ms /* [DA_PROF_NCLASS] host */,
                    int32_t *counts /* [DA_PROF_NCLASS] host */);

/* ---------------------------------------------------------------------------------------
 * Kernel-level entry points (used by the kernel parity tests and the training autograd).
 * ------------------------------------------------------------------------------------- */
/* out[M, ldo] = act(A[M, K] @ W[Nout, K]^T + bias) (+ residual[M, ldo]); A, W, out, residual
 * in act dtype `prec`, bias fp32.  Replaces torch.nn.Linear on the path.                   */
int da_linear(int prec, int M, int K, int Nout, const void *A, int lda, const void *W,
              const float *bias, int act, const void *residual, void *out, int ldo, void *stream);

/* The same layer with a constant weight packed ONCE into the fragment order of the row-panel kernel
 * (da_gemm_xpanel.hip: bf16, K in {128, 256}, 512 <= Nout <= 4096, Nout % 32 == 0) -- what the denoiser does with
 * its Q | K | V | skip projection weights at da_denoiser_create.  da_linear_packed_bytes returns 0 when the shape has
 * no packed form; da_linear_packed falls back to da_linear's kernels when the row-panel kernel does not apply (short
 * inputs, a residual) and gives the same values as da_linear up to the order of the fp32 bias addition.            */
size_t da_linear_packed_bytes(int prec, int K, int Nout);
int da_linear_pack(int prec, int K, int Nout, const void *W, int ldw /* 0 = K */, void *packed, void *stream);
int da_linear_packed(int prec, int M, int K, int Nout, const void *A, int lda, const void *W, const void *packed,
                     const float *bias, int act, const void *residual, void *out, int ldo, void *stream);

/* One PyG TransformerConv attention (Transformer_GNN.py:32,38) over a CSR graph:
 *   qkvs [n_nodes, 4*H*C] act dtype (Q | K | V | skip), out [n_nodes, H*C] act dtype
 *   out_i = sum_e softmax_i(q_i.k_j / sqrt(C)) v_j + skip_i (+ residual_i) ; act applied last.
 *   alpha (nullable) [n_edges, H] fp32 in caller edge order via g->edge_id.                */
int da_attn_csr(int prec, const da_graph *g, int heads, int C, const void *qkvs, const void *residual,
                int act, void *out, float *alpha, void *stream);

/* The same layer through the dense block-diagonal MFMA path (g->dense != 0): fused projection
 * qkvs = x[n_nodes, Din] @ w[4*H*C, Din]^T + b scattered into head-major Q / K / V, then the
 * flash-style attention kernel.  scratch: caller memory of da_attn_dense_scratch_bytes(),
 * zero-filled once by the caller before first use.  Returns nonzero (and an error text) if the
 * shape is not supported by the dense kernels.                                              */
size_t da_attn_dense_scratch_bytes(int prec, const da_graph *g, int heads, int C);
int da_conv_dense(int prec, const da_graph *g, int heads, int C, int Din, const void *x, const void *w,
                  const float *b, const void *residual, int act, void *out, void *scratch, void *stream);

/* The same layer in the two forms the packed denoiser runs it in (da_denoiser_create): with the Q rows of w / b
 * PRE-SCALED by log2(e) / sqrt(C) (DA_CONV_Q_PRESCALED: the attention kernels then take their shift-free softmax paths --
 * exp2 of the raw scores, verified row sums, running-max fallback), and with the value heads FOLDED to 32 channels
 * (DA_CONV_FOLDED_V32, C = 144: w = Wq | Wk | V' [2*H*C + H*32, Din], no skip; out = [H][n_real][32] per-head normalised
 * sum_j softmax_j(q_i.k_j / sqrt(C)) v'_j in the act dtype -- the last conv of the 2D transformer arch composed with
 * final_mlp.0).  Also takes hybrid graphs (adjacency mask + remainder CSR of g).  No reference counterpart
 * beyond da_conv_dense's (Transformer_GNN.py:32,38): kernel-level test entry for those paths.                          */
#define DA_CONV_Q_PRESCALED 1
#define DA_CONV_FOLDED_V32 2
int da_conv_dense_ex(int prec, const da_graph *g, int heads, int C, int Din, const void *x, const void *w,
                     const float *b, const void *residual, int act, void *out, void *scratch, int flags,
                     void *stream);

/* Fallback bookkeeping of the shift-free softmax kernels (no reference counterpart; lets a test assert that the branch it
 * aims at really ran): out[k] = events since the last reset.  Synchronises the device.                                  */
enum {
    DA_DBG_OPT_GEN_WORKGROUPS = 0,         /* k_attn_opt workgroups whose optimistic pass failed its verification   */
    DA_DBG_DENSE_FAST_EXITS = 1,           /* k_attn_dense waves that left the shift-free FAST mode                 */
    DA_DBG_DUAL_GEN_SLABS = 2,             /* k_attn_dual query slabs that left FAST mode                            */
    DA_DBG_OPT_MASKED_GEN_WORKGROUPS = 3,  /* adjacency-masked optimistic kernel: workgroups re-run                  */
    DA_DBG_RES_LAUNCHES = 4,               /* launches of the K / V-resident hidden-layer kernel (k_attn_res; its    */
                                           /* per-WAVE re-runs count under DA_DBG_OPT_GEN_WORKGROUPS)                */
    DA_DBG_VIRT_IN_LAUNCH = 5,             /* masked hidden-layer launches that carried the exophormer's virtual rows */
    DA_DBG_NCOUNTERS = 8
};
int da_debug_counters(int64_t *out /* host [n] */, int n, int reset);

/* ---------------------------------------------------------------------------------------
 * Training: the denoiser forward with saved activations and its backward, fp32.
 * Replaces torch autograd through Eff_GAT.forward_with_feats (efficient_gat.py:121-146) as
 * driven by GNN_Diffusion.p_losses / training_step (spatial_diffusion.py:432-483,707-721); the
 * loss itself (smooth_l1 on [N, c]) and the optimizer stay with the caller, as in the reference.
 *   w      LIVE fp32 parameters (no packing, no copy): per conv layer lin_query | lin_key |
 *          lin_value | lin_skip weights must be contiguous in that order ([4*H*C, Din]), and so
 *          must their biases -- the host keeps the parameters in one flat buffer.
 *   grads  the same struct filled with the gradient pointers (same layout); the library ADDS
 *          into them (zero them per optimizer step).
 *   d_feats nullable [n_real, F]: gradient w.r.t. the piece features, for a trainable encoder.
 * Only the 2D denoiser (arch transformer / exophormer) is implemented; any graph type, through
 * the CSR attention kernels (needs g->out_ptr / g->out_dst).  The forward must precede the
 * backward on the same workspace, with the same weights: besides the activations it leaves the
 * step's weight images there (W^T of every Linear with a dX product; bf16 copies in the bf16 mode),
 * which the backward reads instead of transposing again.
 * Streams: the calls enqueue on `stream`; the weight-gradient products (leaves of the backward)
 * and the forward's weight images run on ONE side stream owned by the library (forked behind an
 * event on `stream`, joined before the call returns control of the gradients: when a forward /
 * backward / backward_stage call returns, everything it enqueued is ordered before whatever the
 * caller enqueues on `stream` next).  DA_TRAIN_SIDE_DW=0 keeps every launch on `stream`.
 * ------------------------------------------------------------------------------------- */
size_t da_train_workspace_bytes(const da_weights *w, const da_graph *g);
/* ABI 18: the size for ONE mode (da_train_workspace_bytes = the exact-fp32 mode's, the larger one).  Hybrid graphs in the
 * bf16-operand mode run flash-style and hold no [n, n] pair matrix: 16 puzzles of 900 pieces need 0.9 GB instead of 3.0 GB.
 * Forward and backward of a step must be given a workspace of at least the size of THEIR mode.                              */
size_t da_train_workspace_bytes_ex(const da_weights *w, const da_graph *g, int mma_precision);
/* mma_precision of the _ex forms: how the matrix-core GEMMs of the step (every Linear forward / dX / dW and the grouped
 * attention GEMMs of complete and hybrid graphs) take their operands.  Storage is fp32 in both modes -- parameters,
 * activations, gradients, the flat buffers the optimizer and the gradient all-reduce see.
 *   DA_TRAIN_MMA_FP32  exact fp32 products (v_mfma_f32_16x16x4_f32): the reference's precision, the mode of the gradient
 *                      fixtures; da_train_forward / da_train_backward are this mode.
 *   DA_TRAIN_MMA_BF16  operands rounded to bf16 (round to nearest even) in registers on their way into the matrix cores,
 *                      products accumulated in fp32 (v_mfma_f32_16x16x16_bf16): what autocast(bfloat16) does to the
 *                      reference's Linear / matmul calls.  Softmax, GELU, bias sums, the CSR edge kernels and the optimizer
 *                      stay fp32.  Use the same mode for the forward and the backward of a step.                       */
#define DA_TRAIN_MMA_FP32 0
#define DA_TRAIN_MMA_BF16 1
int da_train_forward(const da_weights *w, const da_graph *g, const float *x, const int64_t *t,
                     const float *feats, float *out, void *workspace, size_t workspace_bytes,
                     void *stream);
int da_train_backward(const da_weights *w, const da_weights *grads, const da_graph *g, const float *x,
                      const int64_t *t, const float *d_out, float *d_feats, void *workspace,
                      size_t workspace_bytes, void *stream);
int da_train_forward_ex(const da_weights *w, const da_graph *g, const float *x, const int64_t *t,
                        const float *feats, float *out, void *workspace, size_t workspace_bytes,
                        int mma_precision, void *stream);
int da_train_backward_ex(const da_weights *w, const da_weights *grads, const da_graph *g, const float *x,
                         const int64_t *t, const float *d_out, float *d_feats, void *workspace,
                         size_t workspace_bytes, int mma_precision, void *stream);
/* ABI 18: the backward in two halves for a BUCKETED data-parallel exchange (the reference gets this from Lightning's DDP reducer,
 * train_script.py:215-218: gradient buckets all-reduced while the rest of backward still runs).  DA_TRAIN_BWD_EARLY enqueues
 * final_mlp and the convs L-1 .. 1 -- afterwards their gradients are final and may be exchanged on another stream --,
 * DA_TRAIN_BWD_LATE conv 0, the virtual-node embedding, mlp, pos_mlp, time_emb (and d_feats).  EARLY followed by LATE on one
 * stream enqueues exactly the launches of DA_TRAIN_BWD_ALL (= da_train_backward_ex).                                       */
#define DA_TRAIN_BWD_ALL 0
#define DA_TRAIN_BWD_EARLY 1
#define DA_TRAIN_BWD_LATE 2
int da_train_backward_stage(const da_weights *w, const da_weights *grads, const da_graph *g, const float *x,
                            const int64_t *t, const float *d_out, float *d_feats, void *workspace,
                            size_t workspace_bytes, int mma_precision, int stage, void *stream);

/* The elementwise glue of p_losses as single launches (ABI 19):
 *   da_q_sample  == q_sample, spatial_diffusion.py:421-430: x_noisy = extract(sqrt_alphas_cumprod, t) * x_start
 *                   + extract(sqrt_one_minus_alphas_cumprod, t) * noise  (bit-identical to the torch expression);
 *   da_loss_grad == the loss of p_losses, spatial_diffusion.py:470-480 (kind 0 = F.l1_loss, 1 = F.mse_loss,
 *                   2 = F.smooth_l1_loss, all with mean reduction) AND its gradient with respect to the prediction
 *                   (d_pred [n], for a unit upstream gradient) -- one workgroup, fixed summation order.
 *   x_start, noise, x_noisy: [n, c] fp32; t: [n] int64; target, pred, d_pred: [n] fp32; loss: [1] fp32. */
int da_q_sample(int steps, int n, int c, const float *sqrt_alphas_cumprod, const float *sqrt_one_minus_alphas_cumprod,
                const float *x_start, const float *noise, const int64_t *t, float *x_noisy, void *stream);
int da_loss_grad(int kind, size_t n, const float *target, const float *pred, float *loss, float *d_pred, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused Adafactor step over the flat parameter / gradient buffers (one call = one optimizer
 * step for every listed tensor, 4 launches, deterministic, no host sync).  Replaces, for the
 * live denoiser parameters, transformers.optimization.Adafactor with the defaults the reference
 * uses (spatial_diffusion.py:701-705): relative step, scale_parameter, no momentum, no decay.
 *   param_table  device array of n_params records {int64 off; int32 rows, cols, factored, pad;
 *                int64 row_off, col_off; int32 blk0, nblk; int64 colpart_off}  (56 bytes)
 *   block_table  device array of n_blocks records {int32 pid, row0, nrows}       (12 bytes)
 *   state        second-moment statistics (row/col EMAs of matrices, full EMA of vectors)
 *   scratch      >= 2*n_blocks + 4*n_params + sum(nblk*cols over matrices) floats
 *   step         1-based step count; matrices must have cols <= 1280.
 * ------------------------------------------------------------------------------------- */
int da_adafactor_step(int n_params, const void *param_table, int n_blocks, const void *block_table,
                      float *flat, const float *flat_grad, float *state, float *scratch,
                      size_t scratch_floats, int step, float eps1, float eps2, float clip_threshold,
                      float decay_rate, void *stream);

/* ---------------------------------------------------------------------------------------
 * Greedy assignment of predicted positions to grid cells, one workgroup per puzzle, no host sync.
 * Replaces greedy_cost_assignment (spatial_diffusion.py:179-216) as called by validation_step /
 * test_step (:931-955).  Puzzle g owns rows ptr1[g]..ptr1[g+1] of pos1 (row stride ld1 floats, x
 * and y in the first two columns) and rows ptr2[g]..ptr2[g+1] of pos2.  out[ptr1[g] + k] =
 * (row, column, (int64)distance) of the k-th assignment, in the order the reference makes them;
 * indices are local to the puzzle.  Ties: first in row-major order, like the reference.
 * ------------------------------------------------------------------------------------- */
int da_greedy_assign(int n_puzzles, const float *pos1, int ld1, const float *pos2, int ld2,
                     const int32_t *ptr1, const int32_t *ptr2, int max_n, int max_m, long long *out,
                     void *stream);

/* ---------------------------------------------------------------------------------------
 * 2D piece encoder: the P4 group-equivariant ResNet-18 the reference builds
 * for model='resnet18equiv', eval-mode BatchNorm.  Replaces Eff_GAT.visual_features
 * (backbones/efficient_gat.py:149-189) = normalise + ResNet.forward (backbones/resnet_equivariant.py:
 * 93-112) + cat(linear1(out3), linear2(out4)), whose convolutions are groupy's SplitGConv2D
 * (backbones/groupy/gconv/pytorch_gconv/splitgconv2d.py:15-22,70-92).
 *
 * da_encoder_weights holds the PACKED weights (device pointers; built once per checkpoint by the
 * host, diffassemble_amd/encoder.py, from the reference's state-dict tensors):
 *   stem_w [128][27] fp32   rotated filter bank of conv1 (channel o*4+r; taps c*9+ky*3+kx), bn1 folded
 *   stem_b [128]     fp32   folded bn1 bias
 *   conv_w[i] [Cout*4][k*k*Cin*4] act dtype, i in state-dict order (layerL.B.conv1, conv2[, shortcut.0]):
 *             filter bank with the BatchNorm scale folded, K ordered tap-major / channel-minor (NHWC)
 *   conv_b[i] [Cout*4] fp32
 *   lin1_w [544][10*10*256], lin2_w [544][6*6*512] act dtype: linear1 / linear2 re-indexed from the
 *             reference's NCHW flatten to the zero-haloed NHWC maps (halo columns are zero); biases fp32.
 * ------------------------------------------------------------------------------------- */
enum { DA_ENCODER_CONVS = 19, DA_ENCODER_FEATS = 1088 };
typedef struct da_encoder_weights {
    int32_t n_convs;                         /* DA_ENCODER_CONVS                              */
    int32_t reserved0;
    const float *stem_w, *stem_b;
    const void *conv_w[DA_ENCODER_CONVS];
    const float *conv_b[DA_ENCODER_CONVS];
    const void *lin1_w; const float *lin1_b;
    const void *lin2_w; const float *lin2_b;
} da_encoder_weights;

/* Workspace for n_patches pieces processed `chunk` at a time (activations of one chunk + the
 * layer-3 / layer-4 maps of all pieces). */
size_t da_encoder_workspace_bytes(int precision, int n_patches, int chunk);

/* patches [n_patches, 3, 32, 32] fp32 in [0, 1] -> feats [n_patches, 1088] (act dtype, row stride
 * ld_feats elements).  The feature maps carry a zero halo that the kernels never write: pass
 * zero_workspace != 0 on the first call with a (new or foreign-written) workspace, 0 afterwards
 * as long as (precision, n_patches, chunk) stay the same.  Stream-capturable. */
int da_encoder_forward(int precision, const da_encoder_weights *w, int n_patches, const float *patches,
                       void *feats, int ld_feats, void *workspace, size_t workspace_bytes, int chunk,
                       int zero_workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * 3D piece encoder: the reference's vector-neuron DGCNN, eval-mode BatchNorm.
 * Replaces Eff_GAT_3d.pcd_features -> VN_DGCNN.forward (backbones/vnn/vn_dgcnn.py:34-74) with
 * get_graph_feature / knn (:84-120) and VNLinearLeakyReLU / VNBatchNorm (backbones/vnn/vn_layers.py:50-91,
 * 133-154); all fp32.  A stage = kNN(20) in the current feature space -> first VN layer on cat(x_j - x_i, x_i)
 * -> [second VN layer] -> mean over the neighbours; three stages, then conv6 + mean over the points.
 *
 * da_pcd_encoder_weights holds PACKED fp32 device pointers (host: diffassemble_amd/pcd_encoder.py, from the
 * reference's state-dict tensors; Cin = 1, 21, 21 for the three stages, W = map_to_feat / map_to_dir of the
 * stage's first layer, columns [:Cin] act on x_j - x_i and [Cin:] on x_i):
 *   premap[s] [4][21][Cin]    Wf[:, :Cin], Wd[:, :Cin], Wf[:, Cin:] - Wf[:, :Cin], Wd[:, Cin:] - Wd[:, :Cin]
 *   bn_a[s]   [2][21]         eval BatchNorm of the norm as  norm * scale + shift
 *   conv_b[s] [2][21][22] + [2][21]   second layer of the stage: feature map, direction map (rows zero-padded to
 *                             22), then scale, shift; NULL for stage 3 (conv5 stands alone)
 *   conv6     [feat_dim][63] + [63] + [2][feat_dim]   feature map, the ONE shared direction map, scale, shift
 *   linear0   [2 feat_dim][3] + [2 feat_dim]          only read for the invariant output (may be NULL otherwise)
 * ------------------------------------------------------------------------------------- */
enum { DA_PCD_K = 20, DA_PCD_C = 21, DA_PCD_ROW = 64, DA_PCD_STAGES = 3 };
typedef struct da_pcd_encoder_weights {
    int32_t feat_dim;                        /* conv6 output channels (128 in the reference)  */
    int32_t reserved0;
    const float *premap[DA_PCD_STAGES];
    const float *bn_a[DA_PCD_STAGES];
    const float *conv_b[DA_PCD_STAGES];
    const float *conv6;
    const float *linear0;
} da_pcd_encoder_weights;

/* Workspace for fragments of n_points points processed `chunk` at a time. */
size_t da_pcd_encoder_workspace_bytes(int n_points, int chunk, int feat_dim);

/* points [n_parts, n_points, 3] fp32 -> out [n_parts, 6 feat_dim] (inv == 0: the pooled equivariant map twice,
 * vn_dgcnn.py:62-73) or [n_parts, 2 feat_dim] (inv != 0: linear0 path, :68-69); row stride ld_out floats.
 * n_points >= 20.  Stream-capturable. */
int da_pcd_encoder_forward(const da_pcd_encoder_weights *w, int n_parts, int n_points, const float *points, int inv,
                           float *out, int ld_out, void *workspace, size_t workspace_bytes, int chunk, void *stream);

/* The encoder's neighbour search on its own (vn_dgcnn.py:114-120): for every point of every cloud the k nearest
 * points of the same cloud (itself included), nearest first, ties towards the lower index -> idx [n_clouds,
 * n_points, k] (cloud-local).  x rows: dim == 3 with stride ldx >= 3, or 4 <= dim <= 64 as zero-padded 64-float rows. */
int da_knn(int n_clouds, int n_points, int dim, const float *x, int ldx, int k, int32_t *idx, void *stream);

/* Nearest-neighbour squared distances both ways (chamfer_distance.py:148-149 = pytorch3d knn_points K = 1, used by
 * utils_3d.py:1089-1129 calc_part_acc): a [n_clouds, n, 3], b [n_clouds, m, 3] -> d_ab [n_clouds, n],
 * d_ba [n_clouds, m]; either output may be NULL. */
int da_nearest_sq(int n_clouds, int n, int m, const float *a, const float *b, float *d_ab, float *d_ba, void *stream);

/* ---------------------------------------------------------------------------------------
 * Training path of the 2D piece encoder (the encoder runs in EVERY training step,
 * model/spatial_diffusion.py:450): fp32 primitives over the zero-haloed NHWC maps [B][H+2][H+2][C4]
 * (C4 = planes * 4, channel = plane * 4 + rotation; the halo must be zero and is never written).  Together
 * they replace ResNet.forward in train() mode (backbones/resnet_equivariant.py:14-38,93-112: BatchNorm3d on
 * batch statistics) and torch autograd through it, including SplitGConv2D's conv2d / trans_filter
 * (backbones/groupy/gconv/pytorch_gconv/splitgconv2d.py:15-22,70-92).  The network walk is host logic
 * (diffassemble_amd/encoder_train.py).  `scratch`: da_enc_train_scratch_bytes(n_patches) bytes.
 * ------------------------------------------------------------------------------------- */
size_t da_enc_train_scratch_bytes(int n_patches);

/* Y = conv(X, W) + bias [+ res] [ReLU]: the inference kernel unfused.  X [B][Hi+2][Hi+2][Cin], W [Cout][k*k*Cin]
 * (tap-major, channel-minor), Y [B][Hi/stride+2][..][Cout]; k in {1, 3}.  Also the DGRAD of a stride-1
 * convolution when W holds the flipped / transposed bank; `res` (may alias Y) accumulates other paths. */
int da_enc_conv(int precision, int B, const void *X, int Cin, int Hi, const void *W, const float *bias, const void *res,
                void *Y, int Cout, int ksize, int stride, int relu, void *stream);
/* The map-shaped primitives below take `precision` = the STORAGE type of the maps (DA_PREC_F32: the parity mode;
 * DA_PREC_BF16: activations and activation gradients in bf16, all arithmetic and every parameter gradient in fp32). */
/* stem: normalise + P4ConvZ2(3 -> 128 channels, 3x3) + bias [ReLU], patches [B,3,32,32] fp32 -> Y [B][34][34][128] */
int da_enc_stem(int precision, int B, const float *patches, const float *w, const float *bias, void *Y, int relu, void *stream);
/* the stem's input as GEMM rows for its weight gradient: cols [B][34][34][32] (27 taps + zero pad; zero halo) */
int da_enc_stem_im2col(int precision, int B, const float *patches, void *cols, void *stream);
/* nn.BatchNorm3d batch statistics: mean / biased variance per plane over (B, 4, H, W) */
int da_enc_bn_stats(int precision, int B, int H, int C4, const void *Y, float *mean, float *var, void *scratch, void *stream);
/* Z = (Y - mean) * (rsqrt(var + 1e-5) * gamma) + beta [+ res] [ReLU] */
int da_enc_bn_apply(int precision, int B, int H, int C4, const void *Y, const float *mean, const float *var, const float *gamma,
                    const float *beta, const void *res, int relu, void *Z, void *stream);
/* backward of that unit: g = dZ [masked by Z > 0]; dgamma += sum g xhat; dbeta += sum g;
 * dY = gamma rstd (g - mean(g) - xhat mean(g xhat)); dRes = g when non-NULL (gradient of the added tensor) */
int da_enc_bn_backward(int precision, int B, int H, int C4, const void *dZ, const void *Z, const void *Y, const float *mean,
                       const float *var, const float *gamma, int relu, float *dgamma, float *dbeta, void *dY,
                       void *dRes, void *scratch, void *stream);
/* zero-stuffing of a gradient map to twice the resolution (stride-2 layers are differentiated as stride 1) */
int da_enc_upsample2(int precision, int B, int H, int C4, const void *S, void *Up, void *stream);
/* C += A^T B over rows: A [M, N] (lda), B [M, K] (ldb), C [N, K] (ldc); fp32 MFMA, deterministic split over rows.
 * One call per filter tap is a convolution's weight gradient (dY's zero halo makes the haloed maps plain
 * GEMM operands).  scratch: 64 MB. */
int da_gemm_tn_f32(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                   void *scratch, void *stream);
/* the same with bf16 operands (fp32 accumulation and output): v_mfma_f32_32x32x16_bf16, tiles transposed on their way
 * into LDS.  lda / ldb multiples of 8, 16-byte aligned bases. */
int da_gemm_tn_bf16(int M, int N, int K, const void *A, int lda, const void *B, int ldb, float *C, int ldc,
                    void *scratch, void *stream);
/* out[c] += sum_m A[m][c]; scratch: ceil(M / 128) * N floats */
int da_colsum_f32(int M, int N, const float *A, int lda, float *out, void *scratch, void *stream);
/* dW[i] += sum_{r<4} dBank[table[4 i + r]]: backward of the filter-bank gather (trans_filter) */
int da_enc_bank_grad(int n, const int32_t *table, const float *dbank, float *dW, void *stream);

/* ---------------------------------------------------------------------------------------
 * Exphander graphs (dataset/puzzle_dataset.py:115-152 generate_random_regular_graph): the adjacency
 * bit rows of da_graph.mask for n_graphs graphs of n nodes and degree d, from the permutations the generator draws
 * (perms int64 [n_graphs, n]): bit j of row i of graph g = edge j -> i, rows of row_bytes bytes, graph g at
 * g * n * row_bytes.  pos: int32 scratch [n_graphs * n] (receives the inverse permutations).  Two launches. */
int da_expander_mask(int n_graphs, int n, int degree, const int64_t *perms, int32_t *pos, int row_bytes,
                     unsigned char *mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFASSEMBLE_HIP_H */
