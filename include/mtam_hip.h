/*
 * libmtam_hip.so -- C ABI of the MI355X (gfx950) time-aware training path.
 *
 * The reference (cocoandpudding/MTAMRecommender) has no FFI: its hot path is a
 * TensorFlow 1.14 graph run by sess.run (Model/base_model.py:159-164 train,
 * :201-202 eval).  Each entry point below replaces a group of graph ops; the
 * comment on each names the reference lines.  A maintainer binds them with
 * ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative MTAM_E_* code otherwise;
 *     mtam_last_error() gives a thread-local message;
 *   - every pointer is a DEVICE pointer unless named host_*; the library never
 *     allocates, frees or keeps caller memory; scratch is passed in;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and is
 *     asynchronous; calls are re-entrant across streams and safe to capture
 *     into a hipGraph (no allocation, no synchronisation inside);
 *   - matrices are row-major contiguous float32 unless a leading dimension is
 *     given; ids and lengths are int32; times are float32 (raw hours);
 *   - D (num_units) must be 128 in this build (MTAM_D).
 */
#ifndef MTAM_HIP_H
#define MTAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTAM_D 128

#define MTAM_OK 0
#define MTAM_E_ARG (-1)     /* bad argument (shape, alignment, null pointer) */
#define MTAM_E_LAUNCH (-2)  /* hipLaunch failure */
#define MTAM_E_UNSUPPORTED (-3)

const char *mtam_last_error(void);
int mtam_version(void);          /* 1000*major + minor */
const char *mtam_arch(void);     /* "gfx950" */

/* ------------------------------------------------------------------ GEMM
 * C[M,N] = op(A) * op(B) with a fused epilogue; fp32 in, fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain per output element).
 * Replaces tf.layers.dense / tf.matmul on the path
 * (Embedding/Behavior_embedding_time_aware_attention.py:95-103,
 *  Model/Modules/time_aware_attention.py:249-253, Model/base_model.py:316)
 * and their tf.gradients counterparts (Model/base_model.py:292).
 *   trans_a == 0: A is [M,K] (lda >= K);  trans_a == 1: A is [K,M] (lda >= M)
 *   trans_b == 0: B is [K,N] (ldb >= N);  trans_b == 1: B is [N,K] (ldb >= K)
 * Epilogues (acc = the product):
 *   MTAM_EPI_STORE       C = acc
 *   MTAM_EPI_BIAS        C = acc + bias[n]
 *   MTAM_EPI_BIAS_RELU   C = relu(acc + bias[n])
 *   MTAM_EPI_RELU_ADD    aux_out = relu(acc); C = relu(acc) + aux_in[m,n]
 *   MTAM_EPI_ACCUM       C += acc
 *   MTAM_EPI_ACCUM_MASK  C += acc; aux_out = (aux_in[m,n] > 0) ? C : 0
 *   MTAM_EPI_ATOMIC      atomicAdd(C, acc)   (split_k >= 1 slices of K)
 *   MTAM_EPI_ACCUM2_MASK C += acc + add2[m,n]; aux_out = (aux_in[m,n] > 0) ? C : 0
 *                        (add2 is passed in the `bias` argument as an [M, ld_aux] matrix)
 *   MTAM_EPI_STORE_SQ    C = acc; aux_out[4 * tile + wave] = sum of acc^2 over the wave's in-range
 *                        32x32 quadrant: mtam_gemm_sq_partials(M, N) floats whose sum is ||C||^2
 *                        (the dense item gradient's share of tf.global_norm, Model/base_model.py:294)
 * aux_in / aux_out share ld_aux.  split_k > 1 is only valid with ATOMIC.
 * MTAM_GEMM_SPLIT_BF16 OR-ed into `epilogue`: the products are formed on the bf16 matrix cores from the operands
 * split three ways (x = x1 + x2 + x3, six partial products, fp32 accumulation: fp32-equivalent, 2.7 x the matrix
 * rate); the training step's GEMMs ask for it, the evaluation logits (whose ranking contract is the k-ordered
 * fmaf chain) do not.  The grouped weight-gradient launches always use it.  MTAM_GEMM_SPLIT=0 in the environment
 * turns it off everywhere.
 */
enum { MTAM_GEMM_SPLIT_BF16 = 0x100 };
enum {
  MTAM_EPI_STORE = 0,
  MTAM_EPI_BIAS = 1,
  MTAM_EPI_BIAS_RELU = 2,
  MTAM_EPI_RELU_ADD = 3,
  MTAM_EPI_ACCUM = 4,
  MTAM_EPI_ACCUM_MASK = 5,
  MTAM_EPI_ATOMIC = 6,
  MTAM_EPI_ACCUM2_MASK = 7,
  MTAM_EPI_STORE_SQ = 8
};
int mtam_gemm_sq_partials(int M, int N);
int mtam_gemm_f32(int trans_a, int trans_b, int M, int N, int K,
                  const float *A, int lda, const float *B, int ldb,
                  float *C, int ldc, int epilogue, const float *bias,
                  const float *aux_in, float *aux_out, int ld_aux,
                  int split_k, void *stream);

/* C = op(A) op(B) + op(A2) op(B2) (+ epilogue): the same kernel with a second (A2, B2, K2) source whose
 * k-tiles are accumulated into the same output tile after the first source's -- two matmul gradients that
 * meet in one tensor (d x = d(xproj) Wx^T + d(kv) Wkv^T) in one launch.  Same transposes for both sources. */
int mtam_gemm_f32_dual(int trans_a, int trans_b, int M, int N, int K,
                       const float *A, int lda, const float *B, int ldb,
                       int K2, const float *A2, int lda2, const float *B2, int ldb2,
                       float *C, int ldc, int epilogue, const float *bias,
                       const float *aux_in, float *aux_out, int ld_aux, void *stream);

/* Batched GEMM: batch0 x batch1 independent problems of one shape in one launch; problem (z0, z1)
 * uses A + z0*sA0 + z1*sA1 (element offsets; same for B and C).  The L x L products of the
 * self-attention encoder: Q_h K_h^T, (q Wt) k^T, W V and their gradients
 * (Model/Modules/time_aware_attention.py:320-321,380,443).  Epilogue STORE or ACCUM. */
int mtam_gemm_f32_batched(int trans_a, int trans_b, int M, int N, int K,
                          const float *A, int lda, long sA0, long sA1,
                          const float *B, int ldb, long sB0, long sB1,
                          float *C, int ldc, long sC0, long sC1,
                          int batch0, int batch1, int epilogue, void *stream);

/* Grouped weight-gradient GEMMs: n <= MTAM_MAX_GROUP independent problems
 * C[M,N] += A^T B with A [K,M] (lda), B [K,N] (ldb), split-K slices added by
 * atomics, all in ONE launch (the dW of every dense layer after backward;
 * tf.gradients w.r.t. the kernels, Model/base_model.py:292).  `d` is a HOST array. */
#define MTAM_MAX_GROUP 16
typedef struct {
  const float *A; int lda;
  const float *B; int ldb;
  float *C; int ldc;
  int M, N, K, split_k;
} MtamGemmDesc;
int mtam_gemm_tn_atomic_grouped(int n, const MtamGemmDesc *d, void *stream);

/* column sums: out[c] += sum_r in[r, c]  (atomicAdd; bias gradients) */
int mtam_colsum_atomic(const float *in, int rows, int cols, int ld, float *out, void *stream);
/* the same for n <= MTAM_MAX_GROUP jobs in one launch; `jobs` is a HOST array */
typedef struct {
  const float *in; int rows, cols, ld;
  float *out;
} MtamColsumJob;
int mtam_colsum_atomic_multi(int n, const MtamColsumJob *jobs, void *stream);
/* mtam_gemm_tn_atomic_grouped + mtam_colsum_atomic_multi in ONE launch: every kernel gradient and every
 * bias-like gradient of a training step (Model/base_model.py:292).  Both arrays are HOST arrays. */
int mtam_weight_grads(int n_gemm, const MtamGemmDesc *d, int n_colsum, const MtamColsumJob *jobs,
                      void *stream);

/* --------------------------------------------------------- embedding gather
 * tf.nn.embedding_lookup x4 (Embedding/Behavior_embedding_time_aware_attention.py:68,75,82,90)
 * + the concat of :95 + the four tf.nn.l2_loss terms of Model/base_model.py:302-307.
 *   item_cat_out [B*L, 2D]: row r = [item_table[item_ids[r]] | cat_table[cat_ids[r]]]
 *   pos_out      [B*L, D],  user_out [B, D]
 *   l2_partial   [mtam_emb_gather_partials(B, L)] floats: per-wave sums of x^2
 *                (0.5 * their sum = l2_norm; with_user == 0 leaves the user
 *                rows out, as Model/PISTRec_model.py:56-60 does).
 * Ids are clamped into [0, rows) so a bad id cannot fault; validate on the host.
 */
int mtam_emb_gather_partials(int B, int L);
int mtam_emb_gather_fwd(const float *item_table, int item_rows,
                        const float *cat_table, int cat_rows,
                        const float *pos_table, int pos_rows,
                        const float *user_table, int user_rows,
                        const int32_t *item_ids, const int32_t *cat_ids,
                        const int32_t *pos_ids, const int32_t *user_ids,
                        int B, int L, int with_user,
                        float *item_cat_out, float *pos_out, float *user_out,
                        float *l2_partial, void *stream);
/* The same launch, additionally clearing two float ranges (n_a, n_b floats; 16-byte aligned, multiples
 * of 4): the training step's gradient accumulators, zeroed by its first kernel instead of two fill
 * launches (tf.gradients starts from zero accumulators, Model/base_model.py:292). */
int mtam_emb_gather_fwd_clear(const float *item_table, int item_rows,
                              const float *cat_table, int cat_rows,
                              const float *pos_table, int pos_rows,
                              const float *user_table, int user_rows,
                              const int32_t *item_ids, const int32_t *cat_ids,
                              const int32_t *pos_ids, const int32_t *user_ids,
                              int B, int L, int with_user,
                              float *item_cat_out, float *pos_out, float *user_out,
                              float *l2_partial, float *clear_a, size_t n_a, float *clear_b, size_t n_b,
                              void *stream);
/* The same launch with the ITEM rows read from a bf16 image of the item table (item16 [item_rows, 128] bf16
 * bits, e.g. the scoring copy kept by mtam_adam_bf16copy) and widened to fp32: mixed-precision runs then
 * read the item table in bf16 everywhere in the forward.  item16 == NULL: identical to the call above
 * (item_table is still required: it is what a NULL item16 falls back to). */
int mtam_emb_gather_fwd_item16(const float *item_table, const uint16_t *item16, int item_rows,
                               const float *cat_table, int cat_rows, const float *pos_table, int pos_rows,
                               const float *user_table, int user_rows, const int32_t *item_ids,
                               const int32_t *cat_ids, const int32_t *pos_ids, const int32_t *user_ids, int B,
                               int L, int with_user, float *item_cat_out, float *pos_out, float *user_out,
                               float *l2_partial, float *clear_a, size_t n_a, float *clear_b, size_t n_b,
                               void *stream);

/* ---------------------------------------------------- embedding scatter-add
 * Gradient of the four lookups (tf.gradients through embedding_lookup,
 * Model/base_model.py:292) including the L2 term: slot r contributes
 * (d_slot[r] + reg * gathered[r]) to row ids[r] of the table's gradient, by
 * wave-level float atomics (two 128-B row segments per wave instruction).
 * Padded slots (t >= seq_len[b]) carry d_slot == 0 exactly, so they are not
 * scattered one by one: their sum, n_pad * reg * row0, is added once.
 *   d_item_cat [B*L, 2D], d_pos [B*L, D]  upstream gradients
 *   item_cat   [B*L, 2D], pos [B*L, D], user [B, D]  the gathered rows
 *   g_item (already holding the dense scoring gradient), g_cat, g_pos, g_user
 *   slot_sq_partial [mtam_emb_scatter_partials(B, L)]: per-wave sums of
 *     ||contribution||^2 over un-deduplicated slots (TF IndexedSlices norm,
 *     SURVEY.md App D-5), pads included.
 * with_user == 0: the user table gets no gradient (PISTRec).
 */
int mtam_emb_scatter_partials(int B, int L);
int mtam_emb_scatter_add_bwd(const float *d_item_cat, const float *d_pos,
                             const float *item_cat, const float *pos, const float *user,
                             const int32_t *item_ids, const int32_t *cat_ids,
                             const int32_t *pos_ids, const int32_t *user_ids,
                             const int32_t *seq_len, int B, int L, float reg, int with_user,
                             float *g_item, int item_rows, float *g_cat, int cat_rows,
                             float *g_pos, int pos_rows, float *g_user, int user_rows,
                             float *slot_sq_partial, void *stream);
/* The same with the looked-up POSITION rows taken from the table through the ids (pos == NULL, pos_table given):
 * for steps whose forward never wrote them out (mtam_seq_chain_gather_fwd). */
int mtam_emb_scatter_add_bwd_postab(const float *d_item_cat, const float *d_pos, const float *item_cat,
                                    const float *pos, const float *pos_table, const float *user,
                                    const int32_t *item_ids, const int32_t *cat_ids, const int32_t *pos_ids,
                                    const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg,
                                    int with_user, float *g_item, int item_rows, float *g_cat, int cat_rows,
                                    float *g_pos, int pos_rows, float *g_user, int user_rows,
                                    float *slot_sq_partial, void *stream);
/* ... and with the [item | category] gradient rows computed in place (d_item_cat == NULL; d_z [B*L, D] and
 * W4 = dense4emb's kernel [2D, D] given): every 128-slot chunk of the item (category) table forms its own
 * 128 x 128 block of  d_z . W4^T  on the matrix cores -- the d[item | category] GEMM and the round trip of its
 * [B*L, 2D] result through HBM disappear from the step (SURVEY.md 2.1 K11 fused with the dense4emb input gradient). */
int mtam_emb_scatter_add_bwd_fused(const float *d_item_cat, const float *d_z, const float *W4, const float *d_pos,
                                   const float *item_cat, const float *pos, const float *pos_table,
                                   const float *user, const int32_t *item_ids, const int32_t *cat_ids,
                                   const int32_t *pos_ids, const int32_t *user_ids, const int32_t *seq_len, int B,
                                   int L, float reg, int with_user, float *g_item, int item_rows, float *g_cat,
                                   int cat_rows, float *g_pos, int pos_rows, float *g_user, int user_rows,
                                   float *slot_sq_partial, void *stream);
/* ... restricted to a ROW RANGE of the item table (data-parallel row-sharded scoring: a rank owns item rows
 * [item_lo, item_hi) of the gradient): item slots whose id lies outside the range are skipped; every other table as
 * in mtam_emb_scatter_add_bwd_fused.  item_only != 0: ONLY the item slots are applied (another rank's all-gathered
 * d_item_cat / item_cat / item_ids / seq_len: the item halves of its [B*L, 2D] rows); every other pointer may be NULL. */
int mtam_emb_scatter_add_bwd_range(const float *d_item_cat, const float *d_z, const float *W4, const float *d_pos,
                                   const float *item_cat, const float *pos, const float *pos_table,
                                   const float *user, const int32_t *item_ids, const int32_t *cat_ids,
                                   const int32_t *pos_ids, const int32_t *user_ids, const int32_t *seq_len, int B,
                                   int L, float reg, int with_user, float *g_item, int item_rows, float *g_cat,
                                   int cat_rows, float *g_pos, int pos_rows, float *g_user, int user_rows,
                                   float *slot_sq_partial, int item_lo, int item_hi, int item_only, void *stream);
/* The scatter-add with the clip's partial pass RIDING ALONG (mtam_sqnorm_state_loss as extra workgroups of this
 * launch, on the CUs it leaves idle: the dense gradient `norm->g` must be complete before the launch, i.e. the weight
 * gradients come first): partials[offset + i] = sum of squares of 4,096-float block i of g[0 .. n); the Adam state
 * advanced (lr, adam_state; both NULL = not); loss[0..2] = reg * l2 + ce_scale * sum(ce), l2, ce_scale * sum(ce)
 * (NULL = not).  One launch and its gap less per step (2.6 + ~1 us at ml-1m sizes). */
typedef struct MtamNormRider {
  const float *g;
  size_t n;
  float *partials;
  int offset;
  const float *lr;
  float *adam_state;
  const float *l2_partial;
  int n_l2;
  const float *ce;
  int B;
  float reg, ce_scale;
  float *loss;
} MtamNormRider;
int mtam_emb_scatter_add_bwd_norm(const float *d_item_cat, const float *d_z, const float *W4, const float *d_pos,
                                  const float *item_cat, const float *pos, const float *pos_table, const float *user,
                                  const int32_t *item_ids, const int32_t *cat_ids, const int32_t *pos_ids,
                                  const int32_t *user_ids, const int32_t *seq_len, int B, int L, float reg,
                                  int with_user, float *g_item, int item_rows, float *g_cat, int cat_rows,
                                  float *g_pos, int pos_rows, float *g_user, int user_rows, float *slot_sq_partial,
                                  const MtamNormRider *norm, void *stream);
/* out [n, 128]: row ids[i] of the catalog if it lies in the range [row0, row0 + nrows) that table_rows holds (its
 * first row is catalog row row0), else zeros.  Data-parallel "sharded-table" exchange: the owners' rows of every
 * rank's history ids, summed by a reduce-scatter (each row has exactly one owner), replace a replicated item table as
 * the source of the embedding lookups (Embedding/Behavior_embedding_time_aware_attention.py:68-75). */
int mtam_rows_gather_range(const float *table_rows, int row0, int nrows, const int32_t *ids, long n, float *out,
                           void *stream);

/* ----------------------------------------------------------- time-aware GRU
 * dynamic_rnn(TimeAwareGRUCell_decay_new) + gather_indexes(seq_len - 2):
 * Model/Modules/gru.py:69-77, Model/Modules/time_aware_rnn.py:186-269,
 * Model/Modules/net_utils.py:82-92, Model/MTAMRec_model.py:68-79.
 * One workgroup per sample; the recurrent weights live in registers.
 *   xproj   [B*L, 3D] = x @ [Wg_x | Wc_x] + [bg | bc]   (hoisted input projection)
 *   x       [B*L, D], timelast [B*L], seq_len [B] (steps = seq_len - 1)
 *   wh_g    [D, 2D], wh_c [D, D]      recurrent halves of gates/candidate kernels
 *   tvec    [8, D]: _time_kernel_w1, _time_kernel_b1, _time_history_w1, _time_w1,
 *                   _time_b1, _time_kernel_w2, _time_w12, _time_b12
 *           tvec == NULL selects the plain tf GRUCell (Model/Modules/gru.py:13-39,60-67): no time gate
 *   hs      [B*L, D] outputs (zero for t >= seq_len-1), short_out [B, D]
 *   save    [B*L, 5D] or NULL: per step r | u | c | T | h_prev   (for backward)
 */
int mtam_tagru_fwd(const float *xproj, const float *x, const float *timelast,
                   const int32_t *seq_len, const float *wh_g, const float *wh_c,
                   const float *tvec, int B, int L,
                   float *hs, float *short_out, float *save, void *stream);

/* Backward through time of the above.
 *   d_short [B, D]  gradient of short_out
 *   d_hs    [B*L, D] or NULL: gradient of every output row hs[t] (the family members whose decoder
 *           attends over the GRU outputs, Model/MTAMRec_model.py:180,214)
 *   d_xproj [B*L, 3D] out: d(gate pre-act) | d(candidate pre-act), zero for dead steps
 *   rh      [B*L, D] out: r * h_prev (A operand of the candidate-kernel gradient)
 *   d_xt    [B*L, D] out: the time-gate path of d loss / d x (dtw * _time_kernel_w1), zero for dead steps
 *   d_tvec_partial [B, 8, D] out: per-sample gradients of tvec (caller column-sums)
 */
int mtam_tagru_bwd(const float *d_short, const float *d_hs, const float *x, const float *timelast,
                   const int32_t *seq_len, const float *wh_g, const float *wh_c,
                   const float *tvec, const float *save, int B, int L,
                   float *d_xproj, float *rh, float *d_xt, float *d_tvec_partial,
                   void *stream);
/* The same two launches with the decoder's K/V work riding along as EXTRA workgroups (blockIdx >= B), on the CUs the
 * recurrence leaves idle (one sample per workgroup keeps B = 128 of the 256 CUs busy for ~46 us; a GRU workgroup holds
 * or reserves enough LDS that nothing else fits on its CU, so the extra workgroups can only land on the idle ones):
 *   mtam_tagru_fwd_kv:  also kv_out [B*L, n_kv] = relu(x . Wkv + bkv)   (Model/Modules/time_aware_attention.py:251-253;
 *                       x is the same [B*L, 128] the recurrence reads), from the bf16 operand images of Wkv
 *                       (the layout of mtam_seq_chain_fwd's w_images); n_kv a multiple of 32.  The fused lookups +
 *                       projections launch is then called with n_kv = 0.
 *   mtam_tagru_bwd_dkv: also d_x [B*L, 128] += d_kv [B*L, 256] . Wkv^T (the K/V projection's gradient towards x), from
 *                       the images of Wkv's transpose (mtam_seq_chain_bwd's w_images_r layout); n_kv = 256 (one decoder
 *                       block).  mtam_seq_chain_bwd is then called with n_kv = 0.
 * wkv_images / d_kv == NULL: exactly mtam_tagru_fwd / mtam_tagru_bwd. */
int mtam_tagru_fwd_kv(const float *xproj, const float *x, const float *timelast, const int32_t *seq_len,
                      const float *wh_g, const float *wh_c, const float *tvec, int B, int L, float *hs,
                      float *short_out, float *save, const uint16_t *wkv_images, const float *bkv, int n_kv,
                      float *kv_out, const float *gru_w_image, void *stream);
/* gru_w_image (optional, NULL = the kernel brings wh_g / wh_c through LDS itself, ~5 us per launch): the recurrent
 * weights in the order the forward's lanes hold them -- mtam_gru_weight_image_floats() floats, element (k, n) of the
 * gate matrix wh_g [128, 256] (which = 0) / candidate matrix wh_c [128, 128] (which = 1) at
 * mtam_gru_weight_image_pos(which, k, n); a lane's 96 weights are then 24 coalesced 16-byte loads that overlap the
 * staging of the first chunk.  Written by mtam_gru_weight_image, or kept current by mtam_adam_images
 * (MtamWeightImages.gru_which = 1 / 2 for wh_g / wh_c with `images` = the image). */
int mtam_gru_weight_image_floats(void);
int mtam_gru_weight_image_pos(int which, int k, int n);
int mtam_gru_weight_image(const float *wh_g, const float *wh_c, float *image, void *stream);
int mtam_tagru_bwd_dkv(const float *d_short, const float *d_hs, const float *x, const float *timelast,
                       const int32_t *seq_len, const float *wh_g, const float *wh_c, const float *tvec,
                       const float *save, int B, int L, float *d_xproj, float *rh, float *d_xt,
                       float *d_tvec_partial, const float *d_kv, int n_kv, const uint16_t *wkv_images_t, float *d_x,
                       void *stream);

/* The T-SeqRec cell (TimeAwareGRUCell_sigmoid, Model/Modules/time_aware_rnn.py:19-131, used by
 * MTAM_with_T_SeqRec, Model/MTAMRec_model.py:275-306):
 *   now  = x Wk1 + tanh(timenow  w1 + b1) Wt1 + bias1,  last = x Wk2 + tanh(timelast w2 + b2) Wt2 + bias2
 *   h' = u h sigmoid(now) + (1 - u) c sigmoid(last)
 * Neither gate depends on the state, so both are hoisted: xproj5 [B*L, 5 D] = (gates | candidate | now | last)
 * pre-activations, input halves.  save6 [B*L, 6 D]; d_xproj5 [B*L, 5 D] receives all five gradients.
 * mtam_tsr_time_inputs_fwd: tin [R, 2 D] = (tanh(timenow w1 + b1) | tanh(timelast w2 + b2)), tvec4 rows w1, b1,
 * w2, b2.  mtam_tsr_time_inputs_bwd: out [R, 4 D] = (g_now timenow | g_now | g_last timelast | g_last) with
 * g = d_tin (1 - tin^2); its column sums are the gradients of tvec4. */
int mtam_tagru_seqrec_fwd(const float *xproj5, const int32_t *seq_len, const float *wh_g, const float *wh_c,
                          int B, int L, float *hs, float *short_out, float *save6, void *stream);
int mtam_tagru_seqrec_bwd(const float *d_short, const float *d_hs, const int32_t *seq_len, const float *wh_g,
                          const float *wh_c, const float *save6, int B, int L, float *d_xproj5, float *rh,
                          float *d_xt, float *d_tvec_partial, void *stream);
int mtam_tsr_time_inputs_fwd(const float *timenow, const float *timelast, const float *tvec4, int R, float *tin,
                             void *stream);
int mtam_tsr_time_inputs_bwd(const float *d_tin, const float *tin, const float *timenow, const float *timelast,
                             int R, float *out, void *stream);

/* ------------------------------------------ time-aware attention, T_q = 1
 * One decoder block of vanilla_attention: Model/Modules/time_aware_attention.py:215-456
 * with t_querys_length = 1 (Model/MTAMRec_model.py:83-90), including the
 * residual and normalize() (eps 1e-8, :7-34).  One workgroup per sample.
 *   dec_in [B, D] query;  x [B*L, D] raw keys;  kv [B*L, ld_kv] with K at
 *   column k_off and V at column v_off (= relu(x @ Wk + bk), relu(x @ Wv + bv))
 *   t_query [B] target time, t_keys [B*L], seq_len [B] (key_length)
 *   wqt [D, 2D] = [dense/kernel | _time_input_w],  bq [D]
 *   tparams [5, L]: _time_input_w1, _time_input_b1, time_output_w1, time_output_w2, time_output_b
 *   ln_beta, ln_gamma [D]
 *   dec_out [B, D]
 *   save [B, mtam_ta_attn_decode_save_floats(L, H)] or NULL
 *   head_beta, head_gamma [D], pred_out [B, D], head_save [B, D+1] (or all NULL): the model's head
 *   layer_norm (tf.contrib.layers.layer_norm, eps 1e-12, Model/MTAMRec_model.py:91) applied to the LAST
 *   block's output in the same launch; head_save has mtam_layer_norm_fwd's layout (x_hat | rstd).
 */
int mtam_ta_attn_decode_save_floats(int L, int H);
int mtam_ta_attn_decode_fwd(const float *dec_in, const float *x, const float *kv, int ld_kv,
                            int k_off, int v_off, const float *t_query, const float *t_keys,
                            const int32_t *seq_len, const float *wqt, const float *bq,
                            const float *tparams, const float *ln_beta, const float *ln_gamma,
                            int B, int L, int H, float *dec_out, float *save,
                            const float *head_beta, const float *head_gamma, float *pred_out,
                            float *head_save, void *stream);

/*   d_out [B, D] gradient of dec_out
 *   d_dec_in [B, D] out;  d_kv [B*L, ld_kv] out at k_off / v_off (pre-activation
 *   gradients, relu mask applied);  d_x [B*L, D]: = or += (accumulate_dx) the
 *   raw-key path;  d_qt_pre [B, 2D] out: d(Q pre-act) | d(q @ Wt)
 *   d_tparams_partial [B, 5, L], d_ln_partial [B, 2, D] (beta | gamma): per sample
 *   d_pred [B, D], head_gamma [D], head_save [B, D+1], d_head_partial [B, 2, D] (or all NULL): backward
 *   of the fused head layer_norm; with d_pred given, d_out is not read (may be NULL) and
 *   d_head_partial receives the per-sample d beta | d gamma of the head LN.
 */
int mtam_ta_attn_decode_bwd(const float *d_out, const float *dec_in, const float *x,
                            const float *kv, int ld_kv, int k_off, int v_off,
                            const float *t_query, const float *t_keys, const int32_t *seq_len,
                            const float *wqt, const float *tparams, const float *ln_gamma,
                            const float *save, int B, int L, int H, int accumulate_dx,
                            float *d_dec_in, float *d_kv, float *d_x, float *d_qt_pre,
                            float *d_tparams_partial, float *d_ln_partial,
                            const float *d_pred, const float *head_gamma, const float *head_save,
                            float *d_head_partial, void *stream);

/* ------------------------------------------------------------- layer norm
 * y = LN(x [+ resid]) over the last dimension (D = 128), two forms:
 *   form 0  tf.contrib.layers.layer_norm (Model/Modules/net_utils.py:229-232,
 *           Model/MTAMRec_model.py:91): x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma
 *   form 1  Time_Aware_Attention.normalize (Model/Modules/time_aware_attention.py:7-34,451-454):
 *           gamma*(x-mean)/sqrt(var+eps) + beta, with the residual `resid` added first
 *   save [rows, D + 1]: xhat | rstd, or NULL.  The backward is the same for both forms; its d_x is
 *   also the gradient of `resid`.
 */
int mtam_layer_norm_fwd(const float *x, const float *resid, const float *beta, const float *gamma,
                        float eps, int form, int rows, float *y, float *save, void *stream);
/*   d_bg [2, D]: atomicAdd of d_beta | d_gamma */
int mtam_layer_norm_bwd(const float *d_y, const float *gamma, const float *save, int rows,
                        float *d_x, float *d_bg, void *stream);

/* --------------------------------- time-aware attention, T_q = T_k = L (encoder)
 * Row-wise part of one self_attention block (Model/Modules/time_aware_attention.py:320-431,
 * Model/PISTRec_model.py:38-53); the L x L products run through mtam_gemm_f32_batched.
 *   s_raw [B,H,L,L] = Q_h K_h^T;  a [B,L,L]: in (q Wt) k^T, out tanh of it;  t [B,L] times
 *   tparams [5,L,L]: _time_input_w1, _time_input_b1, time_output_w1, time_output_w2, time_output_b
 *   w [B,H,L,L] = softmax_j(mask(s_raw * sigmoid(gate) / sqrt(d))) with query rows >= seq_len zeroed
 *   dk, sg [B,L,L]: saved tanh(decay) and sigmoid(gate)
 * Backward: dw in = d loss / d w, out = d loss / d s_raw;  d_a = d loss / d ((q Wt) k^T);
 *   g_tparams [5,L,L] += gradients (atomics over the batch).
 */
int mtam_ta_selfattn_gate_softmax_fwd(const float *s_raw, float *a, const float *t, const int32_t *seq_len,
                                      const float *tparams, int B, int L, int H, float *w, float *dk,
                                      float *sg, void *stream);
int mtam_ta_selfattn_gate_softmax_bwd(float *dw, const float *w, const float *s_raw, const float *a,
                                      const float *dk, const float *sg, const float *t,
                                      const int32_t *seq_len, const float *tparams, int B, int L, int H,
                                      float *d_a, float *g_tparams, void *stream);

/* gather_indexes(seq [B*L, D], seq_len + offset) -> out [B, D] (Model/Modules/net_utils.py:82-92) and its
 * gradient: d_src [B*L, D] = 0 except row seq_len[b] + offset of sample b = d_out[b]. */
int mtam_seq_row_gather(const float *src, const int32_t *seq_len, int offset, int B, int L, float *out,
                        void *stream);
int mtam_seq_row_scatter(const float *d_out, const int32_t *seq_len, int offset, int B, int L, float *d_src,
                         void *stream);
/* d[i] = y[i] > 0 ? d[i] : 0 (gradient through tf.nn.relu); n a multiple of 4 */
int mtam_relu_bwd_inplace(float *d, const float *y, size_t n, void *stream);

/* --------------------------------------------------- full-catalog softmax CE
 * log_softmax + one-hot cross entropy over logits [B, V]
 * (Model/base_model.py:316-322).  Two launches inside:
 *   lse[b], ce[b] = lse[b] - logits[b, target[b]]
 *   if d_logits != NULL: d_logits[b, v] = (exp(logits - lse) - [v == target]) * grad_scale
 *   (d_logits may alias logits).  grad_scale = 1 / global batch.
 *   partial: scratch of mtam_softmax_ce_partials(B, V) floats.
 */
int mtam_softmax_ce_partials(int B, int V);
int mtam_softmax_ce(const float *logits, int ld, const int32_t *target, int B, int V,
                    float grad_scale, float *lse, float *ce, float *d_logits,
                    float *partial, void *stream);

/* loss[0] = reg * 0.5 * sum(l2_partial) + ce_scale * sum(ce);  loss[1] = 0.5*sum(l2_partial);
 * loss[2] = sum(ce) / B   (Model/base_model.py:322-326) */
int mtam_loss_reduce(const float *l2_partial, int n_l2, const float *ce, int B, float reg,
                     float ce_scale, float *loss, void *stream);

/* mtam_softmax_ce + mtam_loss_reduce in one call; for V <= 16384 also in ONE launch (a workgroup
 * per row keeps the row in registers, the last workgroup to finish reduces the loss).
 * partial: mtam_softmax_ce_partials(B, V) + 4 floats, ZEROED ONCE by the caller before the first
 * call (its first word is an arrival ticket that the kernel resets itself).  loss == NULL: no loss
 * reduction here (mtam_sqnorm_clip_scale can do it at the end of the step instead). */
int mtam_softmax_ce_loss(const float *logits, int ld, const int32_t *target, int B, int V,
                         float grad_scale, float *lse, float *ce, float *d_logits, float *partial,
                         const float *l2_partial, int n_l2, float reg, float ce_scale, float *loss,
                         void *stream);

/* ------------------------------------------- bf16 scoring without logits (BASELINE.json configs[4])
 * The same loss and gradients as the block above (Model/base_model.py:300-328, 290-297) computed from a
 * bf16 copy of the item table with bf16 MFMA and fp32 accumulation, the [B, V] logits never stored:
 *   E16 [V, 128] bf16 bits, P16 [mtam_score16_batch_pad(B), 128] bf16 bits (rows >= B zero) -- both made
 *   by mtam_f32_to_bf16 (round to nearest even; n_dst >= n_src, the tail is zero-filled; counts % 4 == 0).
 *   mtam_score16_lse    lse[b], ce[b] = lse[b] - <P16[b], E16[target[b]]>; partial: scratch of
 *                       mtam_score16_partials(B, V) floats
 *   mtam_score16_bwd    with G = (exp(score - lse) - onehot) * scale (rounded to bf16):
 *                       d_pred [B, 128] += G E16 (fp32 atomics: the caller zeroes it),
 *                       dE [V, 128] = G^T P16 (stored, every row), and if sq_partial != NULL (scratch of
 *                       mtam_score16_sq_partials(V) floats) the per-wave sums of dE^2
 *   mtam_score16_logits evaluation: logits [B, ld] fp32 for mtam_topk
 */
int mtam_f32_to_bf16(const float *src, size_t n_src, uint16_t *dst, size_t n_dst, void *stream);
int mtam_score16_batch_pad(int B);
int mtam_score16_partials(int B, int V);
int mtam_score16_sq_partials(int V);
int mtam_score16_lse(const uint16_t *E16, const uint16_t *P16, const int32_t *target, int B, int V,
                     float *partial, float *lse, float *ce, void *stream);
int mtam_score16_bwd(const uint16_t *E16, const uint16_t *P16, const float *lse, const int32_t *target, int B,
                     int V, float scale, float *d_pred, float *dE, float *sq_partial, void *stream);
int mtam_score16_logits(const uint16_t *E16, const uint16_t *P16, int B, int V, float *logits, long ld,
                        void *stream);

/* ------------------------------------------- fp32 scoring without logits (every catalog size)
 * The pair above with fp32 operands: E [V, 128] and pred [B, 128] as they are (no copies, no padding).  It replaces
 * the logits GEMM, mtam_softmax_ce and the two scoring-gradient GEMMs of a training step
 * (Model/base_model.py:300-328, 290-297); no [B, V] buffer exists.  Two forms of the same two passes:
 *   split-bf16 (default): every product as six v_mfma_f32_32x32x16_bf16 terms of operands split three ways
 *                         (fp32-equivalent, csrc/split_bf16.h); the backward runs in three wave roles with transposed
 *                         LDS reads (csrc/score32.hip);
 *   native fp32:          v_mfma_f32_32x32x2_f32 (each score a k-ordered fmaf chain).
 * Catalogs of at least min_rows rows take the split form (default 1 = always; 0 = never; MTAM_SCORE32_SPLIT_MIN_ROWS
 * at load, the setter at run time).  The buffer sizes below depend on the form: set it before sizing.
 * Evaluation keeps the stored-logits GEMM (its k-ordered fmaf chain is the ranking contract of mtam_topk).
 *   partial: mtam_score32_partials(B, V) floats; sq_partial: mtam_score32_sq_partials(V) floats or NULL;
 *   n_partial / n_sq_partial: how many floats the caller's buffers hold -- checked against the CURRENT form's
 *   counts (n_partial >=, n_sq_partial ==: its consumer sums exactly that many), so a buffer sized before
 *   mtam_score32_set_split_min_rows() changed the form is rejected instead of overrun;
 *   d_pred is accumulated (the caller zeroes it), dE [V, 128] is stored. */
void mtam_score32_set_split_min_rows(long min_rows);
int mtam_score32_partials(int B, int V);
int mtam_score32_sq_partials(int V);
int mtam_score32_lse(const float *E, const float *pred, const int32_t *target, int B, int V, float *partial,
                     int n_partial, float *lse, float *ce, void *stream);
int mtam_score32_bwd(const float *E, const float *pred, const float *lse, const int32_t *target, int B, int V,
                     float scale, float *d_pred, float *dE, float *sq_partial, int n_sq_partial, void *stream);
/* The same two passes over a ROW RANGE of the catalog (data-parallel row-sharded scoring, SURVEY.md 8(e): every rank
 * scores its own V / G rows against the all-gathered pred of the whole batch; Model/base_model.py:309-322): E points
 * at the range's first row, V = rows in the range, row0 = that row's catalog number, `target` holds CATALOG row
 * numbers.  mtam_score32_lse_range: lse[b] = log-sum-exp over the range's rows only and ce[b] = the TARGET LOGIT if
 * the target lies in the range, else 0 (the caller combines the ranks' pairs: lse = log sum_r exp(lse_r), logit =
 * sum_r logit_r).  mtam_score32_bwd_range: `lse` is the combined one; a target outside the range gets no one-hot
 * term; dE [V, 128] covers the range's rows and is COMPLETE (summed over the whole batch); d_pred is the range's
 * share and is summed over ranks by the caller. */
int mtam_score32_lse_range(const float *E, const float *pred, const int32_t *target, int B, int V, int row0,
                           float *partial, int n_partial, float *lse, float *ce, void *stream);
int mtam_score32_bwd_range(const float *E, const float *pred, const float *lse, const int32_t *target, int B, int V,
                           int row0, float scale, float *d_pred, float *dE, float *sq_partial, int n_sq_partial,
                           void *stream);

/* Training's scoring as ONE call (Model/base_model.py:309-328, the loss, and 290-297, its two scoring gradients):
 * ce[b] = lse[b] - logit of the target, d_pred += G E, dE = G^T pred (stored) with G = (softmax - onehot) * scale.
 * With one batch tile (B <= 128) and 8 <= ceil(V / 32) <= 168 (and 7/8 of the device's CU count; 3,709 rows: 116 slabs) this is ONE
 * launch: every 32-row slab has a resident workgroup of its own, the scores stay in registers across a grid barrier,
 * the per-slab shares of d_pred are summed in slab order (deterministic; no float atomics), against
 * 6.3 + 4.7 + 15.8 us for the three launches of lse + bwd.  mtam_score32_train_is_fused() says whether that form
 * applies (MTAM_SCORE32_FUSED=0 turns it off -- required when several processes share one GPU); otherwise the call is
 * mtam_score32_lse followed by mtam_score32_bwd.  `work`: mtam_score32_train_work_floats(B, V) floats, 16-byte aligned,
 * prepared ONCE by mtam_score32_train_work_init (the fused form's exchange buffers must start as "nothing published")
 * and then left to mtam_score32_train; one buffer serves one (B, V) and one stream. */
void mtam_score32_set_fused(int on);      /* run-time form of MTAM_SCORE32_FUSED; work buffers are sized per form */
int mtam_score32_train_is_fused(int B, int V);
long mtam_score32_train_work_floats(int B, int V);
int mtam_score32_train_work_init(float *work, long n_work, int B, int V, void *stream);
int mtam_score32_train(const float *E, const float *pred, const int32_t *target, int B, int V, float scale,
                       float *work, long n_work, float *lse, float *ce, float *d_pred, float *dE, float *sq_partial,
                       int n_sq_partial, void *stream);

/* ------------------------------------------- the forward's three sequence-side projections in one launch
 *   zr = relu(ic W4) ; x = zr + pos                       (Embedding/...attention.py:95-103; = mtam_gemm_f32 RELU_ADD)
 *   kv = relu(x Wkv + bkv)  [R, n_kv]   (n_kv may be 0)    (time_aware_attention.py:251-253;   = BIAS_RELU)
 *   xproj = x Wx + bx       [R, n_x]                       (time_aware_rnn.py:243-256, input halves; = BIAS)
 * ic [R, 256], W4 [256, 128], pos [R, 128], Wkv [128, n_kv], Wx [128, n_x]; n_kv, n_x multiples of 32;
 * every output 16-byte aligned.  A workgroup owns a 32-row stripe; x stays on the CU between the products. */
int mtam_seq_chain_fwd(const float *ic, const float *W4, const float *pos, int R, const float *Wkv,
                       const float *bkv, int n_kv, const float *Wx, const float *bx, int n_x, float *zr, float *x,
                       float *kv, float *xproj, const uint16_t *w_images, void *stream);
/* w_images (both chain entry points): NULL = the three products on v_mfma_f32_32x32x2_f32.  Otherwise the bf16
 * operand images of W4, Wkv and Wx -- one buffer of mtam_seq_chain_images_elems(n_kv, n_x) bf16 values,
 * [W4 | Wx | Wkv] (mtam_seq_chain_image_offset(which, n_x) = where matrix `which` = 0 (W4), 1 (Wkv), 2 (Wx) starts;
 * Wkv last, so a launch with n_kv = 0 reads the same buffer), each matrix as
 * three images (W = W1 + W2 + W3 exactly, bf16 each; image t at t K N) in the order [K / 8][N][8] -- and every
 * product runs as six v_mfma_f32_32x32x16_bf16 terms with fp32 accumulation: fp32-equivalent (the dropped terms
 * are <= 2^-23 |a b|), 448 fp32 matrix instructions of 64 cycles per wave become 336 of 32.  The images are
 * written by the optimizer launch that updates the weights (mtam_adam_images) -- no per-step prepare launch --
 * or, whenever the weights change any other way, by mtam_split_weight_images (one launch per matrix). */
size_t mtam_seq_chain_images_elems(int n_kv, int n_x);
size_t mtam_seq_chain_image_offset(int which, int n_x);
int mtam_split_weight_images(const float *W, int K, int N, uint16_t *images, void *stream);
/* The backward's sequence-side chain in one launch (the mirror of mtam_seq_chain_fwd; replaces the dual-source
 * mtam_gemm_f32_dual(ACCUM2_MASK) and the d[item | category] GEMM; tf.gradients of
 * Embedding/Behavior_embedding_time_aware_attention.py:95-103, Model/Modules/time_aware_attention.py:251-253,
 * Model/Modules/time_aware_rnn.py:243-256):
 *   d_x  [R, 128] in/out: += d_xproj [R, n_x] . Wx^T + d_kv [R, n_kv] . Wkv^T + d_xt [R, 128]
 *   d_z  [R, 128] out   : d_x where zr > 0
 *   d_ic [R, 256] out   : d_z . W4^T
 * n_x, n_kv multiples of 128, n_x + n_kv <= mtam_seq_chain_bwd_max_k() (n_kv = 0, d_kv = NULL: no key / value source;
 * wider models keep the two GEMM launches).  A workgroup owns a 32-row stripe; d_z
 * stays on the CU between the two products; split-bf16 products (fp32-equivalent).  w_images_r: the bf16 images of
 * the three matrices' TRANSPOSES (the B operands of products with W^T), one buffer laid out like w_images
 * ([W4 | Wx | Wkv], the same offsets): matrix W [K, N] as three terms (term t at t K N), element W[k][n] at
 * ((n >> 3) K + k) 8 + (n & 7), N a multiple of 8.  Written by mtam_adam_images (images_r) or mtam_split_weight_rows. */
int mtam_split_weight_rows(const float *W, int K, int N, uint16_t *images_r, void *stream);
int mtam_seq_chain_bwd_max_k(void); /* largest n_x + n_kv the staged stripe holds (640: one decoder block) */
int mtam_seq_chain_bwd(const float *d_xproj, int n_x, const float *d_kv, int n_kv, const float *d_xt, const float *zr,
                       int R, float *d_x, float *d_z, float *d_ic, const uint16_t *w_images_r, void *stream);
/* The same launch with the four embedding lookups folded in (mtam_emb_gather_fwd + mtam_seq_chain_fwd as ONE
 * kernel: SURVEY.md 2.1 K1 + K2): a workgroup gathers its stripe's [item | category] rows straight into the LDS
 * operand of the first product and its position rows into the epilogue registers; the rows are never read back
 * from HBM.  Also produced: the tf.nn.l2_loss partial sums (l2_partial[0 .. mtam_seq_chain_gather_partials(B, L));
 * the rest of the n_l2 entries is zeroed), user_out [B, D], and -- ic_out != NULL, training -- a copy of the
 * [item | category] rows [B*L, 2D] for the dense4emb weight gradient and the scatter-add's L2 term.  The
 * position rows are not written (mtam_emb_scatter_add_bwd_postab reads them through the ids).  clear_a / clear_b:
 * float ranges zeroed on the side (the step's gradient accumulators), as in mtam_emb_gather_fwd_clear.
 * HBM bytes per sequence at L = 50, fp32: (3L+1)(512 + 4) table reads + ids, + L x 1,024 B of ic_out in training
 * (evaluation: none), against (3L+1)(2 x 512 + 4) + L x 1,024 re-read for the two-kernel form. */
int mtam_seq_chain_gather_partials(int B, int L);
int mtam_seq_chain_gather_fwd(const float *item_table, int item_rows, const float *cat_table, int cat_rows,
                              const float *pos_table, int pos_rows, const float *user_table, int user_rows,
                              const int32_t *item_ids, const int32_t *cat_ids, const int32_t *pos_ids,
                              const int32_t *user_ids, int B, int L, int with_user, const float *W4,
                              const float *Wkv, const float *bkv, int n_kv, const float *Wx, const float *bx,
                              int n_x, float *ic_out, float *user_out, float *l2_partial, int n_l2, float *zr,
                              float *x, float *kv, float *xproj, float *clear_a, size_t n_clear_a, float *clear_b,
                              size_t n_clear_b, const uint16_t *w_images, void *stream);

/* ------------------------------------------------------------------ top-K
 * tf.nn.top_k (Model/base_model.py:196-200): for every row the k largest
 * scores, descending, equal values -> lower index first.  k <= 64.
 *   idx_out [rows, k] int32; val_out [rows, k] or NULL
 */
int mtam_topk(const float *scores, int ld, int rows, int V, int k,
              int32_t *idx_out, float *val_out, void *stream);
/* The same result for long rows in two launches: every row is cut into segments that one workgroup each
 * reduces to k candidates, and a second pass picks the k of the row (ties still -> lower index).
 * workspace: mtam_topk_workspace_bytes(rows, V, k) bytes (0 = rows too short to split; NULL = single pass). */
size_t mtam_topk_workspace_bytes(int rows, int V, int k);
int mtam_topk_ws(const float *scores, int ld, int rows, int V, int k, int32_t *idx_out, float *val_out,
                 void *workspace, void *stream);
/* The same result WITHOUT a stored [rows, V] score matrix: the evaluation sess.run (Model/base_model.py:194-202)
 * at catalog sizes where predict_behavior_emb . item_table^T does not fit (25.6 GB at 128 x 50 M).  The caller scores
 * the catalog slab by slab (mtam_gemm_f32 / mtam_score16_logits on a row range of the item table: each score is the
 * same k-ordered fmaf chain as in the one-piece product) into a [rows, ld] scratch and hands every slab to
 * mtam_topk_stream_slab: columns [col0, col0 + width) of the full matrix, col0 a multiple of MTAM_TOPK_STREAM_SEG.
 * Every (row, segment of MTAM_TOPK_STREAM_SEG columns) keeps its k best in `workspace`
 * (mtam_topk_stream_workspace_bytes(rows, V, k) bytes, any contents before the first slab; every segment of
 * [0, V) must be handed in once); mtam_topk_stream_finish then writes the rows' top k.  Lists are identical to
 * mtam_topk on the stored matrix, ties across slab and segment boundaries included. */
#define MTAM_TOPK_STREAM_SEG 65536
int mtam_topk_stream_segments(int V);
size_t mtam_topk_stream_workspace_bytes(int rows, int V, int k);
int mtam_topk_stream_slab(const float *slab_scores, int ld, int rows, int col0, int width, int V, int k,
                          void *workspace, void *stream);
int mtam_topk_stream_finish(void *workspace, int rows, int V, int k, int32_t *idx_out, float *val_out, void *stream);

/* ------------------------------------------------ clip_by_global_norm + Adam
 * tf.clip_by_global_norm + AdamOptimizer.apply_gradients
 * (Model/base_model.py:75-76,294-296) [TF1.14 semantics, SURVEY.md App D-5/6].
 *
 * mtam_sqnorm_partial: partial[i] = per-block sum of g^2 over n floats;
 *   needs mtam_sqnorm_blocks(n) floats.
 * mtam_clip_scale: scale[0] = clip * min(1/norm, 1/clip), scale[1] = norm, with
 *   norm = sqrt(sum of all partials).  With lr / adam_state given it also does
 *   Adam's per-step bookkeeping on the device (so a captured step needs no host
 *   arithmetic): adam_state [8] = {lr_t, beta1, beta2, eps, beta1_power,
 *   beta2_power, -, -}; lr_t = lr[0]*sqrt(1-beta2_power)/(1-beta1_power) is
 *   written from the CURRENT powers, which are then multiplied by beta1/beta2
 *   (AdamOptimizer._prepare/_finish; powers start at beta1, beta2).
 * mtam_adam: p, m, v updated from g * scale[0]; hyper [4] (device) =
 *   lr_t, beta1, beta2, eps (= the head of adam_state).  Elements at index >=
 *   sparse_begin use m = m*b1 + g*(1-b1) (IndexedSlices path, tables); the ones
 *   before it m += (g-m)*(1-b1) (dense ApplyAdam kernel).  sparse_begin must be a
 *   multiple of mtam_adam_block() (or >= n: all dense; 0: all tables).
 */
int mtam_sqnorm_blocks(size_t n);
int mtam_sqnorm_partial(const float *g, size_t n, float *partial, void *stream);
int mtam_clip_scale(const float *partials, int n_partials, float clip_norm, float *scale,
                    const float *lr, float *adam_state, void *stream);
/* Data-parallel training with the item table's gradient reduce-scattered by row range (the reference
 * has no multi-GPU path): every rank sums the squares of the gradient elements it OWNS,
 *   mtam_partials_sum: out[0] (+)= weight * sum(partials[0 .. n))   in double (accumulate != 0: added to out[0]),
 * the per-rank doubles are all-reduced by the caller (RCCL), and
 *   mtam_clip_scale_sq: mtam_clip_scale with norm = sqrt(sq_total[0] + .. + sq_total[n - 1]),  n <= 64. */
int mtam_partials_sum(const float *partials, int n, float weight, double *out, int accumulate, void *stream);
int mtam_clip_scale_sq(const double *sq_total, int n, float clip_norm, float *scale, const float *lr,
                       float *adam_state, void *stream);
/* mtam_sqnorm_partial(g -> partials[offset...]) and mtam_clip_scale over partials[0 .. n_total) in ONE
 * launch (the last workgroup to finish does the reduction).  ticket: one device word, zero on the
 * first call, reset by the kernel.  With `loss` given the same workgroup also does mtam_loss_reduce
 * (the step's reported loss), so a training step needs no separate loss launch. */
int mtam_sqnorm_clip_scale(const float *g, size_t n, float *partials, int offset, int n_total,
                           float clip_norm, float *scale, const float *lr, float *adam_state,
                           unsigned int *ticket, const float *l2_partial, int n_l2, const float *ce,
                           int B, float reg, float ce_scale, float *loss, void *stream);
int mtam_adam_block(void);
int mtam_adam(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
              const float *hyper, size_t sparse_begin, void *stream);
/* mtam_adam that also writes the updated elements [copy_begin, n) as bf16 (round to nearest even) to
 * copy16[0 .. n - copy_begin): the scoring copy of the item table (mtam_score16_*). */
int mtam_adam_bf16copy(float *p, float *m, float *v, const float *g, size_t n, const float *scale,
                       const float *hyper, size_t sparse_begin, uint16_t *copy16, size_t copy_begin,
                       void *stream);
/* mtam_adam (copy16 == NULL) / mtam_adam_bf16copy that ALSO re-writes the bf16 operand images of up to
 * MTAM_MAX_WEIGHT_IMAGES weight matrices lying inside [0, n) -- the B operands of mtam_seq_chain_*'s split-bf16
 * products (see there for the layout) -- from the values it has just updated: the forward of the next step finds
 * its weights already split and laid out as MFMA fragments (Embedding/...attention.py:95-103 dense4emb,
 * time_aware_attention.py:251-253 K / V, time_aware_rnn.py:243-256 input halves).  begin: first element of the
 * row-major [K, N] matrix in the flat space (multiple of 4; K multiple of 8, N of 4). */
#define MTAM_MAX_WEIGHT_IMAGES 6
typedef struct {
  size_t begin;
  int K, N;
  uint16_t *images;   /* [K / 8][N][8] per term: the forward's B fragments (mtam_seq_chain_fwd) */
  uint16_t *images_r; /* the same layout of W^T: the backward's (mtam_seq_chain_bwd); may be NULL */
  int gru_which;      /* 0: the bf16 images above.  1 / 2: W is the GRU's wh_g / wh_c and `images` is the fp32
                         register-order image of mtam_gru_weight_image (images_r unused) */
} MtamWeightImages;
int mtam_adam_images(float *p, float *m, float *v, const float *g, size_t n, const float *scale, const float *hyper,
                     size_t sparse_begin, uint16_t *copy16, size_t copy_begin, const MtamWeightImages *w, int n_w,
                     void *stream);
/* The clip + update pair WITHOUT an arrival ticket (Model/base_model.py:290-297 clip_by_global_norm, :71-80 the
 * optimizer).  mtam_sqnorm_state_loss: partials[offset + i] = sum of squares of block i of g (mtam_sqnorm_blocks(n)
 * blocks) and, in one more workgroup, the Adam state advanced (lr / adam_state as in mtam_sqnorm_clip_scale; NULL =
 * not) and the reported loss reduced (loss NULL = not) -- nothing here waits for the norm.  mtam_adam_images_clip:
 * mtam_adam_images where EVERY workgroup sums norm_partials[0 .. n_partials) itself (float64, one fixed order) and
 * derives the clip scale; scale_out[0] = the scale, [1] = the norm.  n_partials <= mtam_adam_clip_max_partials().
 * 2.6 + 12.4 us against 7.0 + 11.9 for mtam_sqnorm_clip_scale + mtam_adam_images at ml-1m sizes. */
int mtam_sqnorm_state_loss(const float *g, size_t n, float *partials, int offset, const float *lr, float *adam_state,
                           const float *l2_partial, int n_l2, const float *ce, int B, float reg, float ce_scale,
                           float *loss, void *stream);
int mtam_adam_clip_max_partials(void);
int mtam_adam_images_clip(float *p, float *m, float *v, const float *g, size_t n, const float *norm_partials,
                          int n_partials, float clip_norm, float *scale_out, const float *hyper, size_t sparse_begin,
                          uint16_t *copy16, size_t copy_begin, const MtamWeightImages *w, int n_w, void *stream);
/* mtam_adam_images_clip that also hands the NEXT step its feed (Model/base_model.py:150-164: the feed_dict of the
 * next sess.run): feed_ring holds feed_slots packed feed arenas of feed_words int32 words each, already in HBM (a
 * device-resident epoch, or slots a loader fills ahead of the step); ONE more workgroup of the launch copies slot
 * (feed_cursor[0] % feed_slots) into feed_arena -- the arena every kernel of the step reads its ids, times and learning
 * rate from -- and adds one to feed_cursor[0].  Nothing in the optimizer launch reads the arena, so the copy rides in
 * the update's shadow instead of standing in front of the next step's first kernel.  Contract: the slot the cursor
 * points at is complete before this launch starts; the caller primes the first step (slot -> arena, cursor = index of
 * the slot after it).  feed_words a multiple of 4, ring and arena 16-byte aligned and disjoint. */
int mtam_adam_images_clip_feed(float *p, float *m, float *v, const float *g, size_t n, const float *norm_partials,
                               int n_partials, float clip_norm, float *scale_out, const float *hyper,
                               size_t sparse_begin, uint16_t *copy16, size_t copy_begin, const MtamWeightImages *w,
                               int n_w, const int32_t *feed_ring, int feed_slots, int feed_words, int32_t *feed_arena,
                               unsigned int *feed_cursor, void *stream);

/* The other choices of base_model.init_optimizer (Model/base_model.py:71-80):
 * kind 0 GradientDescentOptimizer, 1 AdadeltaOptimizer (rho 0.95, eps 1e-8; slot1 = accum,
 * slot2 = accum_update, both start at 0), 2 RMSPropOptimizer (decay 0.9, momentum 0, eps 1e-10;
 * slot1 = rms starting at ONE, slot2 = momentum starting at 0) [TF1.14 training_ops formulas].
 * p and the slots are updated from g * scale[0] with the raw learning rate lr[0] (device).
 * Elements >= sparse_begin are tables (sparse kernel forms); in [sparse_begin, rowskip_end) a
 * 128-float row whose gradient is entirely zero is not in the batch's IndexedSlices and is left
 * untouched, slots included (TF applies sparse updates to the looked-up rows only).
 */
int mtam_opt_update(int kind, float *p, float *slot1, float *slot2, const float *g, size_t n,
                    const float *scale, const float *lr, size_t sparse_begin, size_t rowskip_end,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MTAM_HIP_H */
