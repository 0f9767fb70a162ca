/* hgn_mp.h -- C ABI of libhgn_mp.so: the MI355X (gfx950) message-passing hot path of HyperGraphNets.
 *
 * The reference (CemOezcan/hyper-graph-nets) has NO native/FFI layer: its hot path is Python calling ATen and the
 * third-party torch_scatter wheel.  Each entry point below therefore cites the reference *Python* code whose
 * arithmetic it replaces (paths relative to the reference root); the Python host in
 * hyper-graph-nets_amd/hgn_amd binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every function returns 0 on success, <0 on failure (HGN_E_*); no exception crosses the ABI;
 *    hgn_last_error() returns a thread-local message for the last failure;
 *  - all pointers are DEVICE pointers unless marked host; fp32 data, int32 indices (int64 only where the
 *    reference hands over int64 ids, i.e. hgn_csr_build / hgn_narrow_gather_i64);
 *  - the library never allocates, frees or synchronises (except hgn_csr_build's range check, which is
 *    topology preprocessing); the caller owns every buffer including workspaces; all work is enqueued on
 *    the hipStream_t passed as `stream` (opaque void*); re-entrant; the only process-wide mutable state is the
 *    optional profiler and ONE word, the meaning of `products == 0` in the argument structs
 *    (hgn_set_matmul_products; every caller that fills `products` itself -- the Python host always does -- is
 *    independent of it);
 *  - latent width is fixed at 128 (reference: src/model/flag.py:57 `latent_size=128`), MLPs have two hidden
 *    layers (flag.py:58 `num_layers=2`): Linear-ReLU-Linear-ReLU-Linear[-LayerNorm].
 */
#ifndef HGN_MP_H
#define HGN_MP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGN_OK 0
#define HGN_E_INVALID (-1)  /* bad argument (null pointer, bad size, unsupported width) */
#define HGN_E_LAUNCH (-2)   /* HIP runtime / launch failure                               */
#define HGN_E_RANGE (-3)    /* a segment id / node index outside [0, num_segments)        */

#define HGN_OP_SUM 0
#define HGN_OP_MEAN 1
#define HGN_OP_MAX 2
#define HGN_OP_MIN 3
#define HGN_F_FP32_MFMA 1
#define HGN_F_GENERAL_FWD 2
#define HGN_F_TILE64_FWD 4   /* A/B switch: 64-row forward tiles (three workgroups per CU) also for launches that would take 128-row tiles */
#define HGN_F_DEFER_LN 8     /* hgn_mlp_bwd / hgn_edge_bwd_fused: leave the LayerNorm partial slabs in ln_ws; the caller sums them later with hgn_ln_reduce_batch */

#define HGN_MAX_SRC 8
#define HGN_MAX_ADD 2
#define HGN_MAX_WTASK 16
/* a pending sum of chunk slabs (hgn_mlp_wgrad_partial / hgn_edge_bwd_fused_partial -> hgn_slab_reduce_batch, see there) */
#define HGN_MAX_WRED 48
typedef struct {
  int32_t type; int32_t K; int32_t n_out; int32_t accumulate; int32_t n_chunks; int32_t reserved;
  float* dW; int64_t ldw; float* db;
  const float* slab; int64_t chunk_stride;   /* first chunk's slab and the distance between two chunks' slabs, in floats */
} hgn_wred_task_t;

const char* hgn_last_error(void);
int hgn_version(void);

/* ------------------------------------------------------------------------------------------------------
 * Topology: receiver-sorted (CSR) view of an edge list.   Replaces the id broadcast
 * `segment_ids.repeat_interleave(...)` of src/util.py:107-110 (half of the reference's CPU step time) and
 * makes every later pass over edge latents a coalesced stream.
 *   ids[E] int64 (as produced by src/util.py:66-67)  ->  perm[E]  (sorted position -> original edge, stable),
 *   seg[E] (sorted ids), rowptr[N+1].
 * ---------------------------------------------------------------------------------------------------- */
int hgn_csr_workspace_bytes(int64_t num_edges, int64_t num_segments, size_t* bytes /*host*/);
int hgn_csr_build(const int64_t* ids, int64_t num_edges, int64_t num_segments, int32_t* perm, int32_t* seg,
                  int32_t* rowptr, void* workspace, size_t workspace_bytes,
                  int32_t* max_rows /*host, nullable: length of the longest row, from the same read-back as the range check*/,
                  void* stream);
/* Fingerprints of index CONTENT (two 64-bit words: one per array; b may be null): equal arrays give equal words whatever
 * tensor object holds them.  The reference builds fresh batched index tensors for every batch (MeshSimulator.py:159-234) although
 * all batches of a trajectory share one mesh: the host keys its topology cache on these words, so the sorts run once per
 * mesh, not once per step.  out_dev: 16 bytes of device scratch; out_host (nullable): the words are copied there and the
 * stream is synchronised (like hgn_csr_build: topology preprocessing). */
int hgn_index_fingerprint(const int64_t* a, const int64_t* b, int64_t n, uint64_t* out_dev, uint64_t* out_host /*host*/,
                          void* stream);
/* dst[i] = (int32) src[perm ? perm[i] : i]   (reorders the *other* endpoint list into sorted order) */
int hgn_narrow_gather_i64(const int64_t* src, const int32_t* perm, int64_t n, int32_t* dst, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * a2: util.unsorted_segment_operation (src/util.py:92-134) + GraphNet.aggregation (graphnet.py:50-70).
 * One pass computes up to four aggregates (order given by ops[]) of data rows grouped by CSR row:
 *   out[n][slot*D + d] = op_slot over j in [rowptr[n], rowptr[n+1]) of data[perm ? perm[j] : j][d]
 * empty segment -> 0 for every op; mean = sum / max(count,1); max/min ties -> first element of the segment.
 * argmax/argmin (int32 [N,D], CSR positions, -1 for empty) are written when non-null (needed by bwd).
 * ---------------------------------------------------------------------------------------------------- */
int hgn_segment_reduce_fwd(const float* data, int64_t ld_data, int D, const int32_t* perm, const int32_t* rowptr,
                           int64_t num_segments, const int32_t* ops /*host*/, int n_ops, float* out,
                           int64_t ld_out, int32_t* argmax, int32_t* argmin, void* stream);
/* d_data[pos][d] = (base ? base[pos][d] : 0) + sum_slot dop_slot(...)   with pos = perm ? perm[j] : j */
int hgn_segment_reduce_bwd(const float* d_out, int64_t ld_out, int D, const int32_t* perm, const int32_t* seg,
                           const int32_t* rowptr, int64_t num_edges, const int32_t* ops /*host*/, int n_ops,
                           const int32_t* argmax, const int32_t* argmin, const float* base, float* d_data,
                           int64_t ld_data, void* stream);

/* hgn_segment_reduce_bwd for D = 128 and rows already in segment order (perm == null), parallel over the N SEGMENTS: rows
 * rowptr[n] .. rowptr[n+1] get base + the aggregation backward of segment n; d_out / argmax / argmin are indexed by n (n < N).  Same result
 * bit for bit; a segment's gradient and arg rows are loaded once instead of once per row. */
int hgn_segment_reduce_bwd_sorted(const float* d_out, int64_t ld_out, const int32_t* rowptr, int64_t N, const int32_t* ops /*host*/, int n_ops,
                                  const int32_t* argmax, const int32_t* argmin, const float* base, float* d_data, int64_t ld_data,
                                  void* stream);

/* Two `sum` reductions of the same [E,128] rows in one pass: out_a[n] = sum of rows rowptr_a[n] .. rowptr_a[n+1] (the rows are sorted by that
 * key), out_b[n] = sum of rows perm_b[rowptr_b[n] ..] (a second key through its CSR permutation); n < N for both.  What the backward of
 * the split first edge layer needs over dz1 (graphnet.py:22-32: the gradient of h[receivers] and of h[senders]); bit-identical to two
 * hgn_segment_reduce_fwd calls, one read of the rows from HBM instead of two on mesh-like graphs. */
int hgn_segment_sum_pair(const float* data, int64_t ld, const int32_t* rowptr_a, const int32_t* perm_b, const int32_t* rowptr_b, int64_t N,
                         float* out_a, int64_t ld_a, float* out_b, int64_t ld_b, void* stream);
/* The fifth operation of util.unsorted_segment_operation, 'std' (src/util.py:129-130 -> torch_scatter.scatter_std with its default
 * unbiased = True; unreachable from the reference's configs).  torch-scatter 2.0.9 (torch_scatter/composite/std.py) as published:
 *   count = max(#rows of the segment, 1);  mean = sum / count;  out = sqrt( sum (x - mean)^2 / (max(count - 1, 1) + 1e-6) )
 * so an empty segment gives 0.  `mean` (nullable, [N, ld_out]) is written for the backward pass, which is the autograd of that
 * composite:  d_data[pos][d] = d_out[n][d] (x - mean[n][d]) / (out[n][d] (max(count - 1, 1) + 1e-6))  -- NaN (0 / 0) for the rows of a
 * segment without variance, exactly like the wheel's sqrt'(0) * 0. */
int hgn_segment_std_fwd(const float* data, int64_t ld_data, int D, const int32_t* perm, const int32_t* rowptr, int64_t N,
                        float* out, int64_t ld_out, float* mean, void* stream);
int hgn_segment_std_bwd(const float* d_out, const float* out, const float* mean, int64_t ld_out, const float* data,
                        int64_t ld_data, int D, const int32_t* perm, const int32_t* seg, const int32_t* rowptr, int64_t E,
                        float* d_data, int64_t ld_d, void* stream);


/* ------------------------------------------------------------------------------------------------------
 * a1/a3/a5: fused MLP  out = [res +] [LN](W3 relu(W2 relu(z1) + b2) + b3),
 *   z1 = b1 + sum_src W1[:, src.col0 : src.col0+src.K] * x_src[idx_src ? idx_src[i] : i]
 *           + sum_add P_add[idx_add[i]][col0 : col0+128]
 * Replaces  _update_edge_features (graphnet.py:22-32: two index_select + cat + 3 Linear + LN + add),
 *           _update_node_features / _update_hyper_node_features / _update_down (graphnet.py:34-48,94-124:
 *           cat + 3 Linear + LN + add) and LazyMLP (meshgraphnet.py:93-108) for encoder / decoder.
 * Weights are nn.Linear layout [out][in] addressed in place (pointer + leading dimension); nothing is
 * repacked.  Saved activations (z1, z2 post-ReLU, xhat, rstd) are written when non-null (training).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
  const float* x;      /* [rows, ld] source rows                                   */
  int64_t ld;
  int32_t K;           /* columns of this source (any K >= 1)                       */
  const int32_t* idx;  /* optional row gather index [M]                             */
  const float* W;      /* &W1[0][col0]: weight columns that multiply this source    */
  const void* Wpk;     /* optional: the same columns packed by hgn_pack_bf16x3 (forward form), K/128 consecutive blocks */
} hgn_src_t;

typedef struct {
  const float* P;      /* pre-projected rows [*, ld]; adds P[idx[i]][0..128)         */
  int64_t ld;
  const int32_t* idx;  /* required                                                   */
} hgn_add_t;

typedef struct {
  int64_t M;                       /* rows                                            */
  int32_t n_src;
  hgn_src_t src[HGN_MAX_SRC];
  int32_t n_add;
  hgn_add_t add[HGN_MAX_ADD];
  int64_t ldw1;                    /* leading dimension of W1 (= in_features)          */
  const float* b1;
  const float* W2; const float* b2;   /* [128][128], [128]                             */
  const float* W3; const float* b3;   /* [out_w][128], [out_w]                         */
  int32_t out_w;                   /* 1..128; must be 128 when LayerNorm is present    */
  const float* ln_g; const float* ln_b;   /* nullable pair: no LayerNorm (decoder)      */
  const float* res; int64_t ld_res;       /* nullable residual [M, out_w]               */
  float* out; int64_t ld_out;
  float* z1; float* z2; float* xhat; float* rstd;   /* nullable saves [M,128] x3, [M]   */
  const void* W2pk; const void* W3pk;     /* optional packed images of W2 / W3 (forward form); see hgn_pack_bf16x3 */
  /* optional save [M][8] words: the ReLU sign pattern of both hidden layers, so that the backward chain reads 32 bytes per
   * row instead of the two [M,128] fp32 activations: word 4*l + q of a row (l = 0: first hidden layer, 1: second) holds,
   * in bit 4*b + u, whether hidden unit 16*b + 4*q + u is active (> 0) */
  uint32_t* relu_bits;
  /* optional, split-bf16 kernel only (hgn_mlp_fwd6_eligible): segment sum of the OUTPUT rows in the same pass --
   *   seg_out[seg_ids[i]][0..128) += out[i]   for rows sorted by seg_ids (the receiver-sorted edge order: this is the `sum`
   * aggregation of graphnet.py:50-70 without a second pass over the edge latents).  seg_out must be zero-filled by the
   * caller (segments without rows stay 0).  Segments inside one 64-row tile are stored, the two at a tile's ends are added
   * atomically: bit-reproducible as long as no segment spans more than two tiles (always true for <= 65 rows per segment;
   * callers with larger segments use hgn_segment_reduce_fwd). */
  float* seg_out; int64_t ld_seg_out; const int32_t* seg_ids;
  /* optional (hgn_mlp_fwd_post_eligible: split-product launches without gathered addends and without seg_out): n_post (1..4) further packed 128 x 128
   * blocks applied to the OUTPUT rows in the same launch,
   *   post_out[i][128 b .. 128 b + 128) = out[i] . post_pk[b]^T        (b < n_post; leading dimension ld_post),
   * bit-identical to hgn_linear_fwd6 on `out` -- the node-level pre-projection of the NEXT edge block (graphnet.py:22-32: the
   * sender / receiver column blocks of its first Linear), formed while the rows are still in registers --, and, when
   * post_zero is given, post_zero[i][0..128) = 0 (that block's aggregate buffer, see seg_out). */
  const void* post_pk[4]; int32_t n_post; float* post_out; int64_t ld_post; float* post_zero; int64_t ld_post_zero;
  /* per-call options (the library keeps no mutable state a call depends on except the DEFAULT below):
   *   products: 0 = the process default (hgn_set_matmul_products), or 6 / 1 / 2 for THIS call -- two models with different precisions,
   *             or two threads, never see each other's setting; packed images must have been built for the same mode;
   *   flags:    HGN_F_FP32_MFMA = the plain fp32-MFMA kernels even where the split-bf16 ones are eligible (what the environment
   *             variable HGN_FP32_MFMA selects for a whole process: the Python host reads it once and passes the flag);
   *             HGN_F_GENERAL_FWD = never the specialised training-edge-block kernel (A/B tests: bit-identical results). */
  int32_t products; int32_t flags;
} hgn_mlp_fwd_t;

int hgn_mlp_fwd(const hgn_mlp_fwd_t* args /*host*/, void* stream);

/* Split-bf16 matrix products (csrc/mlp6.hip).  When every weight block of a call comes with its packed image (Wpk / W2pk /
 * W3pk ...), all widths are multiples of 128 and the output is 128 wide, the fused MLP kernels evaluate each fp32 128x128
 * product as SIX bf16 MFMAs on a 3-way bf16 split of both operands (x = x1+x2+x3 holds all 24 significand bits; products
 * with i+j <= 4; fp32 accumulation): at least fp32-accurate (2.6e-7 vs 4.5e-7 max relative error of a plain fp32 product
 * on these shapes) at 2.7x the fp32 MFMA rate.  hgn_pack_bf16x3 packs blocks of at most 128 x 128 (element (o,i) at
 * W[o*ldw+i], zero padded) into HGN_PACK_BLOCK_BYTES each.  Packs must be refreshed whenever the weights change.
 * The flag HGN_F_FP32_MFMA of a call forces the plain fp32 kernels for that call. */
#define HGN_PACK_BLOCK_BYTES 98304
#define HGN_MAX_PACK 32
typedef struct {
  const float* W; int64_t ldw;     /* block origin (&W[o0][i0]) and leading dimension                       */
  int32_t n_out; int32_t n_in;     /* valid extents of the block (<= 128 each; the rest is zero padded)     */
  int32_t transposed;              /* bit 0: 0 contraction over i (forward products), 1 over o (data gradients);
                                    * | 2: ONE fp16 term in the leading third (forward form of products mode 2);
                                    * | 4: TWO fp16 terms of W * 2^sw in the first two thirds + the exponent sw (int32) at byte
                                    *      32768 of the image (products mode 3; mode 2's backward)                          */
  void* out;                       /* HGN_PACK_BLOCK_BYTES, 16-byte aligned                                 */
} hgn_pack_t;
int hgn_pack_bf16x3(const hgn_pack_t* blocks /*host*/, int n_blocks, void* stream);   /* one launch for up to HGN_MAX_PACK blocks */
/* A training step re-packs EVERY weight block of the model after the optimizer moved them (src/migration/meshgraphnet.py:14-37: 2 MLPs
 * per message-passing block + encoders, forward and transposed form: ~320 blocks for 15 blocks): ONE launch over a descriptor table
 * that lives in device memory -- `blocks_dev` holds the same n_blocks descriptors as `blocks` (which is only validated here), written
 * once by the caller; capturable (no host data is read at replay).  n_blocks <= 65535. */
int hgn_pack_bf16x3_table(const hgn_pack_t* blocks /*host copy*/, const hgn_pack_t* blocks_dev /*device*/, int n_blocks, void* stream);
/* DEFAULT product mode of the split-product kernels: what a call with `products` = 0 gets.
 *   3 (default): each fp32 operand as TWO fp16 terms (hi = rne(x), lo = rne(x - hi): the fp32 value to 2^-24), three
 *      v_mfma_f32_16x16x32_f16 per product (hi*hi + hi*lo + lo*hi, fp32 accumulation).  fp16 has 5 exponent bits, so every operand is
 *      scaled by a power of two first (exact): weights per packed block at pack time (hgn_pack_t.transposed | 4: the exponent is
 *      stored in the image), rows of activations / gradients per row at split time, the operands of a weight gradient per 32-row
 *      block; accumulators are scaled back.  fp32 accurate -- the mode every parity claim of this library refers to.
 *   6: three bf16 terms per operand, six bf16 MFMAs per product; the same accuracy without scales at twice the matrix work.
 *   1: ONE bf16 MFMA per product (both operands rounded to bf16; relative error ~4e-3 per product).
 *   2: the FORWARD products as ONE fp16 MFMA (~5e-4 per product; forward-form packs built with hgn_pack_t.transposed = 2), the
 *      backward / weight-gradient products as in mode 3 (transposed packs with transposed = 1 | 4) -- the "fp16 MFMA edge-MLP" of
 *      BASELINE.json configs[4].  1 and 2 are opt-in, never a default, outside the 1e-5 parity tolerance. */
int hgn_set_matmul_products(int n /* 3, 6, 1 or 2 */);
int hgn_get_matmul_products(void);
int hgn_mlp_fwd6_eligible(const hgn_mlp_fwd_t* args /*host*/);   /* 1 if hgn_mlp_fwd will take the split-bf16 kernel */
int hgn_mlp_fwd_post_eligible(const hgn_mlp_fwd_t* args /*host*/);   /* 1 if hgn_mlp_fwd accepts these args WITH their post_* fields */
int hgn_linear_fwd6(const float* x, int64_t ldx, int64_t M, const void* const* packed_blocks /*host array*/, int n_blocks,
                    float* out, int64_t ld_out, int products /*0 = process default, or 3 / 6 / 1 / 2*/, void* stream);
/* The same launch, which also sets zero_rows[i][0..128) = 0 for i < M (nullable; leading dimension ld_zero >= 128, a multiple of
 * 4; 16-byte aligned): an edge block needs its node-level pre-projection AND a zero-filled aggregate buffer over the same node
 * rows (hgn_mlp_fwd_t.seg_out) -- one pass over the rows instead of a launch of its own for the fill. */
int hgn_linear_fwd6z(const float* x, int64_t ldx, int64_t M, const void* const* packed_blocks /*host array*/, int n_blocks,
                     float* out, int64_t ld_out, float* zero_rows, int64_t ld_zero, int products, void* stream);

/* Backward data-gradient chain of the same MLP (LayerNorm bwd -> W3^T -> relu' -> W2^T -> relu' -> W1^T).
 * Writes dz3, dz2, dz1 ([M,128], consumed by hgn_mlp_wgrad and, for the pre-projected addends, by the
 * sender/receiver segment sums) and, per requested source, dx = dz1 * W1[:, cols] (+ d_out when `residual`). */
typedef struct {
  const float* W;      /* &W1[0][col0]                                               */
  int32_t K;
  float* dx;           /* [M, ld]                                                    */
  int64_t ld;
  int32_t residual;    /* 1: dx += d_out (the `res +` skip connection)                */
  const void* Wpk_t;   /* optional: the same columns packed by hgn_pack_bf16x3 (transposed form), K/128 consecutive blocks */
} hgn_dx_t;

typedef struct {
  int64_t M;
  const float* d_out; int64_t ld_dout; int32_t out_w;
  const float* ln_g; const float* xhat; const float* rstd;    /* nullable triple        */
  const float* z2; const float* z1;
  const float* W3; const float* W2;
  int64_t ldw1;
  float* dz3; float* dz2; float* dz1;
  int32_t n_dx;
  hgn_dx_t dx[HGN_MAX_SRC];
  /* Optional, edge blocks: the aggregation backward (graphnet.py:50-70 / torch_scatter backward) folded into the load
   * of d_out:  d_out_eff[i] = (d_out ? d_out[i] : 0) + sum_slot d(op_slot)(agg_dout[seg[i]][slot*128 ...]).  Rows are in
   * CSR (receiver-sorted) order, so the arg index of max/min is the row index itself.  d_out may be null then. */
  const float* agg_dout; int64_t ld_agg; int32_t n_agg_ops; int32_t agg_ops[4];
  const int32_t* agg_seg; const int32_t* agg_rowptr; const int32_t* agg_argmax; const int32_t* agg_argmin;
  /* Optional: LayerNorm-affine gradients  dgamma[j] = sum_i d_out_eff[i][j]*xhat[i][j],  dbeta[j] = sum_i d_out_eff[i][j]
   * produced by the same pass (deterministic: per-wave shuffles -> per-workgroup slab -> two-level fixed-order sum, the second
   * level by whichever block of the ONE reduction launch finishes last: same order of additions whichever it is).
   * ln_ws: hgn_mlp_bwd_ln_workspace_bytes(M) bytes, contents arbitrary; one workspace per stream that runs backward passes. */
  float* d_gamma; float* d_beta; float* ln_ws; int32_t ln_accumulate;
  const void* W3pk_t; const void* W2pk_t;   /* optional packed images of W3 / W2 (transposed form): split-bf16 kernels */
  const uint32_t* relu_bits;                /* optional: the forward's relu_bits; z1 / z2 may be null then            */
  /* optional, split-bf16 kernel only: seg_dz1[seg_ids[i]][0..128) += dz1[i]  (same contract as hgn_mlp_fwd_t.seg_out): the
   * receiver half of the pre-projection gradient of the split edge layer, without re-reading dz1 */
  float* seg_dz1; int64_t ld_seg_dz1; const int32_t* seg_ids;
  int32_t products; int32_t flags;          /* per-call options, as in hgn_mlp_fwd_t (backward products: those of the forward; mode 2 differentiates with mode 3's) */
} hgn_mlp_bwd_t;

int hgn_mlp_bwd_ln_workspace_bytes(int64_t M, size_t* bytes /*host*/);
/* Deferred LayerNorm-affine sums.  A backward call made with HGN_F_DEFER_LN leaves its per-workgroup slabs (and their count) in its
 * ln_ws -- which must then be a workspace of its OWN, untouched until the sums are taken -- and skips the reduction launch; the
 * gradients of up to HGN_MAX_LN_TASK such calls are summed by ONE launch (same two-level fixed-order sums as the reduction each call
 * would have run: identical results).  A training step of the 15-layer model has 31 of these reductions, 7-9 us each: a tenth of a
 * one-graph step (src/algorithms/MeshSimulator.py:141-152 runs such steps).  Nobody reads a LayerNorm gradient before the optimiser,
 * so the caller may take the sums when the backward pass is over.  Two tasks of ONE batch must not share d_gamma / d_beta (the
 * sums are added to their targets without atomics; launches on one stream serialise). */
#define HGN_MAX_LN_TASK 48
typedef struct {
  float* ln_ws;                 /* the workspace handed to the deferred call (hgn_mlp_bwd_ln_workspace_bytes(M) bytes)        */
  int64_t M;                    /* rows of that call                                                                          */
  float* d_gamma; float* d_beta;/* [128] each                                                                                 */
  int32_t accumulate;           /* 0: overwrite, 1: add (flat gradient buffer)                                                */
  int32_t reserved;
} hgn_ln_task_t;
int hgn_ln_reduce_batch(const hgn_ln_task_t* tasks /*host*/, int n_tasks, void* stream);
int hgn_mlp_bwd(const hgn_mlp_bwd_t* args /*host*/, void* stream);
int hgn_mlp_bwd6_eligible(const hgn_mlp_bwd_t* args /*host*/);   /* 1 if hgn_mlp_bwd will take the split-bf16 kernel */
int hgn_linear_bwd6(const float* g, int64_t ldg, int64_t M, const void* const* packed_blocks_t /*host array, transposed form*/,
                    int n_blocks, float* dx, int64_t ld_dx, int products, void* stream);
/* The same with an ACCUMULATE target: dx (+)= g W.  A node latent h feeds the edge block (graphnet.py:25-26, through this
 * product) and the node update (graphnet.py:43-47): autograd sums the two gradients with a pass of its own over [N, 128]; with
 * accumulate != 0 this launch starts its accumulators from the gradient the node update already wrote instead. */
int hgn_linear_bwd6a(const float* g, int64_t ldg, int64_t M, const void* const* packed_blocks_t /*host array, transposed form*/,
                     int n_blocks, float* dx, int64_t ld_dx, int accumulate, int products, void* stream);

/* Edge-block backward with the weight gradients of the two inner layers in the SAME pass (csrc/fused_bwd.hip): autograd of
 * GraphNet._update_edge_features (graphnet.py:22-32) for one edge set -- everything hgn_mlp_bwd computes for an edge block
 * (dz1, de = dx[0], LayerNorm-affine gradients, the folded aggregation backward) PLUS
 *   dW3 (+)= dz3^T z2, db3 (+)= colsum dz3;  dW2 (+)= dz2^T z1, db2 (+)= colsum dz2
 * without dz3 / dz2 ever being written to memory (`a->dz3`, `a->dz2` are ignored).  `a->dz1` must be given AND hold
 * ceil(M / 64) * 64 rows (the kernel stores whole 64-row tiles; rows >= M are padding): it is read by the
 * sender / receiver sums of the split first layer and by the caller's hgn_mlp_wgrad task for dW1 (= dz1^T x).  One persistent
 * 8-wave workgroup per CU; per-workgroup partials are added in fixed order (deterministic).  Eligible when
 * hgn_mlp_bwd6_eligible(a) holds, a->n_dx == 1 with a 128-wide residual source, and a->seg_dz1 is null.  workspace:
 * hgn_edge_bwd_fused_workspace_bytes(M) bytes, 16-byte aligned; a->ln_ws: hgn_mlp_bwd_ln_workspace_bytes(M) bytes. */
typedef struct {
  const float* z2; const float* z1;   /* [M,128] saved post-ReLU activations of the forward (leading dimension 128)   */
  float* dW3; float* db3;             /* [128][128], [128]                                                              */
  float* dW2; float* db2;
  int32_t accumulate;                 /* 0: results overwrite, 1: results are added (flat gradient buffer)             */
} hgn_wfuse_t;
int hgn_edge_bwd_fused_workspace_bytes(int64_t M, size_t* bytes /*host*/);
int hgn_edge_bwd_fused_eligible(const hgn_mlp_bwd_t* args /*host*/);
int hgn_edge_bwd_fused(const hgn_mlp_bwd_t* args /*host*/, const hgn_wfuse_t* w /*host*/, void* workspace, size_t workspace_bytes,
                       void* stream);
/* the same without the final sums of the dW3 / dW2 slabs (see hgn_mlp_wgrad_partial); red: two entries, written */
int hgn_edge_bwd_fused_partial(const hgn_mlp_bwd_t* a /*host*/, const hgn_wfuse_t* w /*host*/, void* workspace, size_t workspace_bytes,
                                hgn_wred_task_t* red /*host, 2 entries*/, void* stream);

/* Weight / bias / LayerNorm-affine gradients: a list of tasks reduced over all rows in one launch.
 *   type 0:  dW[j][k] = sum_i G[i][j] * A[idxA ? idxA[i] : i][k]   (j < 128, k < K <= 128),  db[j] = sum_i G[i][j]
 *   type 1:  dgamma[j] = sum_i G[i][j] * A[i][j],  dbeta[j] = sum_i G[i][j]        (A = xhat, G = d_out)
 * Results overwrite dW/db (dgamma/dbeta), or are ADDED to them when `accumulate` is set (gradient buffers shared by
 * several uses of one module, or a flat gradient buffer zeroed once per step).  Deterministic (per-chunk slabs +
 * fixed-order reduction). */
typedef struct {
  int32_t type;
  const float* A; int64_t lda; int32_t K; const int32_t* idxA;
  const float* G; int64_t ldg;
  int32_t n_out;        /* rows of dW to write (<=128)                                  */
  float* dW; int64_t ldw;   /* type 0: &dW1[0][col0], leading dim; type 1: dgamma        */
  float* db;                /* nullable; type 1: dbeta                                   */
  int32_t accumulate;       /* 0: dW = result, 1: dW += result                           */
  int32_t products; int32_t flags;   /* per-call options as in hgn_mlp_fwd_t; all tasks of one launch must agree (task 0 decides)  */
} hgn_wtask_t;

int hgn_wgrad_workspace_bytes(int64_t M, int n_tasks, size_t* bytes /*host*/);
int hgn_mlp_wgrad(const hgn_wtask_t* tasks /*host*/, int n_tasks, int64_t M, void* workspace,
                  size_t workspace_bytes, void* stream);

/* Deferred chunk-slab sums.  hgn_mlp_wgrad and hgn_edge_bwd_fused end with a launch that adds the per-workgroup partial slabs of
 * their weight gradients in fixed order (6-8 us, 39 of them per training step of the 15-layer model: a tenth of a one-graph
 * step).  Weight gradients that ACCUMULATE into a gradient buffer are read by nobody before the optimiser: the `_partial` forms run
 * everything but that launch, leave the slabs in `workspace` -- which must then be the call's OWN until the sums are taken -- and
 * describe the pending sums in `red` (host memory, one entry per task; hgn_edge_bwd_fused_partial: two, dW3 / db3 and dW2 / db2).
 * hgn_slab_reduce_batch takes up to HGN_MAX_WRED such sums in one launch: the same additions in the same order as the launch each
 * call would have made (identical results).  Two entries of ONE batch must not share a dW / db target (the sum is added to its
 * target without atomics); launches on one stream serialise. */
int hgn_mlp_wgrad_partial(const hgn_wtask_t* tasks /*host*/, int n_tasks, int64_t M, void* workspace, size_t workspace_bytes,
                          hgn_wred_task_t* red /*host, n_tasks entries, written*/, void* stream);
int hgn_slab_reduce_batch(const hgn_wred_task_t* red /*host*/, int n, void* stream);

/* Single Linear without bias over 128-wide blocks:  out[:, 128*b : 128*b+128] = x * Wb^T  (node pre-projection
 * of the split edge layer:  [h W_s^T | h W_r^T], W_s = W1[:, 0:128], W_r = W1[:, 128:256], graphnet.py:28-30)
 * and its data gradient  dx = sum_b g[:, 128*b : ...] * Wb. */
int hgn_linear_fwd(const float* x, int64_t ldx, int64_t M, const float* const* Wblocks /*host array*/,
                   int n_blocks, int64_t ldw, float* out, int64_t ld_out, void* stream);
int hgn_linear_bwd(const float* g, int64_t ldg, int64_t M, const float* const* Wblocks /*host array*/,
                   int n_blocks, int64_t ldw, float* dx, int64_t ld_dx, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Optimiser step on a flat parameter buffer (torch.optim.Adam semantics, MeshSimulator.py:110), and helpers.
 * ---------------------------------------------------------------------------------------------------- */
int hgn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, float grad_scale, void* stream);
/* Same update with the step counter in DEVICE memory: *step_dev is incremented, then used for the bias corrections, so the
 * call can be captured once into a HIP graph and replayed (a host-side step would be frozen at capture time). */
int hgn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, int32_t* step_dev, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline line).
 * kernel ids: 0 mlp_fwd(edge) 1 mlp_fwd(other) 2 mlp_bwd(edge) 3 mlp_bwd(other) 4 wgrad 5 seg_fwd 6 seg_bwd
 *             7 linear_fwd 8 linear_bwd 9 adam 10 csr 11 wgrad on node rows (hgn_prof_tag(1) before the call)
 *             12 seg_fwd launched as the forward aggregation (hgn_prof_tag(2) before the call)
 *             13 feature kernels  14 hgn_edge_bwd_fused
 * ---------------------------------------------------------------------------------------------------- */
#define HGN_NUM_KERNEL_IDS 15
int hgn_prof_enable(int on);
int hgn_prof_tag(int tag);   /* thread-local: 1 = following hgn_mlp_wgrad launches are node-level, 2 = following
                                hgn_segment_reduce_fwd launches are the forward aggregation, 0 = reset */
int hgn_prof_reset(void);
int hgn_prof_collect(double* total_ms /*host [HGN_NUM_KERNEL_IDS]*/, int64_t* count /*host*/,
                     double* units /*host: rows processed*/);

#ifdef __cplusplus
}
#endif
#endif
