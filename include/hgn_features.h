/* hgn_features.h -- C ABI of the frame -> graph-feature step that feeds the message-passing path
 * ("next" rows f2 feature construction and f3 remote-graph assembly).
 * Same library (libhgn_mp.so), same conventions as hgn_mp.h: int status (0 ok, <0 HGN_E_*), hgn_last_error(),
 * caller-owned device buffers incl. workspace, everything enqueued on the hipStream_t passed in, no allocation.
 * Byte / index work: outputs are bit-exact for the integer parts and fp32 for the features.
 *
 * Replaces, on the reference side (file:line under /root/reference/src):
 *   util.py:50-89                    triangles_to_edges                      -> hgn_cells_to_edges
 *   model/flag.py:67-93, plate.py:165-183, cylinder.py:81-87,
 *   rmp/abstract_connector.py:84-98  relative position features of an edge set -> hgn_rel_edge_features
 *   model/flag.py:68-74, cylinder.py:67-76, plate.py:75-79,186-195
 *                                    velocity + one-hot node features          -> hgn_node_features
 *   model/plate.py:84-110            world edges (cdist + masks + nonzero)      -> hgn_radius_edges_count/_fill
 *   graph_balancer/ricci.py:128-301  balanced Forman curvature kernels (SDRF)   -> hgn_forman_curvature/_post_delta
 *   model/flag.py:178,188, cylinder.py:163,171  target / integrator arithmetic   -> hgn_lincomb3
 *   migration/normalizer.py:40-71    Normalizer.forward / inverse / _accumulate -> hgn_col_stats,
 *                                                                              hgn_normalizer_update, hgn_normalize
 */
#ifndef HGN_FEATURES_H
#define HGN_FEATURES_H
#include <stddef.h>
#include <stdint.h>
#include "hgn_mp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HGN_MAX_FEATURE_WIDTH 32 /* widest row hgn_col_stats / hgn_normalize accept (reference: <= 12) */

/* ---- util.py:50-89 ---------------------------------------------------------------------------------------
 * cells [n_cells, verts] int64 row-major, verts = 3 (triangles) or 4 (`deform=True`: edges 0-1, 1-2, 2-3, 3-0).
 * Unique undirected edges as (max, min) pairs in lexicographic order (what torch.unique(dim=0) returns), then
 * both directions: senders = [max.. ; min..], receivers = [min.. ; max..].
 * senders / receivers: device int64 with capacity 2*verts*n_cells each; *n_unique (HOST) receives the number of
 * undirected edges (valid prefix = 2 * n_unique).  Synchronises the stream (topology step, once per mesh).
 * HGN_E_RANGE if a vertex id is outside [0, 2^31). */
int hgn_cells_to_edges_workspace_bytes(int64_t n_cells, int verts, size_t* bytes);
int hgn_cells_to_edges(const int64_t* cells, int64_t n_cells, int verts, int64_t* senders, int64_t* receivers,
                       int64_t* n_unique /*host*/, void* workspace, size_t ws_bytes, void* stream);

/* ---- relative-position features of an edge set -----------------------------------------------------------
 * row e of feat = [ a[s]-a[r] (da) , |a[s]-a[r]| , b[s]-b[r] (db) , |b[s]-b[r]| ]   (s = senders[e], r = receivers[e])
 * a [n_rows, lda], b [n_rows, ldb] fp32; 1 <= da <= 3; db in 0..3 (db = 0: b unused, row = da+1 floats).
 * feat (nullable) [E, ldf] with ldf >= row width; len_a (nullable) [E] receives |a[s]-a[r]| (flag.py:101-113).
 * Precondition: ids in [0, n_rows) (checked by hgn_csr_build on the same id arrays); out-of-range ids are
 * skipped (row left untouched), never dereferenced. */
int hgn_rel_edge_features(const float* a, int64_t lda, int da, const float* b, int64_t ldb, int db, int64_t n_rows,
                          const int64_t* senders, const int64_t* receivers, int64_t E, float* feat, int64_t ldf,
                          float* len_a, void* stream);

/* ---- node features ------------------------------------------------------------------------------------------
 * out row n = [ v (d floats) | one_hot(cls, n_classes) ]  (vel_first = 1)  or  [ one_hot | v ]  (vel_first = 0)
 *   v   = cur[n] - prev[n]  (prev NULL: cur[n]; cur NULL: zeros);  if vel_mask_type >= 0, v is kept only on rows
 *         whose node_type equals vel_mask_type (plate.py:186-194: obstacle rows), else 0
 *   cls = map[node_type[n]] when map != NULL (device int32 table of map_len entries; flag: type!=0 -> 1;
 *         cylinder.py:71-74; plate.py:78), else node_type[n]; a class outside [0, n_classes) leaves the one-hot 0.
 * node_type int64 with element stride ldt (the reference passes node_type[:, 0]). */
int hgn_node_features(const float* cur, const float* prev, int64_t ld, int d, const int64_t* node_type, int64_t ldt,
                      const int32_t* map, int map_len, int n_classes, int vel_first, int vel_mask_type, int64_t N,
                      float* out, int64_t ldo, void* stream);

/* ---- normalizer.py ------------------------------------------------------------------------------------------
 * hgn_col_stats: batch[0:F] = column sums, batch[F:2F] = column sums of squares of x [rows, F] (contiguous),
 *   accumulated in fp64, fixed order (deterministic), rounded once to fp32 (normalizer.py:57-58).
 * hgn_normalizer_update: the running statistics += batch statistics iff *num_acc < max_acc (normalizer.py:42,
 *   59-62; the gate is evaluated on the device: no host sync); count = rows of the (global) batch.
 * hgn_normalize: out = (x - mean) / max(std, eps)            (inverse = 0, normalizer.py:45,64-71)
 *                out = x * max(std, eps) + mean              (inverse = 1, normalizer.py:47-49)
 *   mean = acc_sum / max(acc_count, 1), std = sqrt(|acc_sumsq / max(acc_count,1) - mean^2|). */
int hgn_col_stats_workspace_bytes(int64_t rows, int F, size_t* bytes);
int hgn_col_stats(const float* x, int64_t rows, int F, float* batch /*[2F]*/, void* workspace, size_t ws_bytes,
                  void* stream);
int hgn_normalizer_update(float* acc_sum, float* acc_sumsq, float* acc_count, float* num_acc, const float* batch,
                          const float* count /*device [1]*/, int F, float max_acc, void* stream);
int hgn_normalize(const float* x, int64_t rows, int F, const float* acc_sum, const float* acc_sumsq,
                  const float* acc_count, float eps, int inverse, float* out, void* stream);

/* ---- world edges by radius (plate.py:84-110) ---------------------------------------------------------------
 * Directed pairs (s, r), s != r, with |pos[s] - pos[r]| < radius, node_type[s] == sender_type, node_type[r] ==
 * receiver_type (a negative type = any), and (s, r) not adjacent in the mesh given as a CSR (nbr_rowptr [N+1],
 * nbr [nnz]: neighbours of node n = nbr[nbr_rowptr[n] .. nbr_rowptr[n+1]); nullable = no exclusion).  Output order =
 * torch.nonzero of the reference's N x N mask: ascending s, then ascending r.  Distances are evaluated as
 * sqrt(sum (pos[s]-pos[r])^2) in fp32 (the reference's cdist switches to the |a|^2+|b|^2-2ab form for N > 25, which
 * differs only for pairs within fp32 noise of the radius).
 * Two calls (the result size is data dependent): _count fills offsets[N+1] (device int32, exclusive prefix of the per-
 * sender counts, offsets[N] = total) and returns the total on the HOST (synchronises the stream); _fill writes
 * senders / receivers [total] int64.  One wavefront per sender row; no N x N matrix is formed. */
int hgn_radius_edges_workspace_bytes(int64_t N, size_t* bytes);
int hgn_radius_edges_count(const float* pos, int64_t ld, int d, const int64_t* node_type, int64_t ldt, int64_t N,
                           float radius, int sender_type, int receiver_type, const int32_t* nbr_rowptr,
                           const int32_t* nbr, int32_t* offsets /*[N+1]*/, int64_t* total /*host*/, void* workspace,
                           size_t ws_bytes, void* stream);
int hgn_radius_edges_fill(const float* pos, int64_t ld, int d, const int64_t* node_type, int64_t ldt, int64_t N,
                          float radius, int sender_type, int receiver_type, const int32_t* nbr_rowptr,
                          const int32_t* nbr, const int32_t* offsets, int64_t* senders, int64_t* receivers,
                          void* stream);

/* ---- balanced Forman curvature (graph_balancer/ricci.py:128-301, the two numba-CUDA kernels of SDRF) -------------
 * Dense fp32 adjacency A [N,N] (row-major, entries 0/1), A2 = A*A [N,N], d_in[i] = column sums, d_out[j] = row sums.
 * hgn_forman_curvature: C[i,j] for the nnz listed pairs (ei[e], ej[e]) with A != 0; C must be zero-filled by the caller
 *   (the reference's dense kernel writes 0 for non-edges).  One wavefront per edge sweeps k = 0..N-1 (count and maximum
 *   of the positive four-cycle terms by wave reduction), then evaluates the closed form in fp64 and rounds to fp32 once per
 *   store -- the typing of the reference's numba kernel (int literal x float32 -> float64), so equal inputs give equal bits
 *   and the argmin/argmax ties of SDRF resolve identically.
 * hgn_forman_post_delta: D[I,J] = curvature of edge (x,y) after inserting (i_nb[I], j_nb[J]); -1000 where i == j or the
 *   pair is already an edge (ricci.py:206-208); d_in_x = sum A[:,x], d_out_y = sum A[y,:]. */
int hgn_forman_curvature(const float* A, const float* A2, const float* d_in, const float* d_out, int64_t N,
                         const int32_t* ei, const int32_t* ej, int64_t nnz, float* C, void* stream);
int hgn_forman_post_delta(const float* A, const float* A2, float d_in_x, float d_out_y, int64_t N, int32_t x, int32_t y,
                          const int32_t* i_nb, int32_t dim_i, const int32_t* j_nb, int32_t dim_j, float* D, void* stream);

/* ---- targets and the one-step integrator ---------------------------------------------------------------------
 * out[i] = (ca*a[i] + cb*b[i]) + cc*c[i]   (c nullable), each product and sum rounded separately (no fma), so that
 *   flag.py:188  target - 2*cur + prev   (a=target, b=cur, c=prev; 1, -2, 1)
 *   flag.py:178  2*cur + acc - prev      (a=cur, b=acc, c=prev; 2, 1, -1)
 *   cylinder.py:163,171                  (a +/- b)
 * come out bit-identical to the reference's left-to-right fp32 evaluation.  n = number of elements (contiguous). */
int hgn_lincomb3(const float* a, float ca, const float* b, float cb, const float* c, float cc, int64_t n, float* out,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif
