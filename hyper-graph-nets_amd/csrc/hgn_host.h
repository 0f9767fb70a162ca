// Host-side helpers shared by the translation units of libhgn_mp.so (error reporting, optional profiler).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hgn_mp.h"

namespace hgn {

int hgn_fail(int code, const char* msg);          // records msg (thread-local) and returns code
int hgn_check_launch(const char* what);
// `per_call`: the call's own `products` field (include/hgn_mp.h), 0 = the process default of hgn_set_matmul_products
int bwd_products(int per_call);                   // products of the backward / weight-gradient kernels: 6, 3, 1 -- and 3 for mode 2 (host.cpp)
int matmul_products(int per_call);                // 6 / 3 (fp32-accurate split products), 1 (single bf16 product) or 2 (single fp16 product, forward only)
bool valid_products(int p);                       // 0, 1, 2, 3 or 6
extern thread_local int g_prof_tag;           // hipGetLastError() -> HGN_OK / HGN_E_LAUNCH

// Fixed-order reductions shared between translation units (wgrad.hip / mlp.hip own the kernels).
//   slab reduce: dW[j][k] (+)= sum over chunks c of slab[c * chunk_stride + j * 128 + k], db[j] (+)= ... slab[c * chunk_stride + 128 * 128 + j]
struct SlabReduceTask { int type; int K; int n_out; int acc; int n_chunks; float* dW; long ldw; float* db; const float* slab; long chunk_stride; };
int launch_slab_reduce(const SlabReduceTask* tasks, int n_tasks, hipStream_t stream);                 // n_tasks <= HGN_MAX_WTASK
//   LayerNorm-affine gradients from n_slabs per-workgroup slabs of 256 floats; `part`: LN_PARTS * 256 floats of scratch
// LayerNorm partial slabs a backward call of M rows may write (one per workgroup: 64-row tiles, or 16-row workgroups of the column-split
// form for small launches); the reduction's own partial slabs lie behind them
inline long ln_slab_capacity(long M) { const long t64 = (M + 63) / 64, t16 = M <= 65536 ? (M + 15) / 16 : 0; return t64 > t16 ? t64 : t16; }
int launch_ln_reduce(float* ws, long n_slabs, float* part, float* d_gamma, float* d_beta, int accumulate, hipStream_t stream);

// Records a HIP event pair around the launches issued in its scope when profiling is enabled.
struct ProfScope {
  int kid; hipStream_t stream; bool on; int slot;
  ProfScope(int kernel_id, double units, hipStream_t s);
  ~ProfScope();
};

}  // namespace hgn
