// Pieces shared by the fp32 (mlp.hip) and split-bf16 (mlp6.hip) backward kernels.
#pragma once
#include "hgn_device.h"
#include "../../include/hgn_mp.h"

namespace hgn {

__device__ __forceinline__ void relu_mask(Act& g, const float* __restrict__ zrow, int kq) {
  HGN_FOR_B(fb) {
    const f32x4 z = *reinterpret_cast<const f32x4*>(zrow + 16 * fb + 4 * kq);
#pragma unroll
    for (int u = 0; u < 4; ++u) g.v[fb][u] = z[u] > 0.f ? g.v[fb][u] : 0.f;
  }
}

// ReLU sign pattern of the lane's 32 values as one word (bit 4*fb + u <-> unit 16*fb + 4*kq + u; hgn_mlp_fwd_t.relu_bits)
__device__ __forceinline__ unsigned relu_bits_of(const Act& a) {
  unsigned m = 0;
  HGN_FOR_B(fb)
#pragma unroll
    for (int u = 0; u < 4; ++u) m |= (a.v[fb][u] > 0.f ? 1u : 0u) << (4 * fb + u);
  return m;
}

// g *= relu'(z) from the sign word: bit -> all-ones / zero by a sign-extending field extract, then one AND (2 instructions per value)
__device__ __forceinline__ void relu_mask_bits(Act& g, unsigned m) {
  HGN_FOR_B(fb)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      g.v[fb][u] = __uint_as_float(__float_as_uint(g.v[fb][u]) & (unsigned)__builtin_amdgcn_sbfe((int)m, 4 * fb + u, 1));
}

// a = relu(a) in place and its sign word (relu_bits_of), three instructions per value: the maximum of the BIT PATTERN and 0 as signed
// integers is relu for every finite float (negative floats are negative integers, -0.0 included; fmaxf costs two instructions: the
// compiler canonicalises its operand first), "z > 0" is "bit pattern of z != 0", i.e. the sign of 0 - bits, shifted into the word
// from the top bit index down.
__device__ __forceinline__ unsigned relu_with_bits(Act& a) {
  unsigned m = 0;
#pragma unroll
  for (int fb = NB - 1; fb >= 0; --fb)
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const int z = max(__float_as_int(a.v[fb][u]), 0);
      a.v[fb][u] = __int_as_float(z);
      m = __builtin_amdgcn_alignbit(m, 0u - (unsigned)z, 31);
    }
  return m;
}
__device__ __forceinline__ void relu_int(Act& a) {
  HGN_FOR_B(fb)
#pragma unroll
    for (int u = 0; u < 4; ++u) a.v[fb][u] = __int_as_float(max(__float_as_int(a.v[fb][u]), 0));
}

// Segment sums of the workgroup's 64 x 128 tile `v` (rows sorted by segment id) into out[seg][0..128): the tile goes through
// LDS (row stride 132 floats: conflict-free 16-byte writes), then thread (half, c) walks column c down 32 rows.  Segments
// that lie inside the tile are stored; the first / last one is added atomically when it continues in the neighbouring tile
// (two contributions onto a zero-filled row: order independent).  The half-1 threads hand the part of a segment that began
// in half 0 over through LDS, so every segment is written once per tile.  `ldsf`: SEG_LDS_FLOATS floats, free for use; the
// caller must put a workgroup barrier between this call and the next write to `ldsf`.
constexpr int SEG_LDS_FLOATS = 64 * 132 + 64 + 128;
// `pre_ids` (optional): the segment ids this call needs, already in LDS (SegPre, filled near kernel entry): [0, 64) the id of
// row tile_row0 + i (-1 past the end), [64] / [65] the ids of the rows before / after the tile (-1: none).  Vector memory retires
// in order, so a load issued HERE is also a wait for every store the caller has in flight (its output rows); with `pre_ids`
// the function waits for no global memory at all and its barriers are LDS-only.  (The ids are parked in LDS rather than in
// registers: a register that lives from kernel entry to here gets spilled, and the reload is a vector-memory load again.)
constexpr int SEG_PRE_INTS = TILE_ROWS + 2;
struct SegPre {
  int id, prev, next;
  __device__ __forceinline__ void load(const int32_t* __restrict__ seg_ids, long tile_row0, long M, int ltid = -1) {
    if (ltid < 0) ltid = threadIdx.x;
    const long r = tile_row0 + ltid;
    id = (ltid < TILE_ROWS && r < M) ? seg_ids[r] : -1;
    prev = (tile_row0 > 0 && tile_row0 <= M) ? seg_ids[tile_row0 - 1] : -1;
    const long e = min(tile_row0 + TILE_ROWS, M);
    next = e < M ? seg_ids[e] : -1;
  }
  __device__ __forceinline__ void stash(int* __restrict__ lds_ids, int ltid = -1) const {      // visible after the caller's next workgroup barrier
    if (ltid < 0) ltid = threadIdx.x;
    if (ltid < TILE_ROWS) lds_ids[ltid] = id;
    if (ltid == 0) { lds_ids[TILE_ROWS] = prev; lds_ids[TILE_ROWS + 1] = next; }
  }
};
// The column walk of tile_segment_sum: the 64 x 128 tile is in `ldsf` (row stride 132 floats), the ids in `ids`, and a workgroup
// barrier lies between those writes and this call.  Threads 0..255 walk (thread = (half, column)); EVERY thread of the workgroup
// must call (one barrier inside; `lds_only`: the barrier does not wait for the caller's global stores).
// `ltid`: the thread's index among the (at least 256) threads that share THIS tile -- threadIdx.x, or the index inside a 256-thread
// group when several groups of one workgroup walk a tile each (every group on its own `ldsf` / `ids`; the barrier is the workgroup's).
__device__ __forceinline__ void tile_segment_walk(float* __restrict__ ldsf, const int* __restrict__ ids, int prev_id, int next_id,
                                                  float* __restrict__ out, long ld, long tile_row0, long M, bool lds_only, int ltid) {
  // Everything that steers the walk is the same for the 64 threads of a wave (they walk the same rows of different columns):
  // the ids go through readfirstlane, so the segment tests compile to SCALAR branches instead of exec-mask sequences (the
  // walk was a hundred `s_and_saveexec` regions per tile before).
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  float* hp = ldsf + 64 * 132 + 64;
  const int rows = (int)max(0L, min((long)TILE_ROWS, M - tile_row0));
  const int half = uni((ltid >> 7) & 1), c = ltid & 127;
  const int r0 = 32 * half, r1 = min(rows, r0 + 32);
  const bool active = uni(ltid < 256 ? 1 : 0) && r0 < rows;
  const bool tile_cont_prev = rows > 0 && tile_row0 > 0 && uni(prev_id) == uni(ids[0]);
  const bool tile_cont_next = rows > 0 && tile_row0 + rows < M && uni(next_id) == uni(ids[rows > 0 ? rows - 1 : 0]);
  int cur = -1;
  float s = 0.f;
  bool first = true;                                        // still inside the segment my range began with
  bool cont_prev = false;                                   // ... and that segment began before my range
  if (active) {
    cur = uni(ids[r0]);
    cont_prev = half ? uni(ids[r0 - 1]) == cur : tile_cont_prev;
    for (int rb = r0; rb < r1; rb += 8) {
      float x[8]; int id[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = min(rb + j, r1 - 1);
        x[j] = ldsf[r * 132 + c]; id[j] = uni(ids[r]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (rb + j < r1) {
          if (id[j] != cur) {
            float* dst = out + (long)cur * ld + c;
            if (first && cont_prev) { if (half) hp[c] = s; else unsafeAtomicAdd(dst, s); }
            else *dst = s;
            cur = id[j]; s = 0.f; first = false;
          }
          s += x[j];
        }
      }
    }
    if (half) {                                             // my open last segment is the tile's last one
      if (first && cont_prev) hp[c] = s;                    // all my rows continue half 0's last segment
      else {
        float* dst = out + (long)cur * ld + c;
        if (tile_cont_next) unsafeAtomicAdd(dst, s); else *dst = s;
      }
    }
  }
  if (lds_only) wg_barrier_lds(); else __syncthreads();
  if (!half && active) {
    const bool joined = rows > 32 && uni(ids[32]) == cur;   // half 1 began inside my last segment
    const float total = s + (joined ? hp[c] : 0.f);
    const bool to_end = joined ? uni(ids[rows - 1]) == cur : rows <= 32;     // the segment runs to the end of the tile
    float* dst = out + (long)cur * ld + c;
    if ((first && cont_prev) || (to_end && tile_cont_next)) unsafeAtomicAdd(dst, total); else *dst = total;
  }
}

__device__ __forceinline__ void tile_segment_sum(const Act& v, float* __restrict__ ldsf, const int32_t* __restrict__ seg_ids,
                                                 float* __restrict__ out, long ld, long tile_row0, long M,
                                                 const int* __restrict__ pre_ids = nullptr, int ltid = -1) {
  if (ltid < 0) ltid = threadIdx.x;
  const int lane = ltid & 63, wave = ltid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int* ids = pre_ids ? pre_ids : reinterpret_cast<const int*>(ldsf + 64 * 132);
  if (pre_ids) wg_barrier_lds(); else __syncthreads();      // every wave is done with the weight stage
  float* wr = ldsf + (wave * WAVE_ROWS + n) * 132 + 4 * kq;
  HGN_FOR_B(fb) *reinterpret_cast<f32x4*>(wr + 16 * fb) = v.v[fb];
  const int rows = (int)min((long)TILE_ROWS, M - tile_row0);
  if (!pre_ids && ltid < 64)
    reinterpret_cast<int*>(ldsf + 64 * 132)[ltid] = ltid < rows ? seg_ids[tile_row0 + ltid] : -1;
  if (pre_ids) wg_barrier_lds(); else __syncthreads();
  const int prev_id = tile_row0 > 0 ? (pre_ids ? pre_ids[TILE_ROWS] : seg_ids[tile_row0 - 1]) : -1;
  const int next_id = tile_row0 + rows < M ? (pre_ids ? pre_ids[TILE_ROWS + 1] : seg_ids[tile_row0 + rows]) : -1;
  tile_segment_walk(ldsf, ids, prev_id, next_id, out, ld, tile_row0, M, pre_ids != nullptr, ltid);
}

// Aggregation backward of the lane's row added to g: sum_slot d(op_slot)(agg_dout[seg[row]][slot * 128 ...]) (graphnet.py:50-70).
// `pre_seg`: a.agg_seg[rc] loaded by the caller ahead of time (kernel entry), or -1: the dependent row loads below then cost one
// memory round trip, not two.  The segment length is only read when a `mean` slot needs it.
__device__ __forceinline__ void add_agg_dout(Act& g, const hgn_mlp_bwd_t& a, long rc, int kq, int pre_seg = -1) {
  if (a.agg_dout) {
    const long r = pre_seg >= 0 ? (long)pre_seg : (long)a.agg_seg[rc];
    bool has_mean = false;
    for (int slot = 0; slot < a.n_agg_ops; ++slot) has_mean = has_mean || a.agg_ops[slot] == HGN_OP_MEAN;
    float inv = 0.f;
    if (has_mean) {
      const int cnt = a.agg_rowptr[r + 1] - a.agg_rowptr[r];
      inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    }
    for (int slot = 0; slot < a.n_agg_ops; ++slot) {
      const float* ar = a.agg_dout + r * a.ld_agg + slot * LAT;
      const int op = a.agg_ops[slot];
      HGN_FOR_B(fb) {
        const int col = 16 * fb + 4 * kq;
        const f32x4 d = *reinterpret_cast<const f32x4*>(ar + col);
        if (op == HGN_OP_SUM) g.v[fb] += d;
        else if (op == HGN_OP_MEAN) g.v[fb] += d * inv;
        else {
          const int* ap = (op == HGN_OP_MAX ? a.agg_argmax : a.agg_argmin) + r * LAT + col;
#pragma unroll
          for (int u = 0; u < 4; ++u) g.v[fb][u] += ap[u] == (int)rc ? d[u] : 0.f;
        }
      }
    }
  }
}

// d_out_eff of the lane's row: d_out (optional) + the aggregation backward scattered back through the CSR row of the edge.
template <bool ACC>
__device__ __forceinline__ void load_dout(Act& g, const hgn_mlp_bwd_t& a, long rc, int kq, int pre_seg = -1) {
  // ACC: add d_out_eff to g (residual path; 128-wide, aligned);  otherwise g = d_out_eff
  if (ACC) {
    if (a.d_out) t_add(g, a.d_out + rc * a.ld_dout, kq);
  } else if (a.d_out) {
    const bool vec_out = (a.out_w == LAT) && ((a.ld_dout & 3) == 0);
    if (vec_out) t_load(g, a.d_out + rc * a.ld_dout, kq); else t_load_masked(g, a.d_out + rc * a.ld_dout, kq, a.out_w);
  } else {
    t_zero(g);
  }
  add_agg_dout(g, a, rc, kq, pre_seg);
}

}  // namespace hgn
