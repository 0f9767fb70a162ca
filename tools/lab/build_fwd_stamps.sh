#!/bin/bash
# Laboratory build of the training edge forward (csrc/mlp6.hip: mlp6_fwd_edge_kernel, mlp6_device.h: gemm6q) with shader-clock stamps of one
# mid-launch workgroup's four waves at every barrier and stage boundary:   tools/lab/build_fwd_stamps.sh
#   ->  hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_fstamp.so   (HGN_LIB=<that file> python tools/lab/fwd_edge_stamps.py)
# The sources are patched into a temporary copy (csrc/ stays as it is); a stamp is a clock read + one dword store of lane 0: the stores
# are younger than every DMA a counted wait is about, so the waits only get stronger.  Order of magnitude only.
set -e
cd "$(dirname "$0")/../../hyper-graph-nets_amd/csrc"
make -s
mkdir -p ../hgn_amd/abl .fst_tmp
python3 - <<'PY'
d = open('mlp6_device.h').read()
m = open('mlp6.hip').read()
def rep(s, old, new, n=1):
    assert s.count(old) == n, (old[:70], s.count(old))
    return s.replace(old, new)
d = rep(d, 'template <int NS, int KEEP>\n__device__ __forceinline__ void gemm6q_landed',
        'namespace est { __device__ unsigned long long g_est[4 * 96]; }\n'
        '#define EST() do { if (blockIdx.x == 5000u && (threadIdx.x & 63u) == 0u) est::g_est[(threadIdx.x >> 6) * 96 + est_i] = clock64(); ++est_i; } while (0)\n'
        'template <int NS, int KEEP>\n__device__ __forceinline__ void gemm6q_landed')
d = rep(d, 'const __bf16* __restrict__ pk_next, bool first, F&& between, G&& at_piece) {\n  bf16x8 xs[NS][3][4];\n  int T[NS];\n  int sw = 0;',
        'const __bf16* __restrict__ pk_next, bool first, F&& between, G&& at_piece, int& est_i) {\n  bf16x8 xs[NS][3][4];\n  int T[NS];\n  int sw = 0;\n  EST();   /* block entry */')
d = rep(d, '  between();\n  if constexpr (!PREWAITED) gemm6q_landed<NS, 0>(acc, b);\n  __builtin_amdgcn_s_barrier();                     // ---- piece 0 has landed for every wave; slot 1 is free\n',
        '  between();\n  EST();   /* loads issued */\n  if constexpr (!PREWAITED) gemm6q_landed<NS, 0>(acc, b);\n  EST();   /* landed */\n  __builtin_amdgcn_s_barrier();\n  EST();   /* barrier 0 */\n')
d = rep(d, '  at_piece(0, b);\n  sweep_piece6<0, NS, NP>(acc, xs, ring + opaque_u(lane * 16u));\n  wait_vm_keep<KEEP1>();\n  __builtin_amdgcn_s_barrier();',
        '  at_piece(0, b);\n  EST();   /* split done */\n  sweep_piece6<0, NS, NP>(acc, xs, ring + opaque_u(lane * 16u));\n  EST();   /* sweep 0 */\n  wait_vm_keep<KEEP1>();\n  EST();   /* piece 1 landed */\n  __builtin_amdgcn_s_barrier();\n  EST();   /* barrier 1 */')
d = rep(d, '  sweep_piece6<1, NS, NP>(acc, xs, ring + opaque_u(lane * 16u + PIECE_BYTES6));\n  wait_vm_keep<KEEP2>();\n  __builtin_amdgcn_s_barrier();',
        '  sweep_piece6<1, NS, NP>(acc, xs, ring + opaque_u(lane * 16u + PIECE_BYTES6));\n  EST();   /* sweep 1 */\n  wait_vm_keep<KEEP2>();\n  EST();\n  __builtin_amdgcn_s_barrier();\n  EST();   /* barrier 2 */')
d = rep(d, '  sweep_piece6<2, NS, NP>(acc, xs, ring + opaque_u(lane * 16u));\n  wait_vm_keep<KEEP3>();\n  __builtin_amdgcn_s_barrier();',
        '  sweep_piece6<2, NS, NP>(acc, xs, ring + opaque_u(lane * 16u));\n  EST();   /* sweep 2 */\n  wait_vm_keep<KEEP3>();\n  EST();\n  __builtin_amdgcn_s_barrier();\n  EST();   /* barrier 3 */')
d = rep(d, '  sweep_piece6<3, NS, NP>(acc, xs, ring + opaque_u(lane * 16u + PIECE_BYTES6));\n  if constexpr (Prod<NP>::SCALED) {\n#pragma unroll\n    for (int u = 0; u < NS; ++u) scale_act(acc[u], -T[u]);\n  }\n}\n\n// ---- latency form',
        '  sweep_piece6<3, NS, NP>(acc, xs, ring + opaque_u(lane * 16u + PIECE_BYTES6));\n  EST();   /* sweep 3 */\n  if constexpr (Prod<NP>::SCALED) {\n#pragma unroll\n    for (int u = 0; u < NS; ++u) scale_act(acc[u], -T[u]);\n  }\n}\n\n// ---- latency form')
# kernel: counter, the three calls, stamps around the save stages and the epilogue
m = rep(m, '  float* st = stage_lds[wave];\n  auto nothing = [](int, Act (&)[NS]) {};\n', '  float* st = stage_lds[wave];\n  auto nothing = [](int, Act (&)[NS]) {};\n  int est_i = 0;\n  EST();   /* kernel: indices loaded */\n')
m = rep(m, '      for (int u = 0; u < NS; ++u) t_load(free_b[u], a.b2, kq);\n    }\n  });', '      for (int u = 0; u < NS; ++u) t_load(free_b[u], a.b2, kq);\n    }\n  }, est_i);')
m = rep(m, '      for (int u = 0; u < NS; ++u) t_load(free_a[u], a.b3, kq);\n    }\n  });', '      for (int u = 0; u < NS; ++u) t_load(free_a[u], a.b3, kq);\n    }\n  }, est_i);')
m = rep(m, '    if (q == 3) { const unsigned ln = lane_now(); t_load32(free_b[1], a.res, row_clamped(1, ln) * 512u + 16u * (ln >> 4)); }      // pieces 0 and 1 consumed\n  });',
        '    if (q == 3) { const unsigned ln = lane_now(); t_load32(free_b[1], a.res, row_clamped(1, ln) * 512u + 16u * (ln >> 4)); }      // pieces 0 and 1 consumed\n  }, est_i);\n  EST();   /* block 3 done */')
m = rep(m, '  if (full) save(std::true_type{}, acc, b, a.z1, 0u); else save(std::false_type{}, acc, b, a.z1, 0u);', '  EST();\n  if (full) save(std::true_type{}, acc, b, a.z1, 0u); else save(std::false_type{}, acc, b, a.z1, 0u);\n  EST();   /* save z1 + landed */')
m = rep(m, '  if (full) save(std::true_type{}, b, acc, a.z2, 4u); else save(std::false_type{}, b, acc, a.z2, 4u);', '  EST();\n  if (full) save(std::true_type{}, b, acc, a.z2, 4u); else save(std::false_type{}, b, acc, a.z2, 4u);\n  EST();   /* save z2 + landed */')
# export
m += '\nextern "C" int hgn_debug_edge_fwd_stamps(unsigned long long* host) {\n  (void)hipDeviceSynchronize();\n  return hipMemcpyFromSymbol(host, HIP_SYMBOL(hgn::est::g_est), 4 * 96 * 8) == hipSuccess ? 0 : 1;\n}\n'
open('.fst_tmp/mlp6_device.h', 'w').write(d)
open('.fst_tmp/mlp6.hip', 'w').write(m)
PY

( cd .fst_tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -I.. -I../../../include -c mlp6.hip -o mlp6_fst.o )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../hgn_amd/abl/libhgn_mp_fstamp.so mlp.o .fst_tmp/mlp6_fst.o fused_bwd.o fused_bwd3.o segment.o wgrad.o features.o host.o
rm -rf .fst_tmp
echo built fstamp
