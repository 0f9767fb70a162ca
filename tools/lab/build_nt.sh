#!/bin/bash
# Laboratory builds of the fused edge backward (csrc/fused_bwd3.hip) with non-temporal hints on its once-read / once-written streams:
#   tools/lab/build_nt.sh 1 2 3 4 8 12 15 ...   ->  hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_nt<N>.so   (HGN_LIB=<that file> selects it)
# bits: 1 operand rows (z2 / z1) LDS-DMA with nt (the product's form since round 5: build 0 for the form without), 2 chain loads of xhat / d(e') nt,
# 4 de stores nt, 8 dz1 stores nt.   Measured (profiles/r05_ab_nt_wg2.log): 1 -3.5 %, 2 +1 %, 4 +1 %, 8 0, 12 +1 %, 15 +11 %.
# The source is patched into a temporary copy: csrc/ (and the kernel-source stamp of the PMC records) stays as it is.
set -e
cd "$(dirname "$0")/../../hyper-graph-nets_amd/csrc"
make -s
mkdir -p ../hgn_amd/abl .nt_tmp
python3 - <<'PY'
s = open('fused_bwd3.hip').read()
helpers = '''
#ifndef HGN_NTX
#define HGN_NTX 0
#endif
namespace hgn {
__device__ __forceinline__ void t_load32nt(Act& a, const float* __restrict__ base, unsigned byte_off) {
  const char* p = reinterpret_cast<const char*>(base);
  HGN_FOR_B(fb) a.v[fb] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (byte_off + 64u * fb)));
}
__device__ __forceinline__ void t_store32nt(const Act& a, float* __restrict__ base, unsigned byte_off) {
  char* p = reinterpret_cast<char*>(base);
  HGN_FOR_B(fb) __builtin_nontemporal_store(a.v[fb], reinterpret_cast<f32x4*>(p + (byte_off + 64u * fb)));
}
}
#if HGN_NTX & 1
#define NT_DMA " nt"
#else
#define NT_DMA ""
#endif
#if HGN_NTX & 2
#define T_LOADX t_load32nt
#else
#define T_LOADX t_load32
#endif
#if HGN_NTX & 4
#define T_STORE_DE t_store32nt
#else
#define T_STORE_DE t_store32
#endif
#if HGN_NTX & 8
#define T_STORE_DZ1 t_store32nt
#else
#define T_STORE_DZ1 t_store32
#endif
'''
def rep(old, new, n=1):
    global s
    assert s.count(old) == n, (old, s.count(old))
    s = s.replace(old, new)
rep('#include "fused_args.h"\n', '#include "fused_args.h"\n' + helpers)
rep('global_load_lds_dwordx4 %0, %1 nt" : : "v"(vo), "s"(A)', 'global_load_lds_dwordx4 %0, %1" NT_DMA : : "v"(vo), "s"(A)')
rep('t_load32(xh, a.xhat,', 'T_LOADX(xh, a.xhat,')
rep('t_load32(dout, a.d_out,', 'T_LOADX(dout, a.d_out,')
rep('t_store32(g, a.dz1,', 'T_STORE_DZ1(g, a.dz1,')
rep('t_store32(g, d.dx,', 'T_STORE_DE(g, d.dx,')
open('.nt_tmp/fused_bwd3_nt.hip', 'w').write(s)
PY
for n in "$@"; do
  ( cd .nt_tmp && mkdir -p t$n && cd t$n && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -I../.. -I../../../../include -DHGN_NTX=$n -save-temps=obj -c ../fused_bwd3_nt.hip -o fused3_nt$n.o \
    && python3 ../../check_fused_counts.py fused_bwd3_nt-hip-amdgcn-amd-amdhsa-gfx950.s \
    && grep -c " nt" fused_bwd3_nt-hip-amdgcn-amd-amdhsa-gfx950.s ) &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../hgn_amd/abl/libhgn_mp_nt$n.so mlp.o mlp6.o fused_bwd.o .nt_tmp/t$n/fused3_nt$n.o segment.o wgrad.o features.o host.o
  echo built nt$n
done
rm -rf .nt_tmp
