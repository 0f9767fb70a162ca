#!/bin/bash
# Timing of the edge kernels (product mode 3) with parts switched off at compile time: tools/build_ablations.sh 1 2 4 8 f1 f2 f4 f8 f32 f64 f128 first.
#   bash tools/fused_ablation.sh > gpurun_out/r05_edge_ablation.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
echo "# forward: HGN_ABL 1 no weight DMA, 2 no MFMA, 4 no stores of saved activations, 8 no row loads"
echo "# fused backward (fused_bwd3.hip): HGN_FEXP 1 no weight DMA, 2 no operand fetch, 4 no dz1 / de stores, 8 no chain row loads, 32 no weight-gradient blocks, 64 no chain products, 128 no publish"
echo "== full"; timeout -k 10 120 python tools/fusedbench.py --iters 6 2>&1 | tail -1
for n in 1 2 4 8 f1 f2 f4 f8 f32 f64 f128; do
  echo "== abl $n"; HGN_LIB=hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_abl$n.so timeout -k 10 120 python tools/fusedbench.py --iters 6 2>&1 | tail -1
done
