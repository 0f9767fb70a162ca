#!/bin/bash
# A/B of laboratory builds on ONE box: bash tools/lab/ab_variants.sh <tag> <variant> ...   (variants: names under hgn_amd/abl/libhgn_mp_<name>.so;
# "base" = the product library).  Each run: tools/fusedbench.py (edge block at 1 188 096 rows, library's own HIP-event profiler).
T=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$T; mkdir -p $O
for v in "$@"; do
  if [ "$v" = base ]; then unset HGN_LIB; else export HGN_LIB=$GRAFT_REPO_ROOT/hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_$v.so; fi
  echo "== $v" | tee -a $O/ab.log
  timeout -k 10 120 python tools/fusedbench.py --iters 8 --reps 2 2>/dev/null | grep rows | tee -a $O/ab.log || exit 1
done
