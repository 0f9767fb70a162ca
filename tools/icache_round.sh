#!/bin/bash
# Instruction-cache and LDS-wait counters of the two edge kernels (46 KB / 55 KB of straight-line code against a 64 KB instruction
# cache shared by two CUs): bash tools/icache_round.sh r05   -> gpurun_out/<tag>_icache/
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R}_icache; mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQC_DCACHE[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_VALU_MFMA[A-Z_]*" $O/counters_available.txt | sort -u > $O/counter_names.txt
cat $O/counter_names.txt | tr '\n' ' '; echo
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES" \
           "SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/ic$i -- python3 tools/fusedbench.py --iters 3 > $O/ic$i.log 2> $O/ic$i.err; echo "pass $i rc=$?"
  python tools/sq_counters.py $O/ic$i 2>/dev/null | grep -A7 "edge_bwd_fused3_kernel\|mlp6_fwd_edge_kernel<" > $O/ic$i.txt; cat $O/ic$i.txt
  rm -rf $O/ic$i/*/*.db 2>/dev/null
done
