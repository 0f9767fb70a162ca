#!/bin/bash
# Memory-side counters of the edge-block kernels (latency of L1->L2 reads / writes, L2->fabric reads, stalls of the texture path):
# bash tools/memlat_round.sh r05   -> gpurun_out/<tag>_memlat/
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R}_memlat; mkdir -p $O
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" \
           "TCC_BUSY_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/m$i -- python3 tools/fusedbench.py --iters 3 > $O/m$i.log 2> $O/m$i.err; echo "pass $i rc=$? ($set)"
  python tools/pmc_kernels.py $O/m$i > $O/m$i.txt 2>&1; cat $O/m$i.txt
  rm -rf $O/m$i/*/*.db 2>/dev/null
done
