#!/bin/bash
# PMC passes over the two PCG kernels (C3, pgo_time_kernel which=2): occupancy / issue / stall picture of the matvec.  Separate passes,
# kernel-trace only (no other trace domains).  Summaries: scripts/rocpd_summary.py pmc <db> <counter> <kernel substring>
export TMPDIR=/tmp
OUT=gpurun_out/pmc_pcg
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc -- python scripts/gpu_kernel_times.py > $OUT/p$i.log 2>&1
  for c in $set; do
    echo "== $c"; python scripts/rocpd_summary.py pmc $(find $OUT/p$i -name "*.db" | head -1) $c mf_spmv; python scripts/rocpd_summary.py pmc $(find $OUT/p$i -name "*.db" | head -1) $c cg_update
  done
done
