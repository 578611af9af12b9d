#!/bin/bash
# Round-2 PMC passes (run on the GPU box through gpurun): HBM traffic of K1 and of the two PCG kernels on C3.  Separate --pmc passes with
# --kernel-trace only.  Output: gpurun_out/pmc_r02/*.txt -> copied into profiles/ by hand.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_r02
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/k1_$c -o pmc -- python scripts/k1_only.py > $OUT/k1_$c.log 2>&1
  python scripts/rocpd_summary.py pmc $(find $OUT/k1_$c -name "*.db" | head -1) $c k1_edges_kernel > $OUT/k1_$c.json
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pcg_$c -o pmc -- python scripts/gpu_pcg_kernel_times.py C3 > $OUT/pcg_$c.log 2>&1
  python scripts/rocpd_summary.py pmc $(find $OUT/pcg_$c -name "*.db" | head -1) $c mf_spmv > $OUT/pcg_spmv_$c.json
  python scripts/rocpd_summary.py pmc $(find $OUT/pcg_$c -name "*.db" | head -1) $c cg_update > $OUT/pcg_update_$c.json
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pcg_l2 -o pmc -- python scripts/gpu_pcg_kernel_times.py C3 > $OUT/pcg_l2.log 2>&1
for c in TCC_HIT_sum TCC_MISS_sum; do python scripts/rocpd_summary.py pmc $(find $OUT/pcg_l2 -name "*.db" | head -1) $c mf_spmv > $OUT/pcg_spmv_$c.json; python scripts/rocpd_summary.py pmc $(find $OUT/pcg_l2 -name "*.db" | head -1) $c cg_update > $OUT/pcg_update_$c.json; done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pcg_sq -o pmc -- python scripts/gpu_pcg_kernel_times.py C3 > $OUT/pcg_sq.log 2>&1
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU; do python scripts/rocpd_summary.py pmc $(find $OUT/pcg_sq -name "*.db" | head -1) $c mf_spmv > $OUT/pcg_spmv_$c.json; python scripts/rocpd_summary.py pmc $(find $OUT/pcg_sq -name "*.db" | head -1) $c cg_update > $OUT/pcg_update_$c.json; done
rm -rf $OUT/*/   # the .db files are large; the JSON summaries are what is kept
ls $OUT
