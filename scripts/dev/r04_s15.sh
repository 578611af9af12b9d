#!/bin/bash
mkdir -p gpurun_out/r04_s15
python scripts/dev/ab_variant.py nouniform "-DPGO_NO_UNIFORM" 3 -- scripts/dev/setup_time.py > gpurun_out/r04_s15/ab_uniform.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_s15/trace -o t -- python scripts/gpu_mg_profile.py > gpurun_out/r04_s15/trace.log 2>&1
python scripts/rocpd_summary.py stats $(find gpurun_out/r04_s15/trace -name "*.db" | head -1) | grep "galerkin\|psTw\|mg_w_\|mg_ps_\|val_f32\|gj_\|dinv\|geometry" > gpurun_out/r04_s15/setup_kernels.txt
rm -rf gpurun_out/r04_s15/trace
cat gpurun_out/r04_s15/ab_uniform.txt gpurun_out/r04_s15/setup_kernels.txt
