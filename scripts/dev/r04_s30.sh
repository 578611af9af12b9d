#!/bin/bash
O=gpurun_out/r04_s30; mkdir -p $O
SHA=$(sha256sum solve_keyframe_pose_graph_amd/libpgo.so | cut -d' ' -f1)
{ echo "# libpgo.so sha256 $SHA"; echo "# PGO_DEBUG_POISON=1 python scripts/gpu_fuzz_soak_multigrid.py 16 77   (library defaults vs multigrid from the first iteration at cg_rel_tolerance 1e-12, 12 LM iterations)";
  PGO_DEBUG_POISON=1 timeout 1500 python scripts/gpu_fuzz_soak_multigrid.py 16 77 2>/dev/null; } > $O/r04_soak_multigrid.txt
tail -4 $O/r04_soak_multigrid.txt
{ echo "# libpgo.so sha256 $SHA"; echo "# PGO_DEBUG_POISON=1 python scripts/archive/gpu_fuzz_soak.py 150 99   (library defaults vs the oracle's exact solve, small random graphs with outliers)";
  PGO_DEBUG_POISON=1 timeout 1500 python scripts/archive/gpu_fuzz_soak.py 150 99 2>/dev/null | tail -12; } > $O/r04_soak_small.txt
tail -6 $O/r04_soak_small.txt
