#!/bin/bash
# Hardware runs of the smoothed keyframe transition (pgo_options::mg_smoothed_fine), through gpurun from the repo root: parts test,c3,poison,variants,c4,verbose,types,prof.
export TMPDIR=/tmp PGO_ENABLE_DEBUG_HOOKS=1
OUT=$PWD/gpurun_out/r05_smoothed_fine
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()" > $OUT/build.log 2>&1
PARTS=${1:-test,c3}
if [[ $PARTS == *test* ]]; then timeout 900 python -m pytest tests/test_gpu_multigrid.py -q -m gpu -k "smoothed_keyframe" -s -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $OUT/test.txt; fi
if [[ $PARTS == *c3* ]]; then timeout 600 python scripts/dev/r05/ab_options.py C3 20 2 "" "mg_smoothed_fine=1" > $OUT/ab_c3.txt 2>&1; fi
if [[ $PARTS == *poison* ]]; then PGO_DEBUG_POISON=1 timeout 900 python -m pytest tests/test_gpu_multigrid.py -q -m gpu -k "smoothed_keyframe" -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 > $OUT/test_poison.txt; fi
if [[ $PARTS == *variants* ]]; then timeout 600 python scripts/dev/r05/ab_options.py C3 20 1 "" "mg_smoothed_fine=1" "mg_smoothed_fine=1,mg_passes=3" "mg_smoothed_fine=1,mg_passes=1" > $OUT/ab_c3_variants.txt 2>&1; fi
if [[ $PARTS == *c4* ]]; then timeout 600 python scripts/dev/r05/ab_options.py C4 20 1 "" "mg_smoothed_fine=1" > $OUT/ab_c4.txt 2>&1; fi
if [[ $PARTS == *verbose* ]]; then timeout 300 python scripts/dev/r05/verbose_ten.py C3 "mg_smoothed_fine=1,verbosity=2" > $OUT/verbose.txt 2>&1; fi
if [[ $PARTS == *types* ]]; then timeout 900 python scripts/dev/r05/opt_types.py "types,C4" "" "mg_smoothed_fine=1" > $OUT/types.txt 2>&1; fi
if [[ $PARTS == *prof* ]]; then
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python scripts/dev/r05/smoothed_fine_iteration_profile.py "mg_smoothed_fine=1,verbosity=1" > $OUT/prof.log 2>&1
  python scripts/rocpd_summary.py stats $(find $OUT/trace -name "*.db" | head -1) > $OUT/prof_kernel_stats.txt; rm -rf $OUT/trace
  grep "multigrid:" $OUT/prof.log | head -3 >> $OUT/prof_kernel_stats.txt
fi
if [[ $PARTS == *typeprof* ]]; then
  for w in plain40k noout60k; do for o in "" "mg_smoothed_fine=1"; do
    n=${w}_${o:+fine}; n=${n%_}
    rocprofv3 --kernel-trace --stats -d $OUT/trace_$n -o t -- python scripts/dev/r05/smoothed_fine_type_profile.py $w "$o" > $OUT/typeprof_$n.log 2>&1
    { echo "## $w  options: ${o:-defaults}"; grep "multigrid: [0-9]* keyframes" $OUT/typeprof_$n.log | head -1; tail -1 $OUT/typeprof_$n.log; python scripts/rocpd_summary.py stats $(find $OUT/trace_$n -name "*.db" | head -1) | head -16 | cut -c1-170; } >> $OUT/typeprof.txt
    rm -rf $OUT/trace_$n
  done; done
fi
for f in $OUT/*.txt; do echo "== $f"; tail -n 30 $f; done
