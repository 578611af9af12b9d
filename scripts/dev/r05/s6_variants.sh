#!/bin/bash
# round 5, call 6: small A/Bs on the multigrid iteration (update kernel's register budget, lane groups of the restriction), and the 2-rank gloo bench with its C5 leg
export TMPDIR=/tmp
export PGO_ENABLE_DEBUG_HOOKS=1
OUT=gpurun_out/r05_s6
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
python scripts/dev/ab_variant.py nobounds "-DPGO_SR_MG_NO_BOUNDS" 3 -- scripts/dev/mg_iteration_time.py C3 > $OUT/ab_nobounds.txt 2>&1
cat $OUT/ab_nobounds.txt
for v in 2.5 10 20; do echo "## PGO_DEBUG_RT_SEG_BLOCKS=$v"; PGO_DEBUG_RT_SEG_BLOCKS=$v python scripts/dev/mg_iteration_time.py C3; done > $OUT/rt_seg.txt 2>&1
echo "## default (5)" >> $OUT/rt_seg.txt; python scripts/dev/mg_iteration_time.py C3 >> $OUT/rt_seg.txt 2>&1
cat $OUT/rt_seg.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --collective gloo --c5-timeout 600 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
python -c "
import json; d=json.load(open('$OUT/bench_gloo2.json')); print(d['value'], d['n_gpus'], json.dumps(d['c5_strong'])[:600])"
tail -3 $OUT/bench_gloo2.err
