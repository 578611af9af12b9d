#!/usr/bin/env python3
"""gpurun_out/r06_final (scripts/profile_r06_final.sh) -> the files committed under profiles/: r06_* copies, r06_pcg_pmc.txt, k1_pmc_r06.json / k1_pmc_latest.json,
pcg_pmc_latest.json, mg_pmc_latest.json — each carrying the sha256 of the libpgo.so that was measured (bench.py refuses a static traffic figure whose sha differs
from the library it has loaded)."""
import json, os, shutil, sys
D = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r06_final'
P = os.path.join(D, 'pmc')
sha = open(os.path.join(D, 'libpgo_sha256.txt')).read().strip()
def load(name):
    return json.load(open(os.path.join(P, name)))
def mean_of(name, sub):
    best = None
    for k, v in load(name).items():
        if sub in k and (best is None or v['launches'] > best['launches']): best = v
    return best['mean'] if best else float('nan')
out = ['# libpgo.so sha256 ' + sha,
       '# PCG kernels on C3, round-6 FINAL build: rocprofv3 --kernel-trace --pmc <counter(s)> -- python scripts/gpu_pcg_kernel_times.py C3   (scripts/profile_r06_final.sh; separate passes)',
       '# means over all launches of the run; FETCH_SIZE x2 = HBM read bytes (gfx950 correction, MI355X_MICROARCH.md)', '']
for title, pre, sub in (('mf_spmv_kernel<false, false> (single-reduction form: w = A u)', 'pcg_spmv_', 'mf_spmv_kernel<false, false>'), ('cg_update_kernel<true> (single-reduction form)', 'pcg_update_', 'cg_update_kernel<true>')):
    f, w = mean_of(pre + 'FETCH_SIZE.json', sub), mean_of(pre + 'WRITE_SIZE.json', sub)
    h, m = mean_of(pre + 'TCC_HIT_sum.json', sub), mean_of(pre + 'TCC_MISS_sum.json', sub)
    wc, wa, bc, va = (mean_of(pre + 'SQ_%s.json' % c, sub) for c in ('WAVE_CYCLES', 'WAIT_ANY', 'BUSY_CYCLES', 'ACTIVE_INST_VALU'))
    out += ['## ' + title, 'FETCH_SIZE %.1f KiB -> %.1f MB read' % (f, 2 * f * 1024 / 1e6), 'WRITE_SIZE %.1f KiB -> %.1f MB written' % (w, w * 1024 / 1e6),
            'TCC_HIT_sum %.0f  TCC_MISS_sum %.0f  -> L2 hit rate %.1f %%' % (h, m, 100 * h / (h + m)),
            'SQ_WAVE_CYCLES %.3g  SQ_WAIT_ANY %.3g  -> %.0f %% of resident wave-cycles waiting;  SQ_BUSY_CYCLES %.3g  SQ_ACTIVE_INST_VALU %.3g' % (wc, wa, 100 * wa / wc, bc, va), '']
# multigrid iteration: (nearly) every launch of the run belongs to one of the timed iterations: bytes per iteration = sum over the iteration's kernels of mean bytes x launches / iterations
fetch, write = load('mg_all_FETCH_SIZE.json'), load('mg_all_WRITE_SIZE.json')
its = max(v['launches'] for k, v in fetch.items() if 'cg_update_mg_kernel' in k)
out += ['## one multigrid-preconditioned PCG iteration (python scripts/gpu_mg_iteration_only.py: %d iterations)' % its]
tot = 0.0
for k in sorted(fetch, key=lambda k: -fetch[k]['mean'] * fetch[k]['launches']):
    if not any(s in k for s in ('mf_spmv_kernel<false, false>', 'cg_update_mg', 'mg_down', 'mg_sdown', 'mg_up', 'mg_dense_solve')): continue
    per_it = max(1, round(fetch[k]['launches'] / its))       # (the run's two LM steps and the set-up's power method add a few launches of the matvec and of mg_smooth_step: whole launches per iteration)
    if 'mf_spmv_kernel' in k: per_it = 1
    b = (2 * fetch[k]['mean'] + write.get(k, {'mean': 0})['mean']) * 1024 * per_it
    tot += b
    out.append('%-62s %d launches / iteration, %6.1f MB / iteration' % (k[:62], per_it, b / 1e6))
out += ['total %.1f MB per iteration' % (tot / 1e6), '']
open('profiles/r06_pcg_pmc.txt', 'w').write('\n'.join(out))
corr = 'x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streams, MI355X_MICROARCH.md)'
pcg = {'libpgo_sha256': sha, 'workload': 'C3', 'fetch_correction': corr}
for key, pre, sub in (('matvec', 'pcg_spmv_', 'mf_spmv_kernel<false, false>'), ('update', 'pcg_update_', 'cg_update_kernel<true>')):
    f, w = mean_of(pre + 'FETCH_SIZE.json', sub), mean_of(pre + 'WRITE_SIZE.json', sub)
    pcg[key] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'hbm_bytes_per_launch': (2 * f + w) * 1024}
json.dump(pcg, open('profiles/pcg_pmc_latest.json', 'w'), indent=1)
json.dump({'libpgo_sha256': sha, 'workload': 'C3', 'hbm_bytes_per_iteration': tot, 'iterations_measured': its, 'fetch_correction': corr}, open('profiles/mg_pmc_latest.json', 'w'), indent=1)
kf, kw = mean_of('k1_FETCH_SIZE.json', 'k1_edges_kernel<true'), mean_of('k1_WRITE_SIZE.json', 'k1_edges_kernel<true')
k1 = {'round': 'r06', 'libpgo_sha256': sha, 'kernel': 'k1_edges_kernel<true, false>', 'workload': 'C3', 'FETCH_SIZE_KiB': kf, 'WRITE_SIZE_KiB': kw, 'fetch_correction': corr,
      'hbm_bytes_per_launch': (2 * kf + kw) * 1024, 'algorithmic_bytes_per_launch': 221600620.0}
json.dump(k1, open('profiles/k1_pmc_r06.json', 'w'), indent=1); json.dump(k1, open('profiles/k1_pmc_latest.json', 'w'), indent=1)
for name in ('r06_bench.json', 'r06_bench_under_rocprof.json', 'r06_bench_kernel_stats.txt', 'r06_mg_kernel_stats.txt', 'r06_k1_400k_kernel_stats.txt', 'r06_session_kernel_stats.txt', 'r06_all_configs.txt',
             'r06_mg_graph_types.txt', 'r06_smoothed_ab.txt', 'r06_session_replay_2deg.jsonl', 'r06_bench_gloo2.json', 'r06_multi_overhead.json', 'r06_multi_overhead_mg.json',
             'r06_session_step_times.txt', 'r06_gate.txt', 'r06_idle_gaps.txt', 'r06_smoothed_fine_rule.txt', 'r06_mg_crossover.txt', 'r06_ranks_c3x4.json', 'r06_ranks_c5x8.json', 'r06_ranks_c3x8.json', 'r06_bench_c5_strong_1gpu.json', 'r06_build_phases.txt'):
    src = os.path.join(D, name)
    if os.path.exists(src): shutil.copy(src, os.path.join('profiles', name))
print('\n'.join(out)); print(json.dumps(k1, indent=1))
