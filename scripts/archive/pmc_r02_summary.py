#!/usr/bin/env python3
"""Condenses gpurun_out/pmc_r02/*.json (scripts/profile_r02_pmc.sh) into profiles/r02_pcg_pmc.txt and profiles/k1_pmc_r02.json / k1_pmc_latest.json."""
import json, os, sys
D = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_r02'
def mean_of(name, sub):
    d = json.load(open(os.path.join(D, name)))
    best = None
    for k, v in d.items():
        if sub in k and (best is None or v['launches'] > best['launches']): best = v
    return best['mean'] if best else float('nan')
out = ['# PCG kernels on C3, round-2 FINAL build (one lane per in-tile edge): rocprofv3 --kernel-trace --pmc <counter(s)> -- python scripts/gpu_pcg_kernel_times.py C3   (scripts/profile_r02_pmc.sh; separate passes)',
       '# means over all launches of the run (2 LM steps of the solve + 3 x 60 timed launches of pgo_time_kernel 2 / 4 / 5); FETCH_SIZE x2 = HBM read bytes (gfx950 correction, MI355X_MICROARCH.md)', '']
design = json.load(open(os.path.join(D, 'design_bytes.json'))) if os.path.exists(os.path.join(D, 'design_bytes.json')) else {}
for title, pre, sub in (('mf_spmv_kernel<true, false>', 'pcg_spmv_', 'mf_spmv_kernel<true, false>'), ('cg_update_kernel', 'pcg_update_', 'cg_update_kernel')):
    f, w = mean_of(pre + 'FETCH_SIZE.json', sub), mean_of(pre + 'WRITE_SIZE.json', sub)
    h, m = mean_of(pre + 'TCC_HIT_sum.json', sub), mean_of(pre + 'TCC_MISS_sum.json', sub)
    wc, wa, bc, va = (mean_of(pre + 'SQ_%s.json' % c, sub) for c in ('WAVE_CYCLES', 'WAIT_ANY', 'BUSY_CYCLES', 'ACTIVE_INST_VALU'))
    out += ['## ' + title,
            'FETCH_SIZE %.1f KiB -> %.1f MB read' % (f, 2 * f * 1024 / 1e6),
            'WRITE_SIZE %.1f KiB -> %.1f MB written' % (w, w * 1024 / 1e6),
            'TCC_HIT_sum %.0f  TCC_MISS_sum %.0f  -> L2 hit rate %.1f %%' % (h, m, 100 * h / (h + m)),
            'SQ_WAVE_CYCLES %.3g  SQ_WAIT_ANY %.3g  -> %.0f %% of resident wave-cycles waiting;  SQ_BUSY_CYCLES %.3g  SQ_ACTIVE_INST_VALU %.3g' % (wc, wa, 100 * wa / wc, bc, va), '']
open('profiles/r02_pcg_pmc.txt', 'w').write('\n'.join(out))
pcg = {}
for key, pre, sub in (('matvec', 'pcg_spmv_', 'mf_spmv_kernel<true, false>'), ('update', 'pcg_update_', 'cg_update_kernel')):
    f, w = mean_of(pre + 'FETCH_SIZE.json', sub), mean_of(pre + 'WRITE_SIZE.json', sub)
    pcg[key] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'hbm_bytes_per_launch': (2 * f + w) * 1024}
pcg['workload'] = 'C3'; pcg['fetch_correction'] = 'x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streams, MI355X_MICROARCH.md)'
json.dump(pcg, open('profiles/pcg_pmc_latest.json', 'w'), indent=1)
kf, kw = mean_of('k1_FETCH_SIZE.json', 'k1_edges_kernel<true'), mean_of('k1_WRITE_SIZE.json', 'k1_edges_kernel<true')
old = json.load(open('profiles/k1_pmc_r02.json'))
old.update({'FETCH_SIZE_KiB': kf, 'WRITE_SIZE_KiB': kw, 'hbm_bytes_per_launch': (2 * kf + kw) * 1024})
json.dump(old, open('profiles/k1_pmc_r02.json', 'w'), indent=1)
json.dump(old, open('profiles/k1_pmc_latest.json', 'w'), indent=1)
print('\n'.join(out)); print(json.dumps(old, indent=1))
