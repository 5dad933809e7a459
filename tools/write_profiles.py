#!/usr/bin/env python3
"""Turn the raw outputs of tools/make_profiles.sh (under gpurun_out/) into the committed
summaries under profiles/:  kernel-stats table of the default bench, the bench line, and the
HBM-traffic table (+ a JSON with per-image traffic that bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, 'gpurun_out')
prof = os.path.join(root, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r4_final'

# ---- 1. kernel stats of the default bench
stats = max(glob.glob(os.path.join(go, 'final_stats', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)   # newest
rows = list(csv.DictReader(open(stats)))
for r in rows:
    r['short'] = re.sub(r'^void ', '', r['Name']).split('(')[0]
ours = [r for r in rows if re.match(r'(void )?k_', r['Name'])]
for r in ours:
    r['short'] = re.sub(r'^void ', '', r['Name']).split('(')[0]
ours.sort(key=lambda r: -float(r['TotalDurationNs']))
lines = ['# %s: rocprofv3 --kernel-trace --stats of the default bench' % tag,
         'command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no_cpu_baseline --steps 20 --warmup 5',
         '(5 warm-up + 20 device-resident + 3 host-loop warm-up + 20 host-loop (the headline) steps of 30 images, 1024x2048, DRN-D-22 fp32, SLIC 200, '
         'mean pooling, k=2)', '',
         '## libspalign kernels (hand-written HIP)', '',
         '| kernel | calls | total ms | avg us | min us | max us |', '|---|---|---|---|---|---|']
for r in ours:
    lines.append('| %s | %s | %.2f | %.1f | %.1f | %.1f |' % (
        r['short'], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
        float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
lines += ['', '## top 12 kernels overall', '', '| kernel | calls | total ms | avg us |', '|---|---|---|---|']
for r in rows[:12]:
    lines.append('| %s | %s | %.2f | %.1f |' % (r['short'][:80], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                             float(r['AverageNs']) / 1e3))
open(os.path.join(prof, tag + '_bench_default_kernel_stats.md'), 'w').write('\n'.join(lines) + '\n')
with open(os.path.join(prof, tag + '_bench_default_kernel_stats.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs'])
    for r in rows:
        if float(r['TotalDurationNs']) > 2e5:
            w.writerow([r['short'][:120], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['MinNs'], r['MaxNs']])

# ---- 2. bench lines
for src, dst in (('final_bench_line.json', tag + '_bench_line.json'),
                 ('final_stats_bench_line.json', tag + '_bench_line_under_rocprof.json')):
    txt = [l for l in open(os.path.join(go, src)).read().splitlines() if l.startswith('{')]
    if txt:
        open(os.path.join(prof, dst), 'w').write(txt[-1] + '\n')

# ---- 3. HBM traffic (batch 8)
def table(path):
    out = {}
    for l in open(path):
        m = re.match(r'(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+launches\s+(\d+)\s+mean\s+([\d.]+)', l)
        if m:
            out[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)))
    return out
fe, wr = table(os.path.join(go, 'final_pmc_fetch.txt')), table(os.path.join(go, 'final_pmc_write.txt'))
B = 8
px = 1024 * 2048
alg = {'k_rgb2lab': 24 * px, 'k_slic_assign': 16 * px, 'k_slic_update': 16 * px, 'k_paint': 6 * px,
       'k_pool_mean': 512 * 128 * 256 * 4 + px * 4, 'k_cell_weights': 4 * px + 128 * 256 * 16,
       'k_conn_relabel': 12 * px, 'k_ccl_merge': 8 * px}
wide = {'k_rgb2lab', 'k_slic_assign', 'k_paint', 'k_pool_mean', 'k_pool_mean_vec<0, 1>', 'k_conn_relabel',
        'k_cell_weights', 'k_run_rows', 'k_bbox_count_lds', 'k_seg_moments', 'k_wino_in', 'k_wino_out<0>', 'k_wino4_in<float4>', 'k_wino4_out<0, float4>',
        'k_conv3x3_f32<0, 256, 1, 256>', 'k_conv3x3_f32<0, 128, 9, 128>'}      # 16 B/lane streaming reads
families = {'connectivity(all)': ('k_run_', 'k_conn_', 'k_small_bbox'),
            'segment_stats(all)': ('k_stats_', 'k_bbox_', 'k_seg_moments', 'k_offsets')}
alias = {'k_pool_mean_vec<0, 1>': 'k_pool_mean', 'k_slic_update<4>': 'k_slic_update', 'k_slic_update2<4>': 'k_slic_update',
         'k_kmeans<double, 2>': 'k_kmeans'}
alg['connectivity(all)'] = 8 * px
alg['segment_stats(all)'] = 4 * px
lines = ['# HBM traffic per launch from PMC counters (%s), batch 8, 1024x2048, MI355X' % tag,
         'command (one pass per counter, as MI355X_MICROARCH.md prescribes):',
         '  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/prof_stages.py --batch 8 --reps 2',
         '  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/prof_stages.py --batch 8 --reps 2',
         'FETCH MB = FETCH_SIZE(KB) x 1024, x 2 for the kernels marked * (gfx950 correction: their reads are wide',
         '16 B/lane streams, whose 128-B requests are tallied at 64 B).  k_slic_update gathers dwords: its requests are',
         '64 B (checked: TCC_EA0_RDREQ x 64 B = FETCH_SIZE, TCC_EA0_RDREQ_32B = 0), no correction.',
         'WRITE MB = WRITE_SIZE(KB) x 1024.  "vs alg" = HBM MB / algorithmic MB of DESIGN.md section 4.', '',
         '```', '%-22s %8s %12s %12s %12s %8s' % ('kernel', 'launches', 'FETCH MB', 'WRITE MB', 'HBM MB', 'vs alg')]
traffic = {}
for k in sorted(fe):
    name = k.replace('void ', '')
    if not name.startswith('k_'):
        continue
    is_wide = name in wide or name.startswith(('k_wino4_in<', 'k_wino4_out<', 'k_wino4_out_s<', 'k_gemm_f16x3<', 'k_gemm_f16x3_stag<', 'k_conv3x3_f32<'))
    f_mb = fe[k][1] * 1024 / 1e6 * (2 if is_wide else 1)
    w_mb = wr.get(k, (0, 0.0))[1] * 1024 / 1e6
    tot = f_mb + w_mb
    a = alg.get(name)
    lines.append('%-22s %8d %11.1f%s %12.1f %12.1f %8s' % (name[:22], fe[k][0], f_mb, '*' if is_wide else ' ',
                                                          w_mb, tot, ('%.2fx' % (tot * 1e6 / (a * B))) if a else '-'))
    traffic[name] = tot * 1e6 / B
    if name in alias:
        traffic[alias[name]] = traffic[name]
# the DRN epilogue kernel on the heavy layers' shapes (prof_stages.py --bias_act): bytes per launch vary with the
# layer, so its entry is the RATIO of HBM bytes to algorithmic bytes (8 B per element, 12 with a residual)
ratio = {}
bkey = [k for k in fe if k.replace('void ', '').startswith('k_bias_act_f32')]
if bkey:
    n_l, f_kb = fe[bkey[0]]
    w_kb = wr.get(bkey[0], (0, 0.0))[1]
    hbm = (f_kb * 2 + w_kb) * 1024                      # 16-byte loads: FETCH_SIZE x 2
    el = lambda C: B * C * (1024 // 8) * (2048 // 8) * 4  # bytes of one float32 activation
    alg_mean = (3 * el(512) + 2 * el(512) + 3 * el(256)) / 3.0
    ratio['k_bias_act(all)'] = hbm / alg_mean
    lines.append('')
    lines.append('k_bias_act_f32 on the 512 ch + residual / 512 ch / 256 ch + residual layer shapes: %.1f MB per launch on average = %.2fx the algorithmic bytes'
                 % (hbm / 1e6, ratio['k_bias_act(all)']))
# one Winograd F(4x4,3x3) layer (prof_stages.py --wino: 512 -> 512, dilation 2): input transform, the batched GEMM launch, output
# transform, each against the bytes it moves by construction (bench.py scales these ratios to its launches' mix)
def hbm_of(kname):
    key = [k for k in fe if k.replace('void ', '').startswith(kname)]           # (pmc_summary truncates long template names)
    fx = 2
    return (fe[key[0]][1] * 1024 * fx + wr.get(key[0], (0, 0.0))[1] * 1024) if key else None
if any(k.replace('void ', '').startswith('k_wino4_in') for k in fe):
    act = 4.0 * B * (1024 // 8) * (2048 // 8) * 512           # one 512-channel activation at 1/8 resolution
    for kname, bench_name, built, what in (
            ('k_wino4_in<', 'k_wino_in', 3.25 * act, 'X read + V = 2.25 X written'),
            ('k_conv3x3_f32<0, 256, 1, 256>', 'k_conv3x3_f32<taps 1>(GEMM form, all)', 4.5 * act, 'V = 2.25 X read + M = 2.25 Y written'),
            ('k_gemm_f16x3_stag<256, 256', 'k_gemm_f16x3_stag<256, 256>', 4.5 * act, 'V = 2.25 X read + M = 2.25 Y written'),
            ('k_gemm_f16x3<256, 256', 'k_gemm_f16x3_stag<256, 256>', 4.5 * act, 'V = 2.25 X read + M = 2.25 Y written'),
            ('k_wino4_out_s<0', 'k_wino_out', 3.25 * act, 'M = 2.25 Y read + Y written'),
            ('k_wino4_out<0', 'k_wino_out', 3.25 * act, 'M = 2.25 Y read + Y written')):
        h = hbm_of(kname)
        if h:
            ratio[bench_name] = h / built
            for tk in [k for k in traffic if k.startswith(kname)]:
                traffic.pop(tk, None)
            lines.append('%-32s %8.1f MB per launch = %.2fx the %.1f MB it moves by construction (%s)' % (kname, h / 1e6, h / built, built / 1e6, what))
passes = fe.get('k_rgb2lab', (1, 0))[0]
lines.append('')
lines.append('families (all launches of one pass summed; %d passes profiled):' % passes)
for fam, prefixes in families.items():
    tot = 0.0
    for k in fe:
        name = k.replace('void ', '')
        if name.startswith(prefixes):
            f_mb = fe[k][1] * fe[k][0] * 1024 / 1e6 * (2 if name in wide else 1)
            w_mb = wr.get(k, (0, 0.0))[1] * wr.get(k, (0, 0.0))[0] * 1024 / 1e6
            tot += (f_mb + w_mb) / passes
    lines.append('%-22s %8s %12s %12s %12.1f %8s' % (fam, '-', '', '', tot, '%.2fx' % (tot * 1e6 / (alg[fam] * B))))
    traffic[fam] = tot * 1e6 / B
lines.append('```')
open(os.path.join(prof, tag + '_pmc_hbm_traffic_b8.md'), 'w').write('\n'.join(lines) + '\n')
json.dump({'note': 'HBM bytes per launch PER IMAGE (1024x2048) from the PMC passes of ' + tag +
                   '_pmc_hbm_traffic_b8.md; bench.py reports roofline.traffic = this x images per launch',
           'bytes_per_image_per_launch': traffic, 'ratio_to_algorithmic_bytes': ratio},
          open(os.path.join(prof, 'pmc_traffic.json'), 'w'), indent=1)
print('\n'.join(lines[-40:]))

# ---- 4. the other operating points (tools/make_profiles.sh: one bench line each)
for src in sorted(glob.glob(os.path.join(go, 'final_variant_*.json'))):
    txt = [l for l in open(src).read().splitlines() if l.startswith('{')]
    if txt:
        name = os.path.basename(src)[len('final_variant_'):-len('.json')]
        open(os.path.join(prof, '%s_bench_%s.json' % (tag, name)), 'w').write(txt[-1] + '\n')
        d = json.loads(txt[-1])
        print('%-28s %9.2f %s  %8.2f ms/step' % (name, d['value'], d['unit'], d['ms_per_step']))
