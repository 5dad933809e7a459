#!/usr/bin/env python
"""SURVEY.md 8d "C3": the drop-in DRIVER measured end to end, not the kernel loop.

    python tools/driver300.py [--n 300] [--height 1024 --width 2048] [--batchsize 30] [--dtype fp32]

Writes N synthetic Cityscapes-shaped PNGs (leftImg8bit) + gtFine labelIds PNGs to a scratch directory, then runs
the reference's command line (utils/create_random300_labels.sh:37-51, README.md:99-107) against this repository's
`batch_spalign_kmeans.py` as a CHILD process:

    python batch_spalign_kmeans.py --superpixel_method slic --n_slic_segments 200 --n_clusters 2
        --resize_shape H W --batchsize 30 --img_file_list ... --label_file_list ... --out_dir ... --no_figure

followed by `utils/mean_result.py result.json`.  Everything the reference does per image is inside the measured
region: PNG decode, DRN forward, superpixels, pooling, k-means, ground-truth decode, nearest resize, two .npy
files, the confusion counts and the result.json line.  Reported:

    wall_images_per_s      N / wall time of the child process (includes interpreter start, model creation and
                           MIOpen's solver search for the first batch)
    steady_images_per_s    images of batches 2.. / time from the first to the last result.json line (the steady
                           state of a long run: train_extra is 20 k images)
    decode_images_per_s_per_core, io_threads: what the input stage needs per GPU

One JSON line on stdout (also written to --out).  This tool never imports the product: it times the scripts a
user of the reference would run.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_one(job):
    i, d, H, W = job
    import importlib
    import numpy as np
    from PIL import Image
    sys.path.insert(0, ROOT)
    synth = importlib.import_module('superpixel-align_amd.synth')
    img = synth.synth_image(5000 + i, H, W, integer_valued=True).astype(np.uint8)
    fn = os.path.join(d, 'leftImg8bit', 'synth', 'synth_%06d_000019_leftImg8bit.png' % i)
    Image.fromarray(img.transpose(1, 2, 0)).save(fn, compress_level=1)
    lf = os.path.join(d, 'gtFine', 'synth', 'synth_%06d_000019_gtFine_labelIds.png' % i)
    Image.fromarray(synth.synth_gt_labels(5000 + i, H, W)).save(lf, compress_level=1)
    return fn, lf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=300)
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--batchsize', type=int, default=30)
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--arch', default='drn_d_22')
    ap.add_argument('--io_threads', type=int, default=16)
    ap.add_argument('--decode_procs', type=int, default=0, help='decode PNGs in this many worker processes (0: threads)')
    ap.add_argument('--n_slic_segments', type=int, default=200)
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--out', default=None)
    ap.add_argument('--extra', nargs='*', default=[], help='extra flags for the driver (e.g. --pool_mode anchor)')
    a = ap.parse_args()

    d = tempfile.mkdtemp(prefix='spa_driver300_')
    for sub in ('leftImg8bit/synth', 'gtFine/synth', 'out'):
        os.makedirs(os.path.join(d, sub))
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
        files = list(ex.map(_make_one, [(i, d, a.height, a.width) for i in range(a.n)], chunksize=2))
    t_gen = time.time() - t0
    with open(os.path.join(d, 'imgs.txt'), 'w') as f:
        f.write('\n'.join(x[0] for x in files) + '\n')
    with open(os.path.join(d, 'labs.txt'), 'w') as f:
        f.write('\n'.join(x[1] for x in files) + '\n')
    png_mb = sum(os.path.getsize(x[0]) for x in files) / 1e6

    # decode rate of ONE core (what the input stage is made of)
    from PIL import Image
    import numpy as np
    t0 = time.time()
    for x in files[:8]:
        with Image.open(x[0]) as im:
            np.asarray(im, dtype=np.uint8)
    decode_rate = 8 / (time.time() - t0)

    out_dir = os.path.join(d, 'out')
    cmd = [sys.executable, os.path.join(ROOT, 'batch_spalign_kmeans.py'),
           '--superpixel_method', 'slic', '--n_slic_segments', str(a.n_slic_segments), '--n_clusters', '2',
           '--resize_shape', str(a.height), str(a.width), '--batchsize', str(a.batchsize),
           '--img_file_list', os.path.join(d, 'imgs.txt'), '--label_file_list', os.path.join(d, 'labs.txt'),
           '--out_dir', out_dir, '--start_index', '0', '--end_index', str(a.n), '--no_figure',
           '--arch', a.arch, '--dtype', a.dtype, '--pool_mode', 'mean', '--io_threads', str(a.io_threads),
           '--decode_procs', str(a.decode_procs)] + a.extra
    t0 = time.time()
    with open(os.path.join(d, 'driver.log'), 'w') as log:
        rc = subprocess.call(cmd, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
    wall = time.time() - t0
    res = os.path.join(out_dir, 'result.json')
    lines = [json.loads(l) for l in open(res)] if os.path.exists(res) else []
    summary = None
    if rc == 0:
        sm = subprocess.run([sys.executable, os.path.join(ROOT, 'utils', 'mean_result.py'), res],
                            capture_output=True, text=True, cwd=ROOT)
        summary = [l for l in sm.stdout.splitlines() if l.startswith(('Road mean IoU', 'Precision', 'Recall'))] if sm.returncode == 0 else ['mean_result.py failed: ' + sm.stderr[-300:]]
    # steady state from the files' own timestamps: first .npy of batch 2 .. last .npy
    npys = sorted((os.path.getmtime(os.path.join(out_dir, f)), f) for f in os.listdir(out_dir) if f.endswith('_all_cluster.npy'))
    steady = None
    if len(npys) > a.batchsize + 1:
        t_first = npys[a.batchsize - 1][0]          # last file of the first batch
        steady = (len(npys) - a.batchsize) / max(1e-9, npys[-1][0] - t_first)
    n_npy = len([f for f in os.listdir(out_dir) if f.endswith('.npy')])
    rec = {
        'tool': 'tools/driver300.py', 'rc': rc, 'n_images': a.n, 'size': [a.height, a.width],
        'batchsize': a.batchsize, 'dtype': a.dtype, 'arch': a.arch, 'io_threads': a.io_threads, 'decode_procs': a.decode_procs,
        'command': ' '.join(os.path.basename(c) if c.startswith(ROOT) else c for c in cmd[1:]).replace(d, '$D'),
        'wall_s': round(wall, 2), 'wall_images_per_s': round(a.n / wall, 2),
        'steady_images_per_s': round(steady, 2) if steady else None,
        'result_json_lines': len(lines), 'npy_files': n_npy, 'png_megabytes': round(png_mb, 1),
        'decode_images_per_s_per_core': round(decode_rate, 1), 'host_cores': os.cpu_count(),
        'generation_s': round(t_gen, 1),
        'mean_result': summary,
        'per_batch_elapsed_s': sorted({round(l['elapsed_time'], 3) for l in lines})[:3] if lines else None,
        'stage_s_per_batch_median': {k: round(float(np.median([l[k] for l in lines])), 4)
                                     for k in ('time_feature_maps', 'time_superpixel', 'time_roialign', 'time_kmeans')
                                     if lines and k in lines[0]},
    }
    print(json.dumps(rec))
    if a.out:
        with open(a.out, 'w') as f:
            f.write(json.dumps(rec) + '\n')
        if rc != 0 or os.environ.get('SPA_DRIVER_TRACE') == '1':
            shutil.copy(os.path.join(d, 'driver.log'), a.out + '.log')
    if not a.keep:
        shutil.rmtree(d, ignore_errors=True)
    return rc


if __name__ == '__main__':
    sys.exit(main())
