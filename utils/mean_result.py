#!/usr/bin/env python
"""Aggregate result.json (JSON lines written by batch_spalign_kmeans.py) into the Road-IoU
summary the reference reports (README "Road IoU", utils/mean_result.py of the reference):
optional de-duplication by img_fn (first occurrence wins), optional truncation to the first
--n_imgs records, nan-aware means, pooled precision/recall, summary.txt next to the input."""
import argparse
import json
import os

import numpy as np


def summarise(path, n_imgs=None, count_duplicated=False, show_failed_fn=False):
    seen, rows = {}, []
    with open(path) as fp:
        for raw in fp:
            raw = raw.strip()
            if not raw:
                continue
            d = json.loads(raw)
            if not count_duplicated:
                if d['img_fn'] in seen:
                    continue
                seen[d['img_fn']] = d['road_iou']
            if show_failed_fn and d['road_iou'] == 0:
                print(d['img_fn'])
            rows.append(d)
    if n_imgs is not None:
        rows = rows[:n_imgs]
    col = lambda k: np.array([np.nan if r[k] is None else r[k] for r in rows], dtype=np.float64)
    riou, niou, prec, rec = col('road_iou'), col('non_road_iou'), col('precision'), col('recall')
    prec[prec == 0] = np.nan          # the reference treats a falsy precision/recall as missing
    rec[rec == 0] = np.nan
    tp, fp_, fn = col('TP').sum(), col('FP').sum(), col('FN').sum()
    out = [('Road mean IoU', np.nanmean(riou)), ('Road min IoU', np.nanmin(riou)),
           ('Road max IoU', np.nanmax(riou)), ('Non-road mean IoU', np.nanmean(niou)),
           ('Non-road min IoU', np.nanmin(niou)), ('Non-road max IoU', np.nanmax(niou)),
           ('Average Precision', np.nanmean(prec)), ('Precision', tp / (tp + fp_)),
           ('Min Precision', np.nanmin(prec)), ('Max Precision', np.nanmax(prec)), ('N', len(prec)),
           ('Average Recall', np.nanmean(rec)), ('Recall', tp / (tp + fn)),
           ('Min Recall', np.nanmin(rec)), ('Max Recall', np.nanmax(rec)), ('N', len(rec))]
    msg = ''.join('%s\t:%s\n' % kv for kv in out) + '\n'
    for fn_, iou in sorted(seen.items(), key=lambda kv: kv[1], reverse=True)[:10]:
        msg += '%s\t%s\n' % (iou, fn_)
    return dict(out[:1] + out[6:8] + out[11:13]), msg


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('result_json', type=str)
    ap.add_argument('--show_failed_fn', action='store_true', default=False)
    ap.add_argument('--count_duplicated', action='store_true', default=False)
    ap.add_argument('--n_imgs', type=int, default=None)
    a = ap.parse_args()
    _, text = summarise(a.result_json, a.n_imgs, a.count_duplicated, a.show_failed_fn)
    print(a.result_json)
    print(text)
    with open(os.path.join(os.path.dirname(a.result_json), 'summary.txt'), 'w') as fp:
        print(text, file=fp)
