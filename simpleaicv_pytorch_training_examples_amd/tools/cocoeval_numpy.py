"""COCO bounding-box evaluation in numpy (no pycocotools: the image has none, and the evaluation is host arithmetic on a few
thousand boxes).  Replaces `COCOeval(coco_true, coco_pred, 'bbox')` + evaluate / accumulate / summarize in the reference's
`evaluate_coco_detection` (tools/scripts.py:742-881) with the published protocol of the COCO API (cocoeval.py, Lin et al. /
pycocotools 2.0, BSD): parity unpinned against the library itself (it is absent here) -- pinned instead on hand-worked cases and the
protocol's invariants (tests/test_r04_host.py).

Protocol, per (image, category):
  * detections sorted by score (stable, descending), cut at 100; ground truth sorted with `ignore` ones last;
    ignore = iscrowd or area outside the area range;
  * IoU in xywh; against a crowd box the union is the detection's own area;
  * per IoU threshold t in 0.50:0.05:0.95, every detection (in score order) takes the unmatched ground truth with the highest
    IoU >= t -- a regular one if any qualifies, else an ignored one; crowd boxes may be matched repeatedly; a detection matched to
    an ignored box, or unmatched with its own area outside the range, is itself ignored;
then per (category, area range, max detections): detections of all images merged by score (stable), cumulative TP / FP over the
non-ignored ones, precision made monotone from the right and sampled at recall 0:0.01:1; recall = final TP / number of regular
ground-truth boxes; the twelve summary numbers are means over all entries > -1."""
import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0.0, 1e10), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e10))      # all, small, medium, large
STAT_NAMES = ('IoU=0.50:0.95,area=all,maxDets=100,mAP', 'IoU=0.50,area=all,maxDets=100,mAP', 'IoU=0.75,area=all,maxDets=100,mAP',
              'IoU=0.50:0.95,area=small,maxDets=100,mAP', 'IoU=0.50:0.95,area=medium,maxDets=100,mAP',
              'IoU=0.50:0.95,area=large,maxDets=100,mAP', 'IoU=0.50:0.95,area=all,maxDets=1,mAR', 'IoU=0.50:0.95,area=all,maxDets=10,mAR',
              'IoU=0.50:0.95,area=all,maxDets=100,mAR', 'IoU=0.50:0.95,area=small,maxDets=100,mAR',
              'IoU=0.50:0.95,area=medium,maxDets=100,mAR', 'IoU=0.50:0.95,area=large,maxDets=100,mAR')


def xywh_iou(dt, gt, iscrowd):
    """[D, 4] x [G, 4] boxes as (x, y, w, h) -> IoU [D, G]; for a crowd ground truth the union is the detection's area."""
    dt, gt = np.asarray(dt, dtype=np.float64).reshape(-1, 4), np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    if dt.shape[0] == 0 or gt.shape[0] == 0:
        return np.zeros((dt.shape[0], gt.shape[0]))
    ix = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    iy = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(ix, 0, None) * np.clip(iy, 0, None)
    da, ga = (dt[:, 2] * dt[:, 3])[:, None], (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(iscrowd, dtype=bool)[None, :], da, da + ga - inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(union > 0, inter / union, 0.0)


def _evaluate_image(dt_boxes, dt_scores, gt_boxes, gt_crowd, gt_area, area_rng, max_det):
    """-> (scores [D], matched [T, D] bool, dt_ignore [T, D] bool, number of regular ground-truth boxes) or None"""
    if len(gt_boxes) == 0 and len(dt_boxes) == 0:
        return None
    gt_ignore = np.asarray(gt_crowd, dtype=bool) | (gt_area < area_rng[0]) | (gt_area > area_rng[1])
    gorder = np.argsort(gt_ignore, kind='mergesort')
    gt_boxes, gt_crowd, gt_ignore = gt_boxes[gorder], np.asarray(gt_crowd, dtype=bool)[gorder], gt_ignore[gorder]
    dorder = np.argsort(-dt_scores, kind='mergesort')[:max_det]
    dt_boxes, dt_scores = dt_boxes[dorder], dt_scores[dorder]
    ious = xywh_iou(dt_boxes, gt_boxes, gt_crowd)
    T, D, G = len(IOU_THRS), len(dt_boxes), len(gt_boxes)
    gtm = np.zeros((T, G), dtype=bool)
    dtm = np.zeros((T, D), dtype=bool)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(IOU_THRS):
        for d in range(D):
            best, m = min(t, 1 - 1e-10), -1
            for g in range(G):
                if gtm[ti, g] and not gt_crowd[g]:
                    continue
                if m > -1 and not gt_ignore[m] and gt_ignore[g]:
                    break                       # a regular match is in hand and only ignored boxes follow
                if ious[d, g] < best:
                    continue
                best, m = ious[d, g], g
            if m == -1:
                continue
            dt_ig[ti, d] = gt_ignore[m]
            dtm[ti, d] = True
            gtm[ti, m] = True
    d_area = dt_boxes[:, 2] * dt_boxes[:, 3] if D else np.zeros(0)
    outside = (d_area < area_rng[0]) | (d_area > area_rng[1])
    dt_ig = dt_ig | (~dtm & outside[None, :])
    return dt_scores, dtm, dt_ig, int((~gt_ignore).sum())


def evaluate_bbox(gts, dts, image_ids=None, category_ids=None):
    """gts / dts: lists of dicts with 'image_id', 'category_id', 'bbox' = [x, y, w, h] (+ 'iscrowd', 'area' for ground truth,
    'score' for detections).  -> (stats [12] in [0, 1] as COCOeval.stats, precision [T, R, K, A, M], recall [T, K, A, M])."""
    image_ids = sorted(set(image_ids if image_ids is not None else [g['image_id'] for g in gts] + [d['image_id'] for d in dts]))
    category_ids = sorted(set(category_ids if category_ids is not None else [g['category_id'] for g in gts]))
    by_gt, by_dt = {}, {}
    for g in gts:
        by_gt.setdefault((g['image_id'], g['category_id']), []).append(g)
    for d in dts:
        by_dt.setdefault((d['image_id'], d['category_id']), []).append(d)
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(category_ids), len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for ki, cat in enumerate(category_ids):
        per_img = {}
        for img in image_ids:
            g, d = by_gt.get((img, cat), []), by_dt.get((img, cat), [])
            if not g and not d:
                continue
            gb = np.array([x['bbox'] for x in g], dtype=np.float64).reshape(-1, 4)
            per_img[img] = (np.array([x['bbox'] for x in d], dtype=np.float64).reshape(-1, 4),
                            np.array([x['score'] for x in d], dtype=np.float64), gb,
                            np.array([x.get('iscrowd', 0) for x in g], dtype=bool),
                            np.array([x.get('area', b[2] * b[3]) for x, b in zip(g, gb)], dtype=np.float64))
        for ai, rng in enumerate(AREA_RNG):
            evals = [e for e in (_evaluate_image(*per_img[img], rng, MAX_DETS[-1]) for img in image_ids if img in per_img) if e]
            if not evals:
                continue
            npig = sum(e[3] for e in evals)
            if npig == 0:
                continue
            for mi, md in enumerate(MAX_DETS):
                scores = np.concatenate([e[0][:md] for e in evals])
                order = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e[1][:, :md] for e in evals], axis=1)[:, order]
                dig = np.concatenate([e[2][:, :md] for e in evals], axis=1)[:, order]
                tps = np.cumsum(dtm & ~dig, axis=1, dtype=np.float64)
                fps = np.cumsum(~dtm & ~dig, axis=1, dtype=np.float64)
                for ti in range(T):
                    tp, fp = tps[ti], fps[ti]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    recall[ti, ki, ai, mi] = rc[-1] if nd else 0
                    pr = np.maximum.accumulate(pr[::-1])[::-1] if nd else pr
                    inds = np.searchsorted(rc, REC_THRS, side='left')
                    q = np.zeros(R)
                    ok = inds < nd
                    q[ok] = pr[inds[ok]]
                    precision[ti, :, ki, ai, mi] = q

    def summ(ap, iou=None, area=0, md=2):
        s = precision[:, :, :, area, md] if ap else recall[:, :, area, md]
        if iou is not None:
            s = s[np.isclose(IOU_THRS, iou)]
        s = s[s > -1]
        return float(s.mean()) if s.size else -1.0

    stats = np.array([summ(1), summ(1, 0.5), summ(1, 0.75), summ(1, area=1), summ(1, area=2), summ(1, area=3),
                      summ(0, md=0), summ(0, md=1), summ(0, md=2), summ(0, area=1), summ(0, area=2), summ(0, area=3)])
    return stats, precision, recall
