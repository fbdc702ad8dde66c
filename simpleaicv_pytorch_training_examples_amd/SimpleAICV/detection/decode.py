"""Evaluation-time decoding (reference SimpleAICV/detection/decode.py): DetNMSMethod (:25-103), DecodeMethod (:106-171),
RetinaDecoder (:174-270) and FCOSDecoder (:273-363) at the end of this file, and DETRDecoder (:366-470): softmax over
the last decoder layer's class logits, arg-max class, drop "no object" and low scores, boxes cxcywh -> xyxy scaled to each
image's (scaled) size, top-n by score into fixed [B, max_object_num] arrays padded with -1 / 0.  Host code on [B, 100]
arrays, as in the reference (the NMS variants the reference offers are not used by any DETR config)."""
import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    'RetinaDecoder',
    'FCOSDecoder',
    'DETRDecoder',
]


class DETRDecoder:

    def __init__(self, num_classes=80, max_object_num=100, min_score_threshold=0.05, topn=100, nms_type=None,
                 nms_threshold=0.5):
        if nms_type:
            raise NotImplementedError('DETR decoding does not use NMS in any reference config')
        self.num_classes, self.max_object_num = num_classes, max_object_num
        self.min_score_threshold, self.topn, self.nms_type = min_score_threshold, topn, nms_type

    def __call__(self, preds, scaled_sizes):
        probs = F.softmax(preds[0][-1].float(), dim=2).cpu().detach().numpy()         # [B, Q, classes + 1]
        boxes = preds[1][-1].float().cpu().detach().numpy()                           # [B, Q, 4] cxcywh in 0..1
        b = probs.shape[0]
        batch_scores = -np.ones((b, self.max_object_num), dtype=np.float32)
        batch_classes = -np.ones((b, self.max_object_num), dtype=np.float32)
        batch_bboxes = np.zeros((b, self.max_object_num, 4), dtype=np.float32)
        for i in range(b):
            classes = np.argmax(probs[i], axis=1)
            scores = probs[i][np.arange(probs.shape[1]), classes]
            h, w = scaled_sizes[i][0], scaled_sizes[i][1]
            cx, cy, bw, bh = boxes[i][:, 0], boxes[i][:, 1], boxes[i][:, 2], boxes[i][:, 3]
            xyxy = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], axis=1) * np.array([[w, h, w, h]], np.float32)
            keep = (classes < self.num_classes) & (scores > self.min_score_threshold)
            scores, classes, xyxy = scores[keep], classes[keep].astype(np.float32), xyxy[keep].astype(np.float32)
            if scores.shape[0] == 0:
                continue
            order = np.argsort(-scores, kind='stable')[:self.topn]       # equal scores keep their candidate (anchor) order
            n = min(self.max_object_num, order.shape[0])
            batch_scores[i, :n] = scores[order][:n]
            batch_classes[i, :n] = classes[order][:n]
            batch_bboxes[i, :n] = xyxy[order][:n]
        return batch_scores, batch_classes, batch_bboxes


# ---------------------------------------------------------------------------------------------- dense detectors
class DetNMSMethod:
    """Greedy non-maximum suppression of score-sorted boxes on the host (reference :25-103): a box survives while its IoU -- for
    'diou_python_nms' IoU minus the squared centre distance over the squared enclosing diagonal -- with every kept box stays BELOW
    the threshold.  One [n, n] overlap matrix, then a scan over the kept boxes; n <= topn (1000).  'torch_nms' (torchvision's
    kernel in the reference: suppress above the threshold) runs the same scan with `<=`."""

    def __init__(self, nms_type='python_nms', nms_threshold=0.5):
        assert nms_type in ['torch_nms', 'python_nms', 'diou_python_nms'], 'wrong nms type!'
        self.nms_type = nms_type
        self.nms_threshold = nms_threshold

    def __call__(self, sorted_bboxes, sorted_scores):
        boxes = np.asarray(sorted_bboxes, dtype=np.float32)
        n = boxes.shape[0]
        if n == 0:
            return np.array([], dtype=np.int32)
        wh = boxes[:, 2:4] - boxes[:, 0:2]
        areas = np.maximum(wh[:, 0] * wh[:, 1], 0)
        lo = np.maximum(boxes[:, None, 0:2], boxes[None, :, 0:2])
        hi = np.minimum(boxes[:, None, 2:4], boxes[None, :, 2:4])
        inter_wh = np.maximum(hi - lo, 0)
        inter = inter_wh[..., 0] * inter_wh[..., 1]
        union = np.maximum(areas[:, None] + areas[None, :] - inter, 1e-4)
        measure = inter / union
        if self.nms_type == 'diou_python_nms':
            hull = np.maximum(np.maximum(boxes[:, None, 2:4], boxes[None, :, 2:4]) - np.minimum(boxes[:, None, 0:2], boxes[None, :, 0:2]), 0)
            c2 = np.maximum((hull ** 2).sum(axis=2), 1e-4)
            centres = (boxes[:, 2:4] + boxes[:, 0:2]) / 2
            p2 = ((centres[:, None, :] - centres[None, :, :]) ** 2).sum(axis=2)
            measure = measure - p2 / c2
        alive = np.ones(n, dtype=bool)
        keep = []
        for i in range(n):
            if not alive[i]:
                continue
            keep.append(i)
            if self.nms_type == 'torch_nms':
                alive[i + 1:] &= measure[i, i + 1:] <= self.nms_threshold
            else:
                alive[i + 1:] &= measure[i, i + 1:] < self.nms_threshold
        return np.array(keep, dtype=np.int32)


class DecodeMethod:
    """score filter -> descending sort -> top-n -> NMS -> the first max_object_num detections into fixed arrays padded with -1 / 0
    (reference :106-171).  Works on whatever candidate rows it is handed: all anchors (as in the reference) or the device-side
    pre-selection of the decoders below."""

    def __init__(self, max_object_num=100, min_score_threshold=0.05, topn=1000, nms_type='python_nms', nms_threshold=0.5):
        self.max_object_num = max_object_num
        self.min_score_threshold = min_score_threshold
        self.topn = topn
        self.nms_function = DetNMSMethod(nms_type=nms_type, nms_threshold=nms_threshold)

    def __call__(self, cls_scores, cls_classes, pred_bboxes):
        b = len(cls_scores)
        batch_scores = -np.ones((b, self.max_object_num), dtype=np.float32)
        batch_classes = -np.ones((b, self.max_object_num), dtype=np.float32)
        batch_bboxes = np.zeros((b, self.max_object_num, 4), dtype=np.float32)
        for i in range(b):
            scores = np.asarray(cls_scores[i])
            live = scores > self.min_score_threshold
            scores = scores[live].astype(np.float32)
            if scores.shape[0] == 0:
                continue
            classes = np.asarray(cls_classes[i])[live].astype(np.float32)
            boxes = np.asarray(pred_bboxes[i])[live].astype(np.float32)
            order = np.argsort(-scores)[:self.topn]
            scores, classes, boxes = scores[order], classes[order], boxes[order]
            keep = self.nms_function(boxes, scores)[:self.max_object_num]
            n = keep.shape[0]
            batch_scores[i, :n], batch_classes[i, :n], batch_bboxes[i, :n] = scores[keep], classes[keep], boxes[keep]
        return [batch_scores, batch_classes, batch_bboxes]


class _DenseDecoder:
    """Shared front of the two decoders.  The reference copies every level's [B, H, W, (anchors,) classes] probabilities to the
    host and takes the arg-max there; here csrc/detloss.hip reduces each level to per-anchor (score, class) where the head wrote
    it, the threshold and the top-n selection run on the device, and only the <= topn surviving candidates per image (score,
    class, four regression values, table row) travel to the host for box decoding and NMS -- which then run on the reference's
    own numpy arithmetic, so the detections are the reference's."""

    def _candidates(self, cls_levels, center_levels, reg_levels, reg_dim):
        from ... import _lib
        L, st = _lib.lib(), _lib.stream()
        b = cls_levels[0].shape[0]
        cls = [t.detach().reshape(b, -1, t.shape[-1]).float().contiguous() for t in cls_levels]
        total = sum(t.shape[1] for t in cls)
        dev = cls[0].device
        scores = torch.empty((b, total), dtype=torch.float32, device=dev)
        classes = torch.empty((b, total), dtype=torch.int32, device=dev)
        off = 0
        for i, t in enumerate(cls):
            ctr = center_levels[i].detach().reshape(b, -1).float().contiguous() if center_levels is not None else None
            _lib.check(L.saicv_det_best_class(t.data_ptr(), ctr.data_ptr() if ctr is not None else None, scores.data_ptr(), classes.data_ptr(),
                                              b, t.shape[1], total, off, t.shape[2], st), 'det_best_class')
            off += t.shape[1]
        reg = torch.cat([t.detach().reshape(b, -1, reg_dim).float() for t in reg_levels], dim=1)
        dm = self.decode_function
        k = min(dm.topn, total)
        masked = torch.where(scores > dm.min_score_threshold, scores, torch.full_like(scores, float('-inf')))
        # ties: (score descending, anchor index ascending) -- a STABLE device sort, then the first k.  torch.topk picks arbitrary
        # members of a tie at the cut (bf16 heads and saturated sigmoids produce many equal scores), which changed which boxes
        # reached NMS from run to run.  The reference argsorts all anchors with numpy's default (unstable) sort, so its own order
        # inside a tie is an accident of that array; lowest anchor index first is the deterministic reading of it.
        order = torch.sort(masked, dim=1, descending=True, stable=True)
        top_scores, top_idx = order.values[:, :k], order.indices[:, :k]
        top_classes = torch.gather(classes, 1, top_idx)
        top_reg = torch.gather(reg, 1, top_idx.unsqueeze(-1).expand(-1, -1, reg_dim))
        top_scores, top_idx, top_classes, top_reg = (t.cpu().numpy() for t in (top_scores, top_idx, top_classes, top_reg))
        out = []
        for i in range(b):
            n = int(np.isfinite(top_scores[i]).sum())
            out.append((top_scores[i, :n], top_classes[i, :n], top_reg[i, :n], top_idx[i, :n]))
        return out


class RetinaDecoder(_DenseDecoder):

    def __init__(self, areas=[[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]], ratios=[0.5, 1, 2],
                 scales=[2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)], strides=[8, 16, 32, 64, 128], max_object_num=100, min_score_threshold=0.05,
                 topn=1000, nms_type='python_nms', nms_threshold=0.5):
        assert nms_type in ['torch_nms', 'python_nms', 'diou_python_nms'], 'wrong nms type!'
        from .models.anchor import RetinaAnchors
        self.anchors = RetinaAnchors(areas=areas, ratios=ratios, scales=scales, strides=strides)
        self.decode_function = DecodeMethod(max_object_num=max_object_num, min_score_threshold=min_score_threshold, topn=topn,
                                            nms_type=nms_type, nms_threshold=nms_threshold)

    def __call__(self, preds):
        cls_preds, reg_preds = preds
        sizes = [[t.shape[2], t.shape[1]] for t in cls_preds]
        table = np.concatenate([a.reshape(-1, 4) for a in self.anchors(sizes)], axis=0)
        cands = self._candidates(cls_preds, None, reg_preds, 4)
        scores = [c[0] for c in cands]
        classes = [c[1] for c in cands]
        boxes = [self.snap_txtytwth_to_x1y1x2y2(c[2][None], table[c[3]][None])[0] for c in cands]
        return self.decode_function(scores, classes, boxes)

    def snap_txtytwth_to_x1y1x2y2(self, reg_preds, anchors):
        """[B, n, 4] offsets on [B, n, 4] anchors -> integer-truncated xyxy boxes (reference :252-270)"""
        wh = anchors[:, :, 2:4] - anchors[:, :, 0:2]
        centre = anchors[:, :, 0:2] + 0.5 * wh
        box_wh = np.exp(reg_preds[:, :, 2:4]) * wh
        box_centre = reg_preds[:, :, :2] * wh + centre
        return np.concatenate([box_centre - 0.5 * box_wh, box_centre + 0.5 * box_wh], axis=2).astype(np.int32)


class FCOSDecoder(_DenseDecoder):

    def __init__(self, strides=[8, 16, 32, 64, 128], max_object_num=100, min_score_threshold=0.05, topn=1000, nms_type='python_nms',
                 nms_threshold=0.6):
        assert nms_type in ['torch_nms', 'python_nms', 'diou_python_nms'], 'wrong nms type!'
        from .models.anchor import FCOSPositions
        self.positions = FCOSPositions(strides=strides)
        self.decode_function = DecodeMethod(max_object_num=max_object_num, min_score_threshold=min_score_threshold, topn=topn,
                                            nms_type=nms_type, nms_threshold=nms_threshold)

    def __call__(self, preds):
        cls_preds, reg_preds, center_preds = preds
        sizes = [[t.shape[2], t.shape[1]] for t in cls_preds]
        table = np.concatenate([p.reshape(-1, 2) for p in self.positions(sizes)], axis=0)
        cands = self._candidates(cls_preds, center_preds, reg_preds, 4)
        scores = [c[0] for c in cands]
        classes = [c[1] for c in cands]
        boxes = [self.snap_ltrb_to_x1y1x2y2(c[2][None], table[c[3]][None])[0] for c in cands]
        return self.decode_function(scores, classes, boxes)

    def snap_ltrb_to_x1y1x2y2(self, reg_preds, points_position):
        """[B, n, 4] log-distances at [B, n, 2] points -> integer-truncated xyxy boxes (reference :349-363)"""
        dist = np.exp(reg_preds)
        return np.concatenate([points_position - dist[:, :, 0:2], points_position + dist[:, :, 2:4]], axis=2).astype(np.int32)
