"""Evaluation-time decoding of DETR outputs (reference SimpleAICV/detection/decode.py:366-470 DETRDecoder): softmax over
the last decoder layer's class logits, arg-max class, drop "no object" and low scores, boxes cxcywh -> xyxy scaled to each
image's (scaled) size, top-n by score into fixed [B, max_object_num] arrays padded with -1 / 0.  Host code on [B, 100]
arrays, as in the reference (the NMS variants the reference offers are not used by any DETR config)."""
import numpy as np
import torch.nn.functional as F

__all__ = [
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
            order = np.argsort(-scores)[:self.topn]
            n = min(self.max_object_num, order.shape[0])
            batch_scores[i, :n] = scores[order][:n]
            batch_classes[i, :n] = classes[order][:n]
            batch_bboxes[i, :n] = xyxy[order][:n]
        return batch_scores, batch_classes, batch_bboxes
