"""Detection losses -- drop-ins for reference SimpleAICV/detection/losses.py: RetinaLoss (:123-433) and FCOSLoss (:434-842), both at
the end of this file on the kernels of csrc/detloss.hip, and DETRLoss (:843-1095):

Same constructor arguments and output keys (`layer_{i}_cls_loss`, `layer_{i}_box_l1_loss`, `layer_{i}_box_iou_loss`
for the 6 decoder layers).  Semantics kept: boxes clamped to [1e-4, 1 - 1e-4]; ONE Hungarian matching per image on
the LAST layer's outputs (cost = 1 * (-p[class]) + 5 * L1 + 2 * (-GIoU), probabilities clamped like the boxes) reused
by every layer; weighted cross-entropy with the no-object class at 0.1; L1 and (1 - GIoU) summed over matched pairs
and divided by the number of ground-truth boxes in the batch.  The assignment itself runs on the host with scipy,
as in the reference (one [100, n_i] matrix per image); everything else is [B, 100, *] tensor arithmetic in fp32.
"""
import math

import numpy as np
import scipy.optimize
import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = [
    'RetinaLoss',
    'FCOSLoss',
    'DETRLoss',
]


def _cxcywh_to_xyxy(boxes):
    cx, cy, w, h = boxes.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def _giou(b1, b2):
    """Generalised IoU of broadcastable [..., 4] xyxy boxes, with the reference's clamps (areas >= 0,
    union and enclosing area >= 1e-4)."""
    area1 = ((b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])).clamp(min=0)
    area2 = ((b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])).clamp(min=0)
    wh = (torch.min(b1[..., 2:], b2[..., 2:]) - torch.max(b1[..., :2], b2[..., :2])).clamp(min=0)
    inter = (wh[..., 0] * wh[..., 1]).clamp(min=0)
    union = (area1 + area2 - inter).clamp(min=1e-4)
    ewh = (torch.max(b1[..., 2:], b2[..., 2:]) - torch.min(b1[..., :2], b2[..., :2])).clamp(min=0)
    enclose = (ewh[..., 0] * ewh[..., 1]).clamp(min=1e-4)
    return inter / union - (enclose - union) / enclose


class _DetrBoxLossFn(torch.autograd.Function):
    """(l1 [L], iou [L]) of DETRLoss.forward_static from the raw regression outputs and the static pair buffers in one kernel each way
    (saicv_detr_box_loss_fwd / _bwd, csrc/detloss.hip): the torch formulation below it in forward_static -- gather, where, the
    cxcywh -> xyxy conversions, _giou, the weighted sums -- was ~55 launches forward and ~130 backward on 4 800 pairs."""

    @staticmethod
    def forward(ctx, reg_preds, gt, src, tgt, w, lo, hi):
        from ... import _lib
        reg = reg_preds.float().contiguous()
        gt = gt.float().contiguous()
        l, b, q = reg.shape[0], reg.shape[1], reg.shape[2]
        t = src.shape[1]
        out = torch.empty(2 * l + 1, dtype=torch.float32, device=reg.device)
        _lib.check(_lib.lib().saicv_detr_box_loss_fwd(_lib.ptr(reg), _lib.ptr(gt), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(w), l, b, q, t,
                                                      float(lo), float(hi), _lib.ptr(out), _lib.stream()), 'detr_box_loss_fwd')
        ctx.save_for_backward(reg, gt, src, tgt, w, out)
        ctx.cfg = (l, b, q, t, float(lo), float(hi), reg_preds.dtype)
        ctx.in_shape = reg_preds.shape
        return out[:l], out[l:2 * l]

    @staticmethod
    def backward(ctx, d_l1, d_iou):
        from ... import _lib
        reg, gt, src, tgt, w, out = ctx.saved_tensors
        l, b, q, t, lo, hi, dtype = ctx.cfg
        d_l1 = None if d_l1 is None else d_l1.float().contiguous()
        d_iou = None if d_iou is None else d_iou.float().contiguous()
        dreg = torch.empty_like(reg)
        _lib.check(_lib.lib().saicv_detr_box_loss_bwd(_lib.ptr(reg), _lib.ptr(gt), _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(w),
                                                      _lib.ptr(d_l1) if d_l1 is not None else None, _lib.ptr(d_iou) if d_iou is not None else None,
                                                      _lib.ptr(out), l, b, q, t, lo, hi, _lib.ptr(dreg), _lib.stream()), 'detr_box_loss_bwd')
        dreg = dreg.view(ctx.in_shape)
        return (dreg if dtype == torch.float32 else dreg.to(dtype)), None, None, None, None, None, None


class DETRLoss(nn.Module):

    def __init__(self, cls_match_cost=1.0, box_match_cost=5.0, giou_match_cost=2.0, cls_loss_weight=1.0,
                 box_l1_loss_weight=5.0, iou_loss_weight=2.0, no_object_cls_weight=0.1, num_classes=80):
        super(DETRLoss, self).__init__()
        self.cls_match_cost = cls_match_cost
        self.box_match_cost = box_match_cost
        self.giou_match_cost = giou_match_cost
        self.cls_loss_weight = cls_loss_weight
        self.box_l1_loss_weight = box_l1_loss_weight
        self.iou_loss_weight = iou_loss_weight
        self.no_object_cls_weight = no_object_cls_weight
        self.num_classes = num_classes
        assert self.cls_match_cost != 0 or self.box_match_cost != 0 or self.giou_match_cost != 0, "all costs cant be 0"

    def forward(self, preds, annotations):
        cls_preds, reg_preds = preds
        raw_reg_preds = reg_preds
        reg_preds = torch.clamp(reg_preds, min=1e-4, max=1. - 1e-4).float()
        cls_preds = cls_preds.float()
        gt, counts = self._valid_targets(annotations)
        indices = self._match(cls_preds[-1], reg_preds[-1], gt, counts)
        # flat (image, query) <-> ground-truth correspondence, shared by all layers; built on the host, one upload
        offs = np.concatenate([[0], np.cumsum(counts)[:-1]]) if counts else np.zeros(0, dtype=np.int64)
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)]).to(cls_preds.device, non_blocking=True)
        src_idx = torch.cat([src for src, _ in indices]).to(cls_preds.device, non_blocking=True)
        rows = torch.cat([j + int(o) for (_, j), o in zip(indices, offs)]).to(cls_preds.device, non_blocking=True)
        matched = gt[rows]                                                                       # [n, 5]
        target_num = int(sum(counts))
        # all decoder layers at once ([L, B, Q, *] tensors): the per-layer arithmetic of the reference's loop
        # (losses.py:905-935) in one pass -- six times fewer tiny launches on a path that is host-bound
        cls_l = self._cls_loss_layers(cls_preds, batch_idx, src_idx, matched[:, 4])
        if raw_reg_preds.is_cuda and target_num > 0:
            # the kernel of the static form (saicv_detr_box_loss_*), fed the flat pair list as ONE image of B * Q queries: the eager and the
            # captured loop then share the per-pair arithmetic bit for bit (they differ in the order of the loss sums only)
            l, b, q = cls_preds.shape[0], cls_preds.shape[1], cls_preds.shape[2]
            n = matched.shape[0]
            l1_l, iou_l = _DetrBoxLossFn.apply(raw_reg_preds.reshape(l, 1, b * q, 4), matched.view(1, n, 5), (batch_idx * q + src_idx).view(1, n),
                                               torch.arange(n, device=matched.device).view(1, n),
                                               torch.ones(1, n, dtype=torch.float32, device=matched.device), 1e-4, 1. - 1e-4)
        else:
            l1_l, iou_l = self._box_losses_layers(reg_preds[:, batch_idx, src_idx], matched[:, 0:4], target_num)
        return self._layer_terms(cls_l, l1_l, iou_l)

    def _layer_terms(self, cls_l, l1_l, iou_l):
        """[L] loss vectors -> the reference's dict of 3 L weighted scalars (losses.py:905-935).  The weights multiply the vectors and
        the scalars are unbind() views: `w * v[idx]` per entry was 18 multiplications forward and, backward, 18 more plus a zero fill, a
        copy and an accumulation per select -- same values, ~100 launches fewer per step."""
        cls_w, l1_w, iou_w = self.cls_loss_weight * cls_l, self.box_l1_loss_weight * l1_l, self.iou_loss_weight * iou_l
        loss_dict = {}
        for idx, (c, a, g) in enumerate(zip(cls_w.unbind(0), l1_w.unbind(0), iou_w.unbind(0))):
            loss_dict[f'layer_{idx}_cls_loss'] = c
            loss_dict[f'layer_{idx}_box_l1_loss'] = a
            loss_dict[f'layer_{idx}_box_iou_loss'] = g
        return loss_dict

    def _cls_loss_layers(self, cls_preds, batch_idx, src_idx, target_classes):
        """Weighted cross-entropy of every layer: [L, B, Q, C+1] logits -> [L] losses (F.cross_entropy(..., weight) per layer:
        sum of w[gt] * nll over sum of w[gt])."""
        l, b, q = cls_preds.shape[0], cls_preds.shape[1], cls_preds.shape[2]
        gt = torch.full((b, q), self.num_classes, dtype=torch.long, device=cls_preds.device)
        gt[batch_idx, src_idx] = target_classes.long()
        weight = torch.ones(self.num_classes + 1, device=cls_preds.device)
        weight[-1] = self.no_object_cls_weight
        nll = -F.log_softmax(cls_preds, dim=-1).gather(-1, gt.view(1, b, q, 1).expand(l, b, q, 1)).squeeze(-1)
        wgt = weight[gt]
        return (nll * wgt).sum(dim=(1, 2)) / wgt.sum()

    def _box_losses_layers(self, matched_preds, target_boxes, target_num):
        """[L, n, 4] matched predictions against [n, 4] targets -> ([L] L1, [L] 1 - GIoU), each summed over the pairs and
        divided by the number of ground-truth boxes."""
        l1 = (matched_preds - target_boxes).abs().sum(dim=(1, 2)) / target_num
        giou = _giou(_cxcywh_to_xyxy(matched_preds), _cxcywh_to_xyxy(target_boxes))
        return l1, (1 - giou).sum(dim=1) / target_num

    def _cls_loss(self, cls_preds, batch_idx, src_idx, target_classes):
        b, q = cls_preds.shape[0], cls_preds.shape[1]
        gt = torch.full((b, q), self.num_classes, dtype=torch.long, device=cls_preds.device)
        gt[batch_idx, src_idx] = target_classes.long()
        weight = torch.ones(self.num_classes + 1, device=cls_preds.device)
        weight[-1] = self.no_object_cls_weight
        return F.cross_entropy(cls_preds.transpose(1, 2), gt, weight)

    def _box_losses(self, matched_preds, target_boxes, target_num):
        l1 = F.l1_loss(matched_preds, target_boxes, reduction='none').sum() / target_num
        giou = _giou(_cxcywh_to_xyxy(matched_preds), _cxcywh_to_xyxy(target_boxes))
        return l1, (1 - giou).sum() / target_num

    # ------------------------------------------------------------------ static-shape form (r05: the whole step as ONE captured graph)
    # The reference runs the Hungarian assignment on the host between forward and loss (scipy), which puts a device -> host copy and
    # a synchronisation into every step and keeps the step from being captured.  Here the assignment can run ON THE DEVICE
    # (assign_device: saicv_detr_assign, scipy's algorithm and tie rules restated in csrc/detloss.hip) and everything around it has a
    # fixed shape: match_inputs() -> assign_device() -> forward_static().  The ground truth is the collater's [B, T, 5] tensor itself
    # (rows with class < 0 are padding), the matched pairs are [B, T] index / weight buffers this module owns (weight 0 = no pair),
    # the number of boxes is a device scalar.  tools.scripts.train_detection takes this path with config.use_step_graph.
    static_form = True

    @torch.no_grad()
    def match_inputs(self, preds, gt_pad):
        """-> (cost [B, Q, T] fp32, valid [B, T] bool): the matching cost of the LAST layer against every row of the padded ground
        truth (the same operations as _match on the valid rows; the columns of padding rows hold garbage nobody reads)."""
        cls_preds, reg_preds = preds
        boxes = torch.clamp(reg_preds[-1], min=1e-4, max=1. - 1e-4).float()                     # [B, Q, 4]
        prob = torch.clamp(F.softmax(cls_preds[-1].float(), dim=-1), min=1e-4, max=1. - 1e-4)   # [B, Q, C + 1]
        gt = gt_pad.float()
        valid = gt[:, :, 4] >= 0
        b, q, t = boxes.shape[0], boxes.shape[1], gt.shape[1]
        cls_idx = gt[:, :, 4].clamp(min=0).long()
        cls_cost = -prob.gather(2, cls_idx[:, None, :].expand(b, q, t))
        box_cost = torch.cdist(boxes, gt[:, :, 0:4], p=1)
        giou_cost = -_giou(_cxcywh_to_xyxy(boxes)[:, :, None, :], _cxcywh_to_xyxy(gt[:, :, 0:4])[:, None, :, :])
        return self.cls_match_cost * cls_cost + self.box_match_cost * box_cost + self.giou_match_cost * giou_cost, valid

    def _pair_buffers(self, b, t, device):
        bufs = getattr(self, '_pairs', None)
        if bufs is None or bufs['src'].shape != (b, t) or bufs['src'].device != device:
            pin = (lambda x: x.pin_memory()) if torch.device(device).type == 'cuda' else (lambda x: x)
            bufs = {'src': torch.zeros(b, t, dtype=torch.int64, device=device), 'tgt': torch.zeros(b, t, dtype=torch.int64, device=device),
                    'w': torch.zeros(b, t, dtype=torch.float32, device=device),
                    'h_src': pin(torch.zeros(b, t, dtype=torch.int64)), 'h_tgt': pin(torch.zeros(b, t, dtype=torch.int64)),
                    'h_w': pin(torch.zeros(b, t, dtype=torch.float32))}
            # constants of forward_static, built HERE (outside any capture: a host -> device copy is not capturable)
            weight = torch.ones(self.num_classes + 1)
            weight[-1] = self.no_object_cls_weight
            bufs['cls_weight'] = weight.to(device)
            bufs['dummy_box'] = torch.tensor([0.5, 0.5, 0.2, 0.2]).to(device)
            self._pairs = bufs
        return bufs

    def assign_device(self, cost, valid):
        """The assignment of every image on the device, no host read: -> (src, tgt, w) [B, T] (the module's static buffers)."""
        from ... import _lib
        b, q, t = cost.shape
        bufs = self._pair_buffers(b, t, cost.device)
        cost = cost.float().contiguous()
        valid = valid.contiguous()
        _lib.check(_lib.lib().saicv_detr_assign(_lib.ptr(cost), _lib.ptr(valid), b, q, t, _lib.ptr(bufs['src']), _lib.ptr(bufs['tgt']),
                                                _lib.ptr(bufs['w']), _lib.stream()), 'detr_assign')
        return bufs['src'], bufs['tgt'], bufs['w']

    def assign_host(self, cost, valid):
        """The same pairs from scipy on the host (the reference's own assignment through its nan / inf wrapper): one device -> host copy
        of (cost, valid), one host -> device copy of the pairs into the module's static buffers -> (src [B, T] query index, tgt [B, T]
        ground-truth row, w [B, T] 1 / 0).  Not on the training path (a synchronising copy per step); it is what the device kernel is
        checked against (tests/test_gpu_r05.py) and what makes the static-shape loss testable without a GPU (tests/test_detr_host.py)."""
        b, q, t = cost.shape
        bufs = self._pair_buffers(b, t, cost.device)
        total = cost.float().cpu().numpy()               # the one synchronising copy of a step
        ok = valid.cpu().numpy()
        hs, ht, hw = bufs['h_src'], bufs['h_tgt'], bufs['h_w']
        hs.zero_(); ht.zero_(); hw.zero_()
        for i in range(b):
            cols = np.nonzero(ok[i])[0]
            if cols.size == 0:
                continue
            rows, cj = self.linear_sum_assignment_with_inf(total[i][:, cols])
            n = len(rows)
            hs[i, :n] = torch.as_tensor(rows, dtype=torch.int64)
            ht[i, :n] = torch.as_tensor(cols[cj], dtype=torch.int64)
            hw[i, :n] = 1.0
        bufs['src'].copy_(hs, non_blocking=True)
        bufs['tgt'].copy_(ht, non_blocking=True)
        bufs['w'].copy_(hw, non_blocking=True)
        return bufs['src'], bufs['tgt'], bufs['w']

    def forward_static(self, preds, gt_pad, src, tgt, w):
        """The loss of forward() from static-shape inputs: pairs (src[i, k], tgt[i, k]) of image i count where w[i, k] = 1.  Same
        per-pair arithmetic; padding pairs are computed on a harmless dummy box and multiplied by 0 (their gradient is exactly 0)."""
        cls_preds, reg_preds = preds
        raw_reg_preds = reg_preds
        reg_preds = torch.clamp(reg_preds, min=1e-4, max=1. - 1e-4).float() if not reg_preds.is_cuda else None
        cls_preds = cls_preds.float()
        gt = gt_pad.float()
        l, b, q = cls_preds.shape[0], cls_preds.shape[1], cls_preds.shape[2]
        t = src.shape[1]
        bidx = torch.arange(b, device=gt.device)[:, None].expand(b, t)
        on = w > 0
        m_cls = gt[bidx, tgt, 4]
        # class map [B, Q]: the no-object class everywhere, the matched class at (i, src); padding pairs write column Q of a wider map
        gmap = torch.full((b, q + 1), self.num_classes, dtype=torch.long, device=gt.device)
        gmap[bidx, torch.where(on, src, torch.full_like(src, q))] = torch.where(on, m_cls.long(), torch.full_like(src, self.num_classes))
        gmap = gmap[:, :q]
        weight = self._pairs['cls_weight']
        nll = -F.log_softmax(cls_preds, dim=-1).gather(-1, gmap.view(1, b, q, 1).expand(l, b, q, 1)).squeeze(-1)
        wgt = weight[gmap]
        cls_l = (nll * wgt).sum(dim=(1, 2)) / wgt.sum()
        # the number of ground-truth boxes of the batch, as forward() and the reference (losses.py:938-954) divide by it -- NOT the
        # number of matched pairs (fewer when an image carries more boxes than queries) and not clamped: a batch without any box
        # gives 0 / 0 = nan here as it does there, and the loop skips the step
        if raw_reg_preds.is_cuda:
            # one kernel each way (late r06); the clamp, the pair gather and the count of ground-truth rows happen inside
            l1_l, iou_l = _DetrBoxLossFn.apply(raw_reg_preds, gt, src, tgt, w, 1e-4, 1. - 1e-4)
        else:
            target_num = (gt[:, :, 4] >= 0).sum().float()
            dummy = self._pairs['dummy_box']
            m_box = gt[bidx, tgt, 0:4]
            pm = torch.where(on[None, :, :, None], reg_preds[:, bidx, src], dummy)                 # [L, B, T, 4]
            tb = torch.where(on[:, :, None], m_box, dummy)                                          # [B, T, 4]
            l1_l = ((pm - tb).abs().sum(dim=-1) * w).sum(dim=(1, 2)) / target_num
            giou = _giou(_cxcywh_to_xyxy(pm), _cxcywh_to_xyxy(tb))
            iou_l = ((1 - giou) * w).sum(dim=(1, 2)) / target_num
        return self._layer_terms(cls_l, l1_l, iou_l)

    # reference-named views of the same computations (used by its tests / tools)
    def compute_batch_cls_loss(self, cls_preds, annotations, indices):
        valid = [a[a[:, 4] >= 0] for a in annotations]
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)]).to(cls_preds.device)
        src_idx = torch.cat([src for src, _ in indices]).to(cls_preds.device)
        classes = torch.cat([v[j.to(v.device), 4] for v, (_, j) in zip(valid, indices)])
        return self._cls_loss(cls_preds, batch_idx, src_idx, classes)

    def compute_batch_l1_iou_loss(self, reg_preds, annotations, indices):
        valid = [a[a[:, 4] >= 0] for a in annotations]
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)]).to(reg_preds.device)
        src_idx = torch.cat([src for src, _ in indices]).to(reg_preds.device)
        boxes = torch.cat([v[j.to(v.device), 0:4] for v, (_, j) in zip(valid, indices)], dim=0)
        return self._box_losses(reg_preds[batch_idx, src_idx], boxes, sum(v.shape[0] for v in valid))

    def transform_cxcywh_box_to_xyxy_box(self, boxes):
        return _cxcywh_to_xyxy(boxes)

    def compute_box_giou(self, boxes1, boxes2):
        """[N, 4] x [M, 4] xyxy -> [N, M] pairwise GIoU."""
        return _giou(boxes1[:, None, :], boxes2[None, :, :])

    @staticmethod
    def _valid_targets(annotations):
        """-> (gt [n, 5] fp32 on the annotations' device: the rows with class >= 0, image after image; boxes per image).
        When the loop attached the host copy the collater produced (`annotations._saicv_host`), the row selection is
        computed there and costs one small upload; else one boolean-mask index per image (a device sync each)."""
        host = getattr(annotations, '_saicv_host', None)
        if host is not None and host.shape == annotations.shape:
            keep = host[:, :, 4] >= 0
            counts = keep.sum(dim=1).tolist()
            rows = keep.flatten().nonzero().squeeze(1)
            return annotations.float().flatten(0, 1)[rows.to(annotations.device, non_blocking=True)], counts
        ann = annotations.float()
        valid = [a[a[:, 4] >= 0] for a in ann]
        return torch.cat(valid, dim=0), [v.shape[0] for v in valid]

    @torch.no_grad()
    def get_matched_pred_target_idxs(self, cls_preds, reg_preds, annotations):
        gt, counts = self._valid_targets(annotations)
        return self._match(cls_preds, reg_preds, gt, counts)

    @torch.no_grad()
    def _match(self, cls_preds, reg_preds, gt, counts):
        b, q = cls_preds.shape[0], cls_preds.shape[1]
        prob = torch.clamp(F.softmax(cls_preds.flatten(0, 1), dim=-1), min=1e-4, max=1. - 1e-4)
        boxes = reg_preds.flatten(0, 1)
        cls_cost = -prob[:, gt[:, 4].long()]
        box_cost = torch.cdist(boxes, gt[:, 0:4], p=1)
        giou_cost = -self.compute_box_giou(_cxcywh_to_xyxy(boxes), _cxcywh_to_xyxy(gt[:, 0:4]))
        total = (self.cls_match_cost * cls_cost + self.box_match_cost * box_cost +
                 self.giou_match_cost * giou_cost).view(b, q, -1).cpu()      # ONE device->host copy per step
        indices = []
        for i, block in enumerate(total.split(counts, -1)):
            rows, cols = self.linear_sum_assignment_with_inf(block[i].numpy())
            indices.append((torch.as_tensor(rows, dtype=torch.int64), torch.as_tensor(cols, dtype=torch.int64)))
        return indices

    def linear_sum_assignment_with_inf(self, cost_matrix):
        """scipy's assignment, tolerating nan (-> 1e5) and one-signed infinities (-> a finite value beyond any
        achievable total), as the reference does."""
        cost_matrix = np.array(cost_matrix, copy=True)
        if np.isnan(cost_matrix).any():
            cost_matrix[np.isnan(cost_matrix)] = 1e5
        min_inf = np.isneginf(cost_matrix).any()
        max_inf = np.isposinf(cost_matrix).any()
        if min_inf and max_inf:
            raise ValueError("matrix contains both inf and -inf")
        if min_inf or max_inf:
            values = cost_matrix[~np.isinf(cost_matrix)]
            lo, hi = values.min(), values.max()
            m = min(cost_matrix.shape)
            positive = m * (hi - lo + np.abs(hi) + np.abs(lo) + 1)
            if max_inf:
                place_holder = (hi + (m - 1) * (hi - lo)) + positive
            else:
                place_holder = (lo + (m - 1) * (lo - hi)) - positive
            cost_matrix[np.isinf(cost_matrix)] = place_holder
        return scipy.optimize.linear_sum_assignment(cost_matrix)


# ---------------------------------------------------------------------------------------------- RetinaNet
def _box_iou(b1, b2, kind):
    """IoU family of row-aligned [N, 4] xyxy boxes with the reference IoUMethod's clamps (losses.py:25-120): areas from clamped
    sides, union / enclosing area / diagonal >= 1e-4."""
    wh1 = (b1[:, 2:4] - b1[:, 0:2]).clamp(min=0)
    wh2 = (b2[:, 2:4] - b2[:, 0:2]).clamp(min=0)
    inter_wh = (torch.min(b1[:, 2:4], b2[:, 2:4]) - torch.max(b1[:, 0:2], b2[:, 0:2])).clamp(min=0)
    inter = inter_wh[:, 0] * inter_wh[:, 1]
    union = (wh1[:, 0] * wh1[:, 1] + wh2[:, 0] * wh2[:, 1] - inter).clamp(min=1e-4)
    iou = inter / union
    if kind == 'IoU':
        return iou
    hull = (torch.max(b1[:, 2:4], b2[:, 2:4]) - torch.min(b1[:, 0:2], b2[:, 0:2])).clamp(min=0)
    if kind == 'GIoU':
        area = (hull[:, 0] * hull[:, 1]).clamp(min=1e-4)
        return iou - (area - union) / area
    diag2 = (hull[:, 0] ** 2 + hull[:, 1] ** 2).clamp(min=1e-4)
    dc = (b1[:, 2:4] + b1[:, 0:2]) / 2 - (b2[:, 2:4] + b2[:, 0:2]) / 2
    centre2 = dc[:, 0] ** 2 + dc[:, 1] ** 2
    if kind == 'DIoU':
        return iou - centre2 / diag2
    if kind == 'CIoU':
        v = (4 / math.pi ** 2) * (torch.atan(wh2[:, 0] / wh2[:, 1]) - torch.atan(wh1[:, 0] / wh1[:, 1])) ** 2
        with torch.no_grad():
            a = v / (1 - iou + v).clamp(min=1e-4)
        return iou - (centre2 / diag2 + v * a)
    dw2, dh2 = (wh2[:, 0] - wh1[:, 0]) ** 2, (wh2[:, 1] - wh1[:, 1]) ** 2          # EIoU
    return iou - (centre2 / diag2 + dw2 / (hull[:, 0] ** 2).clamp(min=1e-4) + dh2 / (hull[:, 1] ** 2).clamp(min=1e-4))


class _RetinaLossFn(torch.autograd.Function):
    """Target assignment, focal loss and SmoothL1 box loss on the HIP kernels of csrc/detloss.hip.  inputs: the anchor table
    [A, 4], annotations [B, G, 5], then the per-level class probabilities [B, A_l, C] and box offsets [B, A_l, 4] (fp32, dense).
    Returns (focal sum / positives, SmoothL1 sum / positives, targets [B, A, 5], positives); both losses are 0 when no anchor
    is positive (losses.py:235-236, :284-285) -- decided on the device, no host synchronisation."""

    @staticmethod
    def forward(ctx, anchors, annots, alpha, gamma, beta, smoothl1, *heads):
        from ... import _lib
        from ...ops import require_gpu
        L, st = _lib.lib(), _lib.stream()
        levels = len(heads) // 2
        cls, reg = heads[:levels], heads[levels:]
        require_gpu(anchors, annots, *heads)
        B, G = annots.shape[0], annots.shape[1]
        sizes = [t.shape[1] for t in cls]
        At = int(anchors.shape[0])
        assert sum(sizes) == At, 'anchor table does not match the pyramid levels'
        dev = annots.device
        targets = torch.empty((B, At, 5), dtype=torch.float32, device=dev)
        sums = torch.zeros(3, dtype=torch.float32, device=dev)                  # positives, focal sum, box sum
        _lib.check(L.saicv_retina_assign(anchors.data_ptr(), annots.data_ptr() if G else None, targets.data_ptr(), sums[0:1].data_ptr(),
                                         B, At, G, int(smoothl1), st), 'retina_assign')
        need_c = [ctx.needs_input_grad[6 + i] for i in range(levels)]
        need_r = [ctx.needs_input_grad[6 + levels + i] for i in range(levels)]
        grads, off = [], 0
        for i in range(levels):
            dc = torch.empty_like(cls[i]) if need_c[i] else None
            _lib.check(L.saicv_focal_loss_level(cls[i].data_ptr(), targets.data_ptr(), dc.data_ptr() if dc is not None else None,
                                                sums[1:2].data_ptr(), B, sizes[i], At, off, cls[i].shape[2], float(alpha), float(gamma), st),
                       'focal_loss_level')
            grads.append(dc)
            off += sizes[i]
        off = 0
        for i in range(levels):
            dr = None
            if smoothl1:
                dr = torch.empty_like(reg[i]) if need_r[i] else None
                _lib.check(L.saicv_smoothl1_level(reg[i].data_ptr(), targets.data_ptr(), dr.data_ptr() if dr is not None else None,
                                                  sums[2:3].data_ptr(), B, sizes[i], At, off, float(beta), st), 'smoothl1_level')
            grads.append(dr)
            off += sizes[i]
        inv = torch.where(sums[0] > 0, 1.0 / sums[0].clamp(min=1.0), torch.zeros_like(sums[0]))
        ctx.save_for_backward(inv, *[g for g in grads if g is not None])
        ctx.present = [g is not None for g in grads]
        ctx.levels = levels
        ctx.mark_non_differentiable(targets)
        return sums[1] * inv, sums[2] * inv, targets, sums[0].clone()

    @staticmethod
    def backward(ctx, g_cls, g_box, _gt, _gp):
        from ... import _lib
        L, st = _lib.lib(), _lib.stream()
        inv, *saved = ctx.saved_tensors
        out, it = [], iter(saved)
        for j, present in enumerate(ctx.present):
            if not present:
                out.append(None)
                continue
            d = next(it)
            scale = ((g_cls if j < ctx.levels else g_box).float() * inv).contiguous()
            res = torch.empty_like(d)
            _lib.check(L.saicv_scale_by_scalar(_lib.F32, d.data_ptr(), scale.data_ptr(), res.data_ptr(), d.numel(), st), 'scale_by_scalar')
            out.append(res)
        return (None, None, None, None, None, None, *out)


class RetinaLoss(nn.Module):
    """Drop-in for the reference RetinaLoss: same constructor, `forward(preds, annotations) -> {'cls_loss', 'reg_loss'}`.
    preds = [cls_heads, reg_heads] of RetinaNet (per level [B, H, W, anchors, classes] probabilities / [B, H, W, anchors, 4]
    offsets), annotations [B, max_annots, 5] padded with -1 rows.  Anchor assignment, the focal loss and the SmoothL1 box loss run
    on csrc/detloss.hip in place on the per-level tensors; the IoU-family box losses decode the few positive anchors and use
    tensor arithmetic on them (`_box_iou`)."""

    def __init__(self, areas=[[32, 32], [64, 64], [128, 128], [256, 256], [512, 512]], ratios=[0.5, 1, 2],
                 scales=[2**0, 2**(1.0 / 3.0), 2**(2.0 / 3.0)], strides=[8, 16, 32, 64, 128], alpha=0.25, gamma=2, beta=1.0 / 9.0,
                 cls_loss_weight=1., box_loss_weight=1., box_loss_type='SmoothL1'):
        super(RetinaLoss, self).__init__()
        assert box_loss_type in ['SmoothL1', 'IoU', 'GIoU', 'DIoU', 'CIoU', 'EIoU'], 'wrong IoU type!'
        from .models.anchor import RetinaAnchors
        self.anchors = RetinaAnchors(areas=areas, ratios=ratios, scales=scales, strides=strides)
        self.alpha, self.gamma, self.beta = alpha, gamma, beta
        self.cls_loss_weight, self.box_loss_weight = cls_loss_weight, box_loss_weight
        self.box_loss_type = box_loss_type
        self._tables = {}
        # SmoothL1: every decision (assignment, positive count) stays on the device and every shape is static -> the training step
        # can be captured as one hipGraph (tools/scripts.py _epoch_loop).  The IoU-family box losses index the positive anchors with
        # a boolean mask (a host synchronisation and a data-dependent shape): eager only.
        self.capturable = box_loss_type == 'SmoothL1'

    def _anchor_table(self, sizes, device):
        key = (tuple(map(tuple, sizes)), str(device))
        if key not in self._tables:
            per_level = self.anchors([list(s) for s in sizes])
            self._tables[key] = torch.cat([torch.from_numpy(a).view(-1, 4) for a in per_level], dim=0).to(device)
        return self._tables[key]

    def forward(self, preds, annotations):
        cls_preds, reg_preds = preds
        sizes = [[t.shape[2], t.shape[1]] for t in cls_preds]                    # [w, h] per level
        anchors = self._anchor_table(sizes, annotations.device)
        b = annotations.shape[0]
        cls = [t.reshape(b, -1, t.shape[-1]).float().contiguous() for t in cls_preds]
        reg = [t.reshape(b, -1, 4).float().contiguous() for t in reg_preds]
        smooth = self.box_loss_type == 'SmoothL1'
        cls_loss, box_loss, targets, positives = _RetinaLossFn.apply(anchors, annotations.float().contiguous(), self.alpha, self.gamma,
                                                                     self.beta, smooth, *cls, *reg)
        if not smooth:
            box_loss = self._iou_box_loss(torch.cat(reg, dim=1), targets, anchors, positives)
        return {'cls_loss': self.cls_loss_weight * cls_loss, 'reg_loss': self.box_loss_weight * box_loss}

    def _iou_box_loss(self, reg, targets, anchors, positives):
        pos = targets[..., 4] > 0
        if not bool(pos.any()):
            return torch.zeros((), dtype=torch.float32, device=reg.device)
        boxes = self.snap_txtytwth_to_xyxy(reg[pos], anchors.unsqueeze(0).expand(reg.shape[0], -1, -1)[pos])
        return (1 - _box_iou(boxes, targets[pos][:, 0:4], self.box_loss_type)).sum() / positives

    @staticmethod
    def snap_txtytwth_to_xyxy(snap_boxes, anchors):
        wh = anchors[:, 2:4] - anchors[:, 0:2]
        centre = anchors[:, 0:2] + 0.5 * wh
        box_wh = torch.exp(snap_boxes[:, 2:4]) * wh
        box_centre = snap_boxes[:, :2] * wh + centre
        return torch.cat([box_centre - 0.5 * box_wh, box_centre + 0.5 * box_wh], dim=1)


# ---------------------------------------------------------------------------------------------- FCOS
class _FocalLevelsFn(torch.autograd.Function):
    """focal loss of per-level probabilities [B, P_l, C] against targets [B, P, 5] (csrc/detloss.hip), times `inv` (1 / positives,
    or 0 when there are none): sum and gradient in one pass per level, no concatenation of the levels."""

    @staticmethod
    def forward(ctx, targets, inv, alpha, gamma, *cls):
        from ... import _lib
        L, st = _lib.lib(), _lib.stream()
        B, P = targets.shape[0], targets.shape[1]
        total = torch.zeros(1, dtype=torch.float32, device=targets.device)
        grads, off = [], 0
        for i, t in enumerate(cls):
            d = torch.empty_like(t) if ctx.needs_input_grad[4 + i] else None
            _lib.check(L.saicv_focal_loss_level(t.data_ptr(), targets.data_ptr(), d.data_ptr() if d is not None else None, total.data_ptr(),
                                                B, t.shape[1], P, off, t.shape[2], float(alpha), float(gamma), st), 'focal_loss_level')
            grads.append(d)
            off += t.shape[1]
        assert off == P, 'targets do not match the pyramid levels'
        ctx.save_for_backward(inv, *[g for g in grads if g is not None])
        ctx.present = [g is not None for g in grads]
        return total[0] * inv

    @staticmethod
    def backward(ctx, gout):
        from ... import _lib
        L, st = _lib.lib(), _lib.stream()
        inv, *saved = ctx.saved_tensors
        scale = (gout.float() * inv).contiguous()
        out, it = [], iter(saved)
        for present in ctx.present:
            if not present:
                out.append(None)
                continue
            d = next(it)
            res = torch.empty_like(d)
            _lib.check(L.saicv_scale_by_scalar(_lib.F32, d.data_ptr(), scale.data_ptr(), res.data_ptr(), d.numel(), st), 'scale_by_scalar')
            out.append(res)
        return (None, None, None, None, *out)


class FCOSLoss(nn.Module):
    """Drop-in for the reference FCOSLoss: same constructor, `forward(preds, annotations) -> {'cls_loss', 'reg_loss',
    'center_ness_loss'}`; preds = [cls_heads, reg_heads, center_heads] of FCOS (per level [B, H, W, classes] probabilities,
    [B, H, W, 4] log-distances, [B, H, W, 1] centre-ness probabilities).  Point assignment and the focal loss run on
    csrc/detloss.hip over all points; the IoU and centre-ness losses are tensor arithmetic on the positive points only."""

    def __init__(self, strides=[8, 16, 32, 64, 128], mi=[[-1, 64], [64, 128], [128, 256], [256, 512], [512, 100000000]], alpha=0.25,
                 gamma=2., cls_loss_weight=1., box_loss_weight=1., center_ness_loss_weight=1., box_loss_iou_type='GIoU',
                 center_sample_radius=1.5, use_center_sample=True):
        super(FCOSLoss, self).__init__()
        assert box_loss_iou_type in ['IoU', 'GIoU', 'DIoU', 'CIoU', 'EIoU'], 'wrong IoU type!'
        from .models.anchor import FCOSPositions
        self.positions = FCOSPositions(strides=strides)
        self.alpha, self.gamma = alpha, gamma
        self.strides, self.mi = strides, mi
        self.cls_loss_weight, self.box_loss_weight = cls_loss_weight, box_loss_weight
        self.center_ness_loss_weight = center_ness_loss_weight
        self.box_loss_iou_type = box_loss_iou_type
        self.center_sample_radius = center_sample_radius
        self.use_center_sample = use_center_sample
        self._tables = {}

    def _point_table(self, sizes, device):
        """[P, 5] = (x, y, stride, range low, range high) of every point of the pyramid, level after level"""
        key = (tuple(map(tuple, sizes)), str(device))
        if key not in self._tables:
            rows = []
            for xy, stride, (lo, hi) in zip(self.positions([list(s) for s in sizes]), self.strides, self.mi):
                xy = torch.from_numpy(xy).view(-1, 2)
                extra = torch.tensor([float(stride), float(lo), float(hi)], dtype=torch.float32).expand(xy.shape[0], 3)
                rows.append(torch.cat([xy, extra], dim=1))
            self._tables[key] = torch.cat(rows, dim=0).contiguous().to(device)
        return self._tables[key]

    def forward(self, preds, annotations):
        from ... import _lib
        cls_preds, reg_preds, center_preds = preds
        dev = annotations.device
        b = annotations.shape[0]
        points = self._point_table([[t.shape[2], t.shape[1]] for t in cls_preds], dev)
        P = points.shape[0]
        annots = annotations.float().contiguous()
        targets = torch.empty((b, P, 5), dtype=torch.float32, device=dev)
        centerness = torch.empty((b, P), dtype=torch.float32, device=dev)
        positives = torch.zeros(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().saicv_fcos_assign(points.data_ptr(), annots.data_ptr() if annots.shape[1] else None, targets.data_ptr(),
                                                centerness.data_ptr(), positives.data_ptr(), b, P, annots.shape[1],
                                                float(self.center_sample_radius), int(self.use_center_sample), _lib.stream()), 'fcos_assign')
        inv = torch.where(positives[0] > 0, 1.0 / positives[0].clamp(min=1.0), torch.zeros_like(positives[0]))
        cls = [t.reshape(b, -1, t.shape[-1]).float().contiguous() for t in cls_preds]
        cls_loss = _FocalLevelsFn.apply(targets, inv, self.alpha, self.gamma, *cls)
        pos = targets[..., 4] > 0
        if bool(pos.any()):
            reg = torch.exp(torch.cat([t.reshape(b, -1, 4) for t in reg_preds], dim=1).float()[pos])
            ctr_pred = torch.cat([t.reshape(b, -1) for t in center_preds], dim=1).float().clamp(min=1e-4, max=1. - 1e-4)[pos]
            xy = points[:, 0:2].unsqueeze(0).expand(b, -1, -1)[pos]
            ltrb, ctr = targets[pos][:, 0:4], centerness[pos]
            pred_boxes = torch.cat([xy - reg[:, 0:2], xy + reg[:, 2:4]], dim=1)
            gt_boxes = torch.cat([xy - ltrb[:, 0:2], xy + ltrb[:, 2:4]], dim=1)
            reg_loss = ((1 - _box_iou(pred_boxes, gt_boxes, self.box_loss_iou_type)) * ctr).sum() * inv
            ctr_loss = -(ctr * torch.log(ctr_pred) + (1. - ctr) * torch.log(1. - ctr_pred)).sum() * inv
        else:
            reg_loss = torch.zeros((), dtype=torch.float32, device=dev)
            ctr_loss = torch.zeros((), dtype=torch.float32, device=dev)
        return {'cls_loss': self.cls_loss_weight * cls_loss, 'reg_loss': self.box_loss_weight * reg_loss,
                'center_ness_loss': self.center_ness_loss_weight * ctr_loss}
