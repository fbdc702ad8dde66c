"""Mixup / CutMix batch collater with soft (smoothed, mixed) labels -- the train_collater of the ViT configs
(reference SimpleAICV/classification/mixupcutmixclassificationcollator.py:99-284; used by
vit_base_patch16_for_self_train_mae_pretrain/train_config.py:77-87).

Same constructor arguments, same output contract ({'image': float32 [B,3,H,W] as an NHWC-strided view, 'label':
float32 [B, num_classes]}) and the same numpy random-draw ORDER as the reference, so a seeded run mixes the same pairs
with the same lambdas and boxes (pinned by tests/golden/mixup_cutmix.pt, produced by the reference collater).
Structure differs: the random plan (who mixes with whom, lambda, box) is drawn first as plain numbers, then applied
with whole-batch tensor operations instead of per-sample python loops.

r03: `plan()` + `apply_on_device()` run the same collater on a batch that is ALREADY in device memory
(fp32 NHWC, or uint8 NHWC with the dataset's normalisation folded in): the numpy draws stay on the host in the reference's
order, the pixels and the soft labels are produced by two HBM-bound HIP kernels (csrc/input.hip) whose arithmetic is
bit-identical to the tensor expressions of `__call__`."""
import numpy as np
import torch


def _box_for(lam, h, w, minmax, correct):
    """CutMix box for one lambda -> (yl, yh, xl, xh, corrected lambda); draw order: (cy, cx) or (ch, cw, yl, xl)."""
    if minmax is not None:
        ch = np.random.randint(int(h * minmax[0]), int(h * minmax[1]))
        cw = np.random.randint(int(w * minmax[0]), int(w * minmax[1]))
        yl = np.random.randint(0, h - ch)
        xl = np.random.randint(0, w - cw)
        yh, xh = yl + ch, xl + cw
    else:
        ratio = np.sqrt(1 - lam)
        ch, cw = int(h * ratio), int(w * ratio)
        cy = np.random.randint(0, h)
        cx = np.random.randint(0, w)
        yl, yh = np.clip(cy - ch // 2, 0, h), np.clip(cy + ch // 2, 0, h)
        xl, xh = np.clip(cx - cw // 2, 0, w), np.clip(cx + cw // 2, 0, w)
    if correct or minmax is not None:
        lam = 1. - (yh - yl) * (xh - xl) / float(h * w)
    return int(yl), int(yh), int(xl), int(xh), lam


def smoothed_one_hot(labels, num_classes, smoothing):
    off = smoothing / num_classes
    return torch.full((labels.numel(), num_classes), off).scatter_(1, labels.view(-1, 1), 1. - smoothing + off)


class MixupCutmixClassificationCollater:

    def __init__(self, use_mixup=True, mixup_alpha=0.8, cutmix_alpha=1.0, cutmix_minmax=None, mixup_cutmix_prob=1.0,
                 switch_to_cutmix_prob=0.5, mode='batch', correct_lam=True, label_smoothing=0.1, num_classes=1000):
        assert mode in ['batch', 'pair', 'elem']
        if cutmix_minmax is not None:
            assert len(cutmix_minmax) == 2
            cutmix_alpha = 1.0
        self.use_mixup, self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = use_mixup, mixup_alpha, cutmix_alpha, cutmix_minmax
        self.mixup_cutmix_prob, self.switch_to_cutmix_prob = mixup_cutmix_prob, switch_to_cutmix_prob
        self.label_smoothing, self.num_classes, self.mode, self.correct_lam = label_smoothing, num_classes, mode, correct_lam

    # ---- random plan -------------------------------------------------------------------------------------------
    def _draw(self, n):
        """per-element lambdas and cutmix switches for n slots (reference _params_per_elem draw order)"""
        use_cut = np.zeros(n, dtype=bool)
        if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
            use_cut = np.random.rand(n) < self.switch_to_cutmix_prob
            lam = np.where(use_cut, np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=n),
                           np.random.beta(self.mixup_alpha, self.mixup_alpha, size=n))
        elif self.mixup_alpha > 0.:
            lam = np.random.beta(self.mixup_alpha, self.mixup_alpha, size=n)
        elif self.cutmix_alpha > 0.:
            use_cut = np.ones(n, dtype=bool)
            lam = np.random.beta(self.cutmix_alpha, self.cutmix_alpha, size=n)
        else:
            raise AssertionError('One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true.')
        lam = np.where(np.random.rand(n) < self.mixup_cutmix_prob, lam.astype(np.float32), np.ones(n, dtype=np.float32))
        return lam, use_cut

    def _draw_batch(self):
        if not (np.random.rand() < self.mixup_cutmix_prob):
            return 1., False
        if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
            cut = np.random.rand() < self.switch_to_cutmix_prob
            a = self.cutmix_alpha if cut else self.mixup_alpha
            return float(np.random.beta(a, a)), bool(cut)
        if self.mixup_alpha > 0.:
            return float(np.random.beta(self.mixup_alpha, self.mixup_alpha)), False
        if self.cutmix_alpha > 0.:
            return float(np.random.beta(self.cutmix_alpha, self.cutmix_alpha)), True
        raise AssertionError('One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true.')

    # ---- application -------------------------------------------------------------------------------------------
    def _apply(self, x, slots, lam, use_cut, mirror):
        """x [B,3,H,W]; sample i in `slots` mixes with B-1-i (and, if mirror, B-1-i with i).  Returns per-slot lambdas."""
        b, _, h, w = x.shape
        src = x.clone()
        lam = lam.copy()
        for k, i in enumerate(slots):
            if lam[k] == 1.:
                continue
            j = b - i - 1
            if use_cut[k]:
                yl, yh, xl, xh, lam[k] = _box_for(lam[k], h, w, self.cutmix_minmax, self.correct_lam)
                x[i, :, yl:yh, xl:xh] = src[j, :, yl:yh, xl:xh]
                if mirror:
                    x[j, :, yl:yh, xl:xh] = src[i, :, yl:yh, xl:xh]
            else:
                x[i] = src[i] * lam[k] + src[j] * (1 - lam[k])
                if mirror:
                    x[j] = src[j] * lam[k] + src[i] * (1 - lam[k])
        return lam

    # ---- the same plan as plain numbers (one entry per sample), for the device path ---------------------------------
    def plan(self, b, h, w):
        """-> list of b tuples (mode, yl, yh, xl, xh, lam, one_minus_lam, label_lam, label_one_minus_lam): mode 0 copy,
        1 mixup, 2 cutmix box; sample i always mixes with sample b-1-i.  Consumes the numpy generator exactly as
        __call__ does for a batch of this size (same draws in the same order), and the float32 / float64 roundings of the
        weights are the ones the tensor expressions of __call__ apply."""
        f32 = np.float32
        keep = (0, 0, 0, 0, 0, f32(1), f32(0), f32(1), f32(0))
        if not self.use_mixup:
            return [keep] * b
        assert b % 2 == 0, 'Batch size should be even when using this'
        if self.mode == 'batch':
            lam, cut = self._draw_batch()
            if lam == 1.:
                return [keep] * b
            if cut:
                yl, yh, xl, xh, lam = _box_for(lam, h, w, self.cutmix_minmax, self.correct_lam)
                return [(2, yl, yh, xl, xh, f32(1), f32(0), f32(lam), f32(1. - lam))] * b
            return [(1, 0, 0, 0, 0, f32(lam), f32(1. - lam), f32(lam), f32(1. - lam))] * b
        out = [keep] * b
        if self.mode == 'elem':
            lam, cut = self._draw(b)
            slots, mirror = range(b), False
        else:
            lam, cut = self._draw(b // 2)
            slots, mirror = range(b // 2), True
        lam = lam.copy()                                   # float32, as in _apply
        for k, i in enumerate(slots):
            if lam[k] == 1.:
                continue
            if cut[k]:
                yl, yh, xl, xh, lam[k] = _box_for(lam[k], h, w, self.cutmix_minmax, self.correct_lam)
                e = (2, yl, yh, xl, xh, f32(1), f32(0), lam[k], f32(1) - lam[k])
            else:
                e = (1, 0, 0, 0, 0, lam[k], f32(1) - lam[k], lam[k], f32(1) - lam[k])
            out[i] = e
            if mirror:
                out[b - 1 - i] = e
        return out                     # (pair mode: both members of a pair carry the same entry, as concat(half, half[::-1]) does)

    def apply_on_device(self, images, labels, scale=None, shift=None):
        """images: [B, H, W, C] on the device, fp32 or uint8 (then `scale` / `shift`: per-channel fp32 device tensors of the
        dataset normalisation v * scale + shift); labels: int64 [B] on the device.  -> the dict __call__ returns, with
        tensors on the device.  No host synchronisation: the plan is uploaded with the launch."""
        import ctypes
        from ..._lib import MixPlan, check, lib, ptr, require_gpu, stream
        require_gpu(images)
        assert images.dim() == 4 and images.is_contiguous() and images.dtype in (torch.float32, torch.uint8)
        b, h, w, c = images.shape
        entries = self.plan(b, h, w)
        arr = (MixPlan * b)()
        for i, e in enumerate(entries):
            (arr[i].mode, arr[i].yl, arr[i].yh, arr[i].xl, arr[i].xh) = (int(v) for v in e[:5])
            (arr[i].lam, arr[i].one_minus_lam, arr[i].label_lam, arr[i].label_one_minus_lam) = (float(v) for v in e[5:])
        plan_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(images.device, non_blocking=True)
        out = torch.empty((b, h, w, c), dtype=torch.float32, device=images.device)
        check(lib().saicv_mixup_cutmix(int(images.dtype == torch.uint8), ptr(images), ptr(plan_dev), ptr(scale), ptr(shift),
                                       ptr(out), b, h, w, c, stream()), 'mixup_cutmix')
        if not self.use_mixup:
            return {'image': out.permute(0, 3, 1, 2), 'label': labels}
        off = self.label_smoothing / self.num_classes
        y = torch.empty((b, self.num_classes), dtype=torch.float32, device=images.device)
        check(lib().saicv_soft_labels(ptr(labels), ptr(plan_dev), ctypes.c_float(off), ctypes.c_float(1. - self.label_smoothing + off),
                                      ptr(y), b, self.num_classes, stream()), 'soft_labels')
        return {'image': out.permute(0, 3, 1, 2), 'label': y}

    def __call__(self, data):
        images = torch.from_numpy(np.array([s['image'] for s in data]).astype(np.float32)).permute(0, 3, 1, 2)
        labels = torch.from_numpy(np.array([s['label'] for s in data]).astype(np.float32)).long()
        if not self.use_mixup:
            return {'image': images, 'label': labels}
        b, _, h, w = images.shape
        assert b % 2 == 0, 'Batch size should be even when using this'
        if self.mode == 'batch':
            lam, cut = self._draw_batch()
            if lam != 1.:
                if cut:
                    yl, yh, xl, xh, lam = _box_for(lam, h, w, self.cutmix_minmax, self.correct_lam)
                    images[:, :, yl:yh, xl:xh] = images.flip(0)[:, :, yl:yh, xl:xh].clone()
                else:
                    images = images * lam + images.flip(0) * (1. - lam)
            lam_t = lam
        elif self.mode == 'elem':
            lam, cut = self._draw(b)
            lam_t = torch.tensor(self._apply(images, range(b), lam, cut, False), dtype=images.dtype).unsqueeze(1)
        else:
            lam, cut = self._draw(b // 2)
            half = self._apply(images, range(b // 2), lam, cut, True)
            lam_t = torch.tensor(np.concatenate((half, half[::-1])), dtype=images.dtype).unsqueeze(1)
        y = smoothed_one_hot(labels, self.num_classes, self.label_smoothing)
        return {'image': images, 'label': y * lam_t + y.flip(0) * (1. - lam_t)}
