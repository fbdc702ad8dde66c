"""float64 arbiter for the whole-model gradient gates.  TEST INFRASTRUCTURE: runs THE REFERENCE
(imported from /root/reference) once more in float64 and stores, next to each existing fixture, what the truth is and how far
the reference's OWN fp32 run is from it.

Build container only:   python oracle/make_golden_fp64.py [names...]

Why: two fp32 runs of one model disagree (summation order), so "HIP within 2 x the reference's reorder noise" cannot tell a
1 % systematic gradient error from noise.  A float64 run of the same modules on the same seeds is the arbiter: both fp32 results
(the reference's and the HIP path's) are noisy estimates of it, and a correct fp32 implementation sits as close to it as the
reference's fp32 run does.

Per fixture `tests/golden/fp64_<name>.pt`:
    idx[n]        up to 1024 evenly spaced flat indices into parameter n's gradient (the whole tensor when smaller)
    g64[n]        float64 gradient at idx (stored as float64)
    ref32_l2[n]   || ref_fp32[idx] - g64[idx] ||_2 / || g64[idx] ||_2      (the reference's own fp32 run, contiguous layout)
    ref32_alt_l2[n]  the same for the reference's second fp32 run (channels_last / other thread count), where one exists
    out64         the float64 output / logits (or their norm and a sub-sample for large outputs), loss64
    all_l2        the relative L2 distance over ALL sampled elements of all parameters, for ref32 and ref32_alt

Cases: resnet50_b32_112, vit_base_patch16_b2_224, sam_b_encoder_256, detr_r50_small  (the builders mirror the generators
that made the fp32 fixtures: oracle/make_golden.py, make_golden_sam.py, make_golden_detr.py -- same seeds, same inputs; the
stored fp32 gradient samples of those fixtures are re-checked here, so a drifted builder cannot go unnoticed).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

NSAMPLE = 1024


def sample_idx(numel):
    if numel <= NSAMPLE:
        return torch.arange(numel)
    return torch.linspace(0, numel - 1, NSAMPLE).round().long()


def _l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def summarise(name, grads32, grads64, grads32_alt, extra, old_fixture):
    """grads*: {param name: gradient tensor}.  Checks the fp32 run against the committed fp32 fixture, stores the arbiter."""
    old = torch.load(os.path.join(OUT, old_fixture + '.pt'), weights_only=False)
    worst_old = 0.0
    for n, s in old['grad_sample'].items():
        if n in grads32 and float(s.abs().max()) > 0:
            worst_old = max(worst_old, float((grads32[n].flatten()[:64] - s).abs().max() / s.abs().max()))
    idx, g64, r32, r32a = {}, {}, {}, {}
    cat32, cat64, cata = [], [], []
    for n, g in grads64.items():
        i = sample_idx(g.numel())
        idx[n] = i
        g64[n] = g.flatten()[i].clone()
        a = grads32[n].flatten()[i]
        r32[n] = _l2(a, g64[n])
        cat32.append(a.double())
        cat64.append(g64[n])
        if grads32_alt is not None and n in grads32_alt:
            b = grads32_alt[n].flatten()[i]
            r32a[n] = _l2(b, g64[n])
            cata.append(b.double())
    cat32, cat64 = torch.cat(cat32), torch.cat(cat64)
    fx = {'name': name, 'of_fixture': old_fixture, 'nsample': NSAMPLE, 'idx': idx, 'g64': g64, 'ref32_l2': r32, 'ref32_alt_l2': r32a,
          'all_l2': {'ref32': _l2(cat32, cat64), 'ref32_alt': _l2(torch.cat(cata), cat64) if cata else None},
          'fp32_rerun_vs_committed_fixture': worst_old, 'torch_version': torch.__version__}
    fx.update(extra)
    path = os.path.join(OUT, 'fp64_' + name + '.pt')
    torch.save(fx, path)
    v = sorted(r32.values())
    print(f'{name}: fp32 rerun vs committed fixture {worst_old:.2e}; reference fp32 vs float64 per parameter: median '
          f'{v[len(v) // 2]:.3e} worst {v[-1]:.3e}; all sampled elements {fx["all_l2"]}; -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)', flush=True)
    assert worst_old < 1e-3, 'the rebuilt fp32 run does not reproduce the committed fixture: builder drift'


class float_is_double:
    """Inside the float64 run `tensor.float()` must not round: the reference's criteria cast their inputs with `.float()`
    (classification/losses.py:24, detection/losses.py:873-875), which would put an fp32 stage into the arbiter."""

    def __enter__(self):
        self.orig = torch.Tensor.float
        torch.Tensor.float = lambda t, *a, **k: t.double()
        torch.set_default_dtype(torch.float64)          # tensors the criteria create on the fly (class weights, eye, ...)

    def __exit__(self, *exc):
        torch.Tensor.float = self.orig
        torch.set_default_dtype(torch.float32)


def _named_grads(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def classification_case(name, factory, criterion, conv_net):
    from oracle.make_golden import make_batch
    old = torch.load(os.path.join(OUT, name + '.pt'), weights_only=False)
    x, y = make_batch(old['data_seed'], tuple(old['shape']), old['num_classes'], old['soft'])

    def run(dtype, channels_last=False):
        torch.manual_seed(old['model_seed'])
        m = factory(**old['kwargs']).train()
        xi, yi = x, y
        if dtype == torch.float64:
            m = m.double()
            xi = x.double()
            yi = y.double() if y.is_floating_point() else y
        if channels_last:
            m = m.to(memory_format=torch.channels_last)
            xi = xi.contiguous(memory_format=torch.channels_last)
        if dtype == torch.float64:
            with float_is_double():
                lg = m(xi)
                ls = criterion(lg, yi)
            assert ls.dtype == torch.float64 and lg.dtype == torch.float64
        else:
            lg = m(xi)
            ls = criterion(lg, yi)
        ls.backward()
        return lg.detach(), float(ls), _named_grads(m)

    lg32, ls32, g32 = run(torch.float32)
    alt = run(torch.float32, channels_last=True)[2] if conv_net else None
    lg64, ls64, g64 = run(torch.float64)
    summarise(name, g32, g64, alt, {'logits64': lg64.clone(), 'loss64': ls64, 'ref32_logits_l2': _l2(lg32, lg64),
                                    'ref32_loss_rel': abs(ls32 - ls64) / abs(ls64)}, name)


def sam_encoder_case(name='sam_b_encoder_256'):
    from oracle.torch_oracle import sam_randomize_zero_init
    from SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
    old = torch.load(os.path.join(OUT, name + '.pt'), weights_only=False)
    kwargs, batch = old['kwargs'], old['batch']
    g = torch.Generator().manual_seed(old['data_seed'])
    x = torch.randn(batch, 3, kwargs['image_size'], kwargs['image_size'], generator=g)

    def run(dtype, threads=8):
        torch.manual_seed(old['model_seed'])
        m = ViTImageEncoder(**kwargs)
        sam_randomize_zero_init(m.named_parameters(), old['model_seed'] + 100)
        m = m.train()
        gg = torch.Generator().manual_seed(old['data_seed'])
        xi = torch.randn(batch, 3, kwargs['image_size'], kwargs['image_size'], generator=gg)
        if dtype == torch.float64:
            m, xi = m.double(), xi.double()
        torch.set_num_threads(threads)
        out = m(xi)
        probe = torch.randn(out.shape, generator=gg)
        ((out * probe.to(out.dtype)).sum()).backward()
        torch.set_num_threads(8)
        return out.detach(), _named_grads(m)

    o32, g32 = run(torch.float32)
    _, galt = run(torch.float32, threads=3)
    o64, g64 = run(torch.float64)
    assert abs(float(x.double().sum()) - old['input_checksum']) < 1e-6
    summarise(name, g32, g64, galt, {'output64_norm': float(o64.norm()), 'output64_sub': o64[:, ::8, ::4, ::4].clone(),
                                      'ref32_output_l2': _l2(o32, o64)}, name)


def detr_case(name='detr_r50_small'):
    from oracle.make_golden_detr import detr_inputs, zero_dropout
    from oracle.make_golden_r03 import _detection_stand_ins
    _detection_stand_ins()
    from SimpleAICV.detection.models import detr
    from SimpleAICV.detection import losses
    old = torch.load(os.path.join(OUT, name + '.pt'), weights_only=False)
    kwargs = old['kwargs']
    crit = losses.DETRLoss(num_classes=kwargs['num_classes'])
    images, masks, annots = detr_inputs(old['batch'], old['data_seed'], num_classes=kwargs['num_classes'])

    def run(dtype, threads=8, fixed_idx=None):
        torch.manual_seed(old['model_seed'])
        m = detr.__dict__[old['factory']](**kwargs)
        zero_dropout(m)
        m = m.train()
        im, an = images, annots
        if dtype == torch.float64:
            m, im, an = m.double(), images.double(), annots.double()
        torch.set_num_threads(threads)
        orig_match = crit.get_matched_pred_target_idxs
        seen = {}

        def match(*a, **k):
            # the float64 run must price the SAME assignment as the fp32 run (a Hungarian tie broken the other way would be a
            # different loss function, not rounding): record it in fp32, replay it in float64
            if fixed_idx is not None:
                return fixed_idx
            seen['idx'] = orig_match(*a, **k)
            return seen['idx']

        crit.get_matched_pred_target_idxs = match
        try:
            if dtype == torch.float64:
                with float_is_double():
                    cls_out, reg_out = m(im, masks)
                    ld = crit([cls_out, reg_out], an)
                    total = sum(ld.values())
                assert total.dtype == torch.float64
            else:
                cls_out, reg_out = m(im, masks)
                ld = crit([cls_out, reg_out], an)
                total = sum(ld.values())
            total.backward()
        finally:
            crit.get_matched_pred_target_idxs = orig_match
            torch.set_num_threads(8)
        return float(total), _named_grads(m), seen.get('idx')

    t32, g32, idx32 = run(torch.float32)
    _, galt, _ = run(torch.float32, threads=3)
    t64, g64, _ = run(torch.float64, fixed_idx=idx32)
    summarise(name, g32, g64, galt, {'total64': t64, 'ref32_total_rel': abs(t32 - t64) / abs(t64)}, name)


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    from SimpleAICV.classification import backbones, losses
    if not only or 'resnet50_b32_112' in only:
        classification_case('resnet50_b32_112', backbones.resnet50, losses.CELoss(), True)
    if not only or 'vit_base_patch16_b2_224' in only:
        classification_case('vit_base_patch16_b2_224', backbones.vit_base_patch16, losses.OneHotLabelCELoss(), False)
    if not only or 'sam_b_encoder_256' in only:
        sam_encoder_case()
    if not only or 'detr_r50_small' in only:
        detr_case()


if __name__ == '__main__':
    main()
