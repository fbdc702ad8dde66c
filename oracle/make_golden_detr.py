"""Generates tests/golden/detr_*.pt by RUNNING THE REFERENCE DETR + DETRLoss (imported from /root/reference).

Build container only:   python oracle/make_golden_detr.py

The reference detection package imports cv2 / torchvision at module scope (dataset code only); neither is in
this image, so empty stand-in modules are registered before the import -- the model and the loss never touch
them.  detr_r18_tiny: resnet18_detr, 20 classes, 20 queries, batch 4 of 192x256 images on a 256x256 canvas
(right / bottom padding -> a non-trivial key-padding mask, applied by the reference as a +1.0 additive bias),
3..5 boxes per image.  Every dropout probability is set to 0 (the masks are RNG-stream specific); the
product-side test does the same.  Stored: class logits / boxes of all 6 decoder layers, the 18 loss terms,
the Hungarian assignment, per-parameter gradient norms + samples, BN buffers, and the reference's own bf16
autocast deviation.
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

DETR_TINY = dict(hidden_inplanes=256, query_nums=20, num_classes=20)
# resnet50_detr as the reference config builds it (res50_detr_yoloresize1024/train_config.py:28-31: 80 classes, 100
# queries, the real ResNet-50 backbone with its 23 conv shapes) on a small canvas, so the CPU reference stays cheap
DETR_R50 = dict(hidden_inplanes=256, query_nums=100, num_classes=80)


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def detr_inputs(batch, data_seed, canvas=256, h=192, w=256, num_classes=20, max_annots=100):
    """Seeded batch in the DETRDetectionCollater contract (detection/common.py:291-363)."""
    g = torch.Generator().manual_seed(data_seed)
    images = torch.zeros(batch, canvas, canvas, 3)
    images[:, :h, :w, :] = torch.randn(batch, h, w, 3, generator=g)
    masks = torch.ones(batch, canvas, canvas, dtype=torch.bool)
    masks[:, :h, :w] = False
    annots = -torch.ones(batch, max_annots, 5)
    for b in range(batch):
        n = 3 + b % 3
        cxcy = torch.rand(n, 2, generator=g) * 0.5 + 0.25
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.1
        cls = torch.randint(0, num_classes, (n, 1), generator=g).float()
        annots[b, :n] = torch.cat([cxcy, wh, cls], dim=1)
    return images.permute(0, 3, 1, 2), masks, annots        # NCHW view over NHWC memory, as the collater returns


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def detr_case(name, factory_name, kwargs, batch, model_seed=0, data_seed=1):
    from SimpleAICV.detection.models import detr
    from SimpleAICV.detection import losses

    def build():
        torch.manual_seed(model_seed)
        m = detr.__dict__[factory_name](**kwargs)
        zero_dropout(m)
        return m.train()

    crit = losses.DETRLoss(num_classes=kwargs['num_classes'])
    images, masks, annots = detr_inputs(batch, data_seed, num_classes=kwargs['num_classes'])
    m = build()
    cls_out, reg_out = m(images, masks)
    with torch.no_grad():
        idx = crit.get_matched_pred_target_idxs(cls_out[-1].float(), torch.clamp(reg_out[-1], 1e-4, 1 - 1e-4).float(),
                                                annots)
    ld = crit([cls_out, reg_out], annots)
    total = sum(ld.values())
    total.backward()
    norms, samples = {}, {}
    for n, p in m.named_parameters():
        gr = p.grad.detach()
        norms[n] = float(gr.norm())
        samples[n] = gr.flatten()[:64].clone()
    buffers = {n: b.detach().clone() for n, b in m.named_buffers() if b.numel() <= 4096}
    m2 = build()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        c16, r16 = m2(images, masks)
        ld16 = crit([c16, r16], annots)
    sum(ld16.values()).backward()
    a = torch.cat([p.grad.flatten()[:64].double() for _, p in m2.named_parameters()])
    b = torch.cat([samples[n].double() for n, _ in m2.named_parameters()])
    # fp32 self-noise: the same model on true-NCHW input with another thread count (other ATen kernels / summation order)
    m3 = build()
    torch.set_num_threads(3)
    c3, r3 = m3(images.contiguous(), masks)
    sum(crit([c3, r3], annots).values()).backward()
    torch.set_num_threads(8)
    reorder = max(_rel(p.grad.flatten()[:64], samples[n]) for n, p in m3.named_parameters() if norms[n] > 1e-7)
    noise = {'fp32_reorder_grad_sample': reorder, 'bf16_cls': _rel(c16.detach().float(), cls_out.detach()), 'bf16_reg': _rel(r16.detach().float(), reg_out.detach()),
             'bf16_loss': abs(float(sum(ld16.values())) - float(total)) / abs(float(total)),
             'bf16_grad_sample_cos': float(a @ b / (a.norm() * b.norm()))}
    fx = {'name': name, 'factory': factory_name, 'kwargs': kwargs, 'batch': batch, 'model_seed': model_seed,
          'data_seed': data_seed, 'input_checksum': float(images.double().sum() + annots.double().sum()),
          'cls_outputs': cls_out.detach().clone(), 'reg_outputs': reg_out.detach().clone(),
          'loss': {k: float(v) for k, v in ld.items()}, 'total': float(total),
          'indices': [(i.clone(), j.clone()) for i, j in idx],
          'grad_norm': norms, 'grad_sample': samples, 'buffers_after': buffers, 'reference_noise': noise,
          'torch_version': torch.__version__}
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: total={float(total):.5f} noise={noise} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    for name in ('cv2', 'torchvision', 'torchvision.transforms'):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'detr_r18_tiny' in only:
        detr_case('detr_r18_tiny', 'resnet18_detr', DETR_TINY, batch=4)
    if not only or 'detr_r50_small' in only:
        detr_case('detr_r50_small', 'resnet50_detr', DETR_R50, batch=2)


if __name__ == '__main__':
    main()
