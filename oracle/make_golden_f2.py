"""Fixtures for the backbones that REUSE the hot-path blocks, produced by running the reference:
  det_resnet50backbone : SimpleAICV/detection/models/backbones/resnet.py resnet50backbone on a [2,3,64,64] batch -> C2..C5
  mae_tiny             : SimpleAICV/masked_image_modeling/models/vit_mae.py VITMAEPretrainModel (encoder 128 planes x 2
                         heads of 64, decoder 64 planes x 2 heads of 32, image 64, patch 16) + losses.MSELoss
  det_vitbackbone_tiny : SimpleAICV/detection/models/backbones/vit.py ViTBackbone (128 planes x 2 heads of 64, 2 blocks,
                         image 64, patch 16) -> [B, 128, 4, 4] and VitPyramidNeck(128, 64) -> P2..P5
Build container only:   python oracle/make_golden_f2.py"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')

MAE_TINY = dict(patch_size=16, image_size=64, mask_ratio=0.75, encoder_embedding_planes=128, encoder_block_nums=2,
                encoder_head_nums=2, decoder_embedding_planes=64, decoder_block_nums=2, decoder_head_nums=2)


def _grads(m):
    norms, samples = {}, {}
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        norms[n] = float(g.norm())
        samples[n] = g.flatten()[:64].clone()
    return norms, samples


def backbone_case():
    from SimpleAICV.detection.models.backbones.resnet import resnet50backbone
    torch.manual_seed(0)
    m = resnet50backbone().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 64, 3, generator=g).permute(0, 3, 1, 2)
    outs = m(x)
    probes = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * p).sum() for o, p in zip(outs, probes)).backward()
    norms, samples = _grads(m)
    # how far the reference moves from ITSELF under another fp32 summation order (true-NCHW input, 3 threads): batch-2
    # BatchNorm backward and ReLU sign flips at near-zero pre-activations amplify rounding differences
    torch.manual_seed(0)
    m2 = resnet50backbone().train()
    torch.set_num_threads(3)
    outs2 = m2(x.contiguous())
    sum((o * p).sum() for o, p in zip(outs2, probes)).backward()
    torch.set_num_threads(8)
    _, samples2 = _grads(m2)
    noise = max(float((samples2[n].double() - samples[n].double()).abs().max() / samples[n].double().abs().max().clamp_min(1e-30))
                for n in samples if norms[n] > 1e-7)
    print('det_resnet50backbone reference reorder noise on gradient samples', noise)
    fx = {'model_seed': 0, 'data_seed': 1, 'shape': (2, 64, 64, 3), 'reference_noise': {'fp32_reorder_grad_sample': noise}, 'outputs': [o.detach().clone() for o in outs],
          'out_channels': m.out_channels, 'grad_norm': norms, 'grad_sample': samples,
          'buffers_after': {n: b.detach().clone() for n, b in m.named_buffers() if b.numel() <= 2048},
          'input_checksum': float(x.double().sum())}
    torch.save(fx, os.path.join(OUT, 'det_resnet50backbone.pt'))
    print('det_resnet50backbone', [tuple(o.shape) for o in outs])


def mae_case():
    from SimpleAICV.masked_image_modeling.models.vit_mae import VITMAEPretrainModel
    from SimpleAICV.masked_image_modeling.losses import MSELoss
    torch.manual_seed(0)
    m = VITMAEPretrainModel(**MAE_TINY).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 64, 64, generator=g)
    torch.manual_seed(77)                       # the forward's only random draw: torch.rand(B, L) in random_masking
    pred, mask = m(x)
    loss = MSELoss()(pred, m.images_to_patch(x), mask)
    loss.backward()
    norms, samples = _grads(m)
    fx = {'kwargs': MAE_TINY, 'model_seed': 0, 'data_seed': 1, 'noise_seed': 77, 'batch': 4, 'pred': pred.detach().clone(),
          'mask': mask.clone(), 'loss': float(loss), 'grad_norm': norms, 'grad_sample': samples,
          'input_checksum': float(x.double().sum())}
    torch.save(fx, os.path.join(OUT, 'mae_tiny.pt'))
    print('mae_tiny loss', float(loss), 'pred', tuple(pred.shape), 'removed', int(mask.sum()))


VITDET_TINY = dict(patch_size=16, embedding_planes=128, block_nums=2, head_nums=2, feedforward_ratio=4, image_size=64)


def vitdet_case():
    from SimpleAICV.detection.models.backbones.vit import ViTBackbone, VitPyramidNeck
    torch.manual_seed(0)
    m = ViTBackbone(**VITDET_TINY).train()
    neck = VitPyramidNeck(128, 64).train()
    init = {('backbone.' + n): (float(p.double().sum()), float(p.double().abs().sum())) for n, p in m.named_parameters()}
    init.update({('neck.' + n): (float(p.double().sum()), float(p.double().abs().sum())) for n, p in neck.named_parameters()})
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 64, 64, 3, generator=g).permute(0, 3, 1, 2)
    feat = m(x)
    outs = neck(feat)
    probes = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * p).sum() for o, p in zip(outs, probes)).backward()
    nb, sb = _grads(m)
    nn_, sn = _grads(neck)
    fx = {'kwargs': VITDET_TINY, 'neck': (128, 64), 'model_seed': 0, 'data_seed': 1, 'shape': (3, 64, 64, 3),
          'init_checksums': init, 'feature': feat.detach().clone(), 'outputs': [o.detach().clone() for o in outs],
          'grad_norm': {**{'backbone.' + k: v for k, v in nb.items()}, **{'neck.' + k: v for k, v in nn_.items()}},
          'grad_sample': {**{'backbone.' + k: v for k, v in sb.items()}, **{'neck.' + k: v for k, v in sn.items()}},
          'input_checksum': float(x.double().sum())}
    torch.save(fx, os.path.join(OUT, 'det_vitbackbone_tiny.pt'))
    print('det_vitbackbone_tiny', tuple(feat.shape), [tuple(o.shape) for o in outs])


def main():
    for name in ('cv2', 'torchvision', 'torchvision.transforms'):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    if not only or 'det_resnet50backbone' in only:
        backbone_case()
    if not only or 'mae_tiny' in only:
        mae_case()
    if not only or 'det_vitbackbone_tiny' in only:
        vitdet_case()


if __name__ == '__main__':
    main()
