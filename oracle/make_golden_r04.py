"""Round-4 fixtures, produced by RUNNING THE REFERENCE (imported from /root/reference).  TEST INFRASTRUCTURE.

Build container only:   python oracle/make_golden_r04.py [names...]

auto_rand_augment         reference AutoAugment / RandAugment (classification/auto_rand_augment.py) on seeded uint8 images: every op of
                          the table at two magnitudes, the four AutoAugment policies, RandAugment (uniform / weighted choice,
                          magnitude noise and cap) -- SHA-256 of the output bytes under fixed `random` / numpy seeds.
convbnact_variants        reference classification ConvBnActBlock (resnet.py:19-48) in the forms round 3 refused:
                          has_bn=False (biased convolution, with and without ReLU) and a depthwise block (groups == channels,
                          BatchNorm + ReLU): state_dict, output, input / parameter gradients, BatchNorm buffers after the step.
det_van_convformer        reference detection VANBackbone / MetaFormerBackbone / Dinov3ConvNeXtBackbone
                          (detection/models/backbones/{van,convformer,dinov3convnext}.py) in tiny
                          geometries, training mode: the four stage outputs, parameter gradients, BatchNorm buffers after the step.
dinov3_detectors          reference dinov3_vit_retinanet.RetinaNet / dinov3_vit_fcos.FCOS (detection/models/) on a two-block DINOv3 trunk
                          registered under a test name: per-level head outputs, parameter gradients.
dinov3_tiny               reference DinoVisionTransformer (detection/models/backbones/dinov3vit.py): GELU-MLP form in training mode
                          (RoPE rescale draw) and SwiGLU form in eval mode; outputs, input and parameter gradients.
random_erasing            reference RandomErasing (classification/common.py:561-640), modes const / rand / pixel, seeded numpy
                          draws: the erased images (the loader-side transform the ViT fine-tuning configs use).
sam_block_relpos_resized  reference segment_anything Block (image_encoder.py:201-239) whose relative-position tables were built
                          for an 8 x 8 grid, run on a 16 x 16 grid: get_rel_pos interpolates the 15-row tables to 31 rows
                          (image_encoder.py:96-103) -- the path ops_tfm.resize_rel_pos restates; output, input gradient and
                          every parameter gradient (incl. the two tables, whose gradient flows back through the interpolation).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)


def convbnact_variants(name='convbnact_variants'):
    from SimpleAICV.classification.backbones.resnet import ConvBnActBlock
    cases = {}
    specs = [('bias_relu_3x3', dict(inplanes=16, planes=32, kernel_size=3, stride=1, padding=1, groups=1, has_bn=False, has_act=True), 16),
             ('bias_only_1x1_s2', dict(inplanes=32, planes=24, kernel_size=1, stride=2, padding=0, groups=1, has_bn=False, has_act=False), 32),
             ('depthwise_bn_relu', dict(inplanes=32, planes=32, kernel_size=3, stride=1, padding=1, groups=32, has_bn=True, has_act=True), 32)]
    for i, (key, kw, cin) in enumerate(specs):
        torch.manual_seed(10 + i)
        blk = ConvBnActBlock(**kw).train()
        with torch.no_grad():                    # non-trivial biases / affine parameters (the defaults are 0 / 1)
            for n, p in blk.named_parameters():
                if p.dim() == 1:
                    p.copy_(torch.randn_like(p) * 0.3 + (1.0 if n == 'layer.1.weight' else 0.0))
        g = torch.Generator().manual_seed(100 + i)
        x = torch.randn(4, cin, 12, 12, generator=g, requires_grad=True)
        sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
        out = blk(x)
        probe = torch.randn(out.shape, generator=g)
        (out * probe).sum().backward()
        cases[key] = {'kwargs': kw, 'state_dict': sd, 'x': x.detach().clone(), 'probe': probe, 'out': out.detach().clone(),
                      'dx': x.grad.clone(), 'grads': {n: p.grad.clone() for n, p in blk.named_parameters()},
                      'buffers_after': {n: b.detach().clone() for n, b in blk.named_buffers()}}
        print(key, [(n, tuple(p.shape)) for n, p in blk.named_parameters()], float(out.norm()))
    path = os.path.join(OUT, name + '.pt')
    torch.save({'name': name, 'cases': cases, 'torch_version': torch.__version__}, path)
    print(f'-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def sam_block_relpos_resized(name='sam_block_relpos_resized'):
    from oracle.torch_oracle import sam_randomize_zero_init
    from SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import Block
    torch.manual_seed(3)
    blk = Block(inplanes=128, head_nums=2, mlp_ratio=4.0, input_size=(8, 8), window_size=0).train()
    sam_randomize_zero_init(blk.named_parameters(), 103)         # the tables are zero-initialised in the reference
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 16, 128, generator=g, requires_grad=True)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    assert sd['attn.rel_pos_h'].shape[0] == 15
    out = blk(x)
    probe = torch.randn(out.shape, generator=g)
    (out * probe).sum().backward()
    fx = {'name': name, 'state_dict': sd, 'x': x.detach().clone(), 'probe': probe, 'out': out.detach().clone(), 'dx': x.grad.clone(),
          'grads': {n: p.grad.clone() for n, p in blk.named_parameters()}, 'torch_version': torch.__version__}
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: out norm {float(out.norm()):.4f}, d rel_pos_h norm {float(fx["grads"]["attn.rel_pos_h"].norm()):.4e} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)')


def random_erasing(name='random_erasing'):
    """reference RandomErasing (SimpleAICV/classification/common.py:561-640) on seeded float32 HWC images: all three modes, prob
    0.7 (both branches over the seeds), one configuration with up to three boxes; per case the numpy seed, the constructor
    arguments and the erased image (the input is regenerated from the seed by the test)."""
    import types
    import numpy as np
    for mod in ('cv2', 'torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    tv = sys.modules['torchvision.transforms']
    for attr in ('ToTensor', 'Normalize', 'Compose'):
        if not hasattr(tv, attr):
            setattr(tv, attr, lambda *a, **k: None)
    sys.modules['torchvision'].transforms = tv
    from SimpleAICV.classification.common import RandomErasing
    cases = []
    for mode in ('const', 'rand', 'pixel'):
        for kw in (dict(prob=0.7, mode=mode), dict(prob=0.7, mode=mode, min_count=1, max_count=4, max_area=0.2)):
            for seed in range(6):
                np.random.seed(1000 + seed)
                image = np.random.standard_normal((40, 48, 3)).astype(np.float32)
                before = image.copy()
                out = RandomErasing(**kw)({'image': image, 'label': 3})
                cases.append({'kwargs': kw, 'seed': 1000 + seed, 'image': torch.from_numpy(out['image'].copy()),
                              'changed': int((out['image'] != before).any(axis=-1).sum())})
    print(name, len(cases), 'cases; erased pixels per case', [c['changed'] for c in cases])
    path = os.path.join(OUT, name + '.pt')
    torch.save({'name': name, 'cases': cases}, path)
    print(f'-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def dinov3_tiny(name='dinov3_tiny'):
    """reference DinoVisionTransformer (SimpleAICV/detection/models/backbones/dinov3vit.py:453-571) in two tiny geometries
    (embedding 128 = 2 heads of 64, 2 blocks, patch 16): 'mlp_train' -- GELU MLP, training mode with the RoPE rescale augmentation
    (one host draw: torch.manual_seed(5) right before the forward), image 64 x 48; 'swiglu_eval' -- SwiGLU FFN (ratio 6), eval
    mode, image 96 x 64.  LayerScale gammas (1e-5 at init: the branches would vanish) are redrawn around 1, biases around 0.  Per
    case: parameter checksums (the test rebuilds the weights from the seeds), output [B, C, H, W], input gradient, norm + first 64
    entries of every parameter gradient for a random probe (all from generator 21 in a fixed order)."""
    import types
    for mod in ('cv2', 'torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    from SimpleAICV.detection.models.backbones.dinov3vit import DinoVisionTransformer
    cases = {}
    for key, kw, train, hw in (('mlp_train', dict(embedding_planes=128, head_nums=2, block_nums=2, ffn_layer='mlp', ffn_ratio=4,
                                                   pos_embed_rope_rescale_coords=2), True, (64, 48)),
                               ('swiglu_eval', dict(embedding_planes=128, head_nums=2, block_nums=2, ffn_layer='swiglu', ffn_ratio=6,
                                                    pos_embed_rope_rescale_coords=2), False, (96, 64))):
        torch.manual_seed(0)
        m = DinoVisionTransformer(**kw)
        g = torch.Generator().manual_seed(21)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith('.gamma'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3 + 1.0)
                elif n.endswith('.bias'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        m.train(train)
        x = torch.randn(2, 3, hw[0], hw[1], generator=g, requires_grad=True)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        torch.manual_seed(5)                      # the rope augmentation's draw (training mode only)
        out = m(x)
        probe = torch.randn(out.shape, generator=g)
        (out * probe).sum().backward()
        # the weights are NOT stored: the test rebuilds them from the same seeds (construction draw order is part of the contract);
        # per-parameter checksums pin that, gradients are stored as norm + first 64 entries
        cases[key] = {'kwargs': kw, 'train': train, 'hw': hw, 'param_sum': {k: float(v.double().sum()) for k, v in sd.items()},
                      'param_abs_sum': {k: float(v.double().abs().sum()) for k, v in sd.items()},
                      'out': out.detach().clone(), 'dx': x.grad.clone(), 'input_checksum': float(x.detach().double().sum()),
                      'grad_norm': {n: float(p.grad.norm()) for n, p in m.named_parameters()},
                      'grad_sample': {n: p.grad.flatten()[:64].clone() for n, p in m.named_parameters()}}
        print(key, tuple(out.shape), float(out.norm()), 'qkv bias grad (k third must be 0):',
              float(m.blocks[0].attn.qkv.bias.grad[128:256].abs().max()))
    path = os.path.join(OUT, name + '.pt')
    torch.save({'name': name, 'cases': cases, 'torch_version': torch.__version__}, path)
    print(f'-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def det_van_convformer(name='det_van_convformer'):
    """reference VANBackbone (SimpleAICV/detection/models/backbones/van.py:32-130) and MetaFormerBackbone
    (.../convformer.py:29-117) and Dinov3ConvNeXtBackbone (.../dinov3convnext.py:120-199), four stages of widths 16..128 / 32..128,
    depths [1, 1, 2, 1], drop-path 0, training mode, image 4 x 3 x 256 x 256: the LAST stage then normalises over 4 x 8 x 8 = 256
    samples per BatchNorm channel (r05; the r04 fixture had 12, where one rounded activation moves a channel's statistics and the
    bf16 gates had to be opened to 50 %).  Layer scales (1e-2 at init in VAN, 1e-6 gammas in ConvNeXt) are redrawn around 0.5 and
    biases around 0 from generator 33 so every branch carries gradient.  Per net: parameter checksums (the test rebuilds the weights
    from the seeds), norm + a strided sample (<= 16384 entries) of the four stage outputs, norm + first 64 entries of every
    parameter gradient for one random probe per output, BatchNorm buffers after the forward, and -- `bf16_drift` -- how far the
    REFERENCE ITSELF moves under torch.autocast('cpu', bfloat16) on the same weights, input and probes (outputs, gradient norms,
    gradient samples, buffers): the yardstick the bf16 test of the HIP path is held to."""
    import types
    for mod in ('cv2', 'torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    from SimpleAICV.detection.models.backbones.van import VANBackbone
    from SimpleAICV.detection.models.backbones.convformer import MetaFormerBackbone
    from SimpleAICV.detection.models.backbones.dinov3convnext import Dinov3ConvNeXtBackbone

    def sample(t):
        f = t.detach().flatten()
        return f[::max(1, f.numel() // 16384)][:16384].clone()

    def build(cls, kw):
        torch.manual_seed(0)
        m = cls(**kw)
        g = torch.Generator().manual_seed(33)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if 'layer_scale' in n or n.endswith('.scale') or n.endswith('.gamma'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.2 + 0.5)
                elif n.endswith('.bias'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        m.train()
        x = torch.randn(4, 3, 256, 256, generator=g)
        return m, x, g

    cases = {}
    for key, cls, kw in (('van', VANBackbone, dict(embedding_planes=[16, 32, 64, 128], mlp_ratios=[8, 8, 4, 4], block_nums=[1, 1, 2, 1])),
                         ('convformer', MetaFormerBackbone, dict(embedding_planes=[32, 64, 96, 128], block_nums=[1, 1, 2, 1])),
                         ('dinov3convnext', Dinov3ConvNeXtBackbone, dict(embedding_planes=[32, 64, 96, 128], block_nums=[1, 1, 2, 1]))):
        m, x, g = build(cls, kw)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        outs = m(x)
        probes = [torch.randn(o.shape, generator=g) for o in outs]
        sum((o * p).sum() for o, p in zip(outs, probes)).backward()
        after = m.state_dict()
        # the reference's own bf16 drift: same weights / input / probes under CPU autocast
        mb, xb, _ = build(cls, kw)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            outs_b = mb(xb)
        sum((o.float() * p).sum() for o, p in zip(outs_b, probes)).backward()
        after_b = mb.state_dict()
        gn = {n: float(p.grad.norm()) for n, p in m.named_parameters()}
        top = max(gn.values())
        drift = {'outs': [float((ob.float() - o).norm() / o.norm()) for ob, o in zip(outs_b, outs)], 'grad_norm': {}, 'grad_sample': {},
                 'buffers': {k: float((after_b[k].float() - after[k]).abs().max() / (after[k].abs().max() + 1e-12)) for k in after if 'running_' in k}}
        for (n, p), (_, pb) in zip(m.named_parameters(), mb.named_parameters()):
            drift['grad_norm'][n] = abs(float(pb.grad.float().norm()) - gn[n]) / max(gn[n], 1e-12)
            ref = p.grad.flatten()[:64]
            scale = max(float(ref.abs().max()), 1e-2 * gn[n])
            drift['grad_sample'][n] = float((pb.grad.float().flatten()[:64] - ref).abs().max()) / max(scale, 1e-12)
        cases[key] = {'kwargs': kw, 'param_sum': {k: float(v.double().sum()) for k, v in sd.items()},
                      'param_abs_sum': {k: float(v.double().abs().sum()) for k, v in sd.items()},
                      'input_shape': tuple(x.shape), 'input_checksum': float(x.double().sum()),
                      'out_shapes': [tuple(o.shape) for o in outs], 'out_norm': [float(o.norm()) for o in outs],
                      'out_sample': [sample(o) for o in outs],
                      'buffers_after': {k: after[k].detach().clone() for k in after if 'running_' in k},
                      'grad_norm': gn, 'grad_sample': {n: p.grad.flatten()[:64].clone() for n, p in m.named_parameters()},
                      'bf16_drift': drift}
        live = [n for n in gn if gn[n] > 1e-3]
        print(key, [tuple(o.shape) for o in outs], [round(float(o.norm()), 3) for o in outs], len(sd), 'state entries;',
              'reference bf16 drift: outs', [f'{v:.1e}' for v in drift['outs']],
              'grad norm max %.2e median %.2e' % (max(drift['grad_norm'][n] for n in live), sorted(drift['grad_norm'][n] for n in live)[len(live) // 2]),
              'grad sample max %.2e median %.2e' % (max(drift['grad_sample'][n] for n in live), sorted(drift['grad_sample'][n] for n in live)[len(live) // 2]),
              'buffers max %.2e' % max(drift['buffers'].values(), default=0.0))
    path = os.path.join(OUT, name + '.pt')
    torch.save({'name': name, 'cases': cases, 'torch_version': torch.__version__}, path)
    print(f'-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def _augment_image(seed, h=48, w=64):
    """a uint8 RGB test card: smooth ramps + a bright box + noise, so that geometric, histogram and colour ops all change it"""
    import numpy as np
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([xx * 255 // (w - 1), yy * 255 // (h - 1), (xx + yy) * 255 // (h + w - 2)], axis=-1).astype(np.int64)
    img[h // 4:h // 2, w // 3:w // 2] = (240, 30, 120)
    return np.clip(img + rng.randint(-20, 21, size=img.shape), 0, 255).astype(np.uint8)


def auto_rand_augment(name='auto_rand_augment'):
    """reference SimpleAICV/classification/auto_rand_augment.py (loaded from its file: no package imports needed).  Cases:
    'ops'   every entry of NAME_TO_OP as AugmentOp(name, prob=1, magnitude m, hparams{translate_const 28, img_mean, magnitude_std .5})
            for m in (3, 9), random.seed(100 + index) before the call;
    'auto'  AutoAugment(policy, resize=64) for the four policies x 6 samples, random.seed(200 + k);
    'rand'  RandAugment in four configurations x 6 samples, random.seed(300 + k), np.random.seed(400 + k).
    Stored per case: SHA-256 of the output image bytes + its mean (a mismatch then says how far off)."""
    import hashlib
    import importlib.util
    import random
    import numpy as np
    from PIL import Image
    spec = importlib.util.spec_from_file_location('ref_auto_rand_augment', os.path.join(REF, 'SimpleAICV/classification/auto_rand_augment.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    def digest(img):
        a = np.asarray(img)
        return {'sha256': hashlib.sha256(a.tobytes()).hexdigest(), 'mean': float(a.mean()), 'shape': list(a.shape)}

    out = {'ops': [], 'auto': [], 'rand': []}
    hp = dict(translate_const=28, img_mean=(124, 116, 104), magnitude_std=0.5)
    for i, opname in enumerate(sorted(ref.NAME_TO_OP)):
        for m in (3, 9):
            random.seed(100 + i)
            res = ref.AugmentOp(opname, prob=1.0, magnitude=m, hparams=hp)(Image.fromarray(_augment_image(i)))
            out['ops'].append({'name': opname, 'magnitude': m, 'image_seed': i, 'seed': 100 + i, **digest(res)})
    for policy in ('original', 'originalr', 'v0', 'v0r'):
        aug = ref.AutoAugment(policy, resize=64, magnitude_std=0.5 if policy.endswith('r') else None)
        for k in range(6):
            random.seed(200 + k)
            res = aug({'image': Image.fromarray(_augment_image(50 + k)), 'label': k})
            out['auto'].append({'policy': policy, 'image_seed': 50 + k, 'seed': 200 + k, **digest(res['image'])})
    for kw in (dict(), dict(integer=False, weight_idx=0, num_layers=3), dict(magnitude=14, magnitude_max=20, magnitude_std=float('inf')),
               dict(magnitude=5, magnitude_std=0, num_layers=4)):
        aug = ref.RandAugment(resize=64, **kw)
        for k in range(6):
            random.seed(300 + k)
            np.random.seed(400 + k)
            res = aug({'image': Image.fromarray(_augment_image(80 + k)), 'label': k})
            out['rand'].append({'kwargs': {a: (str(b) if isinstance(b, float) and b == float('inf') else b) for a, b in kw.items()},
                                'image_seed': 80 + k, 'seed': 300 + k, 'np_seed': 400 + k, **digest(res['image'])})
    import json
    import PIL
    out['pil_version'] = PIL.__version__
    path = os.path.join(OUT, name + '.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=0)
    base = [hashlib.sha256(_augment_image(i).tobytes()).hexdigest() for i in range(100)]
    changed = sum(c['sha256'] not in base for grp in ('ops', 'auto', 'rand') for c in out[grp])
    print(name, {g: len(out[g]) for g in ('ops', 'auto', 'rand')}, f'{changed} outputs differ from their input', f'-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


TINY_DINOV3 = dict(embedding_planes=128, head_nums=2, block_nums=2, ffn_layer='mlp', ffn_ratio=4, pos_embed_rope_rescale_coords=2)


def dinov3_detectors(name='dinov3_detectors'):
    """reference RetinaNet (SimpleAICV/detection/models/dinov3_vit_retinanet.py:28-112) and FCOS (dinov3_vit_fcos.py:28-101) with
    planes 64, 6 classes, on a DINOv3 trunk of two blocks (TINY_DINOV3) registered in the backbones namespace as
    'tiny_dinov3_backbone' (the shipped factories start at 384 channels x 12 blocks), training mode, image 2 x 3 x 128 x 96;
    torch.manual_seed(5) right before the forward (the trunk's RoPE rescale draw).  LayerScale gammas redrawn around 1 and biases
    around 0 from generator 44.  Stored: parameter checksums, every level's outputs, norm + first 64 entries of every parameter
    gradient for one random probe per output."""
    import types
    for mod in ('cv2', 'torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    from SimpleAICV.detection.models import backbones, dinov3_vit_fcos, dinov3_vit_retinanet
    from SimpleAICV.detection.models.backbones.dinov3vit import DinoVisionTransformer
    backbones.__dict__['tiny_dinov3_backbone'] = lambda pretrained_path='', **kw: DinoVisionTransformer(**TINY_DINOV3, **kw)
    cases = {}
    for key, build in (('retinanet', lambda: dinov3_vit_retinanet.RetinaNet('tiny_dinov3_backbone', planes=64, num_classes=6)),
                       ('fcos', lambda: dinov3_vit_fcos.FCOS('tiny_dinov3_backbone', planes=64, num_classes=6))):
        torch.manual_seed(0)
        m = build()
        g = torch.Generator().manual_seed(44)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith('.gamma'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3 + 1.0)
                elif n.endswith('.bias') and 'cls_out' not in n and 'cls_head' not in n:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        m.train()
        x = torch.randn(2, 3, 128, 96, generator=g)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        torch.manual_seed(5)
        outs = m(x)
        flat = [o for group in outs for o in group]
        probes = [torch.randn(o.shape, generator=g) for o in flat]
        sum((o * p).sum() for o, p in zip(flat, probes)).backward()
        cases[key] = {'param_sum': {k: float(v.double().sum()) for k, v in sd.items()},
                      'param_abs_sum': {k: float(v.double().abs().sum()) for k, v in sd.items()},
                      'input_checksum': float(x.double().sum()), 'groups': [len(group) for group in outs],
                      'outs': [o.detach().clone() for o in flat],
                      'grad_norm': {n: float(p.grad.norm()) for n, p in m.named_parameters() if p.grad is not None},
                      'grad_sample': {n: p.grad.flatten()[:64].clone() for n, p in m.named_parameters() if p.grad is not None},
                      'no_grad': [n for n, p in m.named_parameters() if p.grad is None]}
        print(key, [tuple(o.shape) for o in flat][:5], len(sd), 'state entries; without gradient:', cases[key]['no_grad'])
    path = os.path.join(OUT, name + '.pt')
    torch.save({'name': name, 'cases': cases, 'trunk': TINY_DINOV3}, path)
    print(f'-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'convbnact_variants' in only:
        convbnact_variants()
    if not only or 'auto_rand_augment' in only:
        auto_rand_augment()
    if not only or 'sam_block_relpos_resized' in only:
        sam_block_relpos_resized()
    if not only or 'random_erasing' in only:
        random_erasing()
    if not only or 'dinov3_tiny' in only:
        dinov3_tiny()
    if not only or 'det_van_convformer' in only:
        det_van_convformer()
    if not only or 'dinov3_detectors' in only:
        dinov3_detectors()


if __name__ == '__main__':
    main()
