"""Round-3 fixtures at BASELINE-config shapes, produced by RUNNING THE REFERENCE (imported from /root/reference).

Build container only:   python oracle/make_golden_r03.py [names...]

resnet50_b32_112            reference resnet50 at batch 32, 112 x 112 with the per-parameter disagreement of the reference's
                            own two fp32 runs (contiguous vs channels_last).  VERDICT r02 expected that disagreement to
                            collapse with the batch; it does not (see resnet_reorder_noise_detail), so the whole-model
                            gradient gate of tests/test_gpu_models.py is per parameter: 2 x the reference's own noise for
                            THAT parameter, floor 5e-3, plus the relative L2 distance over all samples.
sam_b_blocks_1024           reference ViTImageEncoder at sam_b's real dimensions on ONE 3 x 1024 x 1024 image with two
                            blocks: block 0 windowed (64 x 64 tokens padded to 70 x 70 -> 25 windows of 196), block 1
                            global (N = 4096 tokens, decomposed rel-pos over 64 x 64, tables of 127) -- BASELINE.json
                            configs[4]'s two attention shapes at bench resolution
                            (reference image_encoder.py:201-239, 82-184).
detr_r50_stem_layer1_1333   reference detr_resnet50backbone conv1 + maxpool1 + layer1 on one 3 x 800 x 1333 image
                            (BASELINE.json configs[3]'s resolution; reference detr_resnet.py:256-340): the s2d stem
                            and the 56-stage convolution shapes at odd, non-tile-aligned extents.

Stored per fixture: output (sub-sampled where large) + its norm, per-parameter gradient norms, 64-element
gradient samples, small gradients in full, BN buffers, and the reference's own reorder / bf16 deviations.
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

SAM_B_2BLOCKS_1024 = dict(image_size=1024, patch_size=16, inplanes=3, embedding_planes=768, block_nums=2, head_nums=12,
                          mlp_ratio=4, out_planes=256, window_size=14, global_attn_indexes=[1],
                          use_gradient_checkpoint=False)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _grads(model):
    norms, samples, full = {}, {}, {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        norms[n] = float(g.norm())
        samples[n] = g.flatten()[:64].clone()
        if g.numel() <= 4096:
            full[n] = g.clone()
    return norms, samples, full


def sam_blocks_1024(name='sam_b_blocks_1024', model_seed=0, data_seed=1):
    from oracle.torch_oracle import sam_randomize_zero_init
    from SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
    kwargs = SAM_B_2BLOCKS_1024

    def build():
        torch.manual_seed(model_seed)
        m = ViTImageEncoder(**kwargs)
        sam_randomize_zero_init(m.named_parameters(), model_seed + 100)
        return m.train()

    g = torch.Generator().manual_seed(data_seed)
    x = torch.randn(1, 3, 1024, 1024, generator=g)
    m = build()
    out = m(x)
    probe = torch.randn(out.shape, generator=g)
    loss = (out * probe).sum()
    loss.backward()
    norms, samples, full = _grads(m)
    m2 = build()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        out16 = m2(x)
    (out16.float() * probe).sum().backward()
    a = torch.cat([p.grad.flatten()[:64].double() for _, p in m2.named_parameters()])
    b = torch.cat([samples[n].double() for n, _ in m2.named_parameters()])
    noise = {'bf16_output': _rel(out16.detach().float(), out.detach()),
             'bf16_grad_sample_cos': float(a @ b / (a.norm() * b.norm()))}
    fx = {'name': name, 'kwargs': kwargs, 'batch': 1, 'model_seed': model_seed, 'data_seed': data_seed,
          'input_checksum': float(x.double().sum()), 'probe_checksum': float(probe.double().sum()),
          'output_shape': list(out.shape), 'output_norm': float(out.norm()),
          'output_sub': out.detach()[:, :, ::2, ::2].clone(),          # [1, 256, 32, 32]
          'output_row': out.detach()[:, :, 37, :].clone(),             # one full token row
          'loss': float(loss), 'grad_norm': norms, 'grad_sample': samples, 'grad_full': full,
          'reference_noise': noise, 'torch_version': torch.__version__}
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: loss={float(loss):.5f} out_norm={float(out.norm()):.4f} noise={noise} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)')


def _detection_stand_ins():
    """cv2 / torchvision are imported at module scope by the reference detection package (dataset code only)."""
    for name in ('cv2', 'torchvision', 'torchvision.ops', 'torchvision.transforms', 'pycocotools', 'pycocotools.coco',
                 'pycocotools.cocoeval'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def detr_stem_layer1(name='detr_r50_stem_layer1_1333', model_seed=0, data_seed=1, h=800, w=1333):
    _detection_stand_ins()
    from SimpleAICV.detection.models.backbones import detr_resnet

    def build():
        torch.manual_seed(model_seed)
        m = detr_resnet.detr_resnet50backbone()
        return m.train()

    def run(m, x):
        return m.layer1(m.maxpool1(m.conv1(x)))

    g = torch.Generator().manual_seed(data_seed)
    x = torch.randn(1, 3, h, w, generator=g)
    m = build()
    out = run(m, x)
    probe = torch.randn(out.shape, generator=g)
    loss = (out * probe).sum()
    loss.backward()
    used = [n for n, p in m.named_parameters() if p.grad is not None]
    norms, samples, full = _grads(m)
    buffers = {n: b.detach().clone() for n, b in m.named_buffers()
               if (n.startswith('conv1.') or n.startswith('layer1.')) and b.numel() <= 4096}
    # the reference's own summation-order noise: the same modules in channels_last
    m2 = build().to(memory_format=torch.channels_last)
    out2 = run(m2, x.contiguous(memory_format=torch.channels_last))
    (out2 * probe).sum().backward()
    worst = 0.0
    for n, p in m2.named_parameters():
        if p.grad is not None and norms[n] > 1e-7:
            worst = max(worst, _rel(p.grad.flatten()[:64], samples[n]))
    # ... and its own bf16 autocast deviation (the gate of the bf16 test)
    m3 = build()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        out3 = run(m3, x)
    (out3.float() * probe).sum().backward()
    a = torch.cat([p.grad.flatten()[:64].double() for n, p in m3.named_parameters() if p.grad is not None])
    b = torch.cat([samples[n].double() for n, p in m3.named_parameters() if p.grad is not None])
    noise = {'fp32_reorder_output': _rel(out2.detach(), out.detach()), 'fp32_reorder_grad_sample': worst,
             'bf16_output': _rel(out3.detach().float(), out.detach()), 'bf16_grad_sample_cos': float(a @ b / (a.norm() * b.norm()))}
    fx = {'name': name, 'h': h, 'w': w, 'model_seed': model_seed, 'data_seed': data_seed,
          'input_checksum': float(x.double().sum()), 'output_shape': list(out.shape), 'output_norm': float(out.norm()),
          'output_sub': out.detach()[:, :, ::8, ::8].clone(), 'output_row': out.detach()[:, :, 101, :].clone(),
          'loss': float(loss), 'used_params': used, 'grad_norm': norms, 'grad_sample': samples, 'grad_full': full,
          'buffers_after': buffers, 'reference_noise': noise, 'torch_version': torch.__version__}
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: out {list(out.shape)} loss={float(loss):.4f} noise={noise} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)')


def resnet_reorder_noise_detail(name, factory, criterion):
    """Adds to an oracle.make_golden fixture HOW the reference's own two fp32 runs (contiguous vs channels_last) differ:
    per-parameter sample error, the relative L2 distance and the cosine of the concatenated gradient samples.
    Measured here: the disagreement does NOT fall with the batch (resnet50: b2 224^2 7.7e-2, b32 112^2 9.3e-2,
    b64 160^2 6.0e-2 worst sample; median 2e-2..3e-2; relative L2 over all 25.5 M gradient values 2.2e-2..2.8e-2), so a
    whole-model gradient gate can be no tighter than this, whatever the batch."""
    from oracle.make_golden import make_batch
    path = os.path.join(OUT, name + '.pt')
    fx = torch.load(path, weights_only=False)
    x, y = make_batch(fx['data_seed'], tuple(fx['shape']), fx['num_classes'], fx['soft'])
    torch.manual_seed(fx['model_seed'])
    m = factory(**fx['kwargs']).train().to(memory_format=torch.channels_last)
    criterion(m(x.contiguous(memory_format=torch.channels_last)), y).backward()
    per = {}
    a, b = [], []
    for n, p in m.named_parameters():
        s = p.grad.detach().flatten()[:64]
        per[n] = _rel(s, fx['grad_sample'][n])
        a.append(s.double())
        b.append(fx['grad_sample'][n].double())
    a, b = torch.cat(a), torch.cat(b)
    fx['reference_noise']['fp32_reorder_per_param'] = per
    fx['reference_noise']['fp32_reorder_sample_l2'] = float((a - b).norm() / b.norm())
    fx['reference_noise']['fp32_reorder_sample_cos'] = float(a @ b / (a.norm() * b.norm()))
    torch.save(fx, path)
    v = sorted(per.values())
    print(f'{name}: reorder noise per parameter: median {v[len(v) // 2]:.3e} worst {v[-1]:.3e}; sample L2 '
          f'{fx["reference_noise"]["fp32_reorder_sample_l2"]:.3e} cos {fx["reference_noise"]["fp32_reorder_sample_cos"]:.6f}')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'resnet50_b32_112' in only:
        from oracle import make_golden
        from SimpleAICV.classification import backbones, losses
        make_golden.run_case('resnet50_b32_112', backbones.resnet50, {'num_classes': 1000}, (32, 3, 112, 112), 1000,
                             losses.CELoss(), False)
        resnet_reorder_noise_detail('resnet50_b32_112', backbones.resnet50, losses.CELoss())
    if not only or 'sam_b_blocks_1024' in only:
        sam_blocks_1024()
    if not only or 'detr_r50_stem_layer1_1333' in only:
        detr_stem_layer1()


if __name__ == '__main__':
    main()
