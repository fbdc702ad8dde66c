"""Generates tests/golden/sam_*.pt by RUNNING THE REFERENCE SAM modules (imported from /root/reference).

Build container only:   python oracle/make_golden_sam.py

sam_encoder_tiny: ViTImageEncoder (reference interactive_segmentation/models/segment_anything/
image_encoder.py:259) at image 256, 2 heads x 64, 3 blocks (windowed 7x7 with padding 16 -> 21, one
global block of 256 tokens), neck to 32 planes.  All-zero parameters (pos_embed, rel_pos_*) are
overwritten by oracle.torch_oracle.sam_randomize_zero_init so the relative-position path is live.
The scalar is sum(out * probe) with a seeded probe; stored: output, per-parameter gradient norms,
64-element samples, full small gradients, input-gradient checksum, and the reference's own bf16
autocast deviation (tolerance evidence).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

ENC_TINY = dict(image_size=256, patch_size=16, inplanes=3, embedding_planes=128, block_nums=3, head_nums=2,
                mlp_ratio=4, out_planes=32, window_size=7, global_attn_indexes=[1], use_gradient_checkpoint=False)


# SAM ViT-B image encoder at its REAL dimensions (sam_b, reference segment_anything/sam.py: 768 planes, 12 blocks,
# 12 heads x 64, window 14, global blocks 2/5/8/11, neck to 256) on a 256 x 256 image: 16 x 16 tokens, i.e. the
# windowed blocks pad 16 -> 28 (2 x 2 windows of 196 tokens, rel-pos tables of 27) and the global blocks run the
# decomposed rel-pos bias at Sw = 16 (tables of 31) -- BASELINE.json configs[4]'s kernels at head dim 64 x 12 heads.
ENC_B_256 = dict(image_size=256, patch_size=16, inplanes=3, embedding_planes=768, block_nums=12, head_nums=12,
                 mlp_ratio=4, out_planes=256, window_size=14, global_attn_indexes=[2, 5, 8, 11],
                 use_gradient_checkpoint=False)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def encoder_case(name, kwargs, batch, model_seed=0, data_seed=1):
    from oracle.torch_oracle import sam_randomize_zero_init
    from SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder

    def build():
        torch.manual_seed(model_seed)
        m = ViTImageEncoder(**kwargs)
        sam_randomize_zero_init(m.named_parameters(), model_seed + 100)
        return m.train()

    g = torch.Generator().manual_seed(data_seed)
    x = torch.randn(batch, 3, kwargs['image_size'], kwargs['image_size'], generator=g)
    m = build()
    out = m(x)
    probe = torch.randn(out.shape, generator=g)
    loss = (out * probe).sum()
    loss.backward()
    norms, samples, full = {}, {}, {}
    for n, p in m.named_parameters():
        gr = p.grad.detach()
        norms[n] = float(gr.norm())
        samples[n] = gr.flatten()[:64].clone()
        if gr.numel() <= 4096:
            full[n] = gr.clone()
    m2 = build()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        out16 = m2(x)
    (out16.float() * probe).sum().backward()
    a = torch.cat([p.grad.flatten()[:64].double() for _, p in m2.named_parameters()])
    b = torch.cat([samples[n].double() for n, _ in m2.named_parameters()])
    noise = {'bf16_output': _rel(out16.detach().float(), out.detach()),
             'bf16_grad_sample_cos': float(a @ b / (a.norm() * b.norm()))}
    fx = {'name': name, 'kwargs': kwargs, 'batch': batch, 'model_seed': model_seed, 'data_seed': data_seed,
          'input_checksum': float(x.double().sum()), 'probe_checksum': float(probe.double().sum()),
          'output': out.detach().clone(), 'loss': float(loss), 'grad_norm': norms, 'grad_sample': samples,
          'grad_full': full, 'reference_noise': noise, 'torch_version': torch.__version__}
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: loss={float(loss):.5f} out_norm={float(out.norm()):.4f} noise={noise} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)')


SAM_TINY = dict(image_size=256, patch_size=16, image_encoder_embedding_planes=128, image_encoder_block_nums=2,
                image_encoder_head_nums=2, image_encoder_window_size=7, image_encoder_global_attn_indexes=[1])


def sam_inputs(fx_kwargs, batch, data_seed):
    """Seeded synthetic batch in the SAMBatchCollater contract (common.py:129-232): image [B,3,S,S],
    binary mask [B,1,S,S] (an axis-aligned box per sample), one positive point inside it, its box."""
    g = torch.Generator().manual_seed(data_seed)
    s = fx_kwargs['image_size']
    images = torch.randn(batch, 3, s, s, generator=g)
    masks = torch.zeros(batch, 1, s, s)
    points, boxes = [], []
    for b in range(batch):
        x1, y1 = [int(v) for v in torch.randint(8, s // 2, (2,), generator=g)]
        w, h = [int(v) for v in torch.randint(s // 8, s // 2 - 8, (2,), generator=g)]
        masks[b, 0, y1:y1 + h, x1:x1 + w] = 1.0
        points.append([[x1 + w // 2, y1 + h // 2, 1.0]])
        boxes.append([x1, y1, x1 + w, y1 + h])
    return images, masks, torch.tensor(points, dtype=torch.float32), torch.tensor(boxes, dtype=torch.float32)


def sam_two_pass_loss(model, criterion, images, masks, points, boxes, size, autocast_dtype=None, device_type='cpu'):
    """Deterministic stand-in for one train_sam_segmentation step (tools/interactive_segmentation_scripts.py:
    369-415): encoder, decoder pass with a point + box prompt, a second decoder pass whose extra prompts
    are the centre of the mask (label 1) and the best-IoU mask of pass one at 1/4 resolution, SAMLoss over
    both passes.  Returns (loss dict, total, [mask_preds], [iou_preds])."""
    import contextlib
    import torch.nn.functional as F
    amp = (lambda: torch.autocast(device_type, dtype=autocast_dtype)) if autocast_dtype else contextlib.nullcontext
    with amp():
        emb = model.forward_image_encoder(images)
        prompts = {'prompt_point': points, 'prompt_box': boxes, 'prompt_mask': None}
        m1, i1 = model.forward_prompt_encoder_mask_decoder(emb, prompts, mask_out_idxs=[0, 1, 2, 3])
    with torch.no_grad():
        best = m1[torch.arange(m1.shape[0]), torch.argmax(i1.float(), dim=-1)].unsqueeze(1).float()
        extra = points.clone()
        extra[:, :, 0] += 3.0
        prompts2 = {'prompt_point': torch.cat([points, extra], dim=1), 'prompt_box': boxes,
                    'prompt_mask': F.interpolate(best, size=(size // 4, size // 4), mode='bilinear')}
    with amp():
        m2, i2 = model.forward_prompt_encoder_mask_decoder(emb, prompts2, mask_out_idxs=[0, 1, 2, 3])
        loss = criterion([[m1, m2], [i1, i2]], masks)
    return loss, sum(loss.values()), [m1, m2], [i1, i2]


def sam_case(name, kwargs, batch, model_seed=0, data_seed=1):
    from oracle.torch_oracle import sam_randomize_zero_init
    from SimpleAICV.interactive_segmentation.models.segment_anything import sam
    from SimpleAICV.interactive_segmentation import losses

    def build():
        torch.manual_seed(model_seed)
        m = sam.SAM(**kwargs)
        sam_randomize_zero_init(m.named_parameters(), model_seed + 100)
        return m.train()

    crit = losses.SAMLoss(alpha=0.25, gamma=2, focal_loss_weight=20, dice_loss_weight=1, iou_predict_loss_weight=1,
                          supervise_all_iou=True, mask_threshold=0.0)
    images, masks, points, boxes = sam_inputs(kwargs, batch, data_seed)
    m = build()
    ld, total, mps, ips = sam_two_pass_loss(m, crit, images, masks, points, boxes, kwargs['image_size'])
    total.backward()
    norms, samples, full = {}, {}, {}
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.detach()
        norms[n] = float(gr.norm())
        samples[n] = gr.flatten()[:64].clone()
        if gr.numel() <= 4096:
            full[n] = gr.clone()
    m2 = build()
    ld16, total16, mps16, _ = sam_two_pass_loss(m2, crit, images, masks, points, boxes, kwargs['image_size'],
                                                autocast_dtype=torch.bfloat16)
    total16.backward()
    names = [n for n, p in m2.named_parameters() if p.grad is not None and n in samples]
    a = torch.cat([dict(m2.named_parameters())[n].grad.flatten()[:64].double() for n in names])
    b = torch.cat([samples[n].double() for n in names])
    noise = {'bf16_masks': _rel(mps16[0].detach().float(), mps[0].detach()),
             'bf16_loss': abs(float(total16) - float(total)) / abs(float(total)),
             'bf16_grad_sample_cos': float(a @ b / (a.norm() * b.norm()))}
    fx = {'name': name, 'kwargs': kwargs, 'batch': batch, 'model_seed': model_seed, 'data_seed': data_seed,
          'input_checksum': float(images.double().sum() + masks.double().sum() + points.double().sum()),
          'mask_preds_lowres': [torch.nn.functional.avg_pool2d(t.detach(), 4) for t in mps],
          'mask_preds_sample': [t.detach()[:, :, ::16, ::16].clone() for t in mps],
          'iou_preds': [t.detach().clone() for t in ips],
          'loss': {k: float(v) for k, v in ld.items()}, 'total': float(total),
          'grad_norm': norms, 'grad_sample': samples, 'grad_full': full, 'reference_noise': noise,
          'no_grad_params': [n for n, p in m.named_parameters() if p.grad is None],
          'torch_version': torch.__version__}
    path = os.path.join(OUT, name + '.pt')
    torch.save(fx, path)
    print(f'{name}: loss={fx["loss"]} noise={noise} no_grad={len(fx["no_grad_params"])} -> {path} '
          f'({os.path.getsize(path) / 1024:.0f} KiB)')


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'sam_encoder_tiny' in only:
        encoder_case('sam_encoder_tiny', ENC_TINY, batch=2)
    if not only or 'sam_tiny_two_pass' in only:
        sam_case('sam_tiny_two_pass', SAM_TINY, batch=2)
    if not only or 'sam_b_encoder_256' in only:
        encoder_case('sam_b_encoder_256', ENC_B_256, batch=2)


if __name__ == '__main__':
    main()
