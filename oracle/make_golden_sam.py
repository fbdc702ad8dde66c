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


def main():
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden fixtures can only be (re)generated in the build container')
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    encoder_case('sam_encoder_tiny', ENC_TINY, batch=2)


if __name__ == '__main__':
    main()
