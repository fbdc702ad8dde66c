"""r06: is a ResNet-50 b256 bf16 step bit-reproducible in deterministic mode at the bench's sizes (the streaming kernels, 256 x 256
tiles, persistent launches -- none of which the small trajectory tests reach)?  Two forward+backward passes from the same weights
and input; per-module output bit hashes (first module that differs) and per-parameter gradient equality."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from simpleaicv_pytorch_training_examples_amd import ops  # noqa: E402
from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import resnet50  # noqa: E402


def bits(t):
    t = t.detach().contiguous()
    v = t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)
    return int(v.to(torch.int64).sum())


def one_pass(m, x, y):
    hashes = []
    hooks = [mod.register_forward_hook(lambda mod, i, o, n=n: hashes.append((n, bits(o))) if torch.is_tensor(o) else None)
             for n, mod in m.named_modules() if n]
    m.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = m(x)
        loss = torch.nn.functional.cross_entropy(out.float(), y)
    loss.backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    return hashes, {n: p.grad.clone() for n, p in m.named_parameters()}, float(loss)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ops.set_deterministic(True)
    torch.manual_seed(0)
    m = resnet50(num_classes=1000).cuda().train()
    x = torch.randn(batch, 3, 224, 224).cuda()
    y = torch.randint(0, 1000, (batch,)).cuda()
    runs = [one_pass(m, x, y) for _ in range(4)]
    h0, g0, l0 = runs[0]
    for r, (h, g, l) in enumerate(runs[1:], 1):
        fwd = [(n, a, b) for (n, a), (_, b) in zip(h0, h) if a != b]
        bad = [n for n in g0 if not torch.equal(g0[n], g[n])]
        print(f'run {r}: loss {l0!r} vs {l!r}; forward modules differing {len(fwd)} (first: {fwd[:3]}); gradients differing {len(bad)} of {len(g0)}', flush=True)
        if bad:
            names = list(g0)
            print('   last (closest to the loss) differing parameters:', [n for n in names if n in set(bad)][-6:], flush=True)
            print('   first differing parameters:', [n for n in names if n in set(bad)][:6], flush=True)


if __name__ == '__main__':
    main()
