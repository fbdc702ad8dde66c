"""Parity at BASELINE.json's FULL sizes (per-GPU batch 256 at 224 x 224, SAM at 1024 x 1024, N = 4096 attention),
where the CPU oracle would take minutes to hours: size-independent properties instead of a second computation.

  * batch-slice consistency: nothing on the path couples samples except BatchNorm's batch statistics, so the rows
    of a full-size result that belong to the first few samples must equal the small-batch result of the same
    kernels -- and the small-batch results are what test_gpu_kernels.py / test_gpu_models.py pin to the oracle
    and to the reference fixtures.  Tile geometry, split counts and grid sizes all change with the batch, the
    per-element accumulation order over K does not, so forward / data-gradient rows must agree to rounding
    (asserted at 1e-6 scale-relative in bf16: bit-exact up to the order of fp32 adds inside one MFMA chain);
  * additivity of the weight gradient over the batch: dW(full) = dW(first half) + dW(second half) (fp32 atomics:
    1e-4);
  * BatchNorm scale invariance (train mode, batch statistics): multiplying every pre-BN conv weight by 2 is exact
    in floating point and must leave the logits unchanged (up to the eps = 1e-5 term) -- this exercises the
    full-size statistics epilogue, the partial reduction and the finalize kernels;
  * softmax rows sum to one: attention with V = 1 returns 1 (N = 4096 with rel-pos logits, and DETR's key bias);
  * cross-entropy: the gradient of the classifier bias sums to zero;
  * gradient linearity over the batch for the BN-free ViT: grad(b256) = mean of the two b128 halves.
"""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


# (Cin, Cout, k, stride, H): the extremes of the 23 ResNet-50 shapes at per-GPU batch 256
FULL_CONVS = [
    (8, 64, 7, 2, 224),        # stem on the packed 8-channel input: M = 3.2 M rows, K = 392
    (64, 256, 1, 1, 56),       # widest write-bound pointwise layer
    (256, 64, 1, 1, 56),
    (128, 128, 3, 2, 56),      # strided 3x3 (parity-class data-gradient)
    (256, 256, 3, 1, 14),
    (1024, 2048, 1, 2, 14),    # strided pointwise downsample
    (512, 512, 3, 1, 7),       # deepest K = 4608, smallest M
]


@pytest.mark.parametrize('cin,cout,k,stride,h', FULL_CONVS)
def test_full_size_conv_rows_equal_small_batch_and_wgrad_is_additive(cin, cout, k, stride, h):
    from simpleaicv_pytorch_training_examples_amd import ops
    torch.manual_seed(0)
    B, b = 256, 8
    pad = k // 2
    x = _nhwc(torch.randn(B, cin, h, h, device='cuda').to(BF16))
    w = _nhwc((torch.randn(cout, cin, k, k, device='cuda') * (2.0 / (cin * k * k)) ** 0.5))

    def run(xs, gy=None):
        xs = xs.detach().requires_grad_(True)
        wl = w.detach().clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=BF16):
            y = ops.conv2d(xs, wl, None, stride, pad)
        gy = torch.randn(y.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(1)).to(BF16) if gy is None else gy
        y.backward(_nhwc(gy))
        return y.detach(), xs.grad, wl.grad, gy

    y, dx, dw, gy = run(x)
    ys, dxs, _, _ = run(x[:b].clone(memory_format=torch.channels_last), _nhwc(gy[:b].clone()))
    assert y.dtype == BF16 and tuple(y.shape[:2]) == (B, cout)
    assert torch.isfinite(y.float()).all() and torch.isfinite(dw).all()
    assert rel_err(y[:b], ys) < 1e-6
    assert rel_err(dx[:b], dxs) < 1e-6
    _, _, dw0, _ = run(x[:B // 2].clone(memory_format=torch.channels_last), _nhwc(gy[:B // 2].clone()))
    _, _, dw1, _ = run(x[B // 2:].clone(memory_format=torch.channels_last), _nhwc(gy[B // 2:].clone()))
    assert rel_err(dw, dw0 + dw1) < 1e-4


def _resnet50(seed=0):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
    torch.manual_seed(seed)
    return backbones.resnet50(num_classes=1000).cuda()


def _images(b, seed=0):
    g = torch.Generator('cuda').manual_seed(seed)
    x = torch.randn(b, 224, 224, 3, device='cuda', generator=g)
    return x.permute(0, 3, 1, 2)          # the collater's NCHW-shaped, NHWC-strided fp32 batch


def test_resnet50_b256_eval_rows_equal_small_batch():
    model = _resnet50().eval()
    x = _images(256)
    with torch.no_grad(), torch.autocast('cuda', dtype=BF16):
        full = model(x)
        small = model(x[:8])
    assert tuple(full.shape) == (256, 1000) and torch.isfinite(full).all()
    assert rel_err(full[:8], small) < 1e-6


def test_resnet50_b256_train_step_invariants():
    """One bf16 training step at the bench workload: CE bias-gradient identity, every parameter receives a finite
    gradient, running statistics move."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import losses
    model = _resnet50().train()
    crit = losses.CELoss()
    x = _images(256)
    y = torch.randint(0, 1000, (256,), device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    with torch.autocast('cuda', dtype=BF16):
        logits = model(x)
        loss = crit(logits, y)
    loss.backward()
    assert torch.isfinite(loss) and abs(float(loss.detach()) - 6.9) < 1.0      # ln(1000) = 6.91 at initialisation
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    gb = model.fc.bias.grad
    # sum_c (softmax - onehot) = 0 per sample; d(loss)/d(logits) reaches the bias reduction in bf16 (2^-9 per element)
    assert abs(float(gb.sum())) < 2e-3 * float(gb.abs().sum())
    assert int(model.conv1.layer[1].num_batches_tracked) == 1
    assert float(model.conv1.layer[1].running_mean.abs().max()) > 0


def test_resnet50_b256_batchnorm_scale_invariance():
    """Doubling every conv weight that feeds a BatchNorm is exact in floating point; with batch statistics the
    logits must not move (up to eps / var) and the weight gradients halve.  Run in the fp32 parity mode: under bf16
    the reference's OWN logits move by 10 % between precisions at this depth (fixture reference_noise), so only
    fp32 makes the property sharp.  Exercises the full-size statistics epilogue, partial reduction and finalize."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import losses
    crit = losses.CELoss()
    x = _images(256)
    y = torch.randint(0, 1000, (256,), device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    watch = ('conv1.layer.0.weight', 'layer3.0.conv2.layer.0.weight', 'layer4.2.conv3.layer.0.weight')
    out = []
    for scale in (1.0, 2.0):
        model = _resnet50().train()
        with torch.no_grad():
            for m in model.modules():
                if m.__class__.__name__ == 'ConvBnActBlock':
                    m.layer[0].weight.mul_(scale)
        logits = model(x)
        loss = crit(logits, y)
        loss.backward()
        params = dict(model.named_parameters())
        out.append((logits.detach(), float(loss.detach()), {n: params[n].grad.clone() for n in watch}))
    (l1, s1, g1), (l2, s2, g2) = out
    assert l1.dtype == torch.float32
    print('scale invariance: logits', rel_err(l2, l1), 'loss', abs(s2 - s1) / s1)
    assert rel_err(l2, l1) < 1e-2
    assert abs(s2 - s1) < 1e-3 * s1
    for n in watch:
        # not exact: 1 / sqrt(4 var + eps) != 0.5 / sqrt(var + eps), and the low-variance channels where eps
        # matters carry the LARGEST weight gradients (measured 4-6e-2 of the gradient scale; the reference's own
        # fp32 runs differ by 7.7e-2 on early-layer gradients under a mere re-ordering, fixture reference_noise)
        e = rel_err(2.0 * g2[n], g1[n])
        print(n, e)
        assert e < 0.15, n


def _vit(seed=0):
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
    torch.manual_seed(seed)
    return backbones.vit_base_patch16(image_size=224, drop_path_prob=0.0, global_pool=True, num_classes=1000).cuda()


def test_vit_base_b256_rows_and_gradient_linearity():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import losses
    model = _vit().train()
    crit = losses.CELoss()
    x = _images(256, seed=5)
    y = torch.randint(0, 1000, (256,), device='cuda', generator=torch.Generator('cuda').manual_seed(6))
    watch = ('blocks.0.attn.qkv.weight', 'blocks.11.mlp.fc2.weight', 'pos_embed', 'fc.weight', 'norm.weight')

    def grads(xs, ys):
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=BF16):
            lg = model(xs)
            ls = crit(lg, ys)
        ls.backward()
        return lg.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if n in watch}

    lg, g = grads(x, y)
    lg0, g0 = grads(x[:128], y[:128])
    lg1, g1 = grads(x[128:], y[128:])
    assert torch.isfinite(lg).all()
    assert rel_err(lg[:128], lg0) < 1e-6 and rel_err(lg[128:], lg1) < 1e-6
    for n in watch:
        # bf16 activations: the halves round d(loss)/d(logits) = (p - y) / 128 vs / 256 identically up to a power of
        # two, so the agreement is at accumulation-order level
        assert rel_err(g[n], 0.5 * (g0[n] + g1[n])) < 1e-2, n


@pytest.mark.parametrize('case', ['sam_global', 'detr_encoder'])
def test_full_size_attention_rows_sum_to_one(case):
    """V = 1  =>  softmax(.) V = 1 for every query, whatever the logits: the streaming softmax's running maximum /
    normaliser at N = 4096 with rel-pos logits, and with DETR's key-padding bias."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    torch.manual_seed(0)
    if case == 'sam_global':
        b, n, heads, hd, s = 2, 4096, 12, 64, 64
        rel_h = torch.randn(b * heads, n, s, device='cuda') * 2.0
        rel_w = torch.randn(b * heads, n, s, device='cuda') * 2.0
        kb = None
    else:
        b, n, heads, hd = 8, 42 * 42, 8, 32
        rel_h = rel_w = None
        kb = (torch.rand(b, n, device='cuda') < 0.3).float()          # the reference's +1.0 float mask
    q = (torch.randn(b, n, heads * hd, device='cuda') * 2.0).to(BF16)
    k = (torch.randn(b, n, heads * hd, device='cuda') * 2.0).to(BF16)
    v = torch.ones(b, n, heads * hd, device='cuda', dtype=BF16)
    out, lse = ops_tfm.sattn_fwd(q, k, v, heads, hd ** -0.5, kb, rel_h, rel_w)
    assert torch.isfinite(lse).all()
    assert float((out.float() - 1.0).abs().max()) <= 2.0 ** -7          # one bf16 ulp below 1 at most
    # first rows against the same kernel on a single sample (different grid, same per-row arithmetic)
    o1, l1 = ops_tfm.sattn_fwd(q[:1], k[:1], v[:1], heads, hd ** -0.5, None if kb is None else kb[:1].contiguous(),
                               None if rel_h is None else rel_h[:heads].contiguous(),
                               None if rel_w is None else rel_w[:heads].contiguous())
    assert rel_err(l1, lse[:heads]) < 1e-6


def test_sam_encoder_1024_rows_equal_single_image():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
    from oracle.torch_oracle import sam_randomize_zero_init
    torch.manual_seed(0)
    enc = ViTImageEncoder(image_size=1024, patch_size=16, inplanes=3, embedding_planes=768, block_nums=12, head_nums=12,
                          mlp_ratio=4, out_planes=256, window_size=14, global_attn_indexes=(2, 5, 8, 11)).cuda().eval()
    sam_randomize_zero_init(enc.named_parameters(), 1)
    x = torch.randn(2, 3, 1024, 1024, device='cuda', generator=torch.Generator('cuda').manual_seed(2))
    with torch.no_grad(), torch.autocast('cuda', dtype=BF16):
        full = enc(x)
        one = enc(x[1:])
    assert tuple(full.shape) == (2, 256, 64, 64) and torch.isfinite(full.float()).all()
    assert rel_err(full[1:], one) < 1e-6
