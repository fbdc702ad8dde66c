"""Round 6: the deterministic mode (ops.set_deterministic / csrc/det.h) -- the engine's counterpart of the reference's
`torch.backends.cudnn.deterministic = True` (tools/utils.py:95-107).  Every reduction that adds workgroup partials with fp32 atomics
in the fast path parks them side by side and folds them in index order instead.  Checked three ways: (1) repeated launches of each
kernel family give BIT-IDENTICAL results; (2) the ordered result equals the atomically accumulated one up to fp32 summation order
and the torch fp32 reference of the same op; (3) whole models (every kernel family on the hot path and next to it) repeat their
forward + backward bit for bit from a fresh construction."""
import pytest
import torch

pytestmark = [pytest.mark.gpu]


@pytest.fixture
def deterministic():
    from simpleaicv_pytorch_training_examples_amd import ops
    prev = ops.set_deterministic(True)
    yield
    ops.set_deterministic(prev)


CONV_GEOMS = [(8, 64, 56, 56, 64, 3, 1, 1), (8, 64, 56, 56, 128, 3, 2, 1), (4, 256, 28, 28, 512, 1, 2, 0), (2, 16, 37, 41, 24, 3, 1, 1),
              (3, 8, 19, 23, 16, 7, 2, 3), (16, 512, 7, 7, 512, 3, 1, 1), (64, 64, 32, 32, 64, 3, 1, 1), (32, 256, 14, 14, 1024, 1, 1, 0)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('geom', CONV_GEOMS)
def test_conv_weight_gradient_is_bit_reproducible_in_deterministic_mode(geom, dtype):
    """igemm_tn_kernel (fp32) / igemm_tn_dma_kernel (bf16): the split reduction over the pixels parks split s in part[s] and
    det_fold adds the splits in order.  Three launches: identical bits.  Against the atomic path: within fp32 summation-order noise;
    against torch's fp32 convolution weight gradient: within the dtype's resolution."""
    from simpleaicv_pytorch_training_examples_amd import ops
    n, ci, h, w, co, k, stride, pad = geom
    g = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(n, h, w, ci, generator=g).permute(0, 3, 1, 2).cuda().to(dtype)
    wt = torch.randn(co, ci, k, k, generator=g) * 0.1
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, oh, ow, co, generator=g).permute(0, 3, 1, 2).cuda().to(dtype)

    def grad():
        wp = wt.clone().cuda().requires_grad_(True)
        ops.bump_weights_epoch()
        y = ops.conv2d(x.clone().requires_grad_(True), wp, None, stride, pad)
        y.backward(dy)
        torch.cuda.synchronize()
        return wp.grad.float().clone()

    atomic = grad()
    prev = ops.set_deterministic(True)
    try:
        runs = [grad() for _ in range(3)]
    finally:
        ops.set_deterministic(prev)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), geom
    ref = torch.nn.grad.conv2d_weight(x.float().cpu(), wt.shape, dy.float().cpu(), stride=stride, padding=pad)
    scale = float(ref.abs().max())
    assert float((runs[0] - atomic).abs().max()) <= 2e-5 * scale, geom
    assert float((runs[0].cpu() - ref).abs().max()) <= (2e-2 if dtype == torch.bfloat16 else 1e-4) * scale, geom


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('mkn', [(300, 96, 40), (4096, 768, 768), (513, 200, 1000), (50432, 768, 2304), (129, 32, 30), (12608, 3072, 768)])
def test_linear_weight_and_bias_gradients_are_bit_reproducible(mkn, dtype, deterministic):
    """The same kernels through the linear entry point, with the bias gradient riding in the weight-gradient launch (the dbias
    partials sit behind the tile partials of a split: det.h)."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    m, k, n = mkn
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).to(dtype).cuda()
    w = (torch.randn(n, k, generator=g) * k ** -0.5)
    b = torch.randn(n, generator=g)
    dy = torch.randn(m, n, generator=g).to(dtype).cuda()

    def grads():
        wg, bg = w.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
        ops_tfm.linear_nd(x.clone().requires_grad_(True), wg, bg).backward(dy)
        torch.cuda.synchronize()
        return wg.grad.clone(), bg.grad.clone()

    (w1, b1), (w2, b2), (w3, b3) = grads(), grads(), grads()
    assert torch.equal(w1, w2) and torch.equal(w1, w3) and torch.equal(b1, b2) and torch.equal(b1, b3), mkn
    ref_w = dy.float().cpu().t() @ x.float().cpu()
    ref_b = dy.float().cpu().sum(0)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert float((w1.cpu() - ref_w).abs().max()) <= tol * float(ref_w.abs().max()), mkn
    assert float((b1.cpu() - ref_b).abs().max()) <= tol * float(ref_b.abs().max()), mkn


def test_gradient_norm_is_bit_reproducible(deterministic):
    """grad_stats_kernel (clip_grad_norm_): one partial per wavefront, folded in order."""
    from simpleaicv_pytorch_training_examples_amd import _lib
    g = torch.randn(3 * 1024 * 1024 + 4096, device='cuda')
    outs = []
    for _ in range(4):
        flag, ss = torch.zeros(1, device='cuda'), torch.zeros(1, device='cuda')
        _lib.check(_lib.lib().saicv_grad_stats(_lib.ptr(g), g.numel(), _lib.ptr(flag), _lib.ptr(ss), _lib.stream()), 'grad_stats')
        outs.append(float(ss))
    assert len(set(outs)) == 1 and abs(outs[0] - float((g.double() ** 2).sum())) <= 1e-5 * outs[0], outs


def _twice(build, forward, autocast=False):
    """Two fresh constructions from the same seed, one forward + backward each: -> (outputs equal, list of parameters whose gradients
    differ, number of parameters with a gradient)."""
    res = []
    for _ in range(2):
        torch.manual_seed(0)
        model, inputs = build()
        if autocast:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = forward(model, inputs)
        else:
            out = forward(model, inputs)
        out.float().backward() if out.dim() == 0 else out.float().pow(2).mean().backward()
        torch.cuda.synchronize()
        res.append((out.detach().float().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                    {n: b.detach().clone() for n, b in model.named_buffers()}))
    (o1, g1, b1), (o2, g2, b2) = res
    assert set(g1) == set(g2) and len(g1) > 0
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    bad += ['buffer:' + n for n in b1 if not torch.equal(b1[n], b2[n])]
    return torch.equal(o1, o2), bad, len(g1)


MODEL_CASES = ['resnet18cifar', 'resnet50', 'vit_tiny', 'van_b0', 'convformer', 'darknet19', 'sam_encoder', 'retinanet', 'fcos', 'detr']


@pytest.mark.parametrize('amp', [False, True], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('case', MODEL_CASES)
def test_models_repeat_forward_and_backward_bit_for_bit(case, amp, deterministic):
    """Every kernel family with a reduction, through the models that use it: implicit-GEMM convolutions with BatchNorm statistics in
    the epilogue and the stem's fused pool backward (ResNets), LayerNorm / attention / linears (ViT), depthwise convolutions,
    BatchNorm statistics of block inputs and layer-scale gradients (VAN, ConvFormer), LeakyReLU blocks (Darknet), windowed attention
    with relative-position table gradients (SAM encoder), GroupNorm heads and the pyramid (RetinaNet, FCOS), the DETR transformer."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones

    def images(b, s, seed=1):
        return torch.randn(b, s, s, 3, generator=torch.Generator().manual_seed(seed)).permute(0, 3, 1, 2).cuda()

    if case == 'resnet18cifar':
        build = lambda: (backbones.resnet18cifar(num_classes=100).cuda().train(), images(64, 32))
        fwd = lambda m, x: m(x)
    elif case == 'resnet50':
        build = lambda: (backbones.resnet50(num_classes=100).cuda().train(), images(16, 128))
        fwd = lambda m, x: m(x)
    elif case == 'vit_tiny':
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones import vit
        build = lambda: (vit.ViT(patch_size=16, embedding_planes=192, block_nums=3, head_nums=3, feedforward_ratio=4, image_size=64,
                                 dropout_prob=0., drop_path_prob=0., global_pool=False, num_classes=10).cuda().train(), images(8, 64))
        fwd = lambda m, x: m(x)
    elif case == 'van_b0':
        build = lambda: (backbones.van_b0(num_classes=16).cuda().train(), images(8, 64))
        fwd = lambda m, x: m(x)
    elif case == 'convformer':
        build = lambda: (backbones.convformer_s18(num_classes=16).cuda().train(), images(4, 64))
        fwd = lambda m, x: m(x)
    elif case == 'darknet19':
        build = lambda: (backbones.darknet19(num_classes=16).cuda().train(), images(8, 64))
        fwd = lambda m, x: m(x)
    elif case == 'sam_encoder':
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
        build = lambda: (ViTImageEncoder(image_size=256, patch_size=16, inplanes=3, embedding_planes=128, block_nums=2, head_nums=2, mlp_ratio=4,
                                         out_planes=64, window_size=7, global_attn_indexes=[1]).cuda().train(), images(2, 256))
        fwd = lambda m, x: m(x)
    elif case in ('retinanet', 'fcos'):
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import fcos, retinanet
        factory = retinanet.resnet18_retinanet if case == 'retinanet' else fcos.resnet18_fcos
        build = lambda: (factory(num_classes=20).cuda().train(), images(2, 256))
        fwd = lambda m, x: sum(o.float().pow(2).mean() for outs in m(x) for o in (outs if isinstance(outs, (list, tuple)) else [outs]))
    else:
        from oracle.make_golden_detr import detr_inputs, zero_dropout
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr

        def build():
            m = detr.resnet18_detr(hidden_inplanes=256, query_nums=20, num_classes=20)
            zero_dropout(m)
            im, mask, _ = detr_inputs(2, 1000)
            return m.cuda().train(), (im.cuda(), mask.cuda())
        fwd = lambda m, xm: sum(o.float().pow(2).mean() for o in m(*xm))
    same_out, bad, n = _twice(build, fwd, autocast=amp)
    assert same_out, f'{case}: the outputs of two runs differ'
    assert not bad, f'{case}: {len(bad)} of {n} gradients / buffers differ between two runs, e.g. {bad[:5]}'


def test_losses_and_prompt_path_of_sam_repeat_bit_for_bit(deterministic):
    """The SAM tail: prompt tokens, two-way transformer, hyper-network product, x4 upsampling, the six mask-loss sums per mask and
    their gradient -- one forward + backward of the tiny SAM with SAMLoss, twice."""
    from oracle.make_golden_sam import SAM_TINY, sam_inputs, sam_two_pass_loss
    from oracle.torch_oracle import sam_randomize_zero_init
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation import losses
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything import sam
    crit = losses.SAMLoss(alpha=0.25, gamma=2, focal_loss_weight=20, dice_loss_weight=1, iou_predict_loss_weight=1, supervise_all_iou=True,
                          mask_threshold=0.0)
    res = []
    for _ in range(2):
        torch.manual_seed(0)
        net = sam.SAM(**SAM_TINY)
        sam_randomize_zero_init(net.named_parameters(), 100)
        net = net.cuda().train()
        images, masks, points, boxes = sam_inputs(SAM_TINY, 2, 2000)
        ld, total, _, _ = sam_two_pass_loss(net, crit, images.cuda(), masks.cuda(), points.cuda(), boxes.cuda(), SAM_TINY['image_size'])
        total.backward()
        torch.cuda.synchronize()
        res.append(({k: float(v) for k, v in ld.items()}, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    (l1, g1), (l2, g2) = res
    assert l1 == l2, (l1, l2)
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not bad, bad[:8]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('geom', [(2, 64, 8, 8, 16, 16), (1, 256, 7, 10, 13, 19), (2, 128, 5, 5, 10, 9), (1, 8, 3, 4, 12, 16), (2, 32, 13, 17, 25, 34)])
def test_pyramid_merge_equals_interpolate_plus_lateral(geom, dtype):
    """saicv_resize_bilinear_add_fwd / _bwd (the top-down merge of reference SimpleAICV/detection/models/fpn.py:57-75) against
    F.interpolate(mode='bilinear') + lateral and its autograd gradient in fp32 (1e-6 relative: the same tap arithmetic, the gradient
    gathered in a fixed order instead of scattered with atomics); two launches give identical bits."""
    import torch.nn.functional as F
    from simpleaicv_pytorch_training_examples_amd import ops
    n, c, h, w, H, W = geom
    g = torch.Generator().manual_seed(sum(geom))
    top = torch.randn(n, h, w, c, generator=g).permute(0, 3, 1, 2).cuda().to(dtype)
    lat = torch.randn(n, H, W, c, generator=g).permute(0, 3, 1, 2).cuda().to(dtype)
    dout = torch.randn(n, H, W, c, generator=g).permute(0, 3, 1, 2).cuda()
    runs = []
    for _ in range(2):
        t, l = top.clone().requires_grad_(True), lat.clone().requires_grad_(True)
        out = ops.resize_bilinear_add(t, l)
        out.backward(dout)
        torch.cuda.synchronize()
        runs.append((out.detach().clone(), t.grad.clone(), l.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    t, l = top.float().clone().requires_grad_(True), lat.float().clone().requires_grad_(True)
    ref = F.interpolate(t, size=(H, W), mode='bilinear') + l
    ref.backward(dout)
    out, dt, dl = runs[0]
    assert out.dtype == torch.float32 and dt.dtype == dtype and dl.dtype == dtype
    assert float((out - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-6
    assert float((dt.float() - t.grad).abs().max()) <= tol * float(t.grad.abs().max())
    assert float((dl.float() - l.grad).abs().max()) <= tol * float(l.grad.abs().max())


def _stage(chain):
    """A chain of Bottleneck blocks [(inplanes, planes), ...] as in ResNet-50's layer1 / layer2 (stride 1)."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import Bottleneck, _init_like_reference
    net = torch.nn.Sequential(*[Bottleneck(i, p, 1) for i, p in chain])
    _init_like_reference(net)
    return net


@pytest.mark.parametrize('chain,hw', [([(64, 64), (256, 64)], (61, 47)), ([(256, 128), (512, 128)], (37, 41)), ([(128, 32), (128, 128)], (29, 31))],
                         ids=['stage1', 'stage2', 'k128'])
def test_streaming_pointwise_convolutions_equal_the_tiled_kernel_and_torch(chain, hw, monkeypatch):
    """pw_stream_kernel (csrc/pwstream.hip: ResNet-50 stage-1 / stage-2 1 x 1 convolutions, reference
    SimpleAICV/classification/backbones/resnet.py:33-43,100-155): two Bottleneck blocks forward + backward under bf16 autocast with
    every eligible shape on the streaming kernel -- the 1 x 1 products and, in stage 1, the 3 x 3 / 64 -> 64 convolution as nine taps (BatchNorm statistics in the forward; gated shortcut gradient and
    BatchNorm-backward sums in the data gradient; a row count that is no multiple of 16) against (1) the same blocks on the tiled
    kernel -- same bf16 operands and fp32 accumulation, only the summation order differs -- and (2) the oracle's fp32 blocks on the CPU."""
    h, w = hw
    n = 24
    # (the blocks compute in their input's dtype; in the models the stem hands them bf16)
    x0 = torch.randn(n, h, w, chain[0][0], generator=torch.Generator().manual_seed(5)).permute(0, 3, 1, 2).bfloat16()
    torch.manual_seed(3)
    state = {k: v.clone() for k, v in _stage(chain).state_dict().items()}
    dout = torch.randn(n, h, w, chain[-1][1] * 4, generator=torch.Generator().manual_seed(6)).permute(0, 3, 1, 2)

    def run(stream):
        from simpleaicv_pytorch_training_examples_amd import ops
        monkeypatch.setenv('SAICV_PW_STREAM', '1' if stream else '0')
        monkeypatch.setenv('SAICV_PW_STREAM3', '1' if stream else '0')
        monkeypatch.setenv('SAICV_PW_MIN_ROWS', '1024')
        net = _stage(chain)
        net.load_state_dict(state)
        net = net.cuda().train()
        ops.bump_weights_epoch()
        x = x0.cuda().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = net(x)
        assert y.dtype == torch.bfloat16
        y.backward(dout.cuda().bfloat16())
        torch.cuda.synchronize()
        return y.detach().float().cpu(), x.grad.float().cpu(), {k: p.grad.float().cpu() for k, p in net.named_parameters()}, \
            {k: b.detach().float().cpu() for k, b in net.named_buffers() if 'running' in k}

    ys, dxs, gs, bs = run(True)
    yt, dxt, gt, bt = run(False)
    # the oracle's fp32 restatement of Bottleneck.forward (oracle/torch_oracle.py, reference resnet.py:141-155) on the CPU
    from oracle.torch_oracle import bottleneck
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in state.items()}
    xr = x0.float().requires_grad_(True)
    yr = xr
    for i, (inp, pl) in enumerate(chain):
        yr = bottleneck(yr, sd, str(i), 1, inp != pl * 4, True)
    yr.backward(dout.bfloat16().float())
    gr = {k: sd[k].grad for k in gs}
    rel = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)
    l2 = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
    yr, dxr = yr.detach(), xr.grad
    print(f'[{chain}] streaming vs tiled: y max {rel(ys, yt):.2e} l2 {l2(ys, yt):.2e}; dx max {rel(dxs, dxt):.2e} l2 {l2(dxs, dxt):.2e} | vs fp32 oracle: '
          f'y {l2(ys, yr):.2e} (tiled {l2(yt, yr):.2e}), dx {l2(dxs, dxr):.2e} (tiled {l2(dxt, dxr):.2e})')
    # the two kernels differ by rounding only (another fp32 summation order in front of each bf16 rounding): their distance from each
    # other is a fraction of their common distance from the fp32 oracle (bf16 operands, ReLU gates and BatchNorm-backward cancellation:
    # 0.8 % on the outputs, 12.5 % on the input gradient for either kernel), and neither is further from the oracle than the other
    assert l2(ys, yt) <= 0.5 * l2(yt, yr) and l2(dxs, dxt) <= 0.3 * l2(dxt, dxr), (l2(ys, yt), l2(dxs, dxt))
    assert l2(ys, yr) <= 1.05 * l2(yt, yr) + 1e-4 and l2(dxs, dxr) <= 1.05 * l2(dxt, dxr) + 1e-3
    assert l2(ys, yr) <= 2e-2 and l2(dxs, dxr) <= 0.2, (l2(ys, yr), l2(dxs, dxr))
    worst = 0.0
    for k in gs:
        worst = max(worst, l2(gs[k], gr[k]))
        assert l2(gs[k], gr[k]) <= 1.1 * l2(gt[k], gr[k]) + 5e-3, (k, l2(gs[k], gr[k]), l2(gt[k], gr[k]))
        assert l2(gs[k], gt[k]) <= 0.5 * l2(gt[k], gr[k]) + 2e-3, (k, l2(gs[k], gt[k]), l2(gt[k], gr[k]))
    print(f'    parameter gradients: worst distance from the fp32 oracle {worst:.2e}')
    for k in bs:
        assert rel(bs[k], bt[k]) <= 1e-4, (k, rel(bs[k], bt[k]))


@pytest.mark.parametrize('amp', [False, True], ids=['fp32', 'bf16'])
def test_drop_path_gradient_from_the_layernorm_backward_equals_the_separate_pass(amp, monkeypatch, deterministic):
    """saicv_layernorm_bwd_scaled: the LayerNorm backward of a sub-layer also writes factor * (its output), the gradient the drop-path
    branch in front of it consumes (reference vit.py:160-161, x + drop_path(branch(x))).  A ViT with drop-path 0.4 forward + backward
    with the twin (default) and with the separate saicv_row_scale pass (ops_tfm.LN_SCALED = False): the twin scales the STORED gradient,
    so every parameter gradient is bit-identical; and the twin path really ran (no row_scale launch where a twin existed)."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones import vit
    x = torch.randn(8, 64, 64, 3, generator=torch.Generator().manual_seed(2)).permute(0, 3, 1, 2).cuda()
    calls = []
    real = ops_tfm.row_scale
    monkeypatch.setattr(ops_tfm, 'row_scale', lambda *a: (calls.append(1), real(*a))[1])

    def run(twin):
        monkeypatch.setattr(ops_tfm, 'LN_SCALED', twin)
        torch.manual_seed(0)
        m = vit.ViT(patch_size=16, embedding_planes=192, block_nums=4, head_nums=3, feedforward_ratio=4, image_size=64, dropout_prob=0.,
                    drop_path_prob=0.4, global_pool=False, num_classes=10).cuda().train()
        del calls[:]
        torch.manual_seed(5)                      # the drop-path draws
        if amp:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = m(x)
        else:
            out = m(x)
        out.float().pow(2).mean().backward()
        torch.cuda.synchronize()
        return out.detach().float().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}, len(calls)

    o1, g1, n_twin = run(True)
    o0, g0, n_sep = run(False)
    assert torch.equal(o0, o1)
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not bad, bad[:5]
    # 4 blocks x 2 branches, the first block's rate is 0 (the rates rise linearly): 6 separate passes; with the twin only the last branch
    # (its gradient comes from the final norm, not from a sub-layer) keeps one
    assert n_sep == 6 and n_twin <= 1, (n_sep, n_twin)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('mc', [(300, 256), (1000, 96), (257, 768), (64, 1024)])
def test_fused_residual_dropout_layernorm(mc, dtype):
    """saicv_dropout_add_layernorm_fwd / _bwd (DETR's norm(x + dropout(branch)), reference detection/models/detr.py:89,92,114,118,122).
    p = 0: the autograd function equals LayerNorm(x + branch) and its gradients (torch fp32).  p = 0.3: the stored sum is x + mask / 0.7 *
    branch with a mask whose keep rate is 0.7 within 4 sigma, the output is the LayerNorm of that sum, and the backward's branch gradient is
    the SAME mask / 0.7 times its sum gradient (the mask is regenerated, not stored).  The device-side seed word changes the mask."""
    import torch.nn.functional as F
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    m, c = mc
    g = torch.Generator().manual_seed(m + c)
    x = torch.randn(m, c, generator=g).to(dtype).cuda()
    b = (torch.rand(m, c, generator=g) + 1.0).to(dtype).cuda()            # (in [1, 2): the mask is read off sum - x)
    w = (torch.rand(c, generator=g) + 0.5).cuda()
    bias = torch.randn(c, generator=g).cuda()
    dy = torch.randn(m, c, generator=g).to(dtype).cuda()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    # p = 0 against torch
    xg, bg, wg, biasg = x.clone().requires_grad_(True), b.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y = ops_tfm.dropout_add_layer_norm(xg, bg, wg, biasg, 0.0, 1e-5)
    y.backward(dy)
    xr, br, wr, biasr = (t.detach().float().requires_grad_(True) for t in (x, b, w, bias))
    yr = F.layer_norm((xr + br).to(dtype).float(), (c,), wr, biasr, 1e-5)
    yr.backward(dy.float())
    rel = lambda a, r: float((a.float() - r).abs().max() / (r.abs().max() + 1e-12))
    assert rel(y, yr) <= tol and rel(xg.grad, xr.grad) <= tol and rel(bg.grad, br.grad) <= tol
    assert rel(wg.grad, wr.grad) <= max(tol, 1e-4) and rel(biasg.grad, biasr.grad) <= max(tol, 1e-4)
    assert torch.equal(xg.grad, bg.grad)
    # p = 0.3 through the C-ABI (the sum is an internal of the autograd function)
    from simpleaicv_pytorch_training_examples_amd._lib import check, dtype_code, lib, ptr, stream
    L = lib()
    word = torch.zeros(1, dtype=torch.int32, device='cuda')

    def forward(seed):
        s, out = torch.empty_like(x), torch.empty_like(x)
        mean, rstd = torch.empty(m, device='cuda'), torch.empty(m, device='cuda')
        check(L.saicv_dropout_add_layernorm_fwd(dtype_code(dtype), ptr(x), ptr(b), 0.3, seed, ptr(word), ptr(w), ptr(bias), ptr(s), ptr(out),
                                                ptr(mean), ptr(rstd), m, c, 1e-5, stream()), 'fwd')
        return s, out, mean, rstd
    s, out, mean, rstd = forward(1234)
    mask = (s.float() - x.float()).abs() > 0.5
    keep = float(mask.float().mean())
    assert abs(keep - 0.7) <= 4 * (0.21 / (m * c)) ** 0.5 + 1e-3, keep
    assert rel(s, x.float() + mask.float() * b.float() / 0.7) <= tol
    assert rel(out, F.layer_norm(s.float(), (c,), w, bias, 1e-5)) <= tol
    s2, _, _, _ = forward(1234)
    assert torch.equal(s, s2)
    word.add_(40503)
    s3, _, _, _ = forward(1234)
    assert not torch.equal(s, s3)
    word.zero_()
    dsum, dbranch = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    ws = torch.empty(L.saicv_layernorm_bwd_ws_floats(m, c), device='cuda')
    check(L.saicv_dropout_add_layernorm_bwd(dtype_code(dtype), ptr(dy), ptr(s), ptr(w), ptr(mean), ptr(rstd), 0.3, 1234, ptr(word), ptr(dsum),
                                            ptr(dbranch), ptr(dg), ptr(db), ptr(ws), m, c, 0, stream()), 'bwd')
    torch.cuda.synchronize()
    sr = s.detach().float().requires_grad_(True)
    F.layer_norm(sr, (c,), w, bias, 1e-5).backward(dy.float())
    assert rel(dsum, sr.grad) <= tol
    assert rel(dbranch, mask.float() * dsum.float() / 0.7) <= (8e-3 if dtype == torch.bfloat16 else 1e-6)
    assert bool((dbranch[~mask] == 0).all())


def test_attention_dropout_mask_follows_the_device_seed_word():
    """saicv_attn_desc.seed_device: a captured step freezes the host-side seed of the attention-probability dropout (DETR,
    nn.MultiheadAttention(dropout=0.1)); engine.StepGraph advances the word in device memory before every replay.  Same word -> the same
    output, advanced -> another mask."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(9)
    q, k, v = (torch.randn(2, 70, 256, generator=g).bfloat16().cuda() for _ in range(3))
    a, _ = ops_tfm.sattn_fwd(q, k, v, 8, 32 ** -0.5, dropout_p=0.3, seed=77)
    b, _ = ops_tfm.sattn_fwd(q, k, v, 8, 32 ** -0.5, dropout_p=0.3, seed=77)
    ops_tfm.advance_dropout_step()
    c, _ = ops_tfm.sattn_fwd(q, k, v, 8, 32 ** -0.5, dropout_p=0.3, seed=77)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, c)


# ------------------------------------------------------------------------------------------ DETR launch diet (late r06)
@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(8, 100, 2048), (3, 77, 256), (1, 8)])
def test_relu_dropout_is_dropout_of_relu(shape, dtype):
    """saicv_relu_dropout_fwd / _bwd (the hidden activation of DETR's feed-forward, reference detection/models/detr.py:90-91, 120-121):
    y is 0 or relu(x) / (1 - p), zero wherever x <= 0, kept with probability 1 - p within 4 sigma where x > 0; the same (seed, word)
    gives the same mask, another device word another; the backward passes dy / (1 - p) exactly where y > 0 -- torch's
    F.dropout(relu(x)) backward for the same mask."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    from simpleaicv_pytorch_training_examples_amd._lib import check, dtype_code, lib, ptr, stream
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(dtype).cuda()
    dy = torch.randn(*shape, generator=g).to(dtype).cuda()
    p = 0.25
    word = torch.zeros(1, dtype=torch.int32, device='cuda')
    L = lib()

    def fwd(seed):
        y = torch.empty_like(x)
        check(L.saicv_relu_dropout_fwd(dtype_code(dtype), ptr(x), ptr(y), x.numel(), p, seed, ptr(word), stream()), 'fwd')
        return y
    y = fwd(77)
    pos = x > 0
    assert bool((y[~pos] == 0).all())
    kept = y > 0
    expect = (x.float() / (1 - p)).to(dtype)
    assert torch.equal(y[kept], expect[kept]) and bool((kept <= pos).all())
    n_pos = int(pos.sum())
    if n_pos > 1000:
        rate = float(kept.sum()) / n_pos
        assert abs(rate - (1 - p)) <= 4 * (p * (1 - p) / n_pos) ** 0.5 + 1e-3, rate
    assert torch.equal(y, fwd(77)) and (x.numel() < 64 or not torch.equal(y, fwd(78)))
    word.add_(40503)
    assert x.numel() < 64 or not torch.equal(y, fwd(77))
    dx = torch.empty_like(x)
    check(L.saicv_relu_dropout_bwd(dtype_code(dtype), ptr(dy), ptr(y), ptr(dx), x.numel(), p, stream()), 'bwd')
    ref = torch.where(kept, (dy.float() / (1 - p)), torch.zeros((), device='cuda')).to(dtype)
    assert torch.equal(dx, ref)
    # the autograd function: some mask with the right structure, gradient consistent with ITS output
    xg = x.clone().requires_grad_(True)
    out = ops_tfm.relu_dropout(xg, p)
    out.backward(dy)
    k2 = out > 0
    assert bool((k2 <= pos).all()) and torch.equal(out[k2], expect[k2])
    assert torch.equal(xg.grad, torch.where(k2, dy.float() / (1 - p), torch.zeros((), device='cuda')).to(dtype))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('bnc', [(2, 100, 256, 8), (3, 333, 256, 8), (1, 17, 64, 2)])
def test_attention_over_a_packed_qk_projection_equals_attention_over_its_halves(bnc, dtype):
    """ops_tfm.stream_attention_packed_qk(qk, v): the [q | k] projection enters whole and the backward writes dq / dk into the halves of
    one buffer (DETR self-attention, head dim 32).  Bit-equal to stream_attention on the two slices, forward and all three gradients
    (the slices' autograd route: two zero fills, two copies and an add per call)."""
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    b, n, c, heads = bnc
    g = torch.Generator().manual_seed(b * n + c)
    qk = torch.randn(b, n, 2 * c, generator=g).to(dtype).cuda()
    v = torch.randn(b, n, c, generator=g).to(dtype).cuda()
    bias = (torch.rand(b, n, generator=g) > 0.8).float().cuda() * -1e4
    dout = torch.randn(b, n, c, generator=g).to(dtype).cuda()
    scale = (c // heads) ** -0.5
    a_qk, a_v = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    out_a = ops_tfm.stream_attention_packed_qk(a_qk, a_v, heads, scale, bias, 0.0)
    out_a.backward(dout)
    b_qk, b_v = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    out_b = ops_tfm.stream_attention(b_qk[..., :c], b_qk[..., c:], b_v, heads, scale, bias, 0.0)
    out_b.backward(dout)
    assert torch.equal(out_a, out_b)
    assert torch.equal(a_qk.grad, b_qk.grad) and torch.equal(a_v.grad, b_v.grad)


@pytest.mark.gpu
@pytest.mark.parametrize('lbqt', [(6, 8, 100, 100), (1, 1, 7, 3), (3, 2, 300, 40), (2, 5, 100, 120)])
def test_fused_detr_box_losses_equal_the_torch_formulation(lbqt):
    """saicv_detr_box_loss_fwd / _bwd (DETRLoss.forward_static on the GPU; compute_batch_l1_iou_loss, reference
    SimpleAICV/detection/losses.py:938-954) against the torch formulation it replaces, evaluated in fp32 AND fp64 on the same device
    tensors: per-layer L1 and GIoU losses and the gradient of the raw regression outputs -- including predictions outside the clamp
    range (gradient exactly 0 there), images without boxes, padding pairs, unmatched queries (rows exactly 0).  Gate: the kernel is as
    close to the fp64 value as torch's own fp32 evaluation is, within a factor of 4 (+ 1e-7).  A batch without any box gives nan."""
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection import losses as LS
    l, b, q, t = lbqt
    g = torch.Generator().manual_seed(l * 1000 + q)
    reg = torch.rand(l, b, q, 4, generator=g) * 0.9 + 0.05
    reg[:, :, ::7, 2] = 1.2                         # above the clamp range
    reg[:, :, 1::9, 1] = -0.3                       # below it
    gt = torch.full((b, t, 5), -1.0)
    src = torch.zeros(b, t, dtype=torch.int64)
    tgt = torch.zeros(b, t, dtype=torch.int64)
    w = torch.zeros(b, t)
    for i in range(b):
        n = 0 if (i == 1 and b > 1) else int(torch.randint(1, min(q, t) + 1, (1,), generator=g))
        gt[i, :n, 0:2] = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        gt[i, :n, 2:4] = torch.rand(n, 2, generator=g) * 0.3 + 0.02
        gt[i, :n, 4] = torch.randint(0, 80, (n,), generator=g).float()
        src[i, :n] = torch.randperm(q, generator=g)[:n]
        tgt[i, :n] = torch.randperm(n, generator=g)
        w[i, :n] = 1.0
    reg, gt, src, tgt, w = (x.cuda() for x in (reg, gt, src, tgt, w))

    def torch_form(dtype):
        r = reg.detach().to(dtype).clone().requires_grad_(True)
        p = torch.clamp(r, min=1e-4, max=1. - 1e-4)
        bidx = torch.arange(b, device='cuda')[:, None].expand(b, t)
        on = w > 0
        dummy = torch.tensor([0.5, 0.5, 0.2, 0.2], device='cuda', dtype=dtype)
        pm = torch.where(on[None, :, :, None], p[:, bidx, src], dummy)
        tb = torch.where(on[:, :, None], gt.to(dtype)[bidx, tgt, 0:4], dummy)
        n = (gt[:, :, 4] >= 0).sum().to(dtype)
        l1 = ((pm - tb).abs().sum(-1) * w.to(dtype)).sum((1, 2)) / n
        iou = ((1 - LS._giou(LS._cxcywh_to_xyxy(pm), LS._cxcywh_to_xyxy(tb))) * w.to(dtype)).sum((1, 2)) / n
        return r, l1, iou
    cl1 = torch.randn(l, generator=g).cuda()
    ciou = torch.randn(l, generator=g).cuda()
    res = {}
    for name, dtype in (('f32', torch.float32), ('f64', torch.float64)):
        r, l1, iou = torch_form(dtype)
        ((l1 * cl1.to(dtype)).sum() + (iou * ciou.to(dtype)).sum()).backward()
        res[name] = (l1.detach().double(), iou.detach().double(), r.grad.double())
    rk = reg.clone().requires_grad_(True)
    kl1, kiou = LS._DetrBoxLossFn.apply(rk, gt, src, tgt, w, 1e-4, 1. - 1e-4)
    ((kl1 * cl1).sum() + (kiou * ciou).sum()).backward()
    got = (kl1.detach().double(), kiou.detach().double(), rk.grad.double())
    for a, f32, f64, what in zip(got, res['f32'], res['f64'], ('l1', 'iou', 'gradient')):
        scale = float(f64.abs().max())
        err, own = float((a - f64).abs().max()) / scale, float((f32 - f64).abs().max()) / scale
        assert err <= 4 * own + 1e-7, (what, err, own)
    matched = torch.zeros(b, q, dtype=torch.bool, device='cuda')
    matched[torch.arange(b, device='cuda')[:, None].expand(b, t)[w > 0], src[w > 0]] = True
    assert bool((rk.grad[:, ~matched] == 0).all())
    outside = (reg < 1e-4) | (reg > 1. - 1e-4)
    assert bool((rk.grad[outside] == 0).all()) and bool((rk.grad[:, matched].abs().sum(-1) > 0).all())
    # no box in the whole batch: 0 / 0
    e1, e2 = LS._DetrBoxLossFn.apply(reg, torch.full_like(gt, -1.0), src, tgt, torch.zeros_like(w), 1e-4, 1. - 1e-4)
    assert bool(torch.isnan(e1).all()) and bool(torch.isnan(e2).all())
