"""CPU oracle for the SimpleAICV DDP forward/backward hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 CPU restatement of the reference's model / loss / step math,
written as pure functions over a `state_dict` (no nn.Module graph), so that it can be pinned
against the reference and then used as the checker for the HIP kernels.  Only `tests/`,
`__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may import it; the product
package never does (tests/test_no_oracle_in_product.py enforces that).

Parity pin: the reference ships no golden vectors or unit tests for this path (SURVEY.md
section 8c), so the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF: the script
oracle/make_golden.py imports the reference modules from /root/reference, runs them on seeded
inputs and stores logits / loss / gradients under tests/golden/; tests/test_oracle_golden.py
checks this restatement against those fixtures.

Every function cites the reference lines it follows (paths relative to the reference repo).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------ ResNet
class relu_gates:
    """Context: the ResNet ReLUs below take their open / closed decision from a recorded list of boolean masks (one per ReLU, in
    call order) instead of from the sign of their own input, and report how the two differ.  Purpose (r05, __graft_entry__.smoke):
    two correct fp32 evaluations of a ReLU network disagree on the few gates whose pre-activation is within rounding distance of
    zero (4.5 M gates in ResNet18Cifar at batch 8, ~1e-6 relative arithmetic noise -> a handful), and ONE flipped gate moves the
    relative L2 distance of the gradients by ~1 / sqrt(elements per layer) ~ 1e-3 -- so a gradient comparison tighter than that has
    to be made AT THE SAME GATES.  The report (`flips`: [(relu index, flipped count, max |pre-activation| / rms among the flipped)])
    lets the caller insist that only knife-edge elements were overridden."""
    active = None

    def __init__(self, masks=None):
        """masks: the recorded decisions to impose; None = record this evaluation's own decisions into `.masks` instead."""
        self.record = masks is None
        self.masks, self.k, self.flips = ([] if masks is None else list(masks)), 0, []

    def __enter__(self):
        relu_gates.active = self
        return self

    def __exit__(self, *exc):
        relu_gates.active = None

    def apply(self, y):
        if self.record:
            self.masks.append(y.detach() > 0)
            self.k += 1
            return F.relu(y)
        m = self.masks[self.k].to(y.device)
        assert m.shape == y.shape, (self.k, tuple(m.shape), tuple(y.shape))
        diff = m != (y > 0)
        n = int(diff.sum())
        if n:
            self.flips.append((self.k, n, float(y.detach()[diff].abs().max() / y.detach().pow(2).mean().sqrt())))
        self.k += 1
        return y * m.to(y.dtype)


def _relu(y):
    g = relu_gates.active
    return F.relu(y) if g is None else g.apply(y)


def conv_bn_act(x, sd, prefix, stride, padding, act, training, bn_updates=None, momentum=0.1, eps=1e-5):
    """ConvBnActBlock.forward: Conv2d(bias=False) -> BatchNorm2d -> [ReLU].
    reference SimpleAICV/classification/backbones/resnet.py:19-48."""
    y = F.conv2d(x, sd[prefix + '.layer.0.weight'], None, stride=stride, padding=padding)
    rm, rv = sd[prefix + '.layer.1.running_mean'], sd[prefix + '.layer.1.running_var']
    if training and bn_updates is not None:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(y, rm, rv, sd[prefix + '.layer.1.weight'], sd[prefix + '.layer.1.bias'], training, momentum, eps)
    if training and bn_updates is not None:
        bn_updates[prefix + '.layer.1.running_mean'] = rm
        bn_updates[prefix + '.layer.1.running_var'] = rv
    return _relu(y) if act else y


def basic_block(x, sd, prefix, stride, downsample, training, bn_updates=None):
    """BasicBlock.forward, reference resnet.py:84-97."""
    out = conv_bn_act(x, sd, prefix + '.conv1', stride, 1, True, training, bn_updates)
    out = conv_bn_act(out, sd, prefix + '.conv2', 1, 1, False, training, bn_updates)
    if downsample:
        x = conv_bn_act(x, sd, prefix + '.downsample_conv', stride, 0, False, training, bn_updates)
    return _relu(out + x)


def bottleneck(x, sd, prefix, stride, downsample, training, bn_updates=None):
    """Bottleneck.forward, reference resnet.py:141-155 (stride on the 3x3 conv)."""
    out = conv_bn_act(x, sd, prefix + '.conv1', 1, 0, True, training, bn_updates)
    out = conv_bn_act(out, sd, prefix + '.conv2', stride, 1, True, training, bn_updates)
    out = conv_bn_act(out, sd, prefix + '.conv3', 1, 0, False, training, bn_updates)
    if downsample:
        x = conv_bn_act(x, sd, prefix + '.downsample_conv', stride, 0, False, training, bn_updates)
    return _relu(out + x)


RESNET_SPECS = {
    # name: (block, layer_nums, cifar_stem)   -- factories resnet.py:254-271, resnetforcifar.py:108-125
    'resnet18': ('basic', [2, 2, 2, 2], False),
    'resnet34': ('basic', [3, 4, 6, 3], False),
    'resnet50': ('bottleneck', [3, 4, 6, 3], False),
    'resnet101': ('bottleneck', [3, 4, 23, 3], False),
    'resnet152': ('bottleneck', [3, 8, 36, 3], False),
    'resnet18cifar': ('basic', [2, 2, 2, 2], True),
    'resnet34cifar': ('basic', [3, 4, 6, 3], True),
    'resnet50cifar': ('bottleneck', [3, 4, 6, 3], True),
}


def resnet_forward(name, sd, x, training=True, bn_updates=None):
    """ResNet.forward (resnet.py:226-245) / ResNetCifar.forward (resnetforcifar.py:86-99)."""
    kind, layer_nums, cifar = RESNET_SPECS[name]
    if cifar:
        x = conv_bn_act(x, sd, 'conv1', 1, 1, True, training, bn_updates)
    else:
        x = conv_bn_act(x, sd, 'conv1', 2, 3, True, training, bn_updates)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    expansion = 1 if kind == 'basic' else 4
    inplanes = 64
    for li, (planes, n) in enumerate(zip([64, 128, 256, 512], layer_nums)):
        for bi in range(n):
            stride = (1 if li == 0 else 2) if bi == 0 else 1
            ds = stride != 1 or inplanes != planes * expansion
            fn = basic_block if kind == 'basic' else bottleneck
            x = fn(x, sd, f'layer{li + 1}.{bi}', stride, ds, training, bn_updates)
            inplanes = planes * expansion
    x = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
    return F.linear(x, sd['fc.weight'], sd['fc.bias'])


# ------------------------------------------------------------------------------ losses
def ce_loss(pred, label):
    """CELoss.forward, reference SimpleAICV/classification/losses.py:23-28."""
    return F.cross_entropy(pred.float(), label, reduction='mean')


def one_hot_ce_loss(pred, label):
    """OneHotLabelCELoss.forward, reference losses.py:86-91."""
    return torch.sum(-label * F.log_softmax(pred.float(), dim=-1), dim=-1).mean()


# ------------------------------------------------------------------------------ ViT
def vit_forward(sd, x, patch=16, heads=12, blocks=12, global_pool=True, eps=1e-6):
    """ViT.forward with dropout / drop-path off (reference classification/backbones/vit.py:239-262);
    attention as MultiHeadAttention.forward (:61-80): logits scaled AFTER q@k^T; MLP as
    FeedForward.forward (:91-99) with exact-erf GELU; pre-LN blocks (:159-163)."""
    b = x.shape[0]
    t = F.conv2d(x, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd['cls_token'].expand(b, -1, -1), t), dim=1) + sd['pos_embed']
    c = t.shape[-1]
    hd = c // heads
    for i in range(blocks):
        p = f'blocks.{i}.'
        h = F.layer_norm(t, (c,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps)
        qkv = F.linear(h, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'])
        qkv = qkv.view(b, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
        h = (attn @ v).transpose(1, 2).reshape(b, -1, c)
        t = t + F.linear(h, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'])
        h = F.layer_norm(t, (c,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps)
        h = F.gelu(F.linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias']))
        t = t + F.linear(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
    if global_pool:
        t = F.layer_norm(t[:, 1:, :].mean(dim=1), (c,), sd['norm.weight'], sd['norm.bias'], eps)
    else:
        t = F.layer_norm(t, (c,), sd['norm.weight'], sd['norm.bias'], eps)[:, 0]
    return F.linear(t, sd['fc.weight'], sd['fc.bias'])


# ------------------------------------------------------------------------------ optimizer math
def sgd_momentum_step(params, grads, bufs, lr, momentum, weight_decay):
    """torch.optim.SGD as configured by reference tools/utils.py (build_optimizer, SGD branch):
    d = g + wd*p ; buf = momentum*buf + d (buf = d on the first step) ; p -= lr*buf."""
    for n in params:
        d = grads[n] + weight_decay.get(n, 0.0) * params[n]
        if momentum != 0:
            bufs[n] = d.clone() if n not in bufs else bufs[n] * momentum + d
            d = bufs[n]
        params[n] = params[n] - lr * d
    return params, bufs


def adamw_step(params, grads, state, lr, betas, eps, weight_decay, step):
    """torch.optim.AdamW (decoupled decay), reference tools/utils.py AdamW branch."""
    b1, b2 = betas
    for n in params:
        m, v = state.get(n, (torch.zeros_like(params[n]), torch.zeros_like(params[n])))
        p = params[n] * (1 - lr * weight_decay.get(n, 0.0))
        m = b1 * m + (1 - b1) * grads[n]
        v = b2 * v + (1 - b2) * grads[n] * grads[n]
        denom = v.sqrt() / math.sqrt(1 - b2 ** step) + eps
        params[n] = p - (lr / (1 - b1 ** step)) * m / denom
        state[n] = (m, v)
    return params, state


# ------------------------------------------------------------------------------ helpers
def loss_and_grads(forward_fn, sd, param_names, *inputs, loss_fn=ce_loss, label=None):
    """Runs forward + loss + backward on leaf copies of the parameters; returns
    (logits, loss, {name: grad})."""
    leaves = {k: (v.detach().clone().requires_grad_(True) if k in param_names else v.detach().clone())
              for k, v in sd.items()}
    logits = forward_fn(leaves, *inputs)
    loss = loss_fn(logits, label)
    loss.backward()
    grads = {k: leaves[k].grad for k in param_names}
    return logits.detach(), loss.detach(), grads


# ------------------------------------------------------------------------------ SAM image encoder
def sam_randomize_zero_init(named_parameters, seed):
    """The reference initialises pos_embed / rel_pos_h / rel_pos_w to ZEROS (image_encoder.py:157-160,
    287-289), which would leave the relative-position path untested: fixtures and tests overwrite
    every all-zero parameter with N(0, 0.2) draws from this seeded generator, in registration order."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in named_parameters:
            if float(p.detach().abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)


def _rel_pos_gather(size, table):
    """get_rel_pos for q_size == k_size (reference image_encoder.py:82-113): row q - k + size - 1."""
    r = torch.arange(size)
    return table[(r[:, None] - r[None, :]) + (size - 1)]


def sam_attention(x, sd, p, heads):
    """Attention.forward + add_decomposed_rel_pos (reference image_encoder.py:116-184): the relative
    terms use the UNSCALED q; logits = (q * scale) k^T + rel_h[..., None] + rel_w[..., None, :]."""
    b, h, w, c = x.shape
    hd = c // heads
    qkv = F.linear(x, sd[p + 'qkv.weight'], sd[p + 'qkv.bias']).reshape(b, h * w, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, b * heads, h * w, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    rh = _rel_pos_gather(h, sd[p + 'rel_pos_h'])
    rw = _rel_pos_gather(w, sd[p + 'rel_pos_w'])
    rq = q.reshape(b * heads, h, w, hd)
    rel_h = torch.einsum('bhwc,hkc->bhwk', rq, rh)
    rel_w = torch.einsum('bhwc,wkc->bhwk', rq, rw)
    attn = (attn.view(b * heads, h, w, h, w) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(
        b * heads, h * w, h * w)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).view(b, heads, h, w, hd).permute(0, 2, 3, 1, 4).reshape(b, h, w, c)
    return F.linear(o, sd[p + 'proj.weight'], sd[p + 'proj.bias'])


def sam_layernorm2d(x, weight, bias, eps=1e-6):
    """LayerNorm2d.forward (reference image_encoder.py:250-256)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


def sam_encoder_forward(sd, x, patch=16, heads=12, blocks=12, window=14, global_idx=(2, 5, 8, 11), eps=1e-6):
    """ViTImageEncoder.forward (reference image_encoder.py:318-335) with Block.forward (:222-239),
    window_partition / window_unpartition (:32-79: zero pad AFTER norm1 to a multiple of the window)."""
    t = F.conv2d(x, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], stride=patch).permute(0, 2, 3, 1)
    t = t + sd['pos_embed']
    b, hh, ww, c = t.shape
    for i in range(blocks):
        p = f'blocks.{i}.'
        ws = 0 if i in global_idx else window
        h = F.layer_norm(t, (c,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps)
        if ws > 0:
            ph, pw = (ws - hh % ws) % ws, (ws - ww % ws) % ws
            h = F.pad(h, (0, 0, 0, pw, 0, ph))
            hp, wp = hh + ph, ww + pw
            h = h.view(b, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)
        h = sam_attention(h, sd, p + 'attn.', heads)
        if ws > 0:
            h = h.view(b, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, c)[:, :hh, :ww, :]
        t = t + h
        h = F.layer_norm(t, (c,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps)
        h = F.gelu(F.linear(h, sd[p + 'mlp.lin1.weight'], sd[p + 'mlp.lin1.bias']))
        t = t + F.linear(h, sd[p + 'mlp.lin2.weight'], sd[p + 'mlp.lin2.bias'])
    t = t.permute(0, 3, 1, 2)
    t = sam_layernorm2d(F.conv2d(t, sd['neck.0.weight']), sd['neck.1.weight'], sd['neck.1.bias'])
    t = sam_layernorm2d(F.conv2d(t, sd['neck.2.weight'], padding=1), sd['neck.3.weight'], sd['neck.3.bias'])
    return t


# ------------------------------------------------------------------------------ SAM losses
def sam_per_mask_losses(inputs, targets, pred_ious, alpha=0.25, gamma=2, mask_threshold=0.0):
    """SAMLoss.focal_loss / dice_loss / iou_predict_loss (reference interactive_segmentation/losses.py:136-198)
    -> three [B, M] tensors, each already divided by the batch size."""
    b = inputs.shape[0]
    x = inputs.float()
    t = targets.expand_as(inputs).float()
    bce = F.binary_cross_entropy_with_logits(x, t, reduction='none')
    p = torch.sigmoid(x)
    pt = p * t + (1 - p) * (1 - t)
    focal = ((alpha * t + (1 - alpha) * (1 - t)) * torch.pow(1. - pt, gamma) * bce).flatten(2).mean(dim=-1) / b
    pf, tf = p.flatten(2), t.flatten(2)
    dice = (1. - (2. * (pf * tf).sum(-1) + 1) / (pf.sum(-1) + tf.sum(-1) + 1)) / b
    xi, ti = (x > mask_threshold).flatten(2), (t > mask_threshold).flatten(2)
    inter = torch.sum(xi & ti, dim=-1).float()
    union = torch.sum(xi | ti, dim=-1).float()
    gt = torch.clamp(inter / torch.clamp(union, min=1e-6), min=0.0, max=1.0)
    iou = F.mse_loss(pred_ious.float(), gt, reduction='none') / b
    return focal, dice, iou
