"""Transformer-side autograd Functions over libsaicv_hip.so (LayerNorm, GELU, Linear, fused
attention) and the two fused pre-LN sub-layers of a ViT block.

A sub-layer  out = x + s * f(LN(x))  (reference vit.py:159-163, s = drop-path factor per sample)
is ONE autograd node whose forward fuses the residual add and the drop-path scale into the last
GEMM's epilogue and whose backward lets the residual-stream gradient join inside the LayerNorm
backward kernel -- no stand-alone add / mul / permute / contiguous kernels remain.
"""
import torch

from . import _lib
from ._lib import check, dtype_code, lib, ptr, require_gpu, stream
from .ops import _arena_grad, _grad_ready, packed_weight


def _pad_to(n, e):
    return ((n + e - 1) // e) * e


# ------------------------------------------------------------------------------ raw helpers (no autograd)
def lin_fwd(x2, weight, bias, out_f32=False, addend=None, row_scale=None, rows_per_scale=1, need_wd=True):
    """y[M][N] = addend + row_scale * (x2 W^T + b).  Returns y (out_features padded away)."""
    dt = x2.dtype
    m, k = x2.shape
    o = weight.shape[0]
    e = _lib.epc(dt)
    if k % e:
        raise ValueError(f'linear: in_features={k} must be a multiple of {e}')
    op = _pad_to(o, e)
    wf, _ = packed_weight(weight, dt, k, need_wd, op)
    odt = torch.float32 if (out_f32 or dt == torch.float32) else dt
    y = torch.empty((m, op), dtype=odt, device=x2.device)
    bp = bias
    if bias is not None and op != o:
        bp = torch.zeros(op, dtype=torch.float32, device=x2.device)
        bp[:o] = bias.detach()
    if (addend is not None or row_scale is not None) and op != o:
        raise ValueError('fused residual needs out_features to be a multiple of the chunk width')
    check(lib().saicv_linear_fwd(dtype_code(dt), ptr(x2), ptr(wf), ptr(bp), ptr(y), m, k, op, int(odt == torch.float32),
                                 ptr(addend), ptr(row_scale), rows_per_scale, stream()), 'linear_fwd')
    return y if op == o else y[:, :o]


def lin_bwd(x2, weight, bias, dy, need_dx=True, addend=None):
    """-> (dx or None, dw or None, db or None); a None dw/db means it was accumulated in place
    into the parameter's arena gradient."""
    dt = x2.dtype
    m, k = x2.shape
    o = weight.shape[0]
    e = _lib.epc(dt)
    op = _pad_to(o, e)
    L, st = lib(), stream()
    if op != o:
        dyp = torch.zeros((m, op), dtype=dt, device=dy.device)
        dyp[:, :o] = dy
        dy = dyp
    else:
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
    dx = dw = db = None
    if need_dx:
        _, wd = packed_weight(weight, dt, k, True, op)
        dx = torch.empty((m, k), dtype=dt, device=x2.device)
        check(L.saicv_linear_dgrad(dtype_code(dt), ptr(dy), ptr(wd), ptr(dx), m, k, op, ptr(addend), st), 'linear_dgrad')
    want_b = bias is not None and bias.requires_grad
    gb = _arena_grad(bias) if (want_b and op == o) else None
    tb = (gb if gb is not None else torch.zeros(op, dtype=torch.float32, device=x2.device)) if want_b else None
    if weight.requires_grad:
        gw = _arena_grad(weight) if (op == o and weight.is_contiguous()) else None
        tgt = gw if gw is not None else torch.zeros((op, k), dtype=torch.float32, device=x2.device)
        # the weight-gradient kernel also emits the bias gradient from the dY tiles it streams
        check(L.saicv_linear_wgrad(dtype_code(dt), ptr(dy), ptr(x2), ptr(tgt), ptr(tb), m, k, op, st), 'linear_wgrad')
        if gw is not None:
            _grad_ready(weight)
        else:
            dw = tgt[:o]
    elif want_b:
        check(L.saicv_colsum(dtype_code(dt), ptr(dy), m, op, ptr(tb), st), 'colsum')
    if want_b:
        if gb is not None:
            _grad_ready(bias)
        else:
            db = tb[:o]
    return dx, dw, db


def ln_fwd(x2, weight, bias, eps):
    m, c = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(m, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x2.device)
    check(lib().saicv_layernorm_fwd(dtype_code(x2.dtype), ptr(x2), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd),
                                    m, c, float(eps), stream()), 'layernorm_fwd')
    return y, mean, rstd


def ln_bwd(dy, x2, weight, bias, mean, rstd, addend=None):
    """-> (dx, dgamma or None, dbeta or None)  (None: accumulated into the arena gradient)"""
    m, c = x2.shape
    L = lib()
    dx = torch.empty_like(x2)
    gg, gb = _arena_grad(weight), _arena_grad(bias)
    direct = gg is not None and gb is not None
    dg = gg if direct else torch.empty(c, dtype=torch.float32, device=x2.device)
    db = gb if direct else torch.empty(c, dtype=torch.float32, device=x2.device)
    ws = torch.empty(L.saicv_layernorm_bwd_ws_floats(m, c), dtype=torch.float32, device=x2.device)
    dy = dy.contiguous()
    if dy.dtype != x2.dtype:
        dy = dy.to(x2.dtype)
    check(L.saicv_layernorm_bwd(dtype_code(x2.dtype), ptr(dy), ptr(x2), ptr(weight), ptr(mean), ptr(rstd), ptr(addend),
                                ptr(dx), ptr(dg), ptr(db), ptr(ws), m, c, int(direct), stream()), 'layernorm_bwd')
    if direct:
        _grad_ready(weight)
        _grad_ready(bias)
        return dx, None, None
    return dx, dg, db


def gelu_fwd(x):
    y = torch.empty_like(x)
    check(lib().saicv_gelu_fwd(dtype_code(x.dtype), ptr(x), ptr(y), x.numel(), stream()), 'gelu_fwd')
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    check(lib().saicv_gelu_bwd(dtype_code(x.dtype), ptr(dy), ptr(x), ptr(dx), x.numel(), stream()), 'gelu_bwd')
    return dx


def attn_fwd(qkv, b, n, heads, scale):
    c = qkv.shape[1] // 3
    out = torch.empty((b * n, c), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((b, heads, n), dtype=torch.float32, device=qkv.device)
    check(lib().saicv_attention_fwd(dtype_code(qkv.dtype), ptr(qkv), ptr(out), ptr(lse), b, n, heads, c // heads,
                                    float(scale), stream()), 'attention_fwd')
    return out, lse


def attn_bwd(qkv, out, dout, lse, b, n, heads, scale):
    c = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    check(lib().saicv_attention_bwd(dtype_code(qkv.dtype), ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), b, n,
                                    heads, c // heads, float(scale), stream()), 'attention_bwd')
    return dqkv


def row_scale(x2, scale, rows_per_scale):
    out = torch.empty_like(x2)
    check(lib().saicv_row_scale(dtype_code(x2.dtype), ptr(x2), ptr(scale), ptr(out), x2.shape[0], x2.shape[1],
                                rows_per_scale, stream()), 'row_scale')
    return out


def _as2d(x):
    x = x.contiguous()
    return x.view(-1, x.shape[-1])


# ------------------------------------------------------------------------------ fine-grained Functions
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim (reference vit.py:147,151,225)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        require_gpu(x, weight)
        x2 = _as2d(x)
        y, mean, rstd = ln_fwd(x2, weight, bias, eps)
        ctx.save_for_backward(x2, weight, bias, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias, mean, rstd = ctx.saved_tensors
        dx, dg, db = ln_bwd(_as2d(dy), x2, weight, bias, mean, rstd)
        return dx.view(dy.shape), dg, db, None


def layer_norm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, eps)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_gpu(x)
        x = x.contiguous()
        ctx.save_for_backward(x)
        return gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return gelu_bwd(dy.contiguous().to(x.dtype), x)


def gelu(x):
    return GeluFn.apply(x)


class LinearNdFn(torch.autograd.Function):
    """nn.Linear on [..., K] inputs."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_f32):
        require_gpu(x, weight)
        x2 = _as2d(x)
        y = lin_fwd(x2, weight, bias, out_f32, need_wd=ctx.needs_input_grad[0])
        ctx.save_for_backward(x2, weight, bias)
        ctx.shape = x.shape
        return y.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        dx, dw, db = lin_bwd(x2, weight, bias, _as2d(dy), ctx.needs_input_grad[0])
        return (dx.view(ctx.shape) if dx is not None else None), dw, db, None


def linear_nd(x, weight, bias=None, out_f32=False):
    return LinearNdFn.apply(x, weight, bias, out_f32)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale) v from a packed qkv tensor [B, N, 3*C] -> [B, N, C]
    (MultiHeadAttention.forward, reference vit.py:61-80, without its permutes)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        require_gpu(qkv)
        b, n, c3 = qkv.shape
        q2 = _as2d(qkv)
        out, lse = attn_fwd(q2, b, n, heads, scale)
        ctx.save_for_backward(q2, out, lse)
        ctx.cfg = (b, n, heads, scale)
        return out.view(b, n, c3 // 3)

    @staticmethod
    def backward(ctx, dout):
        q2, out, lse = ctx.saved_tensors
        b, n, heads, scale = ctx.cfg
        d2 = _as2d(dout)
        if d2.dtype != q2.dtype:
            d2 = d2.to(q2.dtype)
        dqkv = attn_bwd(q2, out, d2, lse, b, n, heads, scale)
        return dqkv.view(b, n, -1), None, None


def attention(qkv, heads, scale):
    return AttentionFn.apply(qkv, heads, scale)


# ------------------------------------------------------------------------------ fused ViT sub-layers
class AttnSubLayerFn(torch.autograd.Function):
    """out = x + s * proj(attention(qkv(LN(x))))   -- one node (reference vit.py:160)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, drop_scale, heads, eps):
        require_gpu(x, qkv_w)
        b, n, c = x.shape
        x2 = _as2d(x)
        h, mean, rstd = ln_fwd(x2, ln_w, ln_b, eps)
        qkv = lin_fwd(h, qkv_w, qkv_b)
        scale = (c // heads) ** -0.5
        a, lse = attn_fwd(qkv, b, n, heads, scale)
        out = lin_fwd(a, proj_w, proj_b, addend=x2, row_scale=drop_scale, rows_per_scale=n)
        ctx.save_for_backward(x2, ln_w, ln_b, mean, rstd, h, qkv_w, qkv_b, qkv, a, lse, proj_w, proj_b, drop_scale)
        ctx.cfg = (b, n, c, heads, scale)
        return out.view(b, n, c)

    @staticmethod
    def backward(ctx, dout):
        x2, ln_w, ln_b, mean, rstd, h, qkv_w, qkv_b, qkv, a, lse, proj_w, proj_b, drop_scale = ctx.saved_tensors
        b, n, c, heads, scale = ctx.cfg
        dy = _as2d(dout)
        if dy.dtype != x2.dtype:
            dy = dy.to(x2.dtype)
        dys = row_scale(dy, drop_scale, n) if drop_scale is not None else dy
        da, dpw, dpb = lin_bwd(a, proj_w, proj_b, dys)
        dqkv = attn_bwd(qkv, a, da, lse, b, n, heads, scale)
        dh, dqw, dqb = lin_bwd(h, qkv_w, qkv_b, dqkv)
        dx, dlw, dlb = ln_bwd(dh, x2, ln_w, ln_b, mean, rstd, addend=dy)     # + residual-stream gradient
        return dx.view(b, n, c), dlw, dlb, dqw, dqb, dpw, dpb, None, None, None


class MlpSubLayerFn(torch.autograd.Function):
    """out = x + s * fc2(gelu(fc1(LN(x))))   -- one node (reference vit.py:161)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, drop_scale, eps):
        require_gpu(x, fc1_w)
        b, n, c = x.shape
        x2 = _as2d(x)
        h, mean, rstd = ln_fwd(x2, ln_w, ln_b, eps)
        f1 = lin_fwd(h, fc1_w, fc1_b)
        g = gelu_fwd(f1)
        out = lin_fwd(g, fc2_w, fc2_b, addend=x2, row_scale=drop_scale, rows_per_scale=n)
        ctx.save_for_backward(x2, ln_w, ln_b, mean, rstd, h, fc1_w, fc1_b, f1, g, fc2_w, fc2_b, drop_scale)
        ctx.cfg = (b, n, c)
        return out.view(b, n, c)

    @staticmethod
    def backward(ctx, dout):
        x2, ln_w, ln_b, mean, rstd, h, fc1_w, fc1_b, f1, g, fc2_w, fc2_b, drop_scale = ctx.saved_tensors
        b, n, c = ctx.cfg
        dy = _as2d(dout)
        if dy.dtype != x2.dtype:
            dy = dy.to(x2.dtype)
        dys = row_scale(dy, drop_scale, n) if drop_scale is not None else dy
        dg, d2w, d2b = lin_bwd(g, fc2_w, fc2_b, dys)
        df1 = gelu_bwd(dg, f1)
        dh, d1w, d1b = lin_bwd(h, fc1_w, fc1_b, df1)
        dx, dlw, dlb = ln_bwd(dh, x2, ln_w, ln_b, mean, rstd, addend=dy)
        return dx.view(b, n, c), dlw, dlb, d1w, d1b, d2w, d2b, None, None


def attn_sublayer(x, norm, attn, drop_scale):
    return AttnSubLayerFn.apply(x, norm.weight, norm.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight,
                                attn.proj.bias, drop_scale, attn.head_nums, norm.eps)


def mlp_sublayer(x, norm, mlp, drop_scale):
    return MlpSubLayerFn.apply(x, norm.weight, norm.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias,
                               drop_scale, norm.eps)


# ------------------------------------------------------------------------------ patch embedding
class PatchEmbedFn(torch.autograd.Function):
    """Conv2d(3, C, k=p, stride=p) + bias -> tokens [B, (H/p)*(W/p), C] (PatchEmbeddingBlock,
    reference vit.py:31-43).  The NHWC conv output IS the token layout, so the reference's
    flatten(2).transpose(1, 2) costs nothing here."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        import ctypes
        from .ops import _desc, pack_input
        require_gpu(x, weight)
        xp = pack_input(x)                                  # [B, 8, H, W] NHWC, compute dtype
        dt = xp.dtype
        b, cp, h, w = xp.shape
        k, ci, r, s = weight.shape
        wf, _ = packed_weight(weight, dt, cp, False)
        d = _desc(b, h, w, cp, k, r, s, stride, 0, dt)
        y = torch.empty((b, d.OH, d.OW, k), dtype=dt, device=x.device)
        check(lib().saicv_conv2d_fwd(ctypes.byref(d), ptr(xp), ptr(wf), ptr(bias), ptr(y), 0, 0, 0, stream()),
              'patch_embed_fwd')
        ctx.save_for_backward(xp, weight, bias)
        ctx.cfg = (d, cp)
        return y.view(b, d.OH * d.OW, k)

    @staticmethod
    def backward(ctx, dy):
        import ctypes
        from .ops import _weight_grad
        xp, weight, bias = ctx.saved_tensors
        d, cp = ctx.cfg
        dt = xp.dtype
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        k = weight.shape[0]
        dw = torch.zeros((k, d.R, d.S, cp), dtype=torch.float32, device=dy.device)
        check(lib().saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(xp), ptr(dw), stream()), 'patch_embed_wgrad')
        dwt = _weight_grad(dw, weight, cp)
        db = None
        if bias is not None:
            db = torch.zeros(k, dtype=torch.float32, device=dy.device)
            check(lib().saicv_colsum(dtype_code(dt), ptr(dy), dy.numel() // k, k, ptr(db), stream()), 'colsum')
        return None, dwt, db, None


def patch_embed(x, weight, bias, stride):
    return PatchEmbedFn.apply(x, weight, bias, stride)
