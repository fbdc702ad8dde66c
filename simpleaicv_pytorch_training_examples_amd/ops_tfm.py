"""Transformer-side autograd Functions over libsaicv_hip.so (LayerNorm, GELU, Linear, fused
attention) and the two fused pre-LN sub-layers of a ViT block.

A sub-layer  out = x + s * f(LN(x))  (reference vit.py:159-163, s = drop-path factor per sample)
is ONE autograd node whose forward fuses the residual add and the drop-path scale into the last
GEMM's epilogue and whose backward lets the residual-stream gradient join inside the LayerNorm
backward kernel -- no stand-alone add / mul / permute / contiguous kernels remain.
"""
import os as _os

import torch

from . import _lib, ops
from ._lib import check, dtype_code, lib, ptr, require_gpu, stream
from .ops import KernelTimer, _arena_grad, packed_weight


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
    t0 = KernelTimer.begin('igemm_nt')
    check(lib().saicv_linear_fwd(dtype_code(dt), ptr(x2), ptr(wf), ptr(bp), ptr(y), m, k, op, int(odt == torch.float32),
                                 ptr(addend), ptr(row_scale), rows_per_scale, stream()), 'linear_fwd')
    # algorithmic bytes: each operand once, the output once, the fused addend once
    es = x2.element_size()
    KernelTimer.end(t0, 'igemm_nt', 2.0 * m * k * o,
                    float(m) * k * es + float(o) * k * es + float(m) * o * y.element_size() * (2 if addend is not None else 1))
    return y if op == o else y[:, :o]


GELU_AUX = _os.environ.get('SAICV_GELU_AUX', '1') != '0'


def lin_gelu_fwd(x2, weight, bias, aux=False):
    """(pre, act) = (x2 W^T + b, gelu(pre)) from one GEMM; falls back to two kernels when the output
    width is not a whole number of 16-byte chunks.  aux=True: the first result is gelu'(pre) instead of pre -- the only
    consumer of pre is the activation's backward, which then is one multiply in the next data gradient's epilogue
    (`lin_bwd(..., gelu_dact=...)`).  Returns (first, act, first_is_derivative)."""
    dt = x2.dtype
    m, k = x2.shape
    o = weight.shape[0]
    e = _lib.epc(dt)
    if o % e or k % e:
        pre = lin_fwd(x2, weight, bias)
        return (pre, gelu_fwd(pre), False) if aux else (pre, gelu_fwd(pre))
    wf, _ = packed_weight(weight, dt, k, True, o)
    pre = torch.empty((m, o), dtype=dt, device=x2.device)
    act = torch.empty((m, o), dtype=dt, device=x2.device)
    t0 = KernelTimer.begin('igemm_nt')
    if aux and GELU_AUX:
        check(lib().saicv_linear_gelu_fwd_aux(dtype_code(dt), ptr(x2), ptr(wf), ptr(bias), ptr(pre), ptr(act), m, k, o, stream()),
              'linear_gelu_fwd_aux')
    else:
        check(lib().saicv_linear_gelu_fwd(dtype_code(dt), ptr(x2), ptr(wf), ptr(bias), ptr(pre), ptr(act), m, k, o, stream()),
              'linear_gelu_fwd')
    es = x2.element_size()
    KernelTimer.end(t0, 'igemm_nt', 2.0 * m * k * o, float(m) * k * es + float(o) * k * es + 2.0 * m * o * es)      # two outputs
    return (pre, act, GELU_AUX) if aux else (pre, act)


_AUTO = object()


def lin_bwd(x2, weight, bias, dy, need_dx=True, addend=None, gelu_pre=None, grad_w=_AUTO, grad_b=_AUTO, want_w=None,
            want_bias=None, gelu_dact=None):
    """-> (dx or None, dw or None, db or None); a None dw/db means it was accumulated in place
    into the parameter's arena gradient.  gelu_pre: x2 = gelu(gelu_pre) and the caller wants the gradient
    with respect to gelu_pre -- the activation's backward is applied in the dgrad epilogue.
    grad_w / grad_b / want_w / want_bias: a caller whose weight / bias are row blocks of a larger parameter
    (LinearRowsFn) names the arena-gradient views and whether gradients are wanted itself."""
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
        t0 = KernelTimer.begin('igemm_nt')
        if gelu_dact is not None:                  # x2 = gelu(pre) and gelu_dact = gelu'(pre), stored by the forward
            if addend is not None or k % e:
                raise ValueError('fused GELU backward: no addend, 16-byte aligned rows')
            check(L.saicv_linear_dgrad_mul(dtype_code(dt), ptr(dy), ptr(wd), ptr(gelu_dact), ptr(dx), m, k, op, st), 'linear_dgrad_mul')
        elif gelu_pre is not None:
            if addend is not None or k % e:
                raise ValueError('fused GELU backward: no addend, 16-byte aligned rows')
            check(L.saicv_linear_dgrad_gelu(dtype_code(dt), ptr(dy), ptr(wd), ptr(gelu_pre), ptr(dx), m, k, op, st),
                  'linear_dgrad_gelu')
        else:
            check(L.saicv_linear_dgrad(dtype_code(dt), ptr(dy), ptr(wd), ptr(dx), m, k, op, ptr(addend), st), 'linear_dgrad')
        es = dy.element_size()
        fused_in = 1 if (gelu_dact is not None or gelu_pre is not None or addend is not None) else 0
        KernelTimer.end(t0, 'igemm_nt', 2.0 * m * k * o, float(m) * o * es + float(o) * k * es + float(m) * k * es * (1 + fused_in))
    want_b = (bias is not None and bias.requires_grad) if want_bias is None else (bias is not None and want_bias)
    want_w = weight.requires_grad if want_w is None else want_w
    if grad_b is _AUTO:
        gb = _arena_grad(bias) if (want_b and op == o) else None
    else:
        gb = grad_b if (want_b and op == o) else None
    tb = (gb if gb is not None else torch.zeros(op, dtype=torch.float32, device=x2.device)) if want_b else None
    if want_w:
        if grad_w is _AUTO:
            gw = _arena_grad(weight) if (op == o and weight.is_contiguous()) else None
        else:
            gw = grad_w if (op == o and weight.is_contiguous()) else None
        tgt = gw if gw is not None else torch.zeros((op, k), dtype=torch.float32, device=x2.device)
        # the weight-gradient kernel also emits the bias gradient from the dY tiles it streams
        if gw is not None and (not want_b or gb is not None) and ops.WGRAD_SIDE_STREAM:
            with ops._SideStream(dy, x2):
                t0 = KernelTimer.begin('igemm_tn')
                check(L.saicv_linear_wgrad(dtype_code(dt), ptr(dy), ptr(x2), ptr(tgt), ptr(tb), m, k, op, stream()), 'linear_wgrad')
                KernelTimer.end(t0, 'igemm_tn', 2.0 * m * k * o, float(m) * (k + o) * dy.element_size() + 4.0 * o * k)
        else:
            t0 = KernelTimer.begin('igemm_tn')
            check(L.saicv_linear_wgrad(dtype_code(dt), ptr(dy), ptr(x2), ptr(tgt), ptr(tb), m, k, op, st), 'linear_wgrad')
            KernelTimer.end(t0, 'igemm_tn', 2.0 * m * k * o, float(m) * (k + o) * dy.element_size() + 4.0 * o * k)
        if gw is None:
            dw = tgt[:o]
    elif want_b:
        check(L.saicv_colsum(dtype_code(dt), ptr(dy), m, op, ptr(tb), st), 'colsum')
    if want_b:
        if gb is None:
            db = tb[:o]
    return dx, dw, db


def ln_fwd(x2, weight, bias, eps):
    m, c = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(m, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x2.device)
    t0 = KernelTimer.begin('layernorm_fwd')
    check(lib().saicv_layernorm_fwd(dtype_code(x2.dtype), ptr(x2), ptr(weight), ptr(bias), ptr(y), ptr(mean), ptr(rstd),
                                    m, c, float(eps), stream()), 'layernorm_fwd')
    KernelTimer.end(t0, 'layernorm_fwd', 0, 2.0 * m * c * x2.element_size() + 8.0 * m)      # x read, y written, mean / rstd
    return y, mean, rstd


def ln_bwd(dy, x2, weight, bias, mean, rstd, addend=None, scaled_for=None):
    """-> (dx, dgamma or None, dbeta or None)  (None: accumulated into the arena gradient).
    scaled_for = (row factors [M / rows_per_scale] fp32, rows_per_scale): the drop-path factors of the branch that PRODUCED x2; the same
    pass also writes factor * dx and hangs it on dx as `_saicv_scaled` = (tensor, factors) for that branch's backward to pick up."""
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
    t0 = KernelTimer.begin('layernorm_bwd')
    if scaled_for is not None:
        factors, rows_per = scaled_for
        dxs = torch.empty_like(x2)
        check(L.saicv_layernorm_bwd_scaled(dtype_code(x2.dtype), ptr(dy), ptr(x2), ptr(weight), ptr(mean), ptr(rstd), ptr(addend),
                                           ptr(dx), ptr(dg), ptr(db), ptr(ws), m, c, int(direct), ptr(factors), int(rows_per), ptr(dxs),
                                           stream()), 'layernorm_bwd_scaled')
        dx._saicv_scaled = (dxs, factors)
    else:
        check(L.saicv_layernorm_bwd(dtype_code(x2.dtype), ptr(dy), ptr(x2), ptr(weight), ptr(mean), ptr(rstd), ptr(addend),
                                    ptr(dx), ptr(dg), ptr(db), ptr(ws), m, c, int(direct), stream()), 'layernorm_bwd')
    # dy, x (and the residual-stream addend) read, dx (and its drop-path twin) written; the partial dgamma / dbeta rows are noise next to them
    KernelTimer.end(t0, 'layernorm_bwd', 0, (3.0 + (1.0 if addend is not None else 0.0) + (1.0 if scaled_for is not None else 0.0)) * m * c * x2.element_size() + 8.0 * m)
    if direct:
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
    if c // heads == 32 or n > 256 or (qkv.dtype == torch.bfloat16 and c // heads == 64):
        # bf16: the streaming kernels are faster than the whole-head ones at N = 197, b256 both ways since r04 (forward 92 vs
        # 192 us, backward 255 vs 283 us: csrc/attn_stream.hip); both families keep the same natural-log lse [B, heads, N]
        q3 = qkv.view(b, n, 3 * c)
        out, lse = sattn_fwd(q3[:, :, :c], q3[:, :, c:2 * c], q3[:, :, 2 * c:], heads, scale)
        return out.view(b * n, c), lse.view(b, heads, n)
    out = torch.empty((b * n, c), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((b, heads, n), dtype=torch.float32, device=qkv.device)
    t0 = KernelTimer.begin('attention_fwd')
    check(lib().saicv_attention_fwd(dtype_code(qkv.dtype), ptr(qkv), ptr(out), ptr(lse), b, n, heads, c // heads,
                                    float(scale), stream()), 'attention_fwd')
    KernelTimer.end(t0, 'attention_fwd', 4.0 * b * heads * n * n * (c // heads), 0)
    return out, lse


def attn_bwd(qkv, out, dout, lse, b, n, heads, scale):
    c = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    if c // heads != 64 or n > 256 or qkv.dtype == torch.bfloat16:
        # head dim 32 (the MAE decoder: 512 planes / 16 heads), long sequences, or bf16: the streaming kernels both ways
        q3, g3 = qkv.view(b, n, 3 * c), dqkv.view(b, n, 3 * c)
        sattn_bwd(q3[:, :, :c], q3[:, :, c:2 * c], q3[:, :, 2 * c:], out.view(b, n, c), dout.contiguous().view(b, n, c),
                  lse.view(b * heads, n), heads, scale, g3[:, :, :c], g3[:, :, c:2 * c], g3[:, :, 2 * c:])
        return dqkv
    t0 = KernelTimer.begin('attention_bwd')
    check(lib().saicv_attention_bwd(dtype_code(qkv.dtype), ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), b, n,
                                    heads, c // heads, float(scale), stream()), 'attention_bwd')
    KernelTimer.end(t0, 'attention_bwd', 10.0 * b * heads * n * n * (c // heads), 0)        # five N x N x D products
    return dqkv


# r06 -- dropout inside a captured step.  The masks of this library's dropouts (attention probabilities, the fused residual dropout
# below) are pure functions of a seed drawn on the HOST generator at trace time; a step captured into a hipGraph freezes that seed,
# i.e. every replay would drop the same elements.  The kernels therefore add one word of device memory to the seed, and
# engine.StepGraph advances that word before every replay (an eager step draws fresh host seeds anyway).
_DROPOUT_STEP = {}


def dropout_step_word(device):
    t = _DROPOUT_STEP.get(device)
    if t is None:
        t = _DROPOUT_STEP[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def advance_dropout_step():
    for t in _DROPOUT_STEP.values():
        t.add_(40503)


def _attn_desc(q, k, v, heads, scale, key_bias, rel_h, rel_w, dropout_p=0.0, seed=0):
    """q [B, Nq, H*D], k / v [B, Nk, H*D]: any batch / row strides, unit stride on the last axis."""
    b, nq, c = q.shape
    nk = k.shape[1]
    for t in (q, k, v):
        if t.stride(-1) != 1 or t.dtype != q.dtype:
            raise ValueError('attention operands need a contiguous last axis and one dtype')
    d = _lib.AttnDesc()
    d.q, d.k, d.v = ptr(q), ptr(k), ptr(v)
    d.q_bs, d.q_rs = q.stride(0), q.stride(1)
    d.k_bs, d.k_rs = k.stride(0), k.stride(1)
    d.v_bs, d.v_rs = v.stride(0), v.stride(1)
    d.B, d.H, d.Nq, d.Nk = b, heads, nq, nk
    d.scale = float(scale)
    d.dropout_p, d.seed = float(dropout_p), int(seed) & 0xffffffff
    d.seed_device = ptr(dropout_step_word(q.device)) if dropout_p > 0 else None
    if key_bias is not None:
        if key_bias.dtype != torch.float32 or tuple(key_bias.shape) != (b, nk) or not key_bias.is_contiguous():
            raise ValueError('key_bias must be a contiguous fp32 [B, Nk] tensor')
        d.key_bias = ptr(key_bias)
    if rel_h is not None:
        for t in (rel_h, rel_w):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[0] != b * heads or t.shape[1] != nq:
                raise ValueError('rel_h / rel_w must be contiguous fp32 [B*H, Nq, S]')
        d.rel_h, d.rel_w = ptr(rel_h), ptr(rel_w)
        d.Sh, d.Sw = rel_h.shape[2], rel_w.shape[2]
    return d, c // heads


def sattn_fwd(q, k, v, heads, scale, key_bias=None, rel_h=None, rel_w=None, dropout_p=0.0, seed=0):
    """Streaming attention forward -> (out [B, Nq, C], lse [B*H, Nq])."""
    d, hd = _attn_desc(q, k, v, heads, scale, key_bias, rel_h, rel_w, dropout_p, seed)
    out = torch.empty((d.B, d.Nq, q.shape[2]), dtype=q.dtype, device=q.device)
    lse = torch.empty((d.B * heads, d.Nq), dtype=torch.float32, device=q.device)
    d.out, d.o_bs, d.o_rs, d.lse = ptr(out), out.stride(0), out.stride(1), ptr(lse)
    t0 = KernelTimer.begin('attention_fwd')
    check(lib().saicv_attention_stream_fwd(dtype_code(q.dtype), hd, d, stream()), 'attention_stream_fwd')
    KernelTimer.end(t0, 'attention_fwd', 4.0 * d.B * heads * d.Nq * d.Nk * hd, 0)
    return out, lse


def sattn_bwd(q, k, v, out, dout, lse, heads, scale, dq, dk, dv, key_bias=None, rel_h=None, rel_w=None,
              dropout_p=0.0, seed=0):
    """Streaming attention backward into dq / dk / dv (tensors or views with the strides of q / k / v).
    -> (d_rel_h, d_rel_w) or (None, None)."""
    d, hd = _attn_desc(q, k, v, heads, scale, key_bias, rel_h, rel_w, dropout_p, seed)
    for g, t in ((dq, q), (dk, k), (dv, v)):
        if g.stride() != t.stride() or g.dtype != t.dtype:
            raise ValueError('gradient views must share strides and dtype with their operands')
    if not (out.is_contiguous() and dout.is_contiguous() and dout.dtype == q.dtype):
        raise ValueError('out / dout must be contiguous [B, Nq, C] of the compute dtype')
    dsum = torch.empty_like(lse)
    d.out, d.o_bs, d.o_rs, d.lse = ptr(out), out.stride(0), out.stride(1), ptr(lse)
    d.dout, d.dq, d.dk, d.dv, d.dsum = ptr(dout), ptr(dq), ptr(dk), ptr(dv), ptr(dsum)
    drh = drw = None
    if rel_h is not None:
        drh, drw = torch.empty_like(rel_h), torch.empty_like(rel_w)
        d.d_rel_h, d.d_rel_w = ptr(drh), ptr(drw)
    t0 = KernelTimer.begin('attention_bwd')
    check(lib().saicv_attention_stream_bwd(dtype_code(q.dtype), hd, d, stream()), 'attention_stream_bwd')
    KernelTimer.end(t0, 'attention_bwd', 10.0 * d.B * heads * d.Nq * d.Nk * hd, 0)
    return drh, drw


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


class DropoutAddLayerNormFn(torch.autograd.Function):
    """LayerNorm(x + dropout_p(branch)) -- the post-norm residual of DETR's transformer layers (reference detection/models/detr.py:
    89,92,114,118,122: `norm(src + self.dropout(src2))`) as one kernel each way instead of dropout + add + LayerNorm (and, backwards,
    LayerNorm backward + masked scale + a gradient accumulation).  The mask is a counter-based function of (seed, row, column) drawn
    like the attention dropout's: a host seed at trace time plus the device-side step word (dropout_step_word), so replays of a
    captured step drop different elements.  Same distribution as nn.Dropout (keep with probability 1 - p, survivors / (1 - p)),
    another random stream."""

    @staticmethod
    def forward(ctx, x, branch, weight, bias, p, eps):
        require_gpu(x, branch, weight)
        x2 = _as2d(x)
        b2 = _as2d(branch)
        if b2.dtype != x2.dtype:
            b2 = b2.to(x2.dtype)
        m, c = x2.shape
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())          # host generator: no device synchronisation
        word = dropout_step_word(x2.device)
        s = torch.empty_like(x2)
        y = torch.empty_like(x2)
        mean = torch.empty(m, dtype=torch.float32, device=x2.device)
        rstd = torch.empty(m, dtype=torch.float32, device=x2.device)
        t0 = KernelTimer.begin('layernorm_fwd')
        check(lib().saicv_dropout_add_layernorm_fwd(dtype_code(x2.dtype), ptr(x2), ptr(b2), float(p), seed, ptr(word), ptr(weight), ptr(bias),
                                                    ptr(s), ptr(y), ptr(mean), ptr(rstd), m, c, float(eps), stream()), 'dropout_add_layernorm_fwd')
        KernelTimer.end(t0, 'layernorm_fwd', 0, 4.0 * m * c * x2.element_size() + 8.0 * m)
        ctx.save_for_backward(s, weight, bias, mean, rstd)
        ctx.cfg = (float(p), seed, branch.dtype)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        s, weight, bias, mean, rstd = ctx.saved_tensors
        p, seed, branch_dtype = ctx.cfg
        m, c = s.shape
        L = lib()
        dy2 = _as2d(dy).contiguous()
        if dy2.dtype != s.dtype:
            dy2 = dy2.to(s.dtype)
        dsum = torch.empty_like(s)
        dbranch = torch.empty_like(s)
        gg, gb = _arena_grad(weight), _arena_grad(bias)
        direct = gg is not None and gb is not None
        dg = gg if direct else torch.empty(c, dtype=torch.float32, device=s.device)
        db = gb if direct else torch.empty(c, dtype=torch.float32, device=s.device)
        ws = torch.empty(L.saicv_layernorm_bwd_ws_floats(m, c), dtype=torch.float32, device=s.device)
        t0 = KernelTimer.begin('layernorm_bwd')
        check(L.saicv_dropout_add_layernorm_bwd(dtype_code(s.dtype), ptr(dy2), ptr(s), ptr(weight), ptr(mean), ptr(rstd), p, seed,
                                                ptr(dropout_step_word(s.device)), ptr(dsum), ptr(dbranch), ptr(dg), ptr(db), ptr(ws), m, c,
                                                int(direct), stream()), 'dropout_add_layernorm_bwd')
        KernelTimer.end(t0, 'layernorm_bwd', 0, 4.0 * m * c * s.element_size() + 8.0 * m)
        dbr = dbranch.view(dy.shape)
        if dbr.dtype != branch_dtype:
            dbr = dbr.to(branch_dtype)
        return dsum.view(dy.shape), dbr, (None if direct else dg), (None if direct else db), None, None


def dropout_add_layer_norm(x, branch, weight, bias, p, eps):
    return DropoutAddLayerNormFn.apply(x, branch, weight, bias, p, eps)


class ReluDropoutFn(torch.autograd.Function):
    """dropout_p(relu(x)) -- the hidden activation of DETR's feed-forward (reference detection/models/detr.py:90-91, 120-121) -- as one
    kernel each way instead of clamp + dropout and threshold-backward + masked scale.  The keep decision is the counter-based one of
    DropoutAddLayerNormFn (host seed at trace time + the device-side step word, so replays of a captured step drop different elements);
    the backward reads the OUTPUT, which is positive exactly where the gradient passes."""

    @staticmethod
    def forward(ctx, x, p):
        require_gpu(x)
        x = x.contiguous()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())          # host generator: no device synchronisation
        y = torch.empty_like(x)
        check(lib().saicv_relu_dropout_fwd(dtype_code(x.dtype), ptr(x), ptr(y), x.numel(), float(p), seed, ptr(dropout_step_word(x.device)),
                                           stream()), 'relu_dropout_fwd')
        ctx.save_for_backward(y)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != y.dtype:
            dy = dy.to(y.dtype)
        dx = torch.empty_like(y)
        check(lib().saicv_relu_dropout_bwd(dtype_code(y.dtype), ptr(dy), ptr(y), ptr(dx), y.numel(), ctx.p, stream()), 'relu_dropout_bwd')
        return dx, None


def relu_dropout(x, p):
    """dropout(relu(x), p) in training mode; x: bf16 / fp32 on the GPU with a multiple of 8 / 4 elements"""
    return ReluDropoutFn.apply(x, float(p))


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


class LinearRowsFn(torch.autograd.Function):
    """x @ W[r0:r1]^T + b[r0:r1]: one projection out of a packed parameter (nn.MultiheadAttention.in_proj_weight /
    in_proj_bias, reference detection/models/detr.py:72-78 through F.multi_head_attention_forward).  The parameters
    enter WHOLE, so their row-block gradients go straight into the arena rows -- slicing them outside would cost a
    slice, a zero-filled full-size gradient, a copy and an accumulate per projection and step in autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, r0, r1):
        require_gpu(x, weight)
        x2 = _as2d(x)
        w = weight.detach()[r0:r1]
        if isinstance(weight, torch.nn.Parameter):
            w._saicv_rows_of = (weight, r0, r1)            # packed with every other weight by the step's one batched launch (ops._PackRegistry)
        b = bias.detach()[r0:r1] if bias is not None else None
        y = lin_fwd(x2, w, b, False, need_wd=ctx.needs_input_grad[0])
        ctx.save_for_backward(x2, weight, bias)
        ctx.cfg = (x.shape, r0, r1)
        return y.reshape(*x.shape[:-1], r1 - r0)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        shape, r0, r1 = ctx.cfg
        w = weight.detach()[r0:r1]
        if isinstance(weight, torch.nn.Parameter):
            w._saicv_rows_of = (weight, r0, r1)
        b = bias.detach()[r0:r1] if bias is not None else None
        gw_full = ops._arena_grad(weight) if weight.is_contiguous() else None
        gb_full = ops._arena_grad(bias) if bias is not None else None
        dx, dw, db = lin_bwd(x2, w, b, _as2d(dy), ctx.needs_input_grad[0],
                             grad_w=gw_full[r0:r1] if gw_full is not None else None,
                             grad_b=gb_full[r0:r1] if gb_full is not None else None,
                             want_w=ctx.needs_input_grad[1], want_bias=bias is not None and ctx.needs_input_grad[2])
        if dw is not None:                       # no arena: a full-size gradient with this block filled in
            full = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
            full[r0:r1] = dw
            dw = full
        if db is not None:
            full = torch.zeros(bias.shape, dtype=torch.float32, device=bias.device)
            full[r0:r1] = db
            db = full
        return (dx.view(shape) if dx is not None else None), dw, db, None, None


def linear_rows(x, weight, bias, r0, r1):
    return LinearRowsFn.apply(x, weight, bias, r0, r1)


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


class StreamAttentionFn(torch.autograd.Function):
    """softmax(scale * q k^T + key_bias) v for separate q [B, Nq, C], k / v [B, Nk, C] (views allowed),
    any sequence lengths, head dim 32 or 64 -- the [Nq, Nk] matrix never reaches HBM.
    key_bias [B, Nk] fp32 is ADDED to the logits (DETR's float key_padding_mask, detr.py:252-260)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, key_bias, dropout_p):
        require_gpu(q, k, v)
        # the mask is a pure function of (seed, head, query, key): drawing the seed on the host generator
        # costs no device sync and lets backward (and checkpoint re-forward) regenerate the same mask
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if dropout_p > 0 else 0
        out, lse = sattn_fwd(q, k, v, heads, scale, key_bias, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(q, k, v, out, lse, key_bias)
        ctx.cfg = (heads, scale, dropout_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, key_bias = ctx.saved_tensors
        heads, scale, dropout_p, seed = ctx.cfg
        dout = dout.contiguous()
        if dout.dtype != q.dtype:
            dout = dout.to(q.dtype)
        dq, dk, dv = torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device), \
            torch.empty_strided(k.shape, k.stride(), dtype=k.dtype, device=k.device), \
            torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)
        sattn_bwd(q, k, v, out, dout, lse, heads, scale, dq, dk, dv, key_bias, dropout_p=dropout_p, seed=seed)
        return dq, dk, dv, None, None, None, None


def stream_attention(q, k, v, heads, scale, key_bias=None, dropout_p=0.0):
    return StreamAttentionFn.apply(q, k, v, heads, scale, key_bias, float(dropout_p))


class PackedQKStreamAttentionFn(torch.autograd.Function):
    """stream_attention for q and k that are the two column halves of ONE projection output qk [B, N, 2 C] (self-attention with
    q = k = x + pos, reference detection/models/detr.py:86-88, 110-113).  Slicing qk outside costs autograd two zero fills, two
    copies and an add per call on the way back (SliceBackward of either half); here the halves are views taken inside, and the
    backward kernels write dq and dk straight into the halves of one dqk buffer (they take the operands' strides)."""

    @staticmethod
    def forward(ctx, qk, v, heads, scale, key_bias, dropout_p):
        require_gpu(qk, v)
        qk = qk.contiguous()
        c = qk.shape[-1] // 2
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if dropout_p > 0 else 0
        out, lse = sattn_fwd(qk[..., :c], qk[..., c:], v, heads, scale, key_bias, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(qk, v, out, lse, key_bias)
        ctx.cfg = (heads, scale, dropout_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qk, v, out, lse, key_bias = ctx.saved_tensors
        heads, scale, dropout_p, seed = ctx.cfg
        c = qk.shape[-1] // 2
        dout = dout.contiguous()
        if dout.dtype != qk.dtype:
            dout = dout.to(qk.dtype)
        dqk = torch.empty_like(qk)
        dv = torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)
        sattn_bwd(qk[..., :c], qk[..., c:], v, out, dout, lse, heads, scale, dqk[..., :c], dqk[..., c:], dv, key_bias,
                  dropout_p=dropout_p, seed=seed)
        return dqk, dv, None, None, None, None


def stream_attention_packed_qk(qk, v, heads, scale, key_bias=None, dropout_p=0.0):
    return PackedQKStreamAttentionFn.apply(qk, v, heads, scale, key_bias, float(dropout_p))


# ------------------------------------------------------------------------------ fused ViT sub-layers
# r06 -- the drop-path factor of a branch reaches its gradient without a pass of its own.  `out = x + s * branch(x)`: d branch = s * d out,
# and d out is what the LayerNorm backward of the NEXT sub-layer writes (LayerNorm gradient + residual-stream gradient).  That kernel
# writes s * d out beside it (saicv_layernorm_bwd_scaled): one more store instead of a load + store (24 passes over [B * N, C] per ViT-B
# step).  The plumbing rides on tensor attributes, like the BatchNorm links of ops.py: a sub-layer's output carries its factors
# (`_saicv_drop`), the sub-layer that consumes it hands them to its LayerNorm backward, whose result carries the scaled twin
# (`_saicv_scaled`, with the version counter of the gradient it belongs to: autograd accumulating another gradient into the same tensor,
# or any other route by which a different tensor arrives, falls back to the separate pass).  SAICV_LN_SCALED=0 switches it off.
LN_SCALED = _os.environ.get('SAICV_LN_SCALED', '1') != '0'


def _producer_factors(x):
    """(factors, rows per factor) of the drop-path branch that produced x, or None"""
    return getattr(x, '_saicv_drop', None) if LN_SCALED else None


def _scaled_gradient(dout, drop_scale, dtype):
    """factor * dout if the kernel that produced dout already wrote it (see above), else None"""
    tw = getattr(dout, '_saicv_scaled', None)
    if tw is None or drop_scale is None:
        return None
    twin, factors, version = tw
    if factors.data_ptr() != drop_scale.data_ptr() or version != dout._version or twin.dtype != dtype:
        return None
    return twin


def _with_twin(dx, shape):
    out = dx.view(shape)
    tw = getattr(dx, '_saicv_scaled', None)
    if tw is not None:
        out._saicv_scaled = (tw[0], tw[1], out._version)
    return out


class AttnSubLayerFn(torch.autograd.Function):
    """out = x + s * proj(attention(qkv(LN(x))))   -- one node (reference vit.py:160)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, drop_scale, heads, eps, producer=None):
        require_gpu(x, qkv_w)
        ctx.producer = producer
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
        dys = _scaled_gradient(dout, drop_scale, x2.dtype)
        dys = _as2d(dys) if dys is not None else row_scale(dy, drop_scale, n) if drop_scale is not None else dy
        da, dpw, dpb = lin_bwd(a, proj_w, proj_b, dys)
        dqkv = attn_bwd(qkv, a, da, lse, b, n, heads, scale)
        dh, dqw, dqb = lin_bwd(h, qkv_w, qkv_b, dqkv)
        dx, dlw, dlb = ln_bwd(dh, x2, ln_w, ln_b, mean, rstd, addend=dy, scaled_for=ctx.producer)     # + residual-stream gradient
        return _with_twin(dx, (b, n, c)), dlw, dlb, dqw, dqb, dpw, dpb, None, None, None, None


class MlpSubLayerFn(torch.autograd.Function):
    """out = x + s * fc2(gelu(fc1(LN(x))))   -- one node (reference vit.py:161).  residual=False: the branch alone,
    fc2(gelu(fc1(LN(x)))) -- the token half of a ConvNeXt block, whose shortcut joins after the layer scale
    (reference detection/models/backbones/dinov3convnext.py:103-117)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, drop_scale, eps, residual=True, producer=None):
        require_gpu(x, fc1_w)
        ctx.producer = producer
        b, n, c = x.shape
        x2 = _as2d(x)
        h, mean, rstd = ln_fwd(x2, ln_w, ln_b, eps)
        f1, g, f1_is_dact = lin_gelu_fwd(h, fc1_w, fc1_b, aux=True)      # f1 = gelu'(pre) when the aux form ran, else pre
        out = lin_fwd(g, fc2_w, fc2_b, addend=x2 if residual else None, row_scale=drop_scale, rows_per_scale=n)
        ctx.save_for_backward(x2, ln_w, ln_b, mean, rstd, h, fc1_w, fc1_b, f1, g, fc2_w, fc2_b, drop_scale)
        ctx.cfg = (b, n, c)
        ctx.residual = residual
        ctx.f1_is_dact = f1_is_dact
        return out.view(b, n, c)

    @staticmethod
    def backward(ctx, dout):
        x2, ln_w, ln_b, mean, rstd, h, fc1_w, fc1_b, f1, g, fc2_w, fc2_b, drop_scale = ctx.saved_tensors
        b, n, c = ctx.cfg
        dy = _as2d(dout)
        if dy.dtype != x2.dtype:
            dy = dy.to(x2.dtype)
        dys = _scaled_gradient(dout, drop_scale, x2.dtype)
        dys = _as2d(dys) if dys is not None else row_scale(dy, drop_scale, n) if drop_scale is not None else dy
        df1, d2w, d2b = lin_bwd(g, fc2_w, fc2_b, dys, **({'gelu_dact': f1} if ctx.f1_is_dact else {'gelu_pre': f1}))      # dgrad epilogue applies gelu'
        dh, d1w, d1b = lin_bwd(h, fc1_w, fc1_b, df1)
        dx, dlw, dlb = ln_bwd(dh, x2, ln_w, ln_b, mean, rstd, addend=dy if ctx.residual else None,
                              scaled_for=ctx.producer if ctx.residual else None)
        return _with_twin(dx, (b, n, c)), dlw, dlb, d1w, d1b, d2w, d2b, None, None, None, None


def _tag_drop(out, drop_scale):
    if drop_scale is not None and LN_SCALED:
        out._saicv_drop = (drop_scale, out.shape[1])
    return out


def attn_sublayer(x, norm, attn, drop_scale):
    return _tag_drop(AttnSubLayerFn.apply(x, norm.weight, norm.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight,
                                          attn.proj.bias, drop_scale, attn.head_nums, norm.eps, _producer_factors(x)), drop_scale)


def mlp_sublayer(x, norm, mlp, drop_scale):
    return _tag_drop(MlpSubLayerFn.apply(x, norm.weight, norm.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias,
                                         drop_scale, norm.eps, True, _producer_factors(x)), drop_scale)


def norm_mlp_branch(x, norm, fc1, fc2):
    """fc2(gelu(fc1(norm(x)))) on [B, N, C] tokens as one node: no shortcut, no drop-path factor"""
    return MlpSubLayerFn.apply(x, norm.weight, norm.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, None, norm.eps, False)


# ------------------------------------------------------------------------------ SAM encoder block (windows + rel-pos)
def window_partition(x4, ws):
    """[B, H, W, C] -> ([B*nW, ws*ws, C], (Hp, Wp)); zero pad to a multiple of ws
    (reference segment_anything/image_encoder.py:32-55) in one streaming kernel."""
    b, h, w, c = x4.shape
    x4 = x4.contiguous()
    nwh, nww = (h + ws - 1) // ws, (w + ws - 1) // ws
    out = torch.empty((b * nwh * nww, ws * ws, c), dtype=x4.dtype, device=x4.device)
    check(lib().saicv_window_partition(dtype_code(x4.dtype), ptr(x4), ptr(out), b, h, w, c, ws, stream()), 'window_partition')
    return out, (nwh * ws, nww * ws)


def window_unpartition(win, ws, pad_hw, hw, addend=None):
    """inverse of window_partition, dropping the padding (reference image_encoder.py:58-79); with `addend`
    ([B, H, W, C]) the residual add of Block.forward (:236) happens in the same pass."""
    hp, wp = pad_hw
    h, w = hw
    c = win.shape[-1]
    b = win.shape[0] // ((hp // ws) * (wp // ws))
    win = win.contiguous()
    out = torch.empty((b, h, w, c), dtype=win.dtype, device=win.device)
    check(lib().saicv_window_unpartition(dtype_code(win.dtype), ptr(win), ptr(addend), ptr(out), b, h, w, c, ws, stream()),
          'window_unpartition')
    return out


def _rel_tables_ok(sh, sw, rel_pos_h, rel_pos_w):
    if rel_pos_h.shape[0] != 2 * sh - 1 or rel_pos_w.shape[0] != 2 * sw - 1:
        raise ValueError(f'relative-position tables of length {rel_pos_h.shape[0]} / {rel_pos_w.shape[0]} for a {sh} x {sw} grid: '
                         'resample them first (ops_tfm.resize_rel_pos; sam_attn_sublayer does)')
    if rel_pos_h.shape[1] != 64 or not (rel_pos_h.is_contiguous() and rel_pos_w.is_contiguous()):
        raise NotImplementedError('relative-position tables must be contiguous [2S-1, 64] fp32')


def relpos_fwd(q, heads, sh, sw, rel_pos_h, rel_pos_w):
    """Decomposed relative-position logits (add_decomposed_rel_pos, reference image_encoder.py:116-144) from the
    UNSCALED query view q [Bw, N, C]: -> rel_h [Bw*heads, N, sh], rel_w [Bw*heads, N, sw] (fp32)."""
    bw, n, c = q.shape
    _rel_tables_ok(sh, sw, rel_pos_h, rel_pos_w)
    rel_h = torch.empty((bw * heads, n, sh), dtype=torch.float32, device=q.device)
    rel_w = torch.empty((bw * heads, n, sw), dtype=torch.float32, device=q.device)
    check(lib().saicv_relpos_fwd(dtype_code(q.dtype), ptr(q), q.stride(1), q.stride(0), ptr(rel_pos_h.detach()),
                                 ptr(rel_pos_w.detach()), ptr(rel_h), ptr(rel_w), bw, heads, sh, sw, stream()), 'relpos_fwd')
    return rel_h, rel_w


def relpos_bwd(q, dq, heads, sh, sw, rel_pos_h, rel_pos_w, drh, drw, want_tables):
    """dq += d_rel . tables (in place, layout of q); -> (d rel_pos_h, d rel_pos_w): tensors, or None when the
    gradient was accumulated straight into the parameter's arena gradient / not wanted."""
    bw = q.shape[0]
    gh = gw = th = tw = ws = None
    if want_tables:
        ws = torch.empty(lib().saicv_relpos_bwd_ws_floats(sh, sw), dtype=torch.float32, device=q.device)
        th, tw = _arena_grad(rel_pos_h), _arena_grad(rel_pos_w)
        direct = th is not None and tw is not None
        if not direct:
            th = torch.zeros_like(rel_pos_h, dtype=torch.float32)
            tw = torch.zeros_like(rel_pos_w, dtype=torch.float32)
    check(lib().saicv_relpos_bwd(dtype_code(q.dtype), ptr(q), ptr(dq), q.stride(1), q.stride(0), ptr(rel_pos_h.detach()),
                                 ptr(rel_pos_w.detach()), ptr(drh), ptr(drw), ptr(th), ptr(tw), ptr(ws), bw, heads, sh, sw, stream()),
          'relpos_bwd')
    if want_tables:
        if not direct:
            gh, gw = th, tw
    return gh, gw


class SamAttnSubLayerFn(torch.autograd.Function):
    """out = x + unpartition(proj(rel_pos_attention(qkv(partition(LN(x))))))  -- the attention half of a
    SAM encoder Block as ONE autograd node (reference segment_anything/image_encoder.py:147-184, 222-236).
    x [B, H, W, C]; window = 0 -> global attention over H*W tokens.  The [N, N] logits never reach HBM."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, rel_pos_h, rel_pos_w, heads, eps, window):
        require_gpu(x, qkv_w)
        b, hh, ww, c = x.shape
        x2 = _as2d(x)
        h, mean, rstd = ln_fwd(x2, ln_w, ln_b, eps)
        if window > 0:
            hw, pad_hw = window_partition(h.view(b, hh, ww, c), window)
            sh = sw = window
        else:
            hw, pad_hw = h.view(b, hh * ww, c), (hh, ww)
            sh, sw = hh, ww
        bw, n, _ = hw.shape
        hw2 = hw.view(-1, c)
        qkv = lin_fwd(hw2, qkv_w, qkv_b).view(bw, n, 3 * c)
        q, k, v = qkv[:, :, :c], qkv[:, :, c:2 * c], qkv[:, :, 2 * c:]
        scale = (c // heads) ** -0.5
        rel_h, rel_w = relpos_fwd(q, heads, sh, sw, rel_pos_h, rel_pos_w)
        a, lse = sattn_fwd(q, k, v, heads, scale, None, rel_h, rel_w)
        a2 = a.view(-1, c)
        if window > 0:
            p = lin_fwd(a2, proj_w, proj_b)
            out = window_unpartition(p.view(bw, n, c), window, pad_hw, (hh, ww), addend=x2)     # x + unpartition(p)
        else:
            out = lin_fwd(a2, proj_w, proj_b, addend=x2).view(b, hh, ww, c)
        ctx.save_for_backward(x2, ln_w, ln_b, mean, rstd, hw2, qkv_w, qkv_b, qkv, a, lse, rel_h, rel_w, proj_w,
                              proj_b, rel_pos_h, rel_pos_w)
        ctx.cfg = (b, hh, ww, c, heads, scale, window, pad_hw, sh, sw)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x2, ln_w, ln_b, mean, rstd, hw2, qkv_w, qkv_b, qkv, a, lse, rel_h, rel_w, proj_w, proj_b, rel_pos_h,
         rel_pos_w) = ctx.saved_tensors
        b, hh, ww, c, heads, scale, window, pad_hw, sh, sw = ctx.cfg
        dt = x2.dtype
        dy = dout.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        bw, n, _ = qkv.shape
        dp = window_partition(dy.view(b, hh, ww, c), window)[0].view(-1, c) if window > 0 else dy.view(-1, c)
        da, dpw, dpb = lin_bwd(a.view(-1, c), proj_w, proj_b, dp)
        q, k, v = qkv[:, :, :c], qkv[:, :, c:2 * c], qkv[:, :, 2 * c:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :, :c], dqkv[:, :, c:2 * c], dqkv[:, :, 2 * c:]
        drh, drw = sattn_bwd(q, k, v, a, da.view(bw, n, c), lse, heads, scale, dq, dk, dv, None, rel_h, rel_w)
        # extra query gradient and the table gradients of the relative-position logits
        g_rh, g_rw = relpos_bwd(q, dq, heads, sh, sw, rel_pos_h, rel_pos_w, drh, drw,
                                ctx.needs_input_grad[7] or ctx.needs_input_grad[8])
        dhw, dqw, dqb = lin_bwd(hw2, qkv_w, qkv_b, dqkv.view(-1, 3 * c))
        dh = window_unpartition(dhw.view(bw, n, c), window, pad_hw, (hh, ww)) if window > 0 else dhw
        dx, dlw, dlb = ln_bwd(dh.reshape(-1, c), x2, ln_w, ln_b, mean, rstd, addend=dy.view(-1, c))
        return dx.view(b, hh, ww, c), dlw, dlb, dqw, dqb, dpw, dpb, g_rh, g_rw, None, None, None


_REL_RESIZE = {}


def resize_rel_pos(table, length):
    """A relative-position table [L, C] resampled to [length, C]: the 1-D linear interpolation (half-pixel centres, edge clamp)
    that get_rel_pos applies when a checkpoint's table was trained on another grid (reference segment_anything/
    image_encoder.py:96-103, F.interpolate(mode='linear')).  Written as table' = A . table with the constant [length, L]
    two-diagonal interpolation matrix A: a 127 x 27 x 64 product is tensor glue, and autograd's matmul backward (A^T . d table')
    carries the gradient back to the parameter -- the kernels downstream see an ordinary [2S-1, 64] fp32 table."""
    src_len = table.shape[0]
    if src_len == length:
        return table
    key = (src_len, length, table.device)
    a = _REL_RESIZE.get(key)
    if a is None:
        # fp32 like ATen's upsample_linear1d (scale = L / length as a float, index = scale * (i + 0.5) - 0.5)
        scale = torch.tensor(src_len, dtype=torch.float32) / torch.tensor(length, dtype=torch.float32)
        pos = (scale * (torch.arange(length, dtype=torch.float32) + 0.5) - 0.5).clamp_(min=0.0)
        lo = pos.floor().long().clamp_(max=src_len - 1)
        hi = (lo + 1).clamp_(max=src_len - 1)
        frac = pos - lo.float()
        a = torch.zeros((length, src_len), dtype=torch.float32)
        rows = torch.arange(length)
        a.index_put_((rows, lo), 1.0 - frac, accumulate=True)
        a.index_put_((rows, hi), frac, accumulate=True)
        a = a.to(table.device)
        _REL_RESIZE[key] = a
    with torch.autocast(table.device.type, enabled=False):       # the tables stay fp32 whatever the surrounding autocast says
        return (a @ table.float()).contiguous()


def sam_attn_sublayer(x, norm, attn, window):
    sh, sw = (window, window) if window > 0 else (x.shape[1], x.shape[2])
    rel_h = resize_rel_pos(attn.rel_pos_h, 2 * sh - 1)
    rel_w = resize_rel_pos(attn.rel_pos_w, 2 * sw - 1)
    return SamAttnSubLayerFn.apply(x, norm.weight, norm.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight,
                                   attn.proj.bias, rel_h, rel_w, attn.head_nums, norm.eps, window)


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
        t0 = KernelTimer.begin('patch_embed')
        check(lib().saicv_conv2d_fwd(ctypes.byref(d), ptr(xp), ptr(wf), ptr(bias), ptr(y), 0, 0, 0, stream()),
              'patch_embed_fwd')
        KernelTimer.end(t0, 'patch_embed', 2.0 * b * d.OH * d.OW * k * r * s * ci, 0)
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


# ------------------------------------------------------------------------------ SAM mask-decoder tail (csrc/samtail.hip)
class HyperProductFn(torch.autograd.Function):
    """masks[b, t, p] = <hyper[b, t, :], x[b, p, :]> -- the hyper-network product of the mask decoder (reference
    segment_anything/mask_decoder.py:137-140) as one streaming kernel each way: x [B, P, 32] is read once, the
    [B, T, P] logits written once; backward gives dx in one pass and dhyper by per-block partial sums."""

    @staticmethod
    def forward(ctx, x, hyper):
        require_gpu(x, hyper)
        b, p, c = x.shape
        t = hyper.shape[1]
        x = x.contiguous()
        hyper = hyper.to(x.dtype).contiguous()
        out = torch.empty((b, t, p), dtype=x.dtype, device=x.device)
        check(lib().saicv_hyper_product_fwd(dtype_code(x.dtype), ptr(x), ptr(hyper), ptr(out), b, t, p, c, stream()),
              'hyper_product_fwd')
        ctx.save_for_backward(x, hyper)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, hyper = ctx.saved_tensors
        b, p, c = x.shape
        t = hyper.shape[1]
        dout = dout.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dh = torch.empty((b, t, c), dtype=torch.float32, device=x.device)
        check(lib().saicv_hyper_product_bwd(dtype_code(x.dtype), ptr(x), ptr(hyper), ptr(dout), ptr(dx), ptr(dh), b, t, p, c,
                                            stream()), 'hyper_product_bwd')
        return dx, dh


def hyper_product(x, hyper):
    return HyperProductFn.apply(x, hyper)


class Upsample4Fn(torch.autograd.Function):
    """F.interpolate(x, scale 4, mode="bilinear", align_corners=False) on [B, M, h, w] (reference sam.py:155-158)."""

    @staticmethod
    def forward(ctx, low):
        require_gpu(low)
        b, m, h, w = low.shape
        low = low.contiguous()
        out = torch.empty((b, m, 4 * h, 4 * w), dtype=low.dtype, device=low.device)
        check(lib().saicv_upsample4_fwd(dtype_code(low.dtype), ptr(low), ptr(out), b * m, h, w, stream()), 'upsample4_fwd')
        ctx.shape = (b, m, h, w)
        return out

    @staticmethod
    def backward(ctx, dhi):
        b, m, h, w = ctx.shape
        dhi = dhi.contiguous()
        dlow = torch.empty((b, m, h, w), dtype=dhi.dtype, device=dhi.device)
        check(lib().saicv_upsample4_bwd(dtype_code(dhi.dtype), ptr(dhi), ptr(dlow), b * m, h, w, stream()), 'upsample4_bwd')
        return dlow


def upsample4_bilinear(low):
    return Upsample4Fn.apply(low)


# ---------------------------------------------------------------------------------------------- DINOv3 block pieces
class RopeFn(torch.autograd.Function):
    """Rotary position embedding on the q / k thirds of a packed projection [B, N, 3*C] (reference detection/models/backbones/
    dinov3vit.py:262-276, :331-353): one pass, v and the `prefix` leading tokens copied; the backward is the transposed map."""

    @staticmethod
    def forward(ctx, qkv, sin, cos, heads, prefix):
        require_gpu(qkv, sin, cos)
        b, n, c3 = qkv.shape
        d = c3 // 3 // heads
        qkv = qkv.contiguous()
        sin, cos = sin.float().contiguous(), cos.float().contiguous()
        if tuple(sin.shape) != (n - prefix, d) or tuple(cos.shape) != (n - prefix, d):
            raise ValueError(f'rope tables {tuple(sin.shape)} for {n - prefix} tokens of head dim {d}')
        out = torch.empty_like(qkv)
        check(lib().saicv_rope_apply(dtype_code(qkv.dtype), ptr(qkv), ptr(sin), ptr(cos), ptr(out), b, n, heads, d, prefix, 0,
                                     stream()), 'rope_apply')
        ctx.save_for_backward(sin, cos)
        ctx.cfg = (b, n, heads, d, prefix)
        return out

    @staticmethod
    def backward(ctx, dout):
        sin, cos = ctx.saved_tensors
        b, n, heads, d, prefix = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty_like(dout)
        check(lib().saicv_rope_apply(dtype_code(dout.dtype), ptr(dout), ptr(sin), ptr(cos), ptr(dx), b, n, heads, d, prefix, 1,
                                     stream()), 'rope_apply')
        return dx, None, None, None, None


def rope(qkv, sin, cos, heads, prefix=0):
    return RopeFn.apply(qkv, sin, cos, heads, prefix)


class SwiGluFn(torch.autograd.Function):
    """silu(x1) * x2 (reference dinov3vit.py:137-140) and its gradients, one pass each."""

    @staticmethod
    def forward(ctx, x1, x2):
        require_gpu(x1, x2)
        x1, x2 = x1.contiguous(), x2.contiguous()
        out = torch.empty_like(x1)
        check(lib().saicv_swiglu_fwd(dtype_code(x1.dtype), ptr(x1), ptr(x2), ptr(out), x1.numel(), stream()), 'swiglu_fwd')
        ctx.save_for_backward(x1, x2)
        return out

    @staticmethod
    def backward(ctx, dy):
        x1, x2 = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x1.dtype:
            dy = dy.to(x1.dtype)
        d1, d2 = torch.empty_like(x1), torch.empty_like(x2)
        check(lib().saicv_swiglu_bwd(dtype_code(x1.dtype), ptr(dy), ptr(x1), ptr(x2), ptr(d1), ptr(d2), x1.numel(), stream()), 'swiglu_bwd')
        return d1, d2


def swiglu(x1, x2):
    return SwiGluFn.apply(x1, x2)
