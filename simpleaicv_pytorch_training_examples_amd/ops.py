"""torch.autograd.Functions over the C-ABI of libsaicv_hip.so.

Activations are NCHW-shaped, NHWC-strided (torch.channels_last) tensors in the compute dtype
(bf16 under autocast = perf mode, fp32 otherwise = parity mode).  Statistics, logits, losses
and every parameter gradient are fp32, as under the reference's autocast region
(reference tools/scripts.py:153-156).
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check, dtype_code, lib, ptr, require_gpu, stream

_weights_epoch = [0]


class KernelTimer:
    """HIP-event bracketing of selected launches on torch's current stream (the stream every
    saicv kernel is launched on).  bench.py enables it to price the dominant kernel against
    its roofline from live measurements; it is off by default (zero overhead)."""
    enabled = False
    only = None           # optional set of tags to bracket (every event pair costs host time: ~1400 per ResNet-50 step
                          # with all tags made the bench host-bound and 7 % slower; bench.py brackets the dominant kernel)
    records = []          # (tag, start_event, end_event, algorithmic_flops, algorithmic_bytes)
    PEAK_FLOPS, PEAK_BYTES = 2.5e15, 8.0e12      # dense bf16 MFMA, HBM3E (MI355X_MICROARCH.md)

    @classmethod
    def begin(cls, tag=None):
        if not cls.enabled or (cls.only is not None and tag not in cls.only):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    @classmethod
    def end(cls, e0, tag, flops, nbytes):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        cls.records.append((tag, e0, e1, flops, nbytes))

    @classmethod
    def summary(cls):
        """{tag: dict(calls, ms, flops, bytes, bound_ms)}; call after torch.cuda.synchronize().
        bound_ms = sum over the launches of max(flops / MFMA peak, bytes / HBM peak): what the SHAPES allow."""
        out = {}
        for tag, e0, e1, fl, by in cls.records:
            d = out.setdefault(tag, {'calls': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'bound_ms': 0.0})
            d['calls'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['flops'] += fl
            d['bytes'] += by
            d['bound_ms'] += max(fl / cls.PEAK_FLOPS, by / cls.PEAK_BYTES) * 1e3
        return out


def bump_weights_epoch():
    """Called by the flat-arena optimizer after it rewrote parameters through raw pointers: the step boundary."""
    _weights_epoch[0] += 1
    _ZeroPool.reset()


class _ZeroPool:
    """fp32 scratch that is all zeros when handed out: the few-row buffers BatchNorm statistics are added into with atomics
    (ConvBnActFn, SAICV_BN_INLINE).  One memset per step boundary instead of one per layer; the slices keep their addresses
    from step to step, so a captured step replays against the same memory."""
    CHUNK = 1 << 20            # floats
    LIMIT = 64 << 20           # floats handed out without a step boundary before falling back to torch.zeros per request
    chunks = []                # [tensor, used]
    handed = 0

    @classmethod
    def take(cls, want, device):
        n = (want + 63) // 64 * 64                 # slices start on 256-byte boundaries
        if cls.handed > cls.LIMIT:                 # nobody calls the step boundary (a foreign optimizer): stay bounded
            return torch.zeros(want, dtype=torch.float32, device=device)
        cls.handed += n
        for ch in cls.chunks:
            if ch[0].device == device and ch[1] + n <= ch[0].numel():
                v = ch[0][ch[1]:ch[1] + want]
                ch[1] += n
                return v
        t = torch.zeros(max(n, cls.CHUNK), dtype=torch.float32, device=device)
        cls.chunks.append([t, n])
        return t[:want]

    @classmethod
    def zero_all(cls):
        """Zeroes every chunk in full (a few MB).  A captured step replays its BatchNorm atomics into fixed slices and
        needs them zero whatever ran eagerly since the last step boundary (a train-mode forward without an optimizer step
        takes the same slices and leaves its sums behind): engine.StepGraph captures this call in front of the step."""
        for ch in cls.chunks:
            ch[0].zero_()

    @classmethod
    def reset(cls):
        for ch in cls.chunks:
            if ch[1]:
                ch[0][:ch[1]].zero_()
                ch[1] = 0
        cls.handed = 0


def _stat_rows(tile_rows):
    """Rows the atomically accumulated statistics are spread over: enough to keep the atomics of a many-tile layer apart,
    few enough for every workgroup of the consuming kernel to sum them."""
    return 8 if tile_rows >= 512 else 4 if tile_rows >= 64 else 2 if tile_rows >= 8 else 1


def _arena_grad(t):
    """t.grad when it is a persistent view into the engine's flat gradient arena that kernels
    may accumulate into directly (set up by engine.FlatArena), else None.

    A kernel that wrote in place returns None for that input; autograd still runs the leaf's
    AccumulateGrad node -- exactly once per backward, after EVERY use of the parameter has run its
    backward -- and with it the post-accumulate hook engine.FlatArena registers.  That hook is the
    only "gradient complete" signal the DDP engine and the optimizers act on, so a parameter used
    several times per step (DETR's shared decoder norm, the SAM decoder passes) is never reduced
    or counted early."""
    if t is None or not getattr(t, '_saicv_direct', False):
        return None
    g = t.grad
    if g is None or g.dtype != torch.float32 or g.stride() != t.stride():
        return None
    return g


# ------------------------------------------------------------------------------ side stream for weight gradients
# A layer's weight gradient (igemm_tn, MFMA-bound, accumulates straight into the gradient arena) has no consumer
# until the all-reduce / optimizer, while the data-gradient -> BatchNorm-backward chain of the next layer (HBM-bound
# streaming kernels) is on the critical path.  Launching the weight gradients on a second HIP stream lets the two
# kinds of kernels share the GPU (and, inside a captured step graph, makes them parallel branches).
# Measured on MI355X (r02b): neutral to -2 % on ResNet-50 / ViT-B -- both kernel families fill every CU slot by
# themselves, so the two queues mostly alternate instead of overlapping.  Off by default; SAICV_WGRAD_SIDE=1 turns it on.
import os as _os

WGRAD_SIDE_STREAM = _os.environ.get('SAICV_WGRAD_SIDE', '0') == '1'
# BatchNorm-backward traffic cuts in residual networks: the shortcut gradient travels as
# (dz, ReLU-mask) instead of a masked copy, and a BatchNorm's backward reduction comes out of the epilogue of the data
# gradient that produces its dz.  SAICV_BN_FUSE=0 restores the three-pass form (A/B runs, tests of both paths).
BN_FUSE = _os.environ.get('SAICV_BN_FUSE', '1') == '1'
# BatchNorm statistics added atomically into a few zeroed rows and finalised inside the consuming kernel (no partial-reduce /
# finalize launches); SAICV_BN_INLINE=0 keeps one partial row per tile row and the finalize kernels (bit-reproducible sums)
BN_INLINE = _os.environ.get('SAICV_BN_INLINE', '1') == '1'


def set_deterministic(on=True):
    """The engine's counterpart of `torch.backends.cudnn.deterministic = True` (reference tools/utils.py:106-107): with `on`, every
    reduction of libsaicv_hip.so is ordered (csrc/det.h: partials parked side by side, folded in index order) and the BatchNorm
    statistics of the convolution epilogues take the fixed-order partial rows -- an fp32 step is then bit-reproducible run to run.
    Costs a fold launch per weight gradient; the fast default adds partials with fp32 atomics in completion order.
    tools.utils.set_seed() turns it on (SAICV_DETERMINISTIC=0 in the environment keeps the fast path, as bench.py does);
    SAICV_DETERMINISTIC=1 turns it on at import.  Returns the previous setting."""
    global BN_INLINE
    L = lib()
    prev = bool(L.saicv_set_deterministic(1 if on else 0))
    if on:
        global WGRAD_SIDE_STREAM
        BN_INLINE = False
        WGRAD_SIDE_STREAM = False           # (one partials workspace per stream; a forked branch of a capture would share it)
        if torch.cuda.is_available():
            check(L.saicv_deterministic_prepare(stream()), 'deterministic_prepare')
    else:
        BN_INLINE = _os.environ.get('SAICV_BN_INLINE', '1') == '1'
    return prev


def is_deterministic():
    return bool(lib().saicv_get_deterministic())


if _os.environ.get('SAICV_DETERMINISTIC') == '1' and _lib.available():
    lib().saicv_set_deterministic(1)          # (the workspace is allocated by the first reduction: no device context at import)
    BN_INLINE = False


class _BnLink:
    """What the data gradient of the NEXT conv needs to produce the backward partial sums of a BatchNorm(+ReLU) node,
    and where that node finds them.  Travels forward as an attribute of the node's output tensor."""
    __slots__ = ('y', 'mask', 'mean', 'invstd', 'part', 'rows', 'dx', 'dx_version', 'inline')

    def __init__(self, y, mask, mean, invstd):
        self.y, self.mask, self.mean, self.invstd = y, mask, mean, invstd
        self.part = self.dx = None
        self.rows = self.dx_version = 0
        self.inline = False


class _GateLedger:
    """Gated shortcut gradients handed out in the running backward and not yet consumed.  A gradient that reaches a
    node that does not know about its gate would silently be used unmasked: the end-of-backward check turns that into
    an error."""
    pending = 0
    queued = False

    @classmethod
    def hand_out(cls):
        cls.pending += 1
        if not cls.queued:
            cls.queued = True
            torch.autograd.Variable._execution_engine.queue_callback(cls._check)

    @classmethod
    def consume(cls):
        cls.pending -= 1

    @classmethod
    def _check(cls):
        n, cls.pending, cls.queued = cls.pending, 0, False
        if n != 0:
            raise RuntimeError(f'{n} gated shortcut gradient(s) did not reach a node that applies the gate (a residual '
                               'tensor with several consumers?); rerun with SAICV_BN_FUSE=0')


def _take_gate(t):
    """ReLU-mask that still has to be applied to gradient tensor t (handed out by a residual node), or None."""
    g = getattr(t, '_saicv_gate', None) if t is not None else None
    if g is not None:
        if t._version != t._saicv_gate_version:
            # autograd summed another gradient into this tensor in place: the gate no longer describes its content
            raise RuntimeError('a gated shortcut gradient was accumulated into before its gate was applied (a residual '
                               'tensor with several consumers); rerun with SAICV_BN_FUSE=0')
        _GateLedger.consume()
        t._saicv_gate = None
    return g
_side = {'stream': None, 'dirty': False}


class _SideStream:
    """with _SideStream(tensors...): launches inside run on the side stream, ordered after everything already
    enqueued on the current stream; `tensors` are kept alive for the side stream (record_stream)."""

    def __init__(self, *tensors):
        self.tensors = tensors

    def __enter__(self):
        if _side['stream'] is None:
            _side['stream'] = torch.cuda.Stream()
        side = _side['stream']
        side.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            if t is not None:
                t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        _side['dirty'] = True
        return side

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def side_stream_in_use():
    """the side stream if weight gradients have been launched on it since the last join, else None"""
    return _side['stream'] if _side['dirty'] else None


def join_side_stream():
    """Makes the current stream wait for the weight gradients launched on the side stream.  Called by every consumer
    of the gradient arena (bucket all-reduce, inf/nan check, clipping, optimizer step) and at the end of a captured
    step; cheap when nothing is pending."""
    if _side['dirty']:
        torch.cuda.current_stream().wait_stream(_side['stream'])
        _side['dirty'] = False


def compute_dtype():
    if torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        if dt != torch.bfloat16:
            raise RuntimeError(f'saicv kernels run bf16 or fp32; autocast dtype {dt} is not supported '
                               '(MI355X perf mode is bf16)')
        return dt
    return torch.float32


def _nhwc(x):
    """Returns x as a dense NHWC-strided tensor (no copy when it already is)."""
    if x.dim() != 4:
        raise ValueError('expected a 4-d NCHW-shaped tensor')
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return x


def _empty_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


class ResizeBilinearAddFn(torch.autograd.Function):
    """F.interpolate(top, size=lateral.shape[2:], mode='bilinear') + lateral, the top-down merge of a feature pyramid (reference
    SimpleAICV/detection/models/fpn.py:57-75), as one pass over NHWC tensors; fp32 output as under torch.autocast (which runs the
    resize in fp32).  The gradient towards `top` is a gather in a fixed order -- ATen's upsample backward scatters with atomics, the
    one place where a RetinaNet / FCOS step was not reproducible in deterministic mode."""

    @staticmethod
    def forward(ctx, top, lateral):
        require_gpu(top, lateral)
        top, lateral = _nhwc(top), _nhwc(lateral)
        n, c, h, w = top.shape
        _, _, H, W = lateral.shape
        out = _empty_nhwc(n, c, H, W, torch.float32, top.device)
        check(lib().saicv_resize_bilinear_add_fwd(dtype_code(top.dtype), dtype_code(lateral.dtype), ptr(top), ptr(lateral), ptr(out),
                                                  n, h, w, H, W, c, stream()), 'resize_bilinear_add_fwd')
        ctx.geom = (n, c, h, w, H, W, top.dtype, lateral.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, c, h, w, H, W, tdt, ldt = ctx.geom
        dout = _nhwc(dout.float())
        dtop = None
        if ctx.needs_input_grad[0]:
            dtop = _empty_nhwc(n, c, h, w, tdt, dout.device)
            check(lib().saicv_resize_bilinear_bwd(dtype_code(tdt), ptr(dout), ptr(dtop), n, h, w, H, W, c, stream()), 'resize_bilinear_bwd')
        return dtop, (dout.to(ldt) if ctx.needs_input_grad[1] else None)


def resize_bilinear_add(top, lateral):
    return ResizeBilinearAddFn.apply(top, lateral)


_desc_cache = {}


def _desc(N, H, W, C, K, R, S, stride, pad, dt):
    key = (N, H, W, C, K, R, S, stride, pad, dt)
    d = _desc_cache.get(key)
    if d is None:
        OH = (H + 2 * pad - R) // stride + 1
        OW = (W + 2 * pad - S) // stride + 1
        d = ConvDesc(N, H, W, C, K, R, S, stride, pad, OH, OW, dtype_code(dt))
        _desc_cache[key] = d
    return d


# ------------------------------------------------------------------------------ packing
def pack_input(x, dtype=None, cp=8):
    """NCHW-shaped image batch (any strides) -> NHWC compute-dtype tensor with C padded to cp."""
    require_gpu(x)
    if dtype is None:
        dtype = compute_dtype()
    if x.dtype != torch.float32:
        x = x.float()
    n, c, h, w = x.shape
    cp = max(cp, ((c + _lib.epc(dtype) - 1) // _lib.epc(dtype)) * _lib.epc(dtype))
    out = torch.empty((n, h, w, cp), dtype=dtype, device=x.device)
    sn, sc, sh, sw = x.stride()
    check(lib().saicv_pack_input(dtype_code(dtype), ptr(x), sn, sc, sh, sw, ptr(out), n, c, h, w, cp,
                                 stream()), 'pack_input')
    return out.permute(0, 3, 1, 2)


STEM_S2D = _os.environ.get('SAICV_STEM_S2D', '1') == '1'
# the convolution + BatchNorm shortcut of a residual block hands its raw output to the block's join, which applies the
# shortcut's BatchNorm in the same pass as the main branch's (one write + one read of the widest tensor of the block less);
# SAICV_DS_JOIN_FUSE=0 materialises the normalised shortcut as before
DS_JOIN_FUSE = _os.environ.get('SAICV_DS_JOIN_FUSE', '1') == '1'
# BatchNorm-apply + ReLU + MaxPool of the ResNet stem as one pass (csrc/pool.hip bn_relu_maxpool_*); 0: the unfused pair
STEM_POOL_FUSE = _os.environ.get('SAICV_STEM_POOL_FUSE', '1') == '1'


def pack_stem_input(x, conv, dtype=None):
    """Input of a stem convolution in the layout its kernel wants: for a K x K stride-2 stem (ResNet: 7 x 7, padding 3,
    reference resnet.py:172-174) the space-to-depth image of saicv_pack_input_s2d -- the convolution then runs as a
    stride-1 (K+1)/2-tap one with 4C (padded to 16) channels, K dimension 256 instead of 392 -- else pack_input()."""
    k = conv.kernel_size[0]
    if (STEM_S2D and not x.requires_grad and conv.stride == (2, 2) and conv.kernel_size[0] == conv.kernel_size[1] and k % 2 == 1
            and conv.padding == (k // 2, k // 2) and conv.groups == 1 and 4 * x.shape[1] <= 16):
        require_gpu(x)
        dtype = dtype or compute_dtype()
        if x.dtype != torch.float32:
            x = x.float()
        n, c, h, w = x.shape
        pad = k // 2
        hq, wq, cq = (h + 2 * pad + 1) // 2, (w + 2 * pad + 1) // 2, 16
        out = torch.empty((n, hq, wq, cq), dtype=dtype, device=x.device)
        sn, sc, sh, sw = x.stride()
        check(lib().saicv_pack_input_s2d(dtype_code(dtype), ptr(x), sn, sc, sh, sw, ptr(out), n, c, h, w, pad, cq, stream()),
              'pack_input_s2d')
        out = out.permute(0, 3, 1, 2)
        out._saicv_s2d = (c, h, w, k, pad)
        return out
    return pack_input(x, dtype)


def _packed_weight_s2d(weight, dtype, cq):
    """Wf[O][(R+1)/2][(S+1)/2][cq] of a stride-2 stem weight regrouped for the space-to-depth input; cached like packed_weight."""
    key = (weight._version, _weights_epoch[0], dtype, cq, weight.data_ptr())
    cache = getattr(weight, '_saicv_pack_s2d', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    w = weight.detach()
    o, i, r, s = w.shape
    so, si, sr, ss = w.stride()
    wf = torch.empty((o, (r + 1) // 2, (s + 1) // 2, cq), dtype=dtype, device=w.device)
    check(lib().saicv_pack_weight_s2d(dtype_code(dtype), ptr(w), so, si, sr, ss, o, i, r, s, cq, ptr(wf), stream()),
          'pack_weight_s2d')
    weight._saicv_pack_s2d = (key, wf)
    return wf


class _PackRegistry:
    """Compute-dtype copies of every parameter that went through packed_weight(), refreshed by ONE batched launch
    (saicv_pack_weight_batched) the first time one of them is asked for after the optimizer changed the weights, instead of
    one launch per layer and step.  The copies keep their storage, so a captured step replays against the same pointers."""
    BATCH = _os.environ.get('SAICV_PACK_BATCH', '1') == '1'
    entries = {}            # (id(param), dtype, cin_padded, cout_padded) -> dict
    table = None            # (signature, device descriptor tensor, n, total tiles, dtype)
    # Descriptor tables a CAPTURED step launched with.  The graph keeps the table's device ADDRESS; the table itself was built in the
    # eager warm-up (a host -> device copy cannot be captured), i.e. in the ordinary allocator pool -- were it dropped when a later
    # eager step changes the set of live weights (an evaluation between epochs, an EMA model, one eager iteration between replays),
    # the allocator would hand its memory to the next tensor and the replayed launch would read descriptors out of that: wild
    # writes, a GPU memory fault (r05, found by tests/test_gpu_train_loop.py::test_detr_batch_beyond_max_annots_...).  Never freed;
    # one small tensor per capture.
    pinned_tables = []

    @classmethod
    def get(cls, weight, dtype, cin_padded, cout_padded, need_wd, rows=None):
        """rows = (parameter, r0, r1): `weight` is the row block parameter[r0:r1] of a packed parameter (nn.MultiheadAttention's
        in_proj_weight through ops_tfm.linear_rows) -- an entry of its own, refreshed by the same batched launch from the
        parameter's storage (r06: DETR spent 83 single-matrix pack launches per step on them)."""
        import weakref
        base = weight if rows is None else rows[0]
        k = (id(base), None if rows is None else (rows[1], rows[2]), dtype, cin_padded, cout_padded)
        e = cls.entries.get(k)
        if e is not None and e['ref']() is not base:
            e = None                                            # id() reused by another tensor
        if e is None:
            w = weight.detach()
            if w.dim() == 2:
                o, i, r, s = w.shape[0], w.shape[1], 1, 1
            else:
                o, i, r, s = w.shape
            alloc = torch.empty if cout_padded == o else torch.zeros
            e = {'ref': weakref.ref(base), 'rows': None if rows is None else (rows[1], rows[2]), 'dtype': dtype, 'ip': cin_padded, 'op': cout_padded, 'dims': (o, i, r, s),
                 'wf': alloc((cout_padded, r, s, cin_padded), dtype=dtype, device=w.device), 'wd': None, 'key': None, 'used': 0}
            cls.entries[k] = e
            cls.table = None
        if need_wd and e['wd'] is None:
            o, i, r, s = e['dims']
            alloc = torch.empty if cout_padded == o else torch.zeros
            e['wd'] = alloc((i, r, s, cout_padded), dtype=dtype, device=weight.device)
            e['key'] = None                                     # the new matrix has not been filled yet
            cls.table = None
        e['used'] = _weights_epoch[0]
        return e

    @classmethod
    def refresh(cls):
        """One launch over every live entry; stamps each with the key its weight has NOW."""
        live = [(k, e, e['ref']()) for k, e in cls.entries.items()]
        for k, e, w in live:
            if w is None:
                del cls.entries[k]
                cls.table = None
        # weights nobody asked for since the epoch before last (another model of the process) wait until they are wanted
        live = [(k, e, w) for k, e, w in live if w is not None and w.is_cuda and e['used'] >= _weights_epoch[0] - 1]
        # (an entry of a row block packs the view parameter[r0:r1]; its key is stamped with the parameter's version)
        views = {id(e): (w.detach()[e['rows'][0]:e['rows'][1]] if e.get('rows') else w.detach()) for _, e, w in live}
        if not live:
            return
        by_dtype = {}
        for k, e, w in live:
            by_dtype.setdefault(e['dtype'], []).append((e, w))
        sig = tuple((id(e), views[id(e)].data_ptr(), views[id(e)].stride(), ptr(e['wf']), ptr(e['wd'])) for _, e, w in live)
        if cls.table is None or cls.table[0] != sig:
            tables = []
            for dt, items in by_dtype.items():
                arr = (_lib.PackDesc * len(items))()
                t0 = 0
                for j, (e, w) in enumerate(items):
                    o, i, r, s = e['dims']
                    wv = views[id(e)]
                    if wv.dim() == 2:
                        so, si = wv.stride()
                        sr = ss = 0
                    else:
                        so, si, sr, ss = wv.stride()
                    d = arr[j]
                    d.w, d.sO, d.sI, d.sR, d.sS = wv.data_ptr(), so, si, sr, ss
                    d.O, d.I, d.R, d.S, d.Ip, d.Op = o, i, r, s, e['ip'], e['op']
                    d.wf, d.wd = ptr(e['wf']), ptr(e['wd'])
                    # 64 x 64 tiles with 16-byte reads / 8-byte writes where the input-channel axis is contiguous and aligned
                    vec = (si == 1 and all(v % 4 == 0 for v in (so, sr, ss, i, e['ip'], e['op'])) and wv.data_ptr() % 16 == 0)
                    ts = 64 if vec else 32
                    d.tile = ts
                    d.tiles_i, d.tiles_o = (e['ip'] + ts - 1) // ts, (e['op'] + ts - 1) // ts
                    d.tile_begin = t0
                    t0 += d.tiles_i * d.tiles_o * r * s
                dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(items[0][1].device)
                tables.append((dt, dev, len(items), t0))
            cls.table = (sig, tables)
        if live[0][2].is_cuda and torch.cuda.is_current_stream_capturing() and not any(t is cls.table for t in cls.pinned_tables):
            cls.pinned_tables.append(cls.table)
        for dt, dev, n, tiles in cls.table[1]:
            check(lib().saicv_pack_weight_batched(dtype_code(dt), ptr(dev), n, tiles, stream()), 'pack_weight_batched')
        for _, e, w in live:
            e['key'] = (w._version, _weights_epoch[0], views[id(e)].data_ptr())

    @classmethod
    def used_now(cls):
        """The live entries (asked for since the weight change before last -- the rule of refresh()): what a step that has just been
        captured depends on (its own optimizer step has already bumped the epoch once)."""
        return [e for e in cls.entries.values() if e['used'] >= _weights_epoch[0] - 1]

    @classmethod
    def touch(cls, entries):
        """A replayed step used these entries without running get(): keep them among the live ones, so that an eager step between
        replays refreshes the same set through the same descriptor table (engine.StepGraph calls this after every replay)."""
        for e in entries:
            e['used'] = _weights_epoch[0]


def packed_weight(weight, dtype, cin_padded, need_wd, cout_padded=None):
    """Compute-dtype copies of a conv / linear master weight, cached until the weight changes.

    Returns (wf [Op][R][S][Ip], wd [I][R][S][Op] or None); rows/cols beyond O are zero."""
    cout_padded = cout_padded or weight.shape[0]
    rows = getattr(weight, '_saicv_rows_of', None)          # (parameter, r0, r1): ops_tfm.LinearRowsFn
    if _PackRegistry.BATCH and weight.is_cuda and (isinstance(weight, torch.nn.Parameter) or rows is not None):
        e = _PackRegistry.get(weight, dtype, cin_padded, cout_padded, need_wd, rows)
        if e['key'] != ((weight if rows is None else rows[0])._version, _weights_epoch[0], weight.data_ptr()):
            _PackRegistry.refresh()
        return e['wf'], (e['wd'] if need_wd else None)
    key = (weight._version, _weights_epoch[0], dtype, cin_padded, cout_padded, weight.data_ptr())
    cache = getattr(weight, '_saicv_pack', None)
    if cache is not None and cache[0] == key and (cache[2] is not None or not need_wd):
        return cache[1], cache[2]
    w = weight.detach()
    if w.dim() == 2:
        o, i = w.shape
        r = s = 1
        so, si = w.stride()
        sr = ss = 0
    else:
        o, i, r, s = w.shape
        so, si, sr, ss = w.stride()
    op = cout_padded
    alloc = torch.empty if op == o else torch.zeros
    wf = alloc((op, r, s, cin_padded), dtype=dtype, device=w.device)
    wd = alloc((i, r, s, op), dtype=dtype, device=w.device) if need_wd else None
    check(lib().saicv_pack_weight(dtype_code(dtype), ptr(w), so, si, sr, ss, o, i, r, s, cin_padded, op,
                                  ptr(wf), ptr(wd), stream()), 'pack_weight')
    weight._saicv_pack = (key, wf, wd)
    return wf, wd


def _weight_grad(dw, weight, cin_padded):
    """fp32 dW[O][R][S][Ip] -> gradient laid out like `weight`."""
    if weight.dim() == 2:
        return dw.view(weight.shape[0], cin_padded)[:, :weight.shape[1]] if cin_padded != weight.shape[1] else dw.view(weight.shape)
    o, i, r, s = weight.shape
    if cin_padded == i and weight.is_contiguous(memory_format=torch.channels_last):
        return dw.permute(0, 3, 1, 2)
    g = torch.empty_strided(weight.shape, weight.stride(), dtype=torch.float32, device=weight.device)
    so, si, sr, ss = g.stride()
    check(lib().saicv_unpack_wgrad(ptr(dw), o, i, r, s, cin_padded, ptr(g), so, si, sr, ss, 0, stream()),
          'unpack_wgrad')
    return g


def _weight_grad_s2d(dw, weight, cq, arena_grad):
    """fp32 dW'[O][(R+1)/2][(S+1)/2][cq] of the space-to-depth stem -> the [O, I, R, S] gradient: added straight into the
    arena gradient when there is one (returns None), else a tensor laid out like `weight`."""
    o, i, r, s = weight.shape
    g = arena_grad if arena_grad is not None else torch.empty_strided(weight.shape, weight.stride(), dtype=torch.float32,
                                                                      device=weight.device)
    so, si, sr, ss = g.stride()
    check(lib().saicv_unpack_wgrad_s2d(ptr(dw), o, i, r, s, cq, ptr(g), so, si, sr, ss, int(arena_grad is not None), stream()),
          'unpack_wgrad_s2d')
    return None if arena_grad is not None else g


# ------------------------------------------------------------------------------ conv + BN + act
class ConvBnActFn(torch.autograd.Function):
    """conv -> [BatchNorm2d (train: batch stats, eval: running stats)] -> [+residual] -> [ReLU].

    Mirrors reference ConvBnActBlock (classification/backbones/resnet.py:19-48) plus the
    residual tail of BasicBlock / Bottleneck (:94-95, :152-153)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, bn, stride, pad, relu, want_skip=False, pool=None, defer=False):
        """defer: the node is the convolution + BatchNorm shortcut of a residual block; it returns the RAW convolution output
        tagged with its BatchNorm coefficients (conv_bn_act hangs `_saicv_deferred` on it) and the one consumer -- the join node
        that takes it as `residual` -- applies them in its own pass (csrc/bn.hip bn_act_fwd_join).  The gradient that comes back
        is the one of the normalised shortcut, so the backward below is the ordinary one.
        pool = (kernel, stride, padding): the MaxPool2d behind the block runs in the same pass as BatchNorm-apply + ReLU
        (the ResNet stem; csrc/pool.hip bn_relu_maxpool_*), the full-resolution activation is never written.
        want_skip: also return the (NHWC) input as a second output.  A residual block routes its shortcut
        through that alias, so the shortcut's gradient reaches THIS node's backward and is added in the
        dgrad kernel's epilogue instead of by a separate elementwise add."""
        require_gpu(x, weight)
        in_link = getattr(x, '_saicv_bn', None) if BN_FUSE else None
        xin = x
        res_gate_ok = bool(residual is not None and getattr(residual, '_saicv_gate_ok', False))
        res_affine = getattr(residual, '_saicv_deferred', None) if residual is not None else None
        if defer and (residual is not None or relu or want_skip or pool is not None):
            raise ValueError('a deferred BatchNorm-apply belongs to a plain convolution + BatchNorm shortcut')
        x = _nhwc(x)
        dt = x.dtype
        n, c, h, w = x.shape
        k, ci, r, s = weight.shape
        if c < ci:
            raise ValueError(f'input has {c} channels, weight expects {ci}')
        need_dx = ctx.needs_input_grad[0]
        # this conv's data gradient IS the dz of the BatchNorm(+ReLU) node that produced x (when x has no other consumer):
        # it can leave that node's backward partial sums behind
        ctx.in_link = in_link if (in_link is not None and x is xin and need_dx and c == ci and in_link.y.shape == x.shape
                                  and in_link.y.dtype == dt) else None
        s2d = getattr(xin, '_saicv_s2d', None)
        if s2d is not None:
            # stride-2 stem on the space-to-depth image (pack_stem_input): a stride-1 convolution with (R+1)/2 taps
            if need_dx or (s2d[0], s2d[3], s2d[4]) != (ci, r, pad) or stride != 2 or r != s:
                raise ValueError('space-to-depth stem input does not match this convolution')
            wf, wd = _packed_weight_s2d(weight, dt, c), None
            d = _desc(n, h, w, c, k, (r + 1) // 2, (s + 1) // 2, 1, 0, dt)
        else:
            wf, wd = packed_weight(weight, dt, c, need_dx and c == ci)
            d = _desc(n, h, w, c, k, r, s, stride, pad, dt)
        ctx.s2d = s2d
        L = lib()
        st = stream()
        dev = x.device
        y = _empty_nhwc(n, k, d.OH, d.OW, dt, dev)
        M = n * d.OH * d.OW
        training = bn.training
        scale = torch.empty(k, dtype=torch.float32, device=dev)
        shift = torch.empty(k, dtype=torch.float32, device=dev)
        mean = invstd = None
        if pool is not None and (residual is not None or not relu or want_skip):
            raise ValueError('the fused max-pool follows a plain conv -> BatchNorm -> ReLU block')
        # (the pooled form takes scale / shift from the finalize kernel: one 6 us launch, stem only)
        inline = training and BN_INLINE and k <= 2048 and pool is None
        atomic_rows = inline
        if defer:
            inline = False                   # scale / shift must exist as tensors: the finalize launch stays (over the few rows)
        if training:
            rows = L.saicv_conv2d_stat_rows(ctypes.byref(d))
            t0 = KernelTimer.begin('igemm_nt')
            if atomic_rows:
                rows = _stat_rows(rows)
                stats = _ZeroPool.take(2 * rows * k, dev).view(2, rows, k)
                check(L.saicv_conv2d_fwd_stats(ctypes.byref(d), ptr(x), ptr(wf), ptr(y), ptr(stats[0]), ptr(stats[1]), rows, st),
                      'conv2d_fwd_stats')
            else:
                stats = torch.empty((2, rows, k), dtype=torch.float32, device=dev)
                check(L.saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), 0, ptr(y), 0, ptr(stats[0]),
                                         ptr(stats[1]), st), 'conv2d_fwd')
            es = x.element_size()
            xin_px = M if (r == 1 and stride > 1) else n * h * w          # a strided 1x1 reads a quarter of its input
            KernelTimer.end(t0, 'igemm_nt', 2.0 * M * k * r * s * min(c, ci),
                            float(xin_px) * c * es + float(k) * r * s * c * es + float(M) * k * es)
            mean = torch.empty(k, dtype=torch.float32, device=dev)
            invstd = torch.empty(k, dtype=torch.float32, device=dev)
            if bn.momentum is None:
                raise NotImplementedError('BatchNorm2d(momentum=None) is not supported')
            track = bn.track_running_stats and bn.running_mean is not None
            nbt = bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None
            if not inline:
                ws = torch.empty(L.saicv_bn_ws_floats(k), dtype=torch.float32, device=dev)
                check(L.saicv_bn_finalize_fwd(ptr(stats[0]), ptr(stats[1]), rows, k, float(M), ptr(gamma),
                                              ptr(beta), ptr(bn.running_mean) if track else 0,
                                              ptr(bn.running_var) if track else 0, float(bn.momentum),
                                              float(bn.eps), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                              ptr(ws), ptr(nbt), st), 'bn_finalize_fwd')      # also num_batches_tracked += 1
        else:
            t0 = KernelTimer.begin('igemm_nt')
            check(L.saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), 0, ptr(y), 0, 0, 0, st), 'conv2d_fwd')
            KernelTimer.end(t0, 'igemm_nt', 2.0 * M * k * r * s * min(c, ci), 0)
            check(L.saicv_bn_eval_coeffs(k, ptr(gamma), ptr(beta), ptr(bn.running_mean),
                                         ptr(bn.running_var), float(bn.eps), ptr(scale), ptr(shift), st),
                  'bn_eval_coeffs')
        if pool is not None:
            pk, ps, pp = pool
            poh, pow_ = (d.OH + 2 * pp - pk) // ps + 1, (d.OW + 2 * pp - pk) // ps + 1
            zp = _empty_nhwc(n, k, poh, pow_, dt, dev)
            idx = torch.empty((n, poh, pow_, k), dtype=torch.uint8, device=dev)
            t0 = KernelTimer.begin('bn_act_fwd')
            check(L.saicv_bn_relu_maxpool_fwd(dtype_code(dt), ptr(y), ptr(scale), ptr(shift), ptr(zp), ptr(idx), n, d.OH, d.OW, k,
                                              poh, pow_, pk, ps, pp, st), 'bn_relu_maxpool_fwd')
            es = y.element_size()
            KernelTimer.end(t0, 'bn_act_fwd', 0, float(M) * k * es + float(n) * poh * pow_ * k * (es + 1))
            if training:
                ctx.save_for_backward(x, weight, gamma, y, idx, mean, invstd)
            else:
                ctx.save_for_backward(x, weight, gamma, y, None, None, scale)
            ctx.cfg = (stride, pad, True, False, training, d, wd)
            ctx.pool = (pk, ps, pp, poh, pow_, scale, shift)
            ctx.beta_ref = beta
            ctx.gated_res = False
            ctx.link = None
            ctx.applies_gate = False
            return zp
        ctx.pool = None
        if defer:
            ConvBnActFn._deferred = (scale, shift)
            if training:
                ctx.save_for_backward(x, weight, gamma, y, None, mean, invstd)
            else:
                ctx.save_for_backward(x, weight, gamma, y, None, None, scale)
            ctx.cfg = (stride, pad, False, False, training, d, wd)
            ctx.beta_ref = beta
            ctx.gated_res = False
            ctx.link = None
            ctx.applies_gate = bool(BN_FUSE and training)
            return y
        if residual is not None:
            res_in = residual
            residual = _nhwc(residual)
            if residual.dtype != dt:
                residual = residual.to(dt)
            if res_affine is not None and (residual is not res_in or residual.shape != y.shape):
                # not the tensor the coefficients were made for (a layout / dtype change in between): apply them here
                residual = (residual.float() * res_affine[0].view(1, -1, 1, 1) + res_affine[1].view(1, -1, 1, 1)).to(dt)
                residual = _nhwc(residual)
                res_affine = None
        z = _empty_nhwc(n, k, d.OH, d.OW, dt, dev)
        # backward needs only the sign of z: one byte per 16-byte chunk instead of re-reading z twice
        mask = (torch.empty(M * k // _lib.epc(dt), dtype=torch.uint8, device=dev)
                if (relu and training and any(ctx.needs_input_grad)) else None)
        t0 = KernelTimer.begin('bn_act_fwd')
        if res_affine is not None:
            # the shortcut arrives as a raw convolution output + its BatchNorm coefficients: applied on the fly
            check(L.saicv_bn_act_fwd_join(dtype_code(dt), ptr(y), ptr(residual), ptr(res_affine[0]), ptr(res_affine[1]), ptr(z),
                                          0 if inline else ptr(scale), 0 if inline else ptr(shift),
                                          ptr(stats[0]) if inline else 0, ptr(stats[1]) if inline else 0, rows if inline else 0,
                                          float(M), ptr(gamma), ptr(beta), ptr(bn.running_mean) if (inline and track) else 0,
                                          ptr(bn.running_var) if (inline and track) else 0,
                                          float(bn.momentum) if inline else 0.0, float(bn.eps), ptr(nbt) if inline else 0,
                                          ptr(mean) if inline else 0, ptr(invstd) if inline else 0, M, k, int(relu), ptr(mask), st),
                  'bn_act_fwd_join')
        elif inline:
            # the kernel derives mean / invstd / scale / shift from the few statistics rows itself (and updates the running
            # statistics and num_batches_tracked): no finalize launch between the convolution and this one
            check(L.saicv_bn_act_fwd_stats(dtype_code(dt), ptr(y), ptr(residual), ptr(z), ptr(stats[0]), ptr(stats[1]), rows,
                                           float(M), ptr(gamma), ptr(beta), ptr(bn.running_mean) if track else 0,
                                           ptr(bn.running_var) if track else 0, float(bn.momentum), float(bn.eps), ptr(nbt),
                                           ptr(mean), ptr(invstd), M, k, int(relu), ptr(mask), st), 'bn_act_fwd_stats')
        else:
            check(L.saicv_bn_act_fwd(dtype_code(dt), ptr(y), ptr(residual), ptr(z), ptr(scale), ptr(shift), M,
                                     k, int(relu), ptr(mask), st), 'bn_act_fwd')
        KernelTimer.end(t0, 'bn_act_fwd', 0, float(M) * k * y.element_size() * (3 if residual is not None else 2))
        if training:
            ctx.save_for_backward(x, weight, gamma, y, mask, mean, invstd)
        else:
            # eval-mode backward (frozen statistics) is linear: dy = scale * g
            ctx.save_for_backward(x, weight, gamma, y, None, None, scale)
        ctx.cfg = (stride, pad, bool(relu), residual is not None, training, d, wd)
        ctx.beta_ref = beta
        # the shortcut gradient may come back as (gradient, gate) only from nodes that apply gates: the alias below
        # (its gradient joins in this node's dgrad epilogue) and BatchNorm nodes without a ReLU of their own
        ctx.gated_res = bool(BN_FUSE and res_gate_ok and mask is not None and ctx.needs_input_grad[4]
                             and residual.shape == z.shape)
        # conv_bn_act() below hangs these on the OUTPUT tensors (the objects autograd hands back, not the ones made here)
        ctx.link = _BnLink(y, mask, mean, invstd) if (BN_FUSE and mask is not None) else None
        ctx.applies_gate = bool(BN_FUSE and training and not relu)
        if want_skip:
            return z, x
        return z

    @staticmethod
    def backward(ctx, dz, dskip=None):
        x, weight, gamma, y, mask, mean, invstd = ctx.saved_tensors
        stride, pad, relu, has_res, training, d, wd = ctx.cfg
        if not training:
            raise NotImplementedError('backward through eval-mode BatchNorm is not implemented')
        L = lib()
        st = stream()
        dt = y.dtype
        dev = y.device
        gate_in = _take_gate(dz)          # dz is a shortcut gradient still waiting for the ReLU mask of the block's tail
        dz0 = dz
        dz = _nhwc(dz)
        if dz.dtype != dt:
            dz = dz.to(dt)
        n, k, oh, ow = y.shape
        M = n * oh * ow
        if ctx.pool is not None:
            return ConvBnActFn._backward_pooled(ctx, dz, x, weight, gamma, y, mask, mean, invstd)
        if gate_in is not None:
            if relu:
                raise RuntimeError('a gated shortcut gradient reached a BatchNorm node with its own ReLU')
            relu, mask = True, gate_in      # same [M][C] coordinates: the tail's mask gates this node's dz
        dy = _empty_nhwc(n, k, oh, ow, dt, dev)
        dres = None
        if has_res and ctx.needs_input_grad[4]:
            if ctx.gated_res and dz is dz0:
                # the masked copy g = dz * [z > 0] is not written: the consumer gets dz and the mask
                dres = dz
                dres._saicv_gate = mask
                dres._saicv_gate_version = dres._version
                _GateLedger.hand_out()
            else:
                dres = _empty_nhwc(n, k, oh, ow, dt, dev)
        dres_out = dres if (dres is not None and dres is not dz) else None
        beta = ctx.beta_ref
        gg, gb = _arena_grad(gamma), _arena_grad(beta)
        direct_bn = gg is not None and gb is not None
        if direct_bn:
            dgamma, dbeta = gg, gb
        else:
            dgamma = torch.empty(k, dtype=torch.float32, device=dev)
            dbeta = torch.empty(k, dtype=torch.float32, device=dev)
        ws = torch.empty(L.saicv_bn_bwd_ws_floats(M, k, dtype_code(dt)), dtype=torch.float32, device=dev)
        link = ctx.link
        # dz IS the tensor that data gradient wrote (same memory, never written since): with another consumer of z autograd
        # hands over a sum in a different tensor and the three-pass form runs
        fused_reduce = (link is not None and link.part is not None and link.dx is not None and gate_in is None
                        and dz.data_ptr() == link.dx.data_ptr() and dz.shape == link.dx.shape
                        and dz._version == link.dx_version)
        t0 = KernelTimer.begin('bn_act_bwd')
        if fused_reduce and link.inline:
            # ... as a few atomically accumulated rows: coefficients, dgamma and dbeta come out of the one streaming kernel
            check(L.saicv_bn_act_bwd_inline(dtype_code(dt), ptr(dz), ptr(mask), ptr(y), ptr(gamma), ptr(mean), ptr(invstd),
                                            ptr(link.part[0]), ptr(link.part[1]), link.rows, ptr(dy), ptr(dres_out), ptr(dgamma),
                                            ptr(dbeta), M, k, int(relu), int(direct_bn), st), 'bn_act_bwd_inline')
        elif fused_reduce:
            # the data gradient that wrote dz also left the partial sums of this reduction (no pass over dz and y here)
            check(L.saicv_bn_act_bwd_from_partials(dtype_code(dt), ptr(dz), ptr(mask), ptr(y), ptr(gamma), ptr(mean),
                                                   ptr(invstd), ptr(link.part[0]), ptr(link.part[1]), link.rows, ptr(dy),
                                                   ptr(dres_out), ptr(dgamma), ptr(dbeta), M, k, int(relu), int(direct_bn),
                                                   ptr(ws), st), 'bn_act_bwd_from_partials')
        else:
            check(L.saicv_bn_act_bwd(dtype_code(dt), ptr(dz), 0, ptr(mask), ptr(y), ptr(gamma), ptr(mean), ptr(invstd),
                                     ptr(dy), ptr(dres_out), ptr(dgamma), ptr(dbeta), M, k, int(relu), int(direct_bn),
                                     ptr(ws), st), 'bn_act_bwd')
        if link is not None:
            link.part = link.dx = None
        if direct_bn:
            dgamma = dbeta = None
        # streaming passes over (dz, y) (+ the 1-bit ReLU mask): reduction unless fused away, then apply; dy (and dres) written
        KernelTimer.end(t0, 'bn_act_bwd', 0, float(M) * k * y.element_size() *
                        ((1 if fused_reduce else 2) * (2 + (1.0 / 16 if relu else 0)) + (2 if dres_out is not None else 1)))
        c = x.shape[1]
        flops = 2.0 * M * k * weight.shape[2] * weight.shape[3] * min(c, weight.shape[1])
        dx = None
        if ctx.needs_input_grad[0]:
            if wd is None:
                _, wd = packed_weight(weight, dt, c, True)
            dx = _empty_nhwc(n, c, x.shape[2], x.shape[3], dt, dev)
            t0 = KernelTimer.begin('igemm_nt')
            gate = None
            if dskip is not None:           # gradient of the shortcut alias joins in the dgrad epilogue
                gate = _take_gate(dskip)
                dskip = _nhwc(dskip)
                if dskip.dtype != dt:
                    dskip = dskip.to(dt)
            in_link = ctx.in_link
            if gate is not None or in_link is not None:
                fuse = _lib.DgradFuse()
                fuse.addend, fuse.addend_gate = ptr(dskip), ptr(gate)
                if in_link is not None:
                    rows = L.saicv_conv2d_dgrad_stat_rows(ctypes.byref(d))
                    in_link.inline = BN_INLINE and c <= 2048
                    if in_link.inline:
                        rows = _stat_rows(rows)
                        part = _ZeroPool.take(2 * rows * c, dev).view(2, rows, c)
                        fuse.part_rows = rows
                    else:
                        part = torch.empty((2, rows, c), dtype=torch.float32, device=dev)
                    fuse.bn_y, fuse.bn_mask = ptr(in_link.y), ptr(in_link.mask)
                    fuse.bn_mean, fuse.bn_invstd = ptr(in_link.mean), ptr(in_link.invstd)
                    fuse.part_g, fuse.part_gx = ptr(part[0]), ptr(part[1])
                    in_link.part, in_link.rows, in_link.dx, in_link.dx_version = part, rows, dx, dx._version
                check(L.saicv_conv2d_dgrad_fused(ctypes.byref(d), ptr(dy), ptr(wd), ctypes.byref(fuse), ptr(dx), st),
                      'conv2d_dgrad_fused')
            elif dskip is not None:
                check(L.saicv_conv2d_dgrad_add(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dskip), ptr(dx), st),
                      'conv2d_dgrad_add')
            else:
                check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st), 'conv2d_dgrad')
            es = dy.element_size()
            px = float(n) * x.shape[2] * x.shape[3] * c          # elements of dx (and of every epilogue tensor)
            nbytes = float(M) * k * es + float(k) * d.R * d.S * c * es + px * es
            if dskip is not None:
                nbytes += px * es + (px / 8 if gate is not None else 0)
            if in_link is not None:
                nbytes += px * es + px / 8
            KernelTimer.end(t0, 'igemm_nt', flops, nbytes)
        dwt = None
        if ctx.needs_input_grad[1]:
            gw = _arena_grad(weight)
            direct = (gw is not None and c == weight.shape[1] and
                      weight.is_contiguous(memory_format=torch.channels_last))
            # KRSC fp32 gradient: straight into the arena (atomics accumulate), else a temporary
            dw = gw if direct else torch.zeros((k, d.R, d.S, c), dtype=torch.float32, device=dev)
            if direct and WGRAD_SIDE_STREAM:
                with _SideStream(dy, x):
                    t0 = KernelTimer.begin('igemm_tn')
                    check(L.saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), stream()), 'conv2d_wgrad')
                    KernelTimer.end(t0, 'igemm_tn', flops, 0)
            else:
                t0 = KernelTimer.begin('igemm_tn')
                check(L.saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), st), 'conv2d_wgrad')
                KernelTimer.end(t0, 'igemm_tn', flops, 0)
            if not direct:
                dwt = _weight_grad_s2d(dw, weight, c, gw) if ctx.s2d is not None else _weight_grad(dw, weight, c)
        return (dx, dwt, dgamma if ctx.needs_input_grad[2] else None,
                dbeta if ctx.needs_input_grad[3] else None, dres, None, None, None, None, None, None, None)


def _conv_bn_act_backward_pooled(ctx, dz, x, weight, gamma, y, idx, mean, invstd):
    """backward of the pooled stem block: pooled gradient -> (max-pool backward + ReLU gate + BatchNorm backward in two passes over
    y) -> dy at full resolution -> [data gradient] + weight gradient, as ConvBnActFn.backward does behind its BatchNorm kernels."""
    stride, pad, _, _, _, d, wd = ctx.cfg
    pk, ps, pp, poh, pow_, scale, shift = ctx.pool
    L, st = lib(), stream()
    dt, dev = y.dtype, y.device
    n, k, oh, ow = y.shape
    M = n * oh * ow
    dy = _empty_nhwc(n, k, oh, ow, dt, dev)
    beta = ctx.beta_ref
    gg, gb = _arena_grad(gamma), _arena_grad(beta)
    direct_bn = gg is not None and gb is not None
    dgamma = gg if direct_bn else torch.empty(k, dtype=torch.float32, device=dev)
    dbeta = gb if direct_bn else torch.empty(k, dtype=torch.float32, device=dev)
    ws = torch.empty(L.saicv_bn_relu_maxpool_bwd_ws_floats(k), dtype=torch.float32, device=dev)
    t0 = KernelTimer.begin('bn_act_bwd')
    check(L.saicv_bn_relu_maxpool_bwd(dtype_code(dt), ptr(dz), ptr(idx), ptr(y), ptr(gamma), ptr(mean), ptr(invstd), ptr(scale),
                                      ptr(shift), ptr(dy), ptr(dgamma), ptr(dbeta), int(direct_bn), ptr(ws), n, oh, ow, k, poh, pow_,
                                      pk, ps, pp, st), 'bn_relu_maxpool_bwd')
    es = y.element_size()
    KernelTimer.end(t0, 'bn_act_bwd', 0, 3.0 * M * k * es + 2.0 * n * poh * pow_ * k * (es + 1))       # y twice + dy; dout + idx twice
    if direct_bn:
        dgamma = dbeta = None
    c = x.shape[1]
    flops = 2.0 * M * k * weight.shape[2] * weight.shape[3] * min(c, weight.shape[1])
    dx = None
    if ctx.needs_input_grad[0]:
        if wd is None:
            _, wd = packed_weight(weight, dt, c, True)
        dx = _empty_nhwc(n, c, x.shape[2], x.shape[3], dt, dev)
        t0 = KernelTimer.begin('igemm_nt')
        check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st), 'conv2d_dgrad')
        KernelTimer.end(t0, 'igemm_nt', flops, 0)
    dwt = None
    if ctx.needs_input_grad[1]:
        gw = _arena_grad(weight)
        direct = gw is not None and c == weight.shape[1] and weight.is_contiguous(memory_format=torch.channels_last)
        dw = gw if direct else torch.zeros((k, d.R, d.S, c), dtype=torch.float32, device=dev)
        t0 = KernelTimer.begin('igemm_tn')
        check(L.saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), st), 'conv2d_wgrad')
        KernelTimer.end(t0, 'igemm_tn', flops, 0)
        if not direct:
            dwt = _weight_grad_s2d(dw, weight, c, gw) if ctx.s2d is not None else _weight_grad(dw, weight, c)
    return (dx, dwt, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None,
            None, None, None, None, None, None, None, None)


ConvBnActFn._backward_pooled = staticmethod(_conv_bn_act_backward_pooled)


def conv_bn_act(x, weight, bn, stride, pad, relu, residual=None, want_skip=False, pool=None, defer=False):
    """defer=True: ONLY for a tensor whose single consumer is the `residual` argument of another conv_bn_act call (what comes
    back is the raw convolution output; its values are not the block's output until that consumer applies the coefficients)."""
    # as conv2d below: under autocast the block's convolution computes in the autocast dtype whatever its input's dtype
    if torch.is_autocast_enabled('cuda') and x.is_floating_point() and x.dtype != compute_dtype() and getattr(x, '_saicv_s2d', None) is None:
        x = x.to(compute_dtype())
    if pool is not None:
        return ConvBnActFn.apply(x, weight, bn.weight, bn.bias, None, bn, stride, pad, relu, False, pool)
    if defer:
        ConvBnActFn._deferred = None
        out = ConvBnActFn.apply(x, weight, bn.weight, bn.bias, None, bn, stride, pad, False, False, None, True)
        out._saicv_deferred, ConvBnActFn._deferred = ConvBnActFn._deferred, None
        if BN_FUSE and out.grad_fn is not None and getattr(out.grad_fn, 'applies_gate', False):
            out._saicv_gate_ok = True
        return out
    out = ConvBnActFn.apply(x, weight, bn.weight, bn.bias, residual, bn, stride, pad, relu, want_skip)
    z = out[0] if want_skip else out
    node = z.grad_fn
    if BN_FUSE and node is not None and hasattr(node, 'link'):
        if node.link is not None:
            z._saicv_bn = node.link            # the next conv's data gradient can do this node's backward reduction
        if node.applies_gate:
            z._saicv_gate_ok = True            # as a residual, its gradient may arrive as (dz, ReLU mask)
        if want_skip:
            out[1]._saicv_gate_ok = True       # the alias: its gradient joins in this node's dgrad epilogue
    return out


# ------------------------------------------------------------------------------ plain conv / linear
class ConvFn(torch.autograd.Function):
    """nn.Conv2d (optional bias, no normalisation) on NHWC data: SAM neck convs (reference
    interactive_segmentation/models/segment_anything/image_encoder.py:303-316), DETR input
    projection (reference detection/models/detr.py:301)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        require_gpu(x, weight)
        x = _nhwc(x)
        dt = x.dtype
        n, c, h, w = x.shape
        k, ci, r, s = weight.shape
        if c != ci and not (c > ci and c - ci < 8 and not ctx.needs_input_grad[0]):
            # c > ci: an image batch zero-padded to whole chunks by pack_input() (stems; no input gradient)
            raise ValueError(f'input has {c} channels, weight expects {ci}')
        need_dx = ctx.needs_input_grad[0]
        wf, wd = packed_weight(weight, dt, c, need_dx)
        d = _desc(n, h, w, c, k, r, s, stride, pad, dt)
        y = _empty_nhwc(n, k, d.OH, d.OW, dt, x.device)
        t0 = KernelTimer.begin('igemm_nt')
        check(lib().saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), ptr(bias), ptr(y), 0, 0, 0, stream()),
              'conv2d_fwd')
        KernelTimer.end(t0, 'igemm_nt', 2.0 * n * d.OH * d.OW * k * r * s * c, 0)
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (d, wd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        d, wd = ctx.cfg
        L, st = lib(), stream()
        dt = x.dtype
        dy = _nhwc(dy)
        if dy.dtype != dt:
            dy = dy.to(dt)
        n, c, h, w = x.shape
        k = weight.shape[0]
        e = _lib.epc(dt)
        if k % e:
            # output channels that are not whole 16-byte chunks (RetinaNet's 9 x 4 box offsets, FCOS's 4 + 1 outputs): the
            # backward kernels gather dY by chunks, so dY travels zero-padded to kp channels and the gradients are sliced back
            kp = (k + e - 1) // e * e
            dyp = torch.zeros((n, kp, d.OH, d.OW), dtype=dt, device=dy.device).contiguous(memory_format=torch.channels_last)
            dyp[:, :k] = dy
            dp = _desc(n, h, w, c, kp, d.R, d.S, d.stride, d.pad, dt)
            dx = dwt = db = None
            if ctx.needs_input_grad[0]:
                _, wdp = packed_weight(weight, dt, c, True, kp)
                dx = _empty_nhwc(n, c, h, w, dt, x.device)
                check(L.saicv_conv2d_dgrad(ctypes.byref(dp), ptr(dyp), ptr(wdp), ptr(dx), st), 'conv2d_dgrad')
            if ctx.needs_input_grad[1]:
                dwp = torch.zeros((kp, d.R, d.S, c), dtype=torch.float32, device=x.device)
                check(L.saicv_conv2d_wgrad(ctypes.byref(dp), ptr(dyp), ptr(x), ptr(dwp), st), 'conv2d_wgrad')
                dwt = dwp[:k].permute(0, 3, 1, 2).to(weight.dtype)
            if bias is not None and ctx.needs_input_grad[2]:
                db = dy.float().sum((0, 2, 3)).to(bias.dtype)
            return dx, dwt, db, None, None
        M = n * d.OH * d.OW
        flops = 2.0 * M * k * d.R * d.S * c
        dx = dwt = db = None
        if ctx.needs_input_grad[0]:
            if wd is None:
                _, wd = packed_weight(weight, dt, c, True)
            dx = _empty_nhwc(n, c, h, w, dt, x.device)
            t0 = KernelTimer.begin('igemm_nt')
            check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st), 'conv2d_dgrad')
            KernelTimer.end(t0, 'igemm_nt', flops, 0)
        want_b = bias is not None and ctx.needs_input_grad[2]
        tb = gb = None
        bias_done = False
        if want_b:
            gb = _arena_grad(bias)
            tb = gb if gb is not None else torch.zeros(k, dtype=torch.float32, device=x.device)
        if ctx.needs_input_grad[1]:
            gw = _arena_grad(weight)
            direct = gw is not None and c == weight.shape[1] and weight.is_contiguous(memory_format=torch.channels_last)
            dw = gw if direct else torch.zeros((k, d.R, d.S, c), dtype=torch.float32, device=x.device)
            if direct and WGRAD_SIDE_STREAM:
                with _SideStream(dy, x):
                    t0 = KernelTimer.begin('igemm_tn')
                    check(L.saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), stream()), 'conv2d_wgrad')
                    KernelTimer.end(t0, 'igemm_tn', flops, 0)
            else:
                t0 = KernelTimer.begin('igemm_tn')
                # the bias gradient rides along: column sums of the dY tiles the weight-gradient kernel already holds
                check(L.saicv_conv2d_wgrad_bias(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), ptr(tb), st), 'conv2d_wgrad')
                KernelTimer.end(t0, 'igemm_tn', flops, 0)
                bias_done = want_b
            if not direct:
                dwt = _weight_grad(dw, weight, c)
        if want_b:
            if not bias_done:                   # no weight gradient wanted (or it ran on the side stream): its own pass
                check(L.saicv_colsum(dtype_code(dt), ptr(dy), M, k, ptr(tb), st), 'colsum')
            if gb is None:
                db = tb
        return dx, dwt, db, None, None


def conv2d(x, weight, bias=None, stride=1, pad=0):
    # under autocast a convolution computes in the autocast dtype whatever its input's dtype (torch.autocast casts conv2d's
    # operands): an fp32 activation -- e.g. the output of a bilinear resize, which autocast runs in fp32 -- is cast here
    if torch.is_autocast_enabled('cuda') and x.is_floating_point() and x.dtype != compute_dtype():
        x = x.to(compute_dtype())
    return ConvFn.apply(x, weight, bias, stride, pad)


class DepthwiseConvFn(torch.autograd.Function):
    """nn.Conv2d(C, C, k, stride, padding, dilation, groups=C) on NHWC data (csrc/dwconv.hip): the depthwise layers of
    reference classification/backbones/van.py:30,68,75 and convformer.py.  weight [C, 1, k, k]; HBM-bound streaming kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dilation):
        require_gpu(x, weight)
        x = _nhwc(x)
        dt = x.dtype
        n, c, h, w = x.shape
        if weight.shape[0] != c or weight.shape[1] != 1 or weight.shape[2] != weight.shape[3]:
            raise ValueError(f'depthwise weight {tuple(weight.shape)} for {c} channels')
        k = weight.shape[2]
        oh = (h + 2 * pad - dilation * (k - 1) - 1) // stride + 1
        ow = (w + 2 * pad - dilation * (k - 1) - 1) // stride + 1
        wt = weight.detach().reshape(c, k * k).t().contiguous().to(dt)        # tap-major [k*k][C]
        y = _empty_nhwc(n, c, oh, ow, dt, x.device)
        bf = bias.detach().float() if bias is not None else None
        t0 = KernelTimer.begin('dwconv_fwd')
        check(lib().saicv_dwconv2d_fwd(dtype_code(dt), ptr(x), ptr(wt), ptr(bf), ptr(y), n, h, w, c, oh, ow, k, stride, pad, dilation,
                                       stream()), 'dwconv2d_fwd')
        KernelTimer.end(t0, 'dwconv_fwd', 2.0 * n * oh * ow * c * k * k, float(n) * (h * w + oh * ow) * c * x.element_size())
        ctx.save_for_backward(x, weight, bias, wt)
        ctx.cfg = (n, h, w, c, oh, ow, k, stride, pad, dilation)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, wt = ctx.saved_tensors
        n, h, w, c, oh, ow, k, stride, pad, dilation = ctx.cfg
        L, st = lib(), stream()
        dt = x.dtype
        dy = _nhwc(dy)
        if dy.dtype != dt:
            dy = dy.to(dt)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _empty_nhwc(n, c, h, w, dt, x.device)
            check(L.saicv_dwconv2d_dgrad(dtype_code(dt), ptr(dy), ptr(wt), ptr(dx), n, h, w, c, oh, ow, k, stride, pad, dilation, st),
                  'dwconv2d_dgrad')
        want_b = bias is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            dwt = torch.zeros((k * k, c), dtype=torch.float32, device=x.device)
            tb = torch.zeros(c, dtype=torch.float32, device=x.device) if want_b else None
            check(L.saicv_dwconv2d_wgrad(dtype_code(dt), ptr(dy), ptr(x), ptr(dwt), ptr(tb), n, h, w, c, oh, ow, k, stride, pad, dilation,
                                         st), 'dwconv2d_wgrad')
            if ctx.needs_input_grad[1]:
                dw = dwt.t().reshape(c, 1, k, k).to(weight.dtype)
            db = tb.to(bias.dtype) if want_b else None
        return dx, dw, db, None, None, None


def depthwise_conv2d(x, weight, bias=None, stride=1, pad=0, dilation=1):
    if torch.is_autocast_enabled('cuda') and x.is_floating_point() and x.dtype != compute_dtype():
        x = x.to(compute_dtype())
    return DepthwiseConvFn.apply(x, weight, bias, stride, pad, dilation)


# ------------------------------------------------------------------------------ streaming glue (csrc/elemwise.hip)
ACT_KINDS = {'relu': 0, 'leakyrelu': 1, 'silu': 2}


def _dense(x):
    """x as a dense tensor whose memory order the elementwise kernels may walk: NHWC for 4-d tensors, row-major otherwise."""
    return _nhwc(x) if x.dim() == 4 else x.contiguous()


def _like(x, other):
    """`other` in x's dtype and dense layout"""
    other = _dense(other)
    return other if other.dtype == x.dtype else other.to(x.dtype)


class ActFn(torch.autograd.Function):
    """nn.ReLU / nn.LeakyReLU(slope) / nn.SiLU as one streaming pass (reference darknet.py:16-33, van.py:44,103,
    convformer.py:53,86).  The forward input is kept for the backward (dx = dy * act'(x))."""

    @staticmethod
    def forward(ctx, x, kind, slope):
        require_gpu(x)
        x = _dense(x)
        y = torch.empty_like(x)
        check(lib().saicv_act_fwd(dtype_code(x.dtype), kind, float(slope), ptr(x), ptr(y), x.numel(), stream()), 'act_fwd')
        ctx.save_for_backward(x)
        ctx.cfg = (kind, float(slope))
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        kind, slope = ctx.cfg
        dy = _like(x, dy)
        dx = torch.empty_like(x)
        check(lib().saicv_act_bwd(dtype_code(x.dtype), kind, slope, ptr(dy), ptr(x), ptr(dx), x.numel(), stream()), 'act_bwd')
        return dx, None, None


def act(x, kind, slope=0.):
    return ActFn.apply(x, ACT_KINDS[kind], slope)


class MulFn(torch.autograd.Function):
    """a * b of two activations of one shape (reference van.py:91)."""

    @staticmethod
    def forward(ctx, a, b):
        require_gpu(a, b)
        a = _dense(a)
        b = _like(a, b)
        out = torch.empty_like(a)
        check(lib().saicv_mul_fwd(dtype_code(a.dtype), ptr(a), ptr(b), ptr(out), a.numel(), stream()), 'mul_fwd')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = _like(a, dy)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        check(lib().saicv_mul_bwd(dtype_code(a.dtype), ptr(dy), ptr(a), ptr(b), ptr(da), ptr(db), a.numel(), stream()), 'mul_bwd')
        return da, db


def mul(a, b):
    return MulFn.apply(a, b)


class ScaleAddFn(torch.autograd.Function):
    """x + s[c] * y on NHWC activations, s a per-channel parameter ([C] after flattening; None: plain x + y): the layer-scaled
    residual of reference van.py:181-185 and the residual joins of darknet.py / convformer.py:157-163."""

    @staticmethod
    def forward(ctx, x, y, s):
        require_gpu(x, y)
        y = _nhwc(y)
        x = _like(y, x) if x is not None else None            # None: s[c] * y alone (a layer scale outside a residual join)
        n, c, h, w = y.shape
        sf = s.detach().reshape(-1).float().contiguous() if s is not None else None
        if sf is not None and sf.numel() != c:
            raise ValueError(f'scale of {sf.numel()} values for {c} channels')
        out = torch.empty_like(y)
        check(lib().saicv_channel_scale_add_fwd(dtype_code(y.dtype), ptr(x), ptr(y), ptr(sf), ptr(out), n * h * w, c, stream()),
              'channel_scale_add_fwd')
        ctx.save_for_backward(y if (s is not None and ctx.needs_input_grad[2]) else None, sf, s)
        ctx.cfg = (n, c, h, w, y.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, sf, s = ctx.saved_tensors
        n, c, h, w, dt = ctx.cfg
        dout = _nhwc(dout)
        if dout.dtype != dt:
            dout = dout.to(dt)
        dx = dout if ctx.needs_input_grad[0] else None        # (needs_input_grad[0] is False for x = None)
        if s is None:
            return dx, (dout if ctx.needs_input_grad[1] else None), None
        dy = torch.empty_like(dout) if ctx.needs_input_grad[1] else None
        ds = torch.zeros(c, dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[2] else None
        check(lib().saicv_channel_scale_add_bwd(dtype_code(dt), ptr(dout), ptr(y), ptr(sf), ptr(dy), ptr(ds), n * h * w, c, stream()),
              'channel_scale_add_bwd')
        return dx, dy, (ds.reshape(s.shape).to(s.dtype) if ds is not None else None)


def scale_add(x, y, s=None):
    return ScaleAddFn.apply(x, y, s)


class SampleScaleFn(torch.autograd.Function):
    """x[n] * w[n] on an NHWC activation, w fp32 [N]: stochastic depth of a convolutional residual branch (reference van.py:
    118-150, convformer.py:99-131 DropPathBlock).  The same kernel both ways (saicv_row_scale over [N*H*W][C] rows)."""

    @staticmethod
    def forward(ctx, x, w):
        require_gpu(x, w)
        x = _nhwc(x)
        n, c, h, wd = x.shape
        w = w.detach().reshape(-1).float().contiguous()
        out = torch.empty_like(x)
        check(lib().saicv_row_scale(dtype_code(x.dtype), ptr(x), ptr(w), ptr(out), n * h * wd, c, h * wd, stream()), 'row_scale')
        ctx.save_for_backward(w)
        return out

    @staticmethod
    def backward(ctx, dout):
        (w,) = ctx.saved_tensors
        dout = _nhwc(dout)
        n, c, h, wd = dout.shape
        dx = torch.empty_like(dout)
        check(lib().saicv_row_scale(dtype_code(dout.dtype), ptr(dout), ptr(w), ptr(dx), n * h * wd, c, h * wd, stream()), 'row_scale')
        return dx, None


def sample_scale(x, w):
    return SampleScaleFn.apply(x, w)


class BatchNorm2dFn(torch.autograd.Function):
    """nn.BatchNorm2d on an activation that is not a convolution output (reference van.py:176,178,260; convformer.py:34-35,
    143,149): statistics pass -> the finalize kernel of the fused blocks (running statistics, num_batches_tracked) -> apply;
    backward = the fused blocks' BatchNorm backward without a ReLU gate."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn):
        require_gpu(x, gamma)
        x = _nhwc(x)
        dt = x.dtype
        n, c, h, w = x.shape
        M = n * h * w
        L, st, dev = lib(), stream(), x.device
        scale = torch.empty(c, dtype=torch.float32, device=dev)
        shift = torch.empty(c, dtype=torch.float32, device=dev)
        training = bn.training or not bn.track_running_stats
        if training:
            if bn.momentum is None:
                raise NotImplementedError('BatchNorm2d(momentum=None) is not supported')
            stats = torch.zeros((2, c), dtype=torch.float32, device=dev)
            check(L.saicv_bn_stats(dtype_code(dt), ptr(x), M, c, ptr(stats[0]), ptr(stats[1]), st), 'bn_stats')
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            invstd = torch.empty(c, dtype=torch.float32, device=dev)
            track = bn.training and bn.track_running_stats and bn.running_mean is not None
            nbt = bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None
            ws = torch.empty(L.saicv_bn_ws_floats(c), dtype=torch.float32, device=dev)
            check(L.saicv_bn_finalize_fwd(ptr(stats[0]), ptr(stats[1]), 1, c, float(M), ptr(gamma), ptr(beta),
                                          ptr(bn.running_mean) if track else 0, ptr(bn.running_var) if track else 0,
                                          float(bn.momentum), float(bn.eps), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(ws),
                                          ptr(nbt), st), 'bn_finalize_fwd')
        else:
            check(L.saicv_bn_eval_coeffs(c, ptr(gamma), ptr(beta), ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps),
                                         ptr(scale), ptr(shift), st), 'bn_eval_coeffs')
        z = torch.empty_like(x)
        check(L.saicv_bn_act_fwd(dtype_code(dt), ptr(x), 0, ptr(z), ptr(scale), ptr(shift), M, c, 0, 0, st), 'bn_act_fwd')
        if training:
            ctx.save_for_backward(x, gamma, mean, invstd)
        else:
            # frozen statistics: dx needs only scale; the affine parameters still get gradients (torch does the same in eval mode):
            # dgamma = sum dz * (x - running_mean) * rsqrt(running_var + eps), dbeta = sum dz -- keep x only when one is asked for
            need_affine = bool(gamma.requires_grad or (beta is not None and beta.requires_grad))
            if need_affine:
                rinv = torch.rsqrt(bn.running_var.detach().float() + float(bn.eps))
                ctx.save_for_backward(x, bn.running_mean.detach().float().clone(), rinv, scale)
            else:
                ctx.save_for_backward(None, None, None, scale)
        ctx.training = training
        return z

    @staticmethod
    def backward(ctx, dz):
        x, gamma, mean, invstd = ctx.saved_tensors
        L, st = lib(), stream()
        dz = _nhwc(dz)
        n, c, h, w = dz.shape
        M = n * h * w
        if not ctx.training:
            # frozen statistics: dx = scale * dz (invstd holds scale here); with x saved also the affine gradients
            dt = dz.dtype
            dx = torch.empty_like(dz)
            if x is None:
                check(L.saicv_channel_scale_add_bwd(dtype_code(dt), ptr(dz), 0, ptr(invstd), ptr(dx), 0, M, c, st), 'bn_eval_bwd')
                return dx, None, None, None
            rmean, rinv, scale = gamma, mean, invstd            # the eval-mode save order: (x, running_mean, rsqrt(var + eps), scale)
            if dz.dtype != x.dtype:
                dz = dz.to(x.dtype)
                dx = torch.empty_like(dz)
            sums = torch.zeros((3, c), dtype=torch.float32, device=dz.device)      # sum dz | sum dz^2 (unused) | sum dz * x
            check(L.saicv_bn_stats(dtype_code(dz.dtype), ptr(dz), M, c, ptr(sums[0]), ptr(sums[1]), st), 'bn_eval_bwd_sum')
            check(L.saicv_channel_scale_add_bwd(dtype_code(dz.dtype), ptr(dz), ptr(x), ptr(scale), ptr(dx), ptr(sums[2]), M, c, st),
                  'bn_eval_bwd')
            dbeta = sums[0]
            dgamma = rinv * (sums[2] - rmean * dbeta)
            return dx, dgamma, dbeta.clone(), None
        dt = x.dtype
        if dz.dtype != dt:
            dz = dz.to(dt)
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.saicv_bn_bwd_ws_floats(M, c, dtype_code(dt)), dtype=torch.float32, device=x.device)
        check(L.saicv_bn_act_bwd(dtype_code(dt), ptr(dz), 0, 0, ptr(x), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx), 0, ptr(dgamma),
                                 ptr(dbeta), M, c, 0, 0, ptr(ws), st), 'bn_act_bwd')
        return dx, dgamma, dbeta, None


def batch_norm2d(x, bn):
    """bn: the nn.BatchNorm2d holding the parameters and running statistics"""
    return BatchNorm2dFn.apply(x, bn.weight, bn.bias, bn)


class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm (+ the ReLU behind it) on an NHWC activation in the compute dtype, fp32 arithmetic (csrc/groupnorm.hip): the
    normalisation of the FCOS head towers (reference detection/models/head.py:101-124).  Two streaming passes each way."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        require_gpu(x)
        x = _nhwc(x)
        n, c, h, w = x.shape
        dev = x.device
        mean_rstd = torch.empty((2, n, groups), dtype=torch.float32, device=dev)
        ab = torch.empty((2, n, c), dtype=torch.float32, device=dev)
        ws = torch.empty(lib().saicv_groupnorm_ws_floats(n, c), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        check(lib().saicv_groupnorm_fwd(dtype_code(x.dtype), ptr(x), ptr(weight), ptr(bias), ptr(y), ptr(mean_rstd), ptr(ab), ptr(ws),
                                        n, h * w, c, groups, float(eps), int(relu), stream()), 'groupnorm_fwd')
        ctx.save_for_backward(x, weight, bias, mean_rstd, ab)
        ctx.cfg = (groups, bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean_rstd, ab = ctx.saved_tensors
        groups, relu = ctx.cfg
        n, c, h, w = x.shape
        dy = _nhwc(dy)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dev = x.device
        dx = torch.empty_like(x)
        want_w = weight is not None and ctx.needs_input_grad[1]
        want_b = bias is not None and ctx.needs_input_grad[2]
        gw = _arena_grad(weight) if want_w else None
        gb = _arena_grad(bias) if want_b else None
        dgamma = gw if gw is not None else (torch.zeros(c, dtype=torch.float32, device=dev) if want_w else None)
        dbeta = gb if gb is not None else (torch.zeros(c, dtype=torch.float32, device=dev) if want_b else None)
        ws = torch.empty(lib().saicv_groupnorm_ws_floats(n, c), dtype=torch.float32, device=dev)
        check(lib().saicv_groupnorm_bwd(dtype_code(x.dtype), ptr(dy), ptr(x), ptr(weight), ptr(mean_rstd), ptr(ab), ptr(dx), ptr(dgamma),
                                        ptr(dbeta), ptr(ws), n, h * w, c, groups, int(relu), stream()), 'groupnorm_bwd')
        return (dx, dgamma if (want_w and gw is None) else None, dbeta if (want_b and gb is None) else None, None, None, None)


def group_norm(x, gn, relu=False):
    """gn: the nn.GroupNorm holding the parameters; under autocast the activation is normalised in the autocast dtype's storage
    with fp32 arithmetic (the reference's autocast runs group_norm in fp32 and the next convolution casts its input back)"""
    if torch.is_autocast_enabled('cuda') and x.is_floating_point() and x.dtype != compute_dtype():
        x = x.to(compute_dtype())
    return GroupNormFn.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b on the implicit-GEMM kernel (1x1 geometry).  nn.Linear of resnet.py:204."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_f32):
        require_gpu(x, weight)
        if x.dim() != 2:
            raise ValueError('LinearFn expects a 2-d input')
        x = x.contiguous()
        dt = x.dtype
        b, ci = x.shape
        o = weight.shape[0]
        e = _lib.epc(dt)
        if ci % e:
            raise ValueError(f'linear: in_features={ci} must be a multiple of {e}')
        op = ((o + e - 1) // e) * e              # out_features padded to the 16-byte chunk
        wf, wd = packed_weight(weight, dt, ci, ctx.needs_input_grad[0], op)
        d = _desc(b, 1, 1, ci, op, 1, 1, 1, 0, dt)
        odt = torch.float32 if (out_f32 or dt == torch.float32) else dt
        y = torch.empty((b, op), dtype=odt, device=x.device)
        bp = bias
        if bias is not None and op != o:
            bp = torch.zeros(op, dtype=torch.float32, device=x.device)
            bp[:o] = bias.detach()
        t0 = KernelTimer.begin('linear_head')
        check(lib().saicv_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(wf), ptr(bp), ptr(y),
                                     int(odt == torch.float32), 0, 0, stream()), 'linear_fwd')
        KernelTimer.end(t0, 'linear_head', 2.0 * b * ci * o, 0)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (d, wd, bias is not None, o, op)
        return y if op == o else y[:, :o]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        d, wd, has_bias, o, op = ctx.cfg
        dt = x.dtype
        L = lib()
        st = stream()
        if op != o:
            dyp = torch.zeros((dy.shape[0], op), dtype=dt, device=dy.device)
            dyp[:, :o] = dy
            dy = dyp
        else:
            dy = dy.contiguous()
            if dy.dtype != dt:
                dy = dy.to(dt)
        b, ci = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if wd is None:
                _, wd = packed_weight(weight, dt, ci, True, op)
            dx = torch.empty((b, ci), dtype=dt, device=x.device)
            check(L.saicv_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(wd), ptr(dx), st), 'linear_dgrad')
        if ctx.needs_input_grad[1]:
            dw = torch.zeros((op, ci), dtype=torch.float32, device=x.device)
            check(L.saicv_conv2d_wgrad(ctypes.byref(d), ptr(dy), ptr(x), ptr(dw), st), 'linear_wgrad')
            dw = dw[:o]
        if has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(op, dtype=torch.float32, device=x.device)
            check(L.saicv_colsum(dtype_code(dt), ptr(dy), b, op, ptr(db), st), 'colsum')
            db = db[:o]
        return dx, dw, db, None


def linear(x, weight, bias=None, out_f32=False):
    return LinearFn.apply(x, weight, bias, out_f32)


# ------------------------------------------------------------------------------ pooling
class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(k, s, p) on NHWC (reference resnet.py:184)."""

    @staticmethod
    def forward(ctx, x, k, stride, pad):
        require_gpu(x)
        x = _nhwc(x)
        n, c, h, w = x.shape
        oh = (h + 2 * pad - k) // stride + 1
        ow = (w + 2 * pad - k) // stride + 1
        out = _empty_nhwc(n, c, oh, ow, x.dtype, x.device)
        idx = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device)
        check(lib().saicv_maxpool_fwd(dtype_code(x.dtype), ptr(x), ptr(out), ptr(idx), n, h, w, c, oh, ow, k,
                                      stride, pad, stream()), 'maxpool_fwd')
        ctx.save_for_backward(idx)
        ctx.cfg = (n, c, h, w, oh, ow, k, stride, pad)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        n, c, h, w, oh, ow, k, stride, pad = ctx.cfg
        dout = _nhwc(dout)
        dx = _empty_nhwc(n, c, h, w, dout.dtype, dout.device)
        check(lib().saicv_maxpool_bwd(dtype_code(dout.dtype), ptr(dout), ptr(idx), ptr(dx), n, h, w, c, oh, ow,
                                      k, stride, pad, stream()), 'maxpool_bwd')
        return dx, None, None, None


def max_pool2d(x, k, stride, pad):
    return MaxPoolFn.apply(x, k, stride, pad)


class GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1,1)) + flatten -> [N, C] (reference resnet.py:203,243-244)."""

    @staticmethod
    def forward(ctx, x):
        require_gpu(x)
        x = _nhwc(x)
        n, c, h, w = x.shape
        out = torch.empty((n, c), dtype=x.dtype, device=x.device)
        check(lib().saicv_avgpool_fwd(dtype_code(x.dtype), ptr(x), ptr(out), n, h * w, c, stream()), 'avgpool_fwd')
        ctx.cfg = (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, c, h, w = ctx.cfg
        dout = dout.contiguous()
        dx = _empty_nhwc(n, c, h, w, dout.dtype, dout.device)
        check(lib().saicv_avgpool_bwd(dtype_code(dout.dtype), ptr(dout), ptr(dx), n, h * w, c, stream()),
              'avgpool_bwd')
        return dx


def global_avg_pool(x):
    return GlobalAvgPoolFn.apply(x)


# ------------------------------------------------------------------------------ losses
class SoftmaxCEFn(torch.autograd.Function):
    """mean softmax cross-entropy on fp32 logits; hard (int64) or soft (fp32 [B,C]) labels.

    Reference SimpleAICV/classification/losses.py:21-28 (CELoss), :86-91 (OneHotLabelCELoss)."""

    @staticmethod
    def forward(ctx, logits, label, soft):
        require_gpu(logits, label)
        logits = logits.float().contiguous()
        b, c = logits.shape
        if soft:
            label = label.float().contiguous()
        else:
            label = label.long().contiguous()
        dev = logits.device
        row = torch.empty(b, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0]
        dlog = torch.empty((b, c), dtype=torch.float32, device=dev) if need else None
        check(lib().saicv_softmax_ce_fwd(ptr(logits), ptr(label), int(soft), b, c, ptr(row), ptr(loss), ptr(dlog),
                                         stream()), 'softmax_ce_fwd')
        if need:
            ctx.save_for_backward(dlog)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (dlog,) = ctx.saved_tensors
        gout = gout.float().contiguous()
        out = torch.empty_like(dlog)
        check(lib().saicv_scale_by_scalar(_lib.F32, ptr(dlog), ptr(gout), ptr(out), dlog.numel(), stream()),
              'scale_by_scalar')
        return out, None, None


def softmax_cross_entropy(logits, label, soft=False):
    return SoftmaxCEFn.apply(logits, label, soft)
