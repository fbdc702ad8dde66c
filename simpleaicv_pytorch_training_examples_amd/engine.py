"""Data-parallel training engine for MI355X: flat fp32 arenas, bucketed RCCL gradient
all-reduce overlapped with backward, fused flat optimizers and a sync-free GradScaler.

Replaces, behind the same call surface, what the reference gets from
  * nn.parallel.DistributedDataParallel        (tools/utils.py:193-197, build_training_mode)
  * torch.optim.SGD / AdamW                     (tools/utils.py:292-679, build_optimizer)
  * torch.amp.GradScaler                        (tools/utils.py:199-200)
and removes their per-step host synchronisation: the inf/nan decision, the unscale, the
clip and the parameter update all stay on the device.

Design (MI355X-first, one process per GPU):
  * every parameter is re-pointed into ONE flat fp32 arena (each parameter starts on a
    1024-element boundary), its .grad into a second arena; channels_last conv weights keep
    their strides (as_strided views), so state_dict()/load_state_dict() are unchanged;
  * gradient buckets are contiguous slices of the gradient arena in reverse registration
    order (the order backward produces them); a post-accumulate hook counts parameters and,
    when a bucket is complete, enqueues one all-reduce on the library's own RCCL communicator
    (`saicv_comm_allreduce_bucket`, csrc/comm.hip).  Inside a captured step (StepGraph) it runs on a
    communication stream, ordered by events, and overlaps the rest of backward; in an eagerly
    launched step it runs on the compute stream itself -- on this runtime any cross-stream ordering
    against a busy compute stream costs more than the collective (profiles/r02_ddp_eager_path.md).
    Process groups on another backend (gloo in the CPU tests) go through torch.distributed instead;
  * xGMI is point-to-point (7 links x ~153 GB/s): a ring all-reduce moves 1.75x the bucket
    over one link per GPU, so buckets are large (default 48 MiB) and only the LAST bucket to
    complete (stem + first stage) is small (4 MiB) to shorten the exposed tail;
  * `no_sync()` only suppresses the enqueue; gradients keep accumulating in the arena.
"""
import contextlib
import os
import time

import torch
import torch.distributed as dist

from . import _lib, ops, ops_tfm
from ._lib import check, lib, ptr

ALIGN = 1024          # elements; one optimizer workgroup never straddles two parameters
HYPER = 8             # floats per param group in the device hyper-parameter table


def _dense_numel(p):
    return p.numel()


class FlatArena:
    """Owns the flat fp32 parameter and gradient arenas of a model."""

    def __init__(self, named_params, device):
        self.names, self.params = [], []
        seen = set()
        for n, p in named_params:
            if id(p) in seen:
                continue
            seen.add(id(p))
            if p.dtype != torch.float32:
                raise TypeError(f'{n}: master parameters must be fp32, got {p.dtype}')
            self.names.append(n)
            self.params.append(p)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += ((p.numel() + ALIGN - 1) // ALIGN) * ALIGN
        self.total = off
        self.device = device
        self.flat_param = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                    p.data = p.data.contiguous()
                view = self.flat_param.as_strided(p.shape, p.stride(), o)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad.as_strided(p.shape, p.stride(), o)
                # kernels accumulate into p.grad in place (ops._arena_grad), also for parameters used several times per
                # step (SAM decoder: 1 + decoder_iters passes; DETR's shared decoder norm): every use adds its share, and
                # "complete" is signalled by the post-accumulate hook below, which autograd runs after the LAST use
                p._saicv_direct = True
        # "this parameter's gradient of the current backward is complete": autograd runs a leaf's
        # AccumulateGrad node exactly once per backward, after every use of the leaf (also when the
        # kernels wrote the gradient in place and returned None), and this hook with it
        self.arrived = [False] * len(self.params)
        self.listeners = []             # callables(index), e.g. the DDP engine's bucket counter
        self._mask_cache = {}
        self._flag_cache = {}
        self._block_param = None
        self.zero_listeners = []        # callables() run by zero_grad(): per-backward state of the DDP engine
        # set by DistributedDataParallel(find_unused_parameters=True) after every gradient sync: fp32 [n params], > 0 where
        # ANY rank produced a gradient for the parameter in the backward that was just reduced
        self.global_flags = None
        for i, p in enumerate(self.params):
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(self._make_hook(i))
        ops.bump_weights_epoch()

    def _make_hook(self, i):
        def hook(param):
            self.arrived[i] = True
            for cb in self.listeners:
                cb(i)
        return hook

    def missing(self):
        """Indices of the trainable parameters without a gradient from THIS rank since the last zero_grad()."""
        return tuple(i for i, p in enumerate(self.params) if p.requires_grad and not self.arrived[i])

    def local_flags(self):
        """fp32 [n params] device tensor: 1 where this rank produced a gradient (or the parameter is not trainable)."""
        missing = self.missing()
        f = self._flag_cache.get(missing)
        if f is None:
            t = torch.ones(len(self.params), dtype=torch.float32)
            for i in missing:
                t[i] = 0.0
            f = t.to(self.device)
            if len(self._flag_cache) > 64:
                self._flag_cache.clear()
            self._flag_cache[missing] = f
        return f

    def block_param(self):
        """int64 [total/ALIGN] device table: index of the parameter each 1024-element block belongs to."""
        if self._block_param is None:
            t = torch.zeros(self.total // ALIGN, dtype=torch.int64)
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                t[o // ALIGN:o // ALIGN + (p.numel() + ALIGN - 1) // ALIGN] = i
            self._block_param = t.to(self.device)
        return self._block_param

    def has_grad_mask(self):
        """uint8 [total/ALIGN] device table, 1 for the blocks of parameters that received a gradient since
        the last zero_grad(); None when every trainable parameter did (the common case).  torch.optim
        skips parameters whose .grad is None; the flat kernels skip the blocks this table zeroes.
        Under data parallelism with find_unused_parameters the table is GLOBAL: a parameter this rank did not use
        still carries the averaged gradient of the ranks that did (nn.parallel.DistributedDataParallel sets its
        .grad on every rank), so every rank must step it -- otherwise the replicas drift apart.  The per-parameter
        flags were summed over the ranks with the gradients (device side, no host read)."""
        missing = self.missing()
        if not missing:
            return None                 # used everywhere here => used somewhere: every block is stepped
        if self.global_flags is not None:
            return (self.global_flags > 0).to(torch.uint8)[self.block_param()]
        m = self._mask_cache.get(missing)
        if m is None:
            t = torch.ones(self.total // ALIGN, dtype=torch.uint8)
            for i in missing:
                o, nb = self.offsets[i] // ALIGN, (self.params[i].numel() + ALIGN - 1) // ALIGN
                t[o:o + nb] = 0
            m = t.to(self.device)
            if len(self._mask_cache) > 64:
                self._mask_cache.clear()
            self._mask_cache[missing] = m
        return m

    def zero_grad(self):
        ops.join_side_stream()
        self.flat_grad.zero_()
        self.arrived = [False] * len(self.params)
        for cb in self.zero_listeners:
            cb()
        # re-attach views in case someone set .grad = None (optimizer.zero_grad(set_to_none=True))
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad.as_strided(p.shape, p.stride(), o)

    def block_groups(self, group_of_param):
        """int32 [total/ALIGN] table: optimizer param-group index of each 1024-element block."""
        table = torch.full((self.total // ALIGN,), -1, dtype=torch.int32)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            g = group_of_param.get(id(p), -1)
            nb = (p.numel() + ALIGN - 1) // ALIGN
            table[o // ALIGN:o // ALIGN + nb] = g
        return table.to(self.device)


def _arena_of(model):
    arena = getattr(model, '_saicv_arena', None)
    if arena is None:
        dev = next(model.parameters()).device
        arena = FlatArena(list(model.named_parameters()), dev)
        model._saicv_arena = arena
    return arena


# ------------------------------------------------------------------------------ optimizers
class _FlatOptimizer:
    """torch.optim-compatible surface (param_groups, step, zero_grad, state_dict) on the fused
    flat kernels.  `param_groups[i]['lr']` may be rewritten every iteration by the reference
    Scheduler (tools/utils.py:223-260); values are uploaded to the device table at step()."""

    def __init__(self, model, param_groups):
        self.arena = _arena_of(model)
        if self.arena.device.type != 'cuda':
            raise RuntimeError('the fused flat optimizers run on MI355X only (HIP kernels, no CPU fallback); '
                               'move the model to the GPU before building the optimizer')
        self._pidx = {id(p): k for k, p in enumerate(self.arena.params)}
        self.param_groups = []
        group_of = {}
        for gi, g in enumerate(param_groups):
            g = dict(g)
            g['params'] = list(g['params'])
            for p in g['params']:
                group_of[id(p)] = gi
            self.param_groups.append(g)
        self.block_group = self.arena.block_groups(group_of)
        self._last_hyper = None
        self._hyper_dev = torch.zeros(len(self.param_groups) * HYPER, dtype=torch.float32, device=self.arena.device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=self.arena.device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=self.arena.device)
        self.track_missing_grads = True     # False: treat every parameter as having a gradient (static step graphs)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _upload(self, rows):
        flat = [float(v) for row in rows for v in row]
        if flat == self._last_hyper:
            return
        self._last_hyper = flat
        host = torch.tensor(flat, dtype=torch.float32)
        if self.arena.device.type == 'cuda':
            # a fresh pinned staging tensor per change: the caching host allocator keeps it alive
            # until the async copy has run, so the host may run ahead of the stream safely
            host = host.pin_memory()
        self._hyper_dev.copy_(host, non_blocking=True)

    def _launch(self, inv_scale, found_inf, has_grad):
        raise NotImplementedError

    def check_finite(self):
        """found_inf[0] = 1 if any gradient is inf / nan (device side, no host sync)."""
        ops.join_side_stream()
        self.found_inf.zero_()
        check(lib().saicv_grad_stats(ptr(self.arena.flat_grad), self.arena.total, ptr(self.found_inf), 0,
                                     _lib.stream()), 'grad_stats')

    def clip_grad_norm_(self, max_norm, inv_scale=None):
        """torch.nn.utils.clip_grad_norm_ over the whole arena, fused with the unscale."""
        ops.join_side_stream()
        self.sumsq.zero_()
        check(lib().saicv_grad_stats(ptr(self.arena.flat_grad), self.arena.total, 0, ptr(self.sumsq),
                                     _lib.stream()), 'grad_stats')
        check(lib().saicv_grad_clip_scale(ptr(self.arena.flat_grad), self.arena.total, ptr(self.sumsq),
                                          ptr(inv_scale), float(max_norm), _lib.stream()), 'grad_clip_scale')

    def clip_grad_value_(self, clip_value, inv_scale=None):
        """torch.nn.utils.clip_grad_value_ over the whole arena, fused with the unscale (the reference unscales first)."""
        ops.join_side_stream()
        check(lib().saicv_grad_clip_value(ptr(self.arena.flat_grad), self.arena.total, ptr(inv_scale), float(clip_value),
                                          _lib.stream()), 'grad_clip_value')

    def step(self, inv_scale=None, found_inf=None):
        mask = self.arena.has_grad_mask() if self.track_missing_grads else None
        ops.join_side_stream()
        self.refresh_hyper()
        self._launch(inv_scale, found_inf, mask)
        ops.bump_weights_epoch()

    def refresh_hyper(self):
        """Uploads the per-group hyper-parameter table if the host changed it (the reference Scheduler rewrites
        param_groups[i]['lr'] every iteration).  step() does it by itself; a captured step graph (StepGraph) calls
        it before every replay, because the captured kernels only READ the device table."""
        if self.arena.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return
        self._upload(self._hyper_rows())

    # ---- torch.optim checkpoint layout: state[i] = {per-parameter tensors}, param_groups[g]['params'] = [i, ...]
    # with i the running index over the groups' parameters; per-parameter tensors are views of the flat state
    # arenas with the parameter's own shape and strides, so a reference `latest.pth` loads here and vice versa.
    def _index_of_params(self):
        order, i = {}, 0
        for g in self.param_groups:
            for p in g['params']:
                order[id(p)] = i
                i += 1
        return order

    def _param_view(self, flat, p):
        return flat.as_strided(p.shape, p.stride(), self.arena.offsets[self._pidx[id(p)]])

    def state_dict(self):
        order = self._index_of_params()
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d['params'] = [order[id(p)] for p in g['params']]
            groups.append(d)
        state = {}
        for g in self.param_groups:
            for p in g['params']:
                entry = self._state_of(p)
                if entry is not None:
                    state[order[id(p)]] = entry
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        if 'param_groups' not in sd or 'state' not in sd:
            raise ValueError('optimizer state_dict must have the torch.optim layout {state, param_groups}')
        if len(sd['param_groups']) != len(self.param_groups):
            raise ValueError(f"optimizer state_dict has {len(sd['param_groups'])} param groups, "
                             f'this optimizer {len(self.param_groups)}')
        flat_params = []
        for g, s in zip(self.param_groups, sd['param_groups']):
            if len(s['params']) != len(g['params']):
                raise ValueError('optimizer state_dict: a param group has a different number of parameters')
            g.update({k: v for k, v in s.items() if k != 'params'})
            flat_params.extend(zip(s['params'], g['params']))
        self._reset_state()
        for idx, p in flat_params:
            entry = sd['state'].get(idx, sd['state'].get(str(idx)))
            if entry is not None:
                self._load_state_of(p, entry)
        self._last_hyper = None


class SGD(_FlatOptimizer):
    """torch.optim.SGD(momentum, weight_decay, nesterov) semantics, one launch for all params."""

    def __init__(self, model, param_groups, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        # every key torch.optim.SGD keeps in a param group, so a checkpoint written here loads into torch.optim
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=nesterov,
                        maximize=False, foreach=None, differentiable=False, fused=None)
        super().__init__(model, [{**defaults, **g} for g in param_groups])
        for g in self.param_groups:
            if g['dampening'] != 0 or g['maximize']:
                raise NotImplementedError('flat SGD: dampening / maximize are not used by any reference config')
        self.momentum_buf = torch.zeros_like(self.arena.flat_param)

    def _reset_state(self):
        self.momentum_buf.zero_()

    def _state_of(self, p):
        # torch creates momentum_buffer at a parameter's first step (= its first gradient); an all-zero buffer
        # is indistinguishable from "not created yet" for the update rule, so it is always emitted
        return {'momentum_buffer': self._param_view(self.momentum_buf, p).detach().clone()}

    def _load_state_of(self, p, entry):
        buf = entry.get('momentum_buffer')
        if buf is not None:
            if tuple(buf.shape) != tuple(p.shape):
                raise ValueError(f'momentum_buffer shape {tuple(buf.shape)} does not match parameter {tuple(p.shape)}')
            self._param_view(self.momentum_buf, p).copy_(buf)

    def _hyper_rows(self):
        return [[g['lr'], g['weight_decay'], g['momentum'], 0, 0, 0, 0, 1.0 if g['nesterov'] else 0.0]
                for g in self.param_groups]

    def _launch(self, inv_scale, found_inf, has_grad):
        a = self.arena
        check(lib().saicv_sgd_flat(ptr(a.flat_param), ptr(a.flat_grad), ptr(self.momentum_buf),
                                   ptr(self.block_group), ptr(self._hyper_dev), ptr(inv_scale), ptr(found_inf),
                                   ptr(has_grad), a.total, _lib.stream()), 'sgd_flat')


class AdamW(_FlatOptimizer):
    """torch.optim.AdamW semantics (decoupled decay, bias correction), one launch."""

    def __init__(self, model, param_groups, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        super().__init__(model, [{**defaults, **g} for g in param_groups])
        for g in self.param_groups:
            if g['amsgrad'] or g['maximize']:
                raise NotImplementedError('flat AdamW: amsgrad / maximize are not used by any reference config')
        self.exp_avg = torch.zeros_like(self.arena.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat_param)
        # torch's per-parameter state['step'], one counter per 1024-element block, advanced by the kernel itself
        # only when the block is really updated (a GradScaler-skipped step or a parameter without a gradient does
        # not advance its bias correction) -- nothing about the step count is uploaded by the host
        self.step_blk = torch.zeros(self.arena.total // ALIGN, dtype=torch.float32, device=self.arena.device)

    def _reset_state(self):
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_blk.zero_()

    def _blocks_of(self, p):
        k = self._pidx[id(p)]
        o = self.arena.offsets[k] // ALIGN
        return o, o + (p.numel() + ALIGN - 1) // ALIGN

    def _state_of(self, p):
        b0, b1 = self._blocks_of(p)
        step = self.step_blk[b0:b0 + 1].detach().cpu().reshape(())
        if float(step) == 0:
            return None                      # torch has no state for a parameter that never stepped
        return {'step': step, 'exp_avg': self._param_view(self.exp_avg, p).detach().clone(),
                'exp_avg_sq': self._param_view(self.exp_avg_sq, p).detach().clone()}

    def _load_state_of(self, p, entry):
        for key, flat in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
            if tuple(entry[key].shape) != tuple(p.shape):
                raise ValueError(f'{key} shape {tuple(entry[key].shape)} does not match parameter {tuple(p.shape)}')
            self._param_view(flat, p).copy_(entry[key])
        b0, b1 = self._blocks_of(p)
        self.step_blk[b0:b1] = float(entry['step'])

    def _hyper_rows(self):
        return [[g['lr'], g['weight_decay'], g['betas'][0], g['betas'][1], g['eps'], 1 - g['betas'][0], 1 - g['betas'][1], 0]
                for g in self.param_groups]

    def _launch(self, inv_scale, found_inf, has_grad):
        a = self.arena
        check(lib().saicv_adamw_flat(ptr(a.flat_param), ptr(a.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                     ptr(self.block_group), ptr(self._hyper_dev), ptr(inv_scale), ptr(found_inf),
                                     ptr(has_grad), ptr(self.step_blk), a.total, _lib.stream()), 'adamw_flat')


# ------------------------------------------------------------------------------ GradScaler
class GradScaler:
    """torch.amp.GradScaler surface without host syncs: the scale lives on the device, the
    inf/nan test sets a device flag that the optimizer kernel honours, update() is a kernel."""

    def __init__(self, device='cuda', init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5,
                 growth_interval=2000, enabled=True):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        # state = [scale, growth_tracker, 1/scale]
        self.state = torch.tensor([init_scale, 0.0, 1.0 / init_scale], dtype=torch.float32, device=device)
        self._unscaled = False

    def scale(self, loss):
        return loss * self.state[0] if self.enabled else loss

    def get_scale(self):
        return float(self.state[0])

    def unscale_(self, optimizer, max_norm=None):
        """Folds 1/scale (and, if given, clip_grad_norm_) into the gradient arena."""
        if not self.enabled:
            return
        optimizer.check_finite()
        optimizer.clip_grad_norm_(max_norm if max_norm is not None else float('inf'), self.state[2:3])
        self._unscaled = True

    def step(self, optimizer):
        if not self.enabled:
            optimizer.step()
            return
        if self._unscaled:
            optimizer.step(None, optimizer.found_inf)
        else:
            optimizer.check_finite()
            optimizer.step(self.state[2:3], optimizer.found_inf)
        self._found_inf = optimizer.found_inf

    def update(self):
        if not self.enabled:
            return
        check(lib().saicv_scaler_update(ptr(self.state), ptr(self._found_inf), self.growth_factor,
                                        self.backoff_factor, self.growth_interval, _lib.stream()), 'scaler_update')
        self._unscaled = False

    def state_dict(self):
        s = self.state.detach().cpu()
        return {'scale': float(s[0]), 'growth_tracker': int(s[1]), 'growth_factor': self.growth_factor,
                'backoff_factor': self.backoff_factor, 'growth_interval': self.growth_interval}

    def load_state_dict(self, sd):
        self.state.copy_(torch.tensor([sd['scale'], float(sd['growth_tracker']), 1.0 / sd['scale']]))


# ------------------------------------------------------------------------------ step graph
def any_nonfinite(*tensors):
    """bool 0-d device tensor: some element of the given tensors is inf / nan -- the device-side form of the reference loops'
    `torch.isinf / isnan` host branches on the batch (tools/scripts.py:147-151).  Dense fp32 tensors (the image batch: 154 MB at
    256 x 3 x 224 x 224) go through ONE pass of the gradient-statistics kernel; ATen's isfinite().all() is abs + two compares + an
    and + a reduction, two extra passes over the batch and ~8 launches (0.25 ms per step)."""
    dev = tensors[0].device
    flag = torch.zeros(1, dtype=torch.float32, device=dev)
    other = None
    for t in tensors:
        if (t.is_cuda and t.dtype == torch.float32 and t.numel() % 4 == 0 and t.numel() > 0 and t.data_ptr() % 16 == 0
                and (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)))):
            check(lib().saicv_grad_stats(ptr(t), t.numel(), ptr(flag), 0, _lib.stream()), 'grad_stats')
        else:
            b = ~torch.isfinite(t).all()
            other = b if other is None else (other | b)
    bad = flag[0] != 0
    return bad if other is None else (bad | other)


class StepGraph:
    """One training step as a hipGraph: `fn(*tensors) -> tensor | tuple of tensors` runs eagerly `warmup` times,
    is then captured once (torch.cuda.CUDAGraph on a side stream; every saicv kernel is launched on torch's current
    stream, so the whole forward / backward / optimizer launch sequence lands in the graph) and replayed afterwards
    with the inputs copied into the captured step's static input buffers.

    Why: a ResNet-50 step is ~700 kernel launches; issued from Python through ctypes they cost ~18 ms of host time
    per step against ~20 ms of GPU time, a replay costs tens of microseconds.  What must hold for `fn`: no host
    synchronisation (.item(), .tolist(), prints of device values), the same shapes every call, and the same set of
    parameters receiving gradients every call (the optimizer's no-gradient mask is frozen at capture).  Values the
    host changes between steps must live in device buffers refreshed by `before_replay` (e.g.
    optimizer.refresh_hyper for the scheduler's learning rates).  Returned tensors are static buffers overwritten
    by the next replay: clone what must outlive a step."""

    _warned_packets = False

    def __init__(self, fn, warmup=3, before_replay=()):
        self.fn, self.warmup, self.before_replay = fn, warmup, tuple(before_replay)
        # Two captured steps replayed WRONGLY under ROCm's graph packet capture (the SAM step; a deterministic ResNet-50 step whose
        # weight-gradient partials workspace is reused layer after layer -- DESIGN.md section 3k).  The package switches it off when
        # it is imported before the process's first HIP call; in a process where that came too late the step is never captured.
        from . import GRAPH_PACKET_CAPTURE_OFF
        self.eager_only = not GRAPH_PACKET_CAPTURE_OFF and os.environ.get('SAICV_STEP_GRAPH_WITH_PACKETS') != '1'     # (=1: the failing A/B leg)
        if self.eager_only and not StepGraph._warned_packets:
            StepGraph._warned_packets = True
            import warnings
            warnings.warn('StepGraph: ROCm graph packet capture is on (DEBUG_CLR_GRAPH_PACKET_CAPTURE != 0 when HIP started): captured '
                          'steps can replay wrongly with it, so the step runs eagerly.  Import simpleaicv_pytorch_training_examples_amd '
                          'before the first HIP call, or export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0', RuntimeWarning, stacklevel=2)
        self.calls = 0
        self.graph = None
        self.static_in = self.static_out = None
        self._pack_entries = []          # ops._PackRegistry entries the captured step reads (kept "live" across replays)
        self.replays, self.replay_host_s = 0, 0.0        # host time spent issuing replays (input copies + hipGraphLaunch)

    def __call__(self, *inputs):
        if self.graph is None:
            if self.calls < self.warmup or self.eager_only:
                self.calls += 1
                return self.fn(*inputs)
            self._capture(inputs)
        t0 = time.perf_counter()
        for s, x in zip(self.static_in, inputs):
            if s is not None and s.data_ptr() != x.data_ptr():
                s.copy_(x, non_blocking=True)
        for cb in self.before_replay:
            cb()
        ops_tfm.advance_dropout_step()      # the captured dropout seeds are frozen: their device-side part moves on (ops_tfm.py)
        self.graph.replay()
        ops.bump_weights_epoch()        # the replayed optimizer kernels rewrote the parameters
        ops._PackRegistry.touch(self._pack_entries)     # ... and the replayed step used its compute-dtype copies
        self.replays += 1
        self.replay_host_s += time.perf_counter() - t0
        return self.static_out

    def _capture(self, inputs):
        self.static_in = [x.clone() if torch.is_tensor(x) else None for x in inputs]    # clone keeps NHWC strides
        args = [s if s is not None else x for s, x in zip(self.static_in, inputs)]
        for cb in self.before_replay:
            cb()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        dump = os.environ.get('SAICV_GRAPH_DUMP')       # path: write the captured graph as DOT (hipGraphDebugDotPrint) -- which
        if dump:                                        # kernels sit on which branch, e.g. the bucket all-reduces (tests/test_gpu_ddp.py)
            graph.enable_debug_mode()
        failed = None
        try:
            with torch.cuda.graph(graph):
                ops._ZeroPool.zero_all()    # the statistics scratch starts a replay all-zero, whatever ran eagerly in between
                try:
                    out = self.fn(*args)
                except BaseException as e:      # the capture still has to END with every forked stream rejoined: a stream left
                    failed = e                  # capturing refuses every later allocation and copy of the process
                ops.join_side_stream()      # every forked stream rejoins before the capture ends
        except BaseException as e:
            failed = failed or e
        if failed is not None:
            torch.cuda.synchronize()
            raise failed
        if dump:
            graph.debug_dump(dump)
        self.graph, self.static_out = graph, out
        self._pack_entries = ops._PackRegistry.used_now()
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------ DDP engine
class NativeComm:
    """The library's RCCL communicator (include/saicv_hip.h, saicv_comm_*) for one process group member.

    Bootstrap: rank 0 draws the RCCL unique id and publishes it in the torch.distributed store (the rendezvous the
    reference's launcher already set up: tools/train_classification_model.py:76-84, `init_process_group`), the other
    ranks read it there; `saicv_comm_create` is then the collective every rank joins.  torch.distributed carries no
    gradient traffic afterwards."""

    _serial = 0

    @classmethod
    def exchange_id(cls, world, rank, make_id):
        """128 bytes drawn by rank 0 (`make_id()`), identical on every rank afterwards: through the torch.distributed
        store (TCPStore / FileStore of the rendezvous), no collective and no device involved."""
        key = f'saicv_comm_id_{cls._serial}'
        cls._serial += 1
        if world == 1 or not (dist.is_available() and dist.is_initialized()):
            return bytes(make_id())
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            raw = bytes(make_id())
            store.set(key, raw)
            return raw
        return bytes(store.get(key))                       # blocks until rank 0 has published it

    def __init__(self, world, rank):
        import ctypes
        L = lib()

        def make_id():
            b = ctypes.create_string_buffer(128)
            check(L.saicv_comm_unique_id(b), 'comm_unique_id')
            return b.raw

        raw = self.exchange_id(world, rank, make_id)
        if len(raw) != 128:
            raise RuntimeError(f'RCCL unique id of {len(raw)} bytes came out of the store (expected 128)')
        buf = ctypes.create_string_buffer(raw, 128)
        handle = ctypes.c_void_p()
        check(L.saicv_comm_create(buf, world, rank, ctypes.byref(handle)), 'comm_create')
        self.handle, self.world, self.rank = handle, world, rank

    def allreduce_bucket(self, view, producer_stream, average=True):
        check(lib().saicv_comm_allreduce_bucket(self.handle, ptr(view), view.numel(), int(average),
                                                producer_stream.cuda_stream), 'comm_allreduce_bucket')

    def reduce_scatter_all_gather(self, view, producer_stream, average=True):
        """The same result as allreduce_bucket as its two halves, in place: every rank reduces the 1/world slice it owns
        (saicv_comm_reduce_scatter), then the slices are gathered (saicv_comm_all_gather).  Over the point-to-point xGMI mesh
        each half keeps every link busy with 1/world of the bucket; which form is faster for which
        bucket size can only be measured on a multi-GPU node -- DistributedDataParallel takes it for buckets of at least
        SAICV_DDP_RSAG_MIB MiB when that variable is set."""
        n = view.numel()
        assert n % self.world == 0, 'bucket length must be a multiple of the world size'
        per = n // self.world
        shard = view[self.rank * per:(self.rank + 1) * per]
        s = producer_stream.cuda_stream
        check(lib().saicv_comm_reduce_scatter(self.handle, ptr(view), ptr(shard), per, int(average), s), 'comm_reduce_scatter')
        check(lib().saicv_comm_all_gather(self.handle, ptr(shard), ptr(view), per, s), 'comm_all_gather')

    def allreduce_now(self, t, average=False):
        """Small fp32 tensor written on the current stream: reduced on the communication stream (behind every bucket
        already enqueued there), and the current stream waits for it."""
        self.allreduce_bucket(t, torch.cuda.current_stream(), average=average)
        self.join()

    def broadcast(self, t, root=0):
        check(lib().saicv_comm_broadcast(self.handle, ptr(t), t.numel() * t.element_size(), root, _lib.stream()),
              'comm_broadcast')

    def join(self):
        check(lib().saicv_comm_join(self.handle, _lib.stream()), 'comm_join')

    def stats(self):
        import ctypes
        w, r = ctypes.c_int(), ctypes.c_int()
        nb, by = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        check(lib().saicv_comm_stats(self.handle, ctypes.byref(w), ctypes.byref(r), ctypes.byref(nb), ctypes.byref(by)),
              'comm_stats')
        return {'world': w.value, 'rank': r.value, 'buckets': nb.value, 'bytes': by.value}

    def close(self):
        if self.handle:
            lib().saicv_comm_destroy(self.handle)
            self.handle = None

    def self_check(self, device):
        """One small all-reduce with a known answer: sum over ranks of (rank + 1)."""
        t = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device=device)
        self.allreduce_bucket(t, torch.cuda.current_stream(), average=False)
        self.join()
        return bool((t == self.world * (self.world + 1) / 2).all().item())


class DistributedDataParallel(torch.nn.Module):
    _slots_env_set = False

    def __del__(self):
        if self._slots_env_set:          # the CU-slot share this wrapper asked the weight-gradient kernel for ends with it
            os.environ.pop('SAICV_TN_SLOTS_PCT', None)

    """Drop-in for nn.parallel.DistributedDataParallel on the flat gradient arena.

    Honours the surface the reference loop uses (tools/scripts.py:124,155,173,185,219;
    tools/train_classification_model.py:217-227): forward pass-through, `.module`, `no_sync()`,
    `module.`-prefixed state_dict keys, mean-over-ranks gradients, rank-0 buffer broadcast
    before forward (`broadcast_buffers`), tolerance of parameters that receive no gradient."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False,
                 process_group=None, bucket_cap_mb=48, last_bucket_cap_mb=4, broadcast_buffers=True):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.find_unused_parameters = find_unused_parameters
        self.broadcast_buffers = broadcast_buffers
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # SAICV_DDP_FORCE_SYNC=1: run the whole bucket / all-reduce machinery in a world of one (a mean over one rank is
        # the identity): how the RCCL path is exercised end to end on a single-GPU box (tests/test_gpu_ddp.py)
        self._active = self.world > 1 or os.environ.get('SAICV_DDP_FORCE_SYNC') == '1'
        self.arena = _arena_of(module)
        self.comm = self._native_comm(process_group)
        if self.world > 1 and self.comm is None and self.arena.device.type == 'cuda':
            # torch.distributed's own RCCL kernels will share the GPU with backward: the library only knows about communicators
            # it created itself (g_saicv_comm_world), so tell the weight-gradient kernel to leave CU slots free (csrc/igemm.hip)
            # (set for the life of THIS wrapper only -- ADVICE r05: it used to stay in os.environ and slowed every later single-GPU
            # model of the process; __del__ / close() restore it)
            if 'SAICV_TN_SLOTS_PCT' not in os.environ:
                os.environ['SAICV_TN_SLOTS_PCT'] = '85'
                self._slots_env_set = True
        self._sync = True
        self._works = []
        self._next_bucket = 0           # buckets are launched in list order on every rank
        self._callback_queued = False
        self._done = True               # no backward with pending collectives
        self._build_buckets(int(bucket_cap_mb * 2 ** 20 // 4), int(last_bucket_cap_mb * 2 ** 20 // 4))
        self._flatten_buffers()
        if self._active:
            # identical start on every rank (DDP ctor broadcast, C3 in SURVEY.md section 2.4)
            self._broadcast(self.arena.flat_param)
            if self.flat_buffers is not None:
                self._broadcast(self.flat_buffers)
            for b in self.module.buffers():
                if not b.dtype.is_floating_point:
                    self._broadcast(b)
            ops.bump_weights_epoch()
        self.arena.listeners.append(self._on_grad_complete)
        self.arena.zero_listeners.append(self._reset_backward_state)
        # find_unused_parameters: which parameters got a gradient on ANY rank travels with the gradients (one more
        # tiny all-reduce per step) -- see FlatArena.has_grad_mask
        self._flags = None
        if self._active and find_unused_parameters:
            self._flags = torch.zeros(len(self.arena.params), dtype=torch.float32, device=self.arena.device)

    def _reset_backward_state(self):
        """A backward that raised (out of memory, a failing check in a criterion) never runs autograd's final callbacks:
        `_callback_queued` would stay set and the bucket counters keep the aborted pass's arrivals, so that a loop which
        catches the error and goes on would skip the gradient sync or launch buckets early.  zero_grad() -- which every
        loop calls between two backward passes that are meant to be independent -- therefore starts from a clean slate.
        Collectives already enqueued by the aborted pass stay matched across the ranks only if every rank aborted at the
        same point; a rank-local failure remains fatal for the job, as it is with nn.parallel.DistributedDataParallel."""
        if self._callback_queued or not self._done:
            if self.comm is not None and self._works:
                self.comm.join()
            for w, _ in self._works:
                if w is not None:
                    w.wait()
        self._callback_queued = False
        self._done = True
        self._works = []
        self._next_bucket = 0
        for b in self.buckets:
            b['count'] = 0
        self.arena.global_flags = None

    def _native_comm(self, process_group):
        """The library's own RCCL communicator when the job runs on RCCL over the default group; None (torch.distributed
        carries the collectives) for gloo groups, sub-groups, or SAICV_NATIVE_COMM=0.  Every rank must reach the same
        decision: the outcome of creation + a known-answer all-reduce is agreed on with one MIN all-reduce."""
        if not self._active or os.environ.get('SAICV_NATIVE_COMM', '1') == '0' or process_group is not None:
            return None
        if not (dist.is_available() and dist.is_initialized()):
            if self.world == 1 and torch.cuda.is_available():
                comm = NativeComm(1, 0)
                return comm if comm.self_check(self.arena.flat_grad.device) else None
            return None
        if dist.get_backend() != 'nccl':
            return None
        # rank-local preconditions first (RCCL resolvable in this process): agreed on BEFORE anyone enters the collective
        # communicator creation, so a rank that cannot load RCCL sends every rank to the torch.distributed path instead of
        # leaving the others blocked inside ncclCommInitRank / the store read of the unique id
        pre = torch.tensor([int(lib().saicv_comm_available() == 0)], dtype=torch.int32, device=self.arena.flat_grad.device)
        dist.all_reduce(pre, op=dist.ReduceOp.MIN)
        if int(pre.item()) != 1:
            if dist.get_rank() == 0:
                print('[saicv] RCCL not loadable on every rank; gradients go through torch.distributed')
            return None
        comm, ok = None, 1
        try:
            comm = NativeComm(self.world, dist.get_rank())
            ok = int(comm.self_check(self.arena.flat_grad.device))
        except Exception as e:                                     # noqa: BLE001 -- any failure means "use torch.distributed"
            ok = 0
            if dist.get_rank() == 0:
                print(f'[saicv] native RCCL communicator unavailable ({e}); gradients go through torch.distributed')
        flag = torch.tensor([ok], dtype=torch.int32, device=self.arena.flat_grad.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            if comm is not None:
                comm.close()
            return None
        return comm

    def _broadcast(self, t):
        if self.comm is not None:
            self.comm.broadcast(t, 0)
        elif self.world > 1:
            dist.broadcast(t, 0, group=self.process_group)

    # buckets are contiguous arena ranges; walking parameters in REVERSE registration order
    def _build_buckets(self, cap, last_cap):
        a = self.arena
        size = lambda i: ((a.params[i].numel() + ALIGN - 1) // ALIGN) * ALIGN
        idxs = [i for i, p in enumerate(a.params) if p.requires_grad]
        # the parameters inside the first `last_cap` elements of the arena (stem + first layers,
        # whose gradients arrive last) form their own small bucket
        tail = [i for i in idxs if a.offsets[i] + size(i) <= last_cap]
        main = [i for i in idxs if i not in set(tail)]
        self.buckets = []          # dicts: start, end (arena range), params (indices), count
        for group in (main, tail):
            cur, used = None, 0
            for i in reversed(group):
                if cur is None or used + size(i) > cap:
                    cur = {'start': a.offsets[i], 'end': a.offsets[i] + size(i), 'params': [], 'count': 0}
                    self.buckets.append(cur)
                    used = 0
                cur['start'] = min(cur['start'], a.offsets[i])
                cur['end'] = max(cur['end'], a.offsets[i] + size(i))
                cur['params'].append(i)
                used += size(i)
        self.bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for i in b['params']:
                self.bucket_of[i] = bi

    def _on_grad_complete(self, i):
        """Arena listener: parameter i's gradient of the running backward is complete (fires once per
        backward per parameter, after its last use).  Runs inside autograd's backward."""
        if not self._active:
            return
        if not self._callback_queued:
            # the reference loop (tools/scripts.py:183-226) calls optimizer.step() right after backward():
            # the wait for the in-flight buckets therefore happens in autograd's end-of-backward callback
            # (as nn.parallel.DistributedDataParallel does), not in a call the loop would have to add
            self._callback_queued = True
            self._done = False
            torch.autograd.Variable._execution_engine.queue_callback(self._on_backward_end)
        if not self._sync:
            return
        bi = self.bucket_of.get(i)
        if bi is None:
            return
        self.buckets[bi]['count'] += 1
        self._launch_ready_buckets()

    def _launch_ready_buckets(self, force=False):
        # strictly in list order: every rank issues the same sequence of collectives even when the ranks'
        # graphs complete their buckets in different orders (SAM draws its prompt type per rank and step)
        while self._next_bucket < len(self.buckets):
            b = self.buckets[self._next_bucket]
            if not force and b['count'] != len(b['params']):
                break
            self._reduce_bucket(b)
            self._next_bucket += 1

    def _on_backward_end(self):
        self._callback_queued = False
        if self._sync:
            self.finish_gradient_sync()

    def _reduce_bucket(self, b):
        view = self.arena.flat_grad[b['start']:b['end']]
        if self.comm is not None:
            # weight gradients may be produced on the side stream, BatchNorm / LayerNorm / bias gradients on the compute
            # stream: the communication stream is ordered after BOTH by recording its event on the side stream once
            # that has waited for the compute stream -- the compute stream itself never waits here
            side = ops.side_stream_in_use()
            producer = torch.cuda.current_stream()
            if side is not None:
                side.wait_stream(producer)
                producer = side
            rsag = os.environ.get('SAICV_DDP_RSAG_MIB')
            if rsag and view.numel() * 4 >= float(rsag) * 2 ** 20 and view.numel() % self.comm.world == 0:
                self.comm.reduce_scatter_all_gather(view, producer)
            else:
                self.comm.allreduce_bucket(view, producer)
            self._works.append((None, None))
            return
        if self.world == 1:
            return
        backend = dist.get_backend(self.process_group)
        if backend == 'nccl':
            side = ops.side_stream_in_use()
            if side is not None:
                # weight gradients are produced on the side stream, BatchNorm / LayerNorm / bias gradients on the
                # compute stream: RCCL's stream is ordered after BOTH by issuing the collective from the side stream
                # once that has waited for the compute stream -- the compute stream itself never waits here
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.process_group, async_op=True)
            else:
                w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.process_group, async_op=True)
            self._works.append((w, None))
        else:
            ops.join_side_stream()
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            self._works.append((w, view))

    def _reduce_flags(self):
        """Sum over the ranks of "this rank produced a gradient for parameter i": the last collective of the step."""
        self._flags.copy_(self.arena.local_flags())
        if self.comm is not None:
            self.comm.allreduce_bucket(self._flags, torch.cuda.current_stream(), average=False)
            self._works.append((None, None))
        elif self.world > 1:
            if dist.get_backend(self.process_group) != 'nccl':
                ops.join_side_stream()
            self._works.append((dist.all_reduce(self._flags, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True), None))
        self.arena.global_flags = self._flags

    def finish_gradient_sync(self):
        """Makes the compute stream wait for every in-flight bucket of the backward that just ran, after
        reducing the buckets whose parameters produced no gradient (find_unused_parameters semantics).
        Runs by itself as autograd's end-of-backward callback; calling it again afterwards is a no-op."""
        if self._done:
            return
        if self._active and self._sync:
            self._launch_ready_buckets(force=True)
            if self._flags is not None:
                self._reduce_flags()
        if self.comm is not None and self._works:
            self.comm.join()
        for w, view in self._works:
            if w is not None:
                w.wait()
            if view is not None:
                view.div_(self.world)
        self._works = []
        self._next_bucket = 0
        for b in self.buckets:
            b['count'] = 0
        self._done = True

    def allreduce_grads(self):
        """Stand-in for the reference's manual per-parameter all_reduce loop
        (tools/interactive_segmentation_scripts.py:446-449): one bucketed pass."""
        self._done = False
        old, self._sync = self._sync, True
        try:
            self.finish_gradient_sync()
        finally:
            self._sync = old

    @contextlib.contextmanager
    def no_sync(self):
        old = self._sync
        self._sync = False
        try:
            yield
        finally:
            self._sync = old

    def _flatten_buffers(self):
        """Re-points every floating-point buffer (BN running statistics) into one flat tensor so
        the per-forward rank-0 buffer sync (DDP broadcast_buffers, C2 in SURVEY.md) is ONE
        broadcast instead of one per buffer."""
        entries = []
        for mod in self.module.modules():
            for name, b in mod._buffers.items():
                if b is not None and b.dtype == torch.float32:
                    entries.append((mod, name, b))
        if not entries:
            self.flat_buffers = None
            return
        total = sum(((b.numel() + 3) // 4) * 4 for _, _, b in entries)
        self.flat_buffers = torch.zeros(total, dtype=torch.float32, device=entries[0][2].device)
        o = 0
        with torch.no_grad():
            for mod, name, b in entries:
                view = self.flat_buffers[o:o + b.numel()].view(b.shape)
                view.copy_(b)
                mod._buffers[name] = view
                o += ((b.numel() + 3) // 4) * 4

    def forward(self, *args, **kwargs):
        if self._active and self.broadcast_buffers and self.module.training and self.flat_buffers is not None:
            self._broadcast(self.flat_buffers)
        return self.module(*args, **kwargs)
