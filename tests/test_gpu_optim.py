"""Numerical parity of the flat-arena optimizer path (csrc/optim.hip, engine.SGD / AdamW / GradScaler) with the
reference's own optimizer objects -- torch.optim.SGD / torch.optim.AdamW / torch.nn.utils.clip_grad_norm_ /
torch.amp.GradScaler, as configured by reference tools/utils.py:581-600 (build_optimizer) and used by
tools/scripts.py:195-259 -- on identical parameters and gradients.  The torch.optim side runs on CPU in fp32;
tolerance 1e-6 of each tensor's scale (the two sides differ only in fp32 operation order)."""
import copy
import math

import pytest
import torch
import torch.nn as nn

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-6


class _Net(nn.Module):
    """Parameters of every layout the arenas hold: channels_last conv weights, 2-d, 1-d, a tiny 0-size tail."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.conv = nn.Conv2d(8, 24, 3, bias=False)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.bn = nn.BatchNorm2d(24)
        self.fc1 = nn.Linear(300, 1500)          # > 1024 elements: several optimizer blocks
        self.fc2 = nn.Linear(1500, 7)
        self.unused = nn.Linear(33, 5)           # never receives a gradient in some steps
        with torch.no_grad():
            self.bn.weight.uniform_(0.5, 1.5)
            self.bn.bias.uniform_(-0.2, 0.2)


def _pair():
    """-> (gpu model re-pointed into arenas later, cpu twin with identical values)"""
    gpu = _Net().cuda()
    cpu = copy.deepcopy(gpu).cpu()
    for a, b in zip(gpu.parameters(), cpu.parameters()):
        assert a.stride() == b.stride()
    return gpu, cpu


def _groups(model):
    decay = [p for n, p in model.named_parameters() if p.ndim > 1]
    plain = [p for n, p in model.named_parameters() if p.ndim <= 1]
    return decay, plain


def _set_grads(gpu, cpu, arena, seed, skip=(), scale=1.0, poison=False):
    """Writes the same random gradient into both twins; parameters named in `skip` get none (None on the torch
    side, not-arrived on the arena side -- what a step that did not use them looks like)."""
    g = torch.Generator().manual_seed(seed)
    names = [n for n, _ in gpu.named_parameters()]
    arena.zero_grad()
    for i, ((n, pg), pc) in enumerate(zip(gpu.named_parameters(), cpu.parameters())):
        if any(n.startswith(s) for s in skip):
            pc.grad = None
            continue
        grad = torch.randn(pc.shape, generator=g) * scale
        if poison and n == 'fc1.weight':
            grad.view(-1)[17] = float('inf')
        pc.grad = grad.clone().contiguous(memory_format=torch.channels_last) if pc.dim() == 4 else grad.clone()
        pg.grad.copy_(grad.cuda())
        arena.arrived[arena.names.index(n)] = True


def _compare(gpu, cpu, what):
    for (n, a), b in zip(gpu.named_parameters(), cpu.parameters()):
        e = rel_err(a, b)
        assert e < TOL, (what, n, e)


@pytest.mark.parametrize('nesterov', [False, True])
def test_sgd_flat_matches_torch_optim_sgd(nesterov):
    from simpleaicv_pytorch_training_examples_amd import engine
    gpu, cpu = _pair()
    dg, pg = _groups(gpu)
    dc, pc = _groups(cpu)
    opt = engine.SGD(gpu, [{'params': dg, 'weight_decay': 5e-4, 'lr': 0.1}, {'params': pg, 'weight_decay': 0.0, 'lr': 0.03}],
                     lr=0.1, momentum=0.9, nesterov=nesterov)
    ref = torch.optim.SGD([{'params': dc, 'weight_decay': 5e-4, 'lr': 0.1}, {'params': pc, 'weight_decay': 0.0, 'lr': 0.03}],
                          lr=0.1, momentum=0.9, nesterov=nesterov)
    arena = opt.arena
    for step in range(4):
        skip = ('unused',) if step in (1, 2) else ()
        _set_grads(gpu, cpu, arena, 100 + step, skip=skip)
        if step == 2:                                   # the reference Scheduler rewrites lr every iteration
            for o in (opt, ref):
                o.param_groups[0]['lr'] = 0.05
        opt.step()
        ref.step()
        torch.cuda.synchronize()
        _compare(gpu, cpu, f'sgd step {step}')
    for (n, p), c in zip(gpu.named_parameters(), cpu.parameters()):
        buf = ref.state[c]['momentum_buffer']
        mine = opt._param_view(opt.momentum_buf, p)
        assert rel_err(mine, buf) < TOL, n
    # a GradScaler-skipped step (found_inf != 0) leaves parameters and momentum untouched
    before_p, before_m = arena.flat_param.clone(), opt.momentum_buf.clone()
    _set_grads(gpu, cpu, arena, 999)
    opt.step(None, torch.ones(1, device='cuda'))
    torch.cuda.synchronize()
    assert torch.equal(before_p, arena.flat_param) and torch.equal(before_m, opt.momentum_buf)


def test_adamw_flat_matches_torch_optim_adamw():
    from simpleaicv_pytorch_training_examples_amd import engine
    gpu, cpu = _pair()
    dg, pg = _groups(gpu)
    dc, pc = _groups(cpu)
    kw = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8)
    opt = engine.AdamW(gpu, [{'params': dg, 'weight_decay': 0.05}, {'params': pg, 'weight_decay': 0.0, 'lr': 1e-4}], **kw)
    ref = torch.optim.AdamW([{'params': dc, 'weight_decay': 0.05}, {'params': pc, 'weight_decay': 0.0, 'lr': 1e-4}], **kw)
    arena = opt.arena
    found = torch.zeros(1, device='cuda')
    for step in range(5):
        # step 1: `unused` gets no gradient (torch skips it: no decay, no moment update, no step count)
        # step 3: GradScaler found an inf -> nobody steps, bias corrections must NOT advance
        skip = ('unused',) if step == 1 else ()
        _set_grads(gpu, cpu, arena, 200 + step, skip=skip)
        found.fill_(1.0 if step == 3 else 0.0)
        opt.step(None, found)
        if step != 3:
            ref.step()
        torch.cuda.synchronize()
        _compare(gpu, cpu, f'adamw step {step}')
    for (n, p), c in zip(gpu.named_parameters(), cpu.parameters()):
        st = ref.state[c]
        assert rel_err(opt._param_view(opt.exp_avg, p), st['exp_avg']) < TOL, n
        assert rel_err(opt._param_view(opt.exp_avg_sq, p), st['exp_avg_sq']) < TOL, n
        b0, b1 = opt._blocks_of(p)
        assert float(opt.step_blk[b0]) == float(st['step']), (n, float(opt.step_blk[b0]), float(st['step']))
        assert bool((opt.step_blk[b0:b1] == opt.step_blk[b0]).all())
    assert float(ref.state[cpu.unused.weight]['step']) == 3 and float(ref.state[cpu.fc1.weight]['step']) == 4


def test_optimizer_state_dict_round_trips_through_torch_optim():
    """`latest.pth` carries optimizer.state_dict() (reference tools/train_classification_model.py:260): the flat
    optimizers emit and accept torch.optim's per-parameter layout, so either side resumes the other's checkpoint."""
    from simpleaicv_pytorch_training_examples_amd import engine
    for kind in ('sgd', 'adamw'):
        gpu, cpu = _pair()
        dg, pg = _groups(gpu)
        dc, pc = _groups(cpu)
        if kind == 'sgd':
            opt = engine.SGD(gpu, [{'params': dg, 'weight_decay': 1e-4}, {'params': pg, 'weight_decay': 0.0}], lr=0.1, momentum=0.9)
            ref = torch.optim.SGD([{'params': dc, 'weight_decay': 1e-4}, {'params': pc, 'weight_decay': 0.0}], lr=0.1, momentum=0.9)
        else:
            opt = engine.AdamW(gpu, [{'params': dg, 'weight_decay': 0.05}, {'params': pg, 'weight_decay': 0.0}], lr=1e-3)
            ref = torch.optim.AdamW([{'params': dc, 'weight_decay': 0.05}, {'params': pc, 'weight_decay': 0.0}], lr=1e-3)
        for step in range(2):
            _set_grads(gpu, cpu, opt.arena, 300 + step)
            opt.step()
            ref.step()
        torch.cuda.synchronize()
        # torch -> flat: a fresh flat optimizer loads the torch.optim checkpoint and continues identically
        gpu2, cpu2 = _pair()
        with torch.no_grad():
            for a, b in zip(gpu2.parameters(), cpu.parameters()):
                a.copy_(b)
        dg2, pg2 = _groups(gpu2)
        if kind == 'sgd':
            opt2 = engine.SGD(gpu2, [{'params': dg2, 'weight_decay': 1e-4}, {'params': pg2, 'weight_decay': 0.0}], lr=0.1, momentum=0.9)
        else:
            opt2 = engine.AdamW(gpu2, [{'params': dg2, 'weight_decay': 0.05}, {'params': pg2, 'weight_decay': 0.0}], lr=1e-3)
        opt2.load_state_dict(ref.state_dict())
        # flat -> torch: torch.optim loads what the flat optimizer saved
        sd = opt.state_dict()
        assert set(sd.keys()) == {'state', 'param_groups'}
        assert sd['param_groups'][0]['params'] == list(range(len(dc)))
        ref2 = type(ref)([{'params': dc}, {'params': pc}], lr=0.1)
        ref2.load_state_dict(sd)
        _set_grads(gpu2, cpu, opt2.arena, 400)
        opt2.step()
        ref2.step()
        torch.cuda.synchronize()
        _compare(gpu2, cpu, f'{kind} resumed from the other side\'s checkpoint')
        with pytest.raises(ValueError):
            opt2.load_state_dict({'steps': 3})


def test_clip_grad_norm_and_unscale_match_torch():
    from simpleaicv_pytorch_training_examples_amd import engine
    gpu, cpu = _pair()
    opt = engine.SGD(gpu, [{'params': list(gpu.parameters())}], lr=0.1, momentum=0.0)
    arena = opt.arena
    for max_norm, scale in ((0.1, 1.0), (1e9, 1.0), (1.0, 65536.0)):
        _set_grads(gpu, cpu, arena, 500, scale=scale)
        inv = torch.tensor([1.0 / scale], device='cuda')
        opt.clip_grad_norm_(max_norm, inv if scale != 1.0 else None)
        for p in cpu.parameters():                  # GradScaler.unscale_ then clip (reference scripts.py:205-216)
            p.grad.mul_(1.0 / scale)
        total = torch.nn.utils.clip_grad_norm_(list(cpu.parameters()), max_norm)
        torch.cuda.synchronize()
        assert abs(math.sqrt(float(opt.sumsq)) / scale - float(total)) < 1e-5 * float(total)
        for (n, a), b in zip(gpu.named_parameters(), cpu.parameters()):
            # the clip factor carries the fp32 rounding of a 450 k-term sum of squares (different order on each side)
            assert rel_err(a.grad, b.grad) < 5e-6, (max_norm, scale, n)


def test_clip_grad_value_and_unscale_match_torch():
    """clip_grad_value of the loops (reference tools/scripts.py:211-218: GradScaler.unscale_, then
    torch.nn.utils.clip_grad_value_, then -- if configured -- the norm clip)."""
    from simpleaicv_pytorch_training_examples_amd import engine
    gpu, cpu = _pair()
    opt = engine.SGD(gpu, [{'params': list(gpu.parameters())}], lr=0.1, momentum=0.0)
    arena = opt.arena
    for value, scale in ((0.05, 1.0), (1e9, 1.0), (0.3, 65536.0)):
        _set_grads(gpu, cpu, arena, 600, scale=scale)
        inv = torch.tensor([1.0 / scale], device='cuda')
        opt.clip_grad_value_(value, inv if scale != 1.0 else None)
        for p in cpu.parameters():
            p.grad.mul_(1.0 / scale)
        torch.nn.utils.clip_grad_value_(list(cpu.parameters()), value)
        torch.cuda.synchronize()
        clipped = 0
        for (n, a), b in zip(gpu.named_parameters(), cpu.parameters()):
            assert torch.equal(a.grad.cpu(), b.grad), (value, scale, n)          # a product and a clamp: bit-identical
            clipped += int((b.grad.abs() == value).sum())
        assert (clipped > 0) == (value < 1e9)


def test_grad_scaler_follows_torch_amp_grad_scaler():
    """Same sequence of clean / overflowing steps through torch.amp.GradScaler (on plain GPU tensors and
    torch.optim.SGD) and through engine.GradScaler + the flat SGD: identical scale after every update, identical
    skip decisions, identical parameters."""
    from simpleaicv_pytorch_training_examples_amd import engine
    gpu, cpu = _pair()
    twin = copy.deepcopy(gpu)                      # lives on the GPU, trained by torch's own scaler + SGD
    opt = engine.SGD(gpu, [{'params': list(gpu.parameters())}], lr=0.05, momentum=0.9)
    arena = opt.arena
    ref_opt = torch.optim.SGD(twin.parameters(), lr=0.05, momentum=0.9)
    kw = dict(init_scale=2.0 ** 10, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    mine, ref = engine.GradScaler(device='cuda', **kw), torch.amp.GradScaler('cuda', **kw)
    ref.scale(torch.zeros(1, device='cuda'))       # torch creates its device-side scale lazily
    pattern = [False, False, True, False, False, False, True, True, False, False, False, False]
    for step, bad in enumerate(pattern):
        s = float(ref.get_scale())
        assert mine.get_scale() == s, (step, mine.get_scale(), s)
        _set_grads(gpu, cpu, arena, 600 + step, scale=s, poison=bad)
        for pt, pc in zip(twin.parameters(), cpu.parameters()):
            pt.grad = pc.grad.cuda()
        # torch side: scaler.step() unscales, checks, steps or skips; update() moves the scale
        ref.unscale_(ref_opt)
        ref.step(ref_opt)
        ref.update()
        mine.step(opt)
        mine.update()
        torch.cuda.synchronize()
        for (n, a), b in zip(gpu.named_parameters(), twin.parameters()):
            assert rel_err(a, b) < TOL, (step, n)
    assert mine.get_scale() == float(ref.get_scale())
    sd = mine.state_dict()
    assert sd['scale'] == float(ref.get_scale()) and sd['growth_tracker'] == ref.state_dict()['_growth_tracker']
