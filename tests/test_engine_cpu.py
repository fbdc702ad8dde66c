"""Host logic of the data-parallel engine on CPU: flat arenas, bucket construction, gloo
world_size-2 gradient averaging vs a single-process full batch, no_sync accumulation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from simpleaicv_pytorch_training_examples_amd import engine


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(12, 2000), nn.ReLU(), nn.Linear(2000, 300), nn.ReLU(), nn.Linear(300, 5))


def test_arena_repoints_parameters_and_preserves_values():
    m = _toy()
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    arena = engine.FlatArena(list(m.named_parameters()), torch.device('cpu'))
    assert arena.total % engine.ALIGN == 0
    for (n, p), o in zip(m.named_parameters(), arena.offsets):
        assert o % engine.ALIGN == 0
        assert torch.equal(p, ref[n])
        assert p.data_ptr() == arena.flat_param.data_ptr() + 4 * o
        assert p.grad.data_ptr() == arena.flat_grad.data_ptr() + 4 * o
    x = torch.randn(4, 12)
    m(x).sum().backward()
    assert float(arena.flat_grad.abs().sum()) > 0
    arena.zero_grad()
    assert float(arena.flat_grad.abs().sum()) == 0
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys()) or set(ref.keys()) <= set(sd.keys())


def test_arena_keeps_channels_last_strides():
    conv = nn.Conv2d(8, 16, 3)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    w = conv.weight.detach().clone()
    arena = engine.FlatArena(list(conv.named_parameters()), torch.device('cpu'))
    assert conv.weight.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(conv.weight, w)
    assert conv.weight.grad.stride() == conv.weight.stride()


def test_buckets_cover_every_parameter_once():
    m = _toy()
    ddp = engine.DistributedDataParallel(m, bucket_cap_mb=0.05, last_bucket_cap_mb=0.01)
    seen = sorted(i for b in ddp.buckets for i in b['params'])
    assert seen == list(range(len(ddp.arena.params)))
    assert len(ddp.buckets) >= 3
    for b in ddp.buckets:
        lo = min(ddp.arena.offsets[i] for i in b['params'])
        assert b['start'] == lo and b['end'] > b['start']
    # first-registered parameters (ready last in backward) sit in the last, small bucket
    assert 0 in ddp.buckets[-1]['params']
    assert list(ddp.state_dict().keys())[0].startswith('module.')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _toy()
    if rank == 1:                      # ranks start different; the ctor broadcast must fix that
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    ddp = engine.DistributedDataParallel(m, bucket_cap_mb=0.05, last_bucket_cap_mb=0.01)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 12, generator=g)
    y = torch.randn(8, 5, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    # step A: plain synchronous step
    ddp.arena.zero_grad()
    ((ddp(xs) - ys) ** 2).mean().backward()
    ddp.finish_gradient_sync()
    ga = ddp.arena.flat_grad.clone()
    # step B: accumulate one micro-batch under no_sync, then a synced one
    ddp.arena.zero_grad()
    with ddp.no_sync():
        ((ddp(xs) - ys) ** 2).mean().backward()
    local_only = ddp.arena.flat_grad.clone()
    ((ddp(xs * 0.5) - ys) ** 2).mean().backward()
    ddp.finish_gradient_sync()
    gb = ddp.arena.flat_grad.clone()
    q.put((rank, ga.numpy(), gb.numpy(), local_only.numpy(), ddp.arena.flat_param.detach().clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gloo_world2_matches_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [(r,) + tuple(torch.from_numpy(a) for a in rest) for r, *rest in res]
    (_, ga0, gb0, lo0, p0), (_, ga1, gb1, lo1, p1) = res
    assert torch.equal(p0, p1)                       # ctor broadcast made the ranks identical
    assert torch.allclose(ga0, ga1, atol=1e-7) and torch.allclose(gb0, gb1, atol=1e-7)
    assert not torch.allclose(lo0, lo1)              # no_sync really stayed local

    m = _toy()
    arena = engine.FlatArena(list(m.named_parameters()), torch.device('cpu'))
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 12, generator=g)
    y = torch.randn(8, 5, generator=g)
    ((m(x) - y) ** 2).mean().backward()
    assert torch.allclose(arena.flat_grad, ga0, rtol=1e-4, atol=1e-6)    # mean over ranks == full batch
    arena.zero_grad()
    ((m(x) - y) ** 2).mean().backward()
    ((m(x * 0.5) - y) ** 2).mean().backward()
    assert torch.allclose(arena.flat_grad, gb0, rtol=1e-4, atol=1e-6)    # accumulated, averaged once


class _Shared(nn.Module):
    """One layer used three times per forward (like DETR's decoder_norm, reference detection/models/detr.py:263)
    plus a branch that only some ranks / steps take (like SAM's prompt encoders)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.inp = nn.Linear(6, 1100)
        self.shared = nn.Linear(1100, 1100)
        self.rare = nn.Linear(1100, 1100)
        self.out = nn.Linear(1100, 3)

    def forward(self, x, use_rare):
        h = self.inp(x)
        for _ in range(3):
            h = torch.tanh(self.shared(h))
        if use_rare:
            h = h + self.rare(h)
        return self.out(h)


def _worker_shared(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _Shared()
    # one bucket per parameter: a multi-use parameter reduced after its FIRST contribution, or buckets issued in
    # per-rank completion order, would corrupt the result / deadlock
    ddp = engine.DistributedDataParallel(m, bucket_cap_mb=0.001, last_bucket_cap_mb=0.0005)
    assert len(ddp.buckets) >= 6
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 6, generator=g)
    xs = x[rank * 4:(rank + 1) * 4]
    ddp.arena.zero_grad()
    # rank 0 takes the rare branch, rank 1 does not; NO explicit finish_gradient_sync(): the reference loop
    # (tools/scripts.py:183-226) goes from backward() straight to the optimizer
    ddp(xs, use_rare=(rank == 0)).pow(2).mean().backward()
    grads = ddp.arena.flat_grad.clone()
    arrived = list(ddp.arena.arrived)
    q.put((rank, grads.numpy(), arrived))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gloo_world2_shared_and_rank_dependent_parameters_reference_loop_shape():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shared, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, a0), (_, g1, a1) = res
    g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
    assert torch.equal(g0, g1)
    # single-process reference: mean over the two half batches of the per-rank losses
    m = _Shared()
    arena = engine.FlatArena(list(m.named_parameters()), torch.device('cpu'))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 6, generator=g)
    (0.5 * m(x[:4], True).pow(2).mean() + 0.5 * m(x[4:], False).pow(2).mean()).backward()
    assert torch.allclose(arena.flat_grad, g0, rtol=1e-4, atol=1e-7)
    names = arena.names
    assert all(a0)                                                     # rank 0 used everything
    assert [n for n, a in zip(names, a1) if not a] == ['rare.weight', 'rare.bias']   # rank 1 never touched `rare`


def _worker_unused(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _Shared()
    m.never = nn.Linear(1100, 1100)            # used by no rank
    ddp = engine.DistributedDataParallel(m, bucket_cap_mb=0.001, last_bucket_cap_mb=0.0005, find_unused_parameters=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 6, generator=g)
    out = []
    for step in range(2):
        ddp.arena.zero_grad()
        if step == 1:
            # a backward that raises half-way (the skip-batch pattern): the next zero_grad() must restore a clean state
            try:
                # the failing node sits at the INPUT: it runs after every parameter reported (and every bucket was launched)
                xin = _Boom.apply(x[rank * 4:(rank + 1) * 4].clone().requires_grad_(True))
                ddp(xin, use_rare=True).sum().backward()
            except RuntimeError:
                pass
            assert ddp._callback_queued                      # autograd did not run the end-of-backward callback
            ddp.arena.zero_grad()
            assert not ddp._callback_queued and ddp._next_bucket == 0 and all(b['count'] == 0 for b in ddp.buckets)
        ddp(x[rank * 4:(rank + 1) * 4], use_rare=(rank == 0)).pow(2).mean().backward()
        mask = ddp.arena.has_grad_mask()
        out.append((ddp.arena.flat_grad.clone().numpy(), list(ddp.arena.arrived), None if mask is None else mask.numpy()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


class _Boom(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError('boom')


@pytest.mark.timeout(300)
def test_gloo_world2_unused_parameters_are_stepped_wherever_any_rank_used_them():
    """with find_unused_parameters a parameter that THIS rank did not use still receives the averaged
    gradient of the ranks that did, so the optimizer's skip mask must be global -- identical on every rank, 1 for `rare`
    (rank 0 used it), 0 only for `never` (nobody did).  Also ADVICE r02 (medium): after a backward that raised, zero_grad()
    restores the per-backward state and the next step synchronises as usual."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unused, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    m = _Shared()
    m.never = nn.Linear(1100, 1100)
    arena = engine.FlatArena(list(m.named_parameters()), torch.device('cpu'))
    off = {n: o // engine.ALIGN for n, o in zip(arena.names, arena.offsets)}
    for step in range(2):
        (g0, a0, m0), (g1, a1, m1) = res[0][1][step], res[1][1][step]
        assert torch.equal(torch.from_numpy(g0), torch.from_numpy(g1))
        assert [n for n, a in zip(arena.names, a1) if not a] == ['rare.weight', 'rare.bias', 'never.weight', 'never.bias']
        assert m0 is not None and m1 is not None and (m0 == m1).all()          # the SAME table on both ranks
        assert m1[off['rare.weight']] == 1 and m1[off['rare.bias']] == 1       # rank 1 steps what rank 0 used
        assert m1[off['never.weight']] == 0 and m1[off['never.bias']] == 0     # nobody used it: torch.optim skips it too
        assert m1[off['shared.weight']] == 1 and m1[off['out.bias']] == 1
    assert float(torch.from_numpy(res[0][1][0][0]).abs().sum()) > 0
    assert torch.allclose(torch.from_numpy(res[0][1][0][0]), torch.from_numpy(res[0][1][1][0]), atol=1e-7)   # step 2 == step 1


def test_arena_tracks_which_parameters_received_gradients():
    m = _Shared()
    arena = engine.FlatArena(list(m.named_parameters()), torch.device('cpu'))
    assert arena.has_grad_mask() is not None and int(arena.has_grad_mask().sum()) == 0
    m(torch.randn(2, 6), False).sum().backward()
    mask = arena.has_grad_mask()
    off = {n: o // engine.ALIGN for n, o in zip(arena.names, arena.offsets)}
    assert mask[off['rare.weight']] == 0 and mask[off['rare.bias']] == 0
    assert mask[off['shared.weight']] == 1 and mask[off['out.bias']] == 1
    m(torch.randn(2, 6), True).sum().backward()                        # accumulation: now everything has a gradient
    assert arena.has_grad_mask() is None
    arena.zero_grad()
    assert int(arena.has_grad_mask().sum()) == 0


def _worker_id_exchange(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from simpleaicv_pytorch_training_examples_amd import engine
    # every byte value incl. NUL: the store must carry the id as binary, not as text
    first = engine.NativeComm.exchange_id(world, rank, lambda: bytes(range(128)))
    second = engine.NativeComm.exchange_id(world, rank, lambda: bytes((255 - i) % 256 for i in range(128)))
    q.put((rank, first, second))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_unique_id_travels_through_the_process_group_store():
    """The communicator bootstrap of the N-rank path (engine.NativeComm.exchange_id): rank 0 publishes 128 binary bytes
    in the torch.distributed store, every rank ends up with the same 128 bytes, a second communicator gets a fresh key."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_id_exchange, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == bytes(range(128))
    assert res[0][2] == res[1][2] == bytes((255 - i) % 256 for i in range(128))


# ------------------------------------------------------------------ leaving the captured DETR step is a collective decision
def _worker_agree(rank, world, port, q):
    import types
    from simpleaicv_pytorch_training_examples_amd.tools import scripts
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    config = types.SimpleNamespace(gpus_num=world, group=None)
    # iteration 1: every rank's batch fits the captured shapes; 2: rank 1's does not (an image with more boxes than max_annots); 3: rank 0's
    fits = [(True, True), (True, False), (False, True), (True, True)]
    out = [scripts.all_ranks_agree(f[rank], config) for f in fits]
    q.put((rank, out, config._saicv_host_group is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_every_rank_leaves_the_captured_step_when_one_rank_must():
    """tools/scripts.py train_detection -- a DETR batch beyond config.max_annots takes the eager step; under DDP the ranks
    must take it TOGETHER (the eager and the captured step issue different collective sequences).  all_ranks_agree: a MIN over a gloo
    group, here the default group itself (world of two)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_agree, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == [True, False, False, True], res


def test_one_rank_decides_alone():
    import types
    from simpleaicv_pytorch_training_examples_amd.tools import scripts
    assert scripts.all_ranks_agree(True, types.SimpleNamespace(gpus_num=1)) is True
    assert scripts.all_ranks_agree(False, types.SimpleNamespace(gpus_num=1)) is False
