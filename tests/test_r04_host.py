"""Round-4 host-side logic (CPU): the relative-position table resampling against torch's own linear interpolation, and the
ConvBnActBlock switches' parameter contract against the reference-generated fixture (no kernels run here)."""
import torch
import torch.nn.functional as F

from conftest import load_golden


def test_resize_rel_pos_equals_linear_interpolation_and_its_gradient():
    from simpleaicv_pytorch_training_examples_amd import ops_tfm
    g = torch.Generator().manual_seed(0)
    for src, dst in [(27, 127), (127, 27), (27, 63), (63, 64), (13, 127), (15, 31)]:
        t = torch.randn(src, 64, generator=g, requires_grad=True)
        ref = F.interpolate(t.reshape(1, src, -1).permute(0, 2, 1), size=dst, mode='linear').reshape(-1, dst).permute(1, 0)
        out = ops_tfm.resize_rel_pos(t, dst)
        assert out.shape == (dst, 64) and out.is_contiguous() and out.dtype == torch.float32
        assert float((out - ref).abs().max()) < 1e-5
        probe = torch.randn(dst, 64, generator=g)
        (g_ref,) = torch.autograd.grad(ref, t, probe, retain_graph=True)
        (g_out,) = torch.autograd.grad(out, t, probe)
        assert float((g_out - g_ref).abs().max()) < 1e-4
    t = torch.randn(31, 64)
    assert ops_tfm.resize_rel_pos(t, 31) is t                     # the native grid: the parameter itself, no copy


def test_convbnact_block_switches_keep_the_reference_parameter_contract():
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification.backbones.resnet import ConvBnActBlock
    cases = load_golden('convbnact_variants')['cases']
    for key, c in cases.items():
        blk = ConvBnActBlock(**c['kwargs'])
        sd = blk.state_dict()
        assert list(sd.keys()) == list(c['state_dict'].keys()), key
        for k in sd:
            assert sd[k].shape == c['state_dict'][k].shape, (key, k)
        blk.load_state_dict(c['state_dict'])


def _vote(world, finish_after, seconds=1.0):
    """Drive bench.WatchdogVote with `world` in-process ranks over one HashStore: rank r calls finished() after
    finish_after[r] seconds (None: never).  -> (decisions, acted)."""
    import threading
    import time
    import torch.distributed as dist
    import bench
    store = dist.HashStore()
    acted = []
    lock = threading.Lock()

    def act(secs, why, r):
        with lock:
            acted.append((r, why))

    votes = [bench.WatchdogVote(store, r, world, seconds, epoch=7, act=lambda s, w, r=r: act(s, w, r), poll=0.02) for r in range(world)]

    def finisher(r):
        if finish_after[r] is not None:
            time.sleep(finish_after[r])
            votes[r].finished()

    ts = [threading.Thread(target=finisher, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for v in votes:
        v.thread.join(seconds + 10)
        assert not v.thread.is_alive()
    return [v.decision for v in votes], sorted(acted)


def test_watchdog_vote_everyone_finished_nobody_reexecutes():
    decisions, acted = _vote(3, [0.05, 0.1, 0.2])
    assert decisions == ['ok', 'ok', 'ok'] and acted == []


def test_watchdog_vote_one_stalled_rank_sends_every_rank_to_eager():
    decisions, acted = _vote(3, [0.05, None, 0.1])
    assert decisions == ['reexec'] * 3
    assert [r for r, _ in acted] == [0, 1, 2] and all(w == 'collective decision' for _, w in acted)


def test_watchdog_vote_is_one_decision_even_when_a_rank_finishes_just_after_the_deadline():
    """The race ADVICE r03 describes: rank 0 done in time, rank 1 a moment too late.  Per-rank timers would send only rank 1 to
    eager; the vote gives every rank the same answer (here: reexec, because rank 0 published before rank 1's key existed)."""
    decisions, acted = _vote(2, [0.1, 1.3], seconds=1.0)
    assert len(set(decisions)) == 1
    assert (decisions[0] == 'reexec') == (len(acted) == 2)
