"""the product modules under the reference's own multi-GPU wrappers.

The reference trainer wraps the model in nn.DataParallel (/root/reference/train_CLC.py:74-79, 472-473: CustomDataParallel) and its
recipe launches DistributedDataParallel processes (/root/reference/run_ddp.sh:7).  A one-GPU box can only place one replica, so
what is asserted here is that wrapping changes nothing: same loss terms, same gradients, bit for bit, as the bare module — and that
the deferred machinery of clc_amd.ops (per-device state) is idle / consistent when no TrainEngine drives the step.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model_and_batch(dev, R=1, B=2):
    from clc_amd import models as pm
    from clc_amd.recipe import apply_weight_recipe, synthetic_image

    m = pm.CLC(N=64, num_ref_frames=R)
    apply_weight_recipe(m, 0)
    m = m.to(dev).eval()   # eval: deterministic rounding, so two passes are comparable bit for bit
    x = synthetic_image(B, 256, 256, 100, smooth=True).to(dev)
    refs = [synthetic_image(B, 256, 256, 101 + i, smooth=True).to(dev) for i in range(R)]
    return m, x, refs


def _loss_and_grads(module, model, x, refs):
    from clc_amd import ops
    from clc_amd.train import RateDistortionLoss

    model.zero_grad(set_to_none=True)
    out = RateDistortionLoss(0.0067)(module(x, refs), x)
    out["loss"].backward()
    ops.join_side_streams()
    torch.cuda.synchronize()
    return ({k: v.item() for k, v in out.items()},
            {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})


def test_dataparallel_single_device_equals_bare_module(dev):
    """The reference's CustomDataParallel forwards unknown attributes to .module (train_CLC.py:74-79): aux_loss / update are reached
    the same way here."""
    m, x, refs = _model_and_batch(dev)
    l0, g0 = _loss_and_grads(m, m, x, refs)

    class CustomDataParallel(torch.nn.DataParallel):
        def __getattr__(self, key):
            try:
                return super().__getattr__(key)
            except AttributeError:
                return getattr(self.module, key)

    dp = CustomDataParallel(m, device_ids=[0])
    l1, g1 = _loss_and_grads(dp, m, x, refs)
    assert l0 == l1, (l0, l1)
    assert g0.keys() == g1.keys() and len(g0) > 1000
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    assert torch.isfinite(dp.aux_loss()).item()
    # state_dict keys carry the "module." prefix the reference strips when it resumes (train_CLC.py:458-464)
    assert all(k.startswith("module.") for k in dp.state_dict())


def test_distributed_data_parallel_single_rank_equals_bare_module(dev):
    """Stock DDP (1-rank gloo group, find_unused_parameters=True because the cc_* / lrp_* twins and the dormant CLM modules never
    receive gradients, SURVEY 7 'DDP with dormant parameters'): gradients equal the bare module's, dormant ones stay None."""
    import torch.distributed as dist

    m, x, refs = _model_and_batch(dev)
    l0, g0 = _loss_and_grads(m, m, x, refs)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    import warnings

    try:
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            l1, g1 = _loss_and_grads(ddp, m, x, refs)
        # DDP's gradient layout contract: every gradient arrives with the strides of its parameter's bucket view (a 1x1 filter's
        # gradient used to come out with channels_last strides, [Co, 1, Co, Co], next to a bucket view of [Ci, 1, 1, 1])
        bad = [str(w.message)[:200] for w in caught if "strides do not match" in str(w.message)]
        assert not bad, bad
    finally:
        if created:
            dist.destroy_process_group()
    assert l0 == l1, (l0, l1)
    assert g0.keys() == g1.keys()
    worst = 0.0
    for n in g0:   # DDP's reducer copies gradients through its buckets (and divides by world size 1): values unchanged
        worst = max(worst, (g0[n] - g1[n]).abs().max().item())
    assert worst == 0.0, worst
    dormant = [n for n, p in m.named_parameters() if p.grad is None]
    assert any(n.startswith("cc_mean_transforms") for n in dormant) and any(n.startswith("feature_alignment") for n in dormant)


def test_ops_state_is_per_device_and_idle_outside_the_engine(dev):
    """clc_amd.ops keeps its queues per device (nn.DataParallel runs one thread per GPU in one process); a plain autograd step
    leaves nothing queued, and release_workspaces() drops the 642 MiB stream-K workspace."""
    from clc_amd import ops

    m, x, refs = _model_and_batch(dev, B=1)
    _loss_and_grads(m, m, x, refs)
    S = ops._S()
    assert not S.pending and not S.pending_post and not S.pending_reduce and not S.keepalive
    assert ops._S() is S and list(ops._STATES) == [torch.cuda.current_device()]
    ops.release_workspaces()
    assert not ops._STATES


def test_group_workspace_is_one_buffer_per_device(dev):
    """the stream-K workspace used to be allocated once per (device, stream): warm-up stream, capture stream, default
    stream ... each 642 MiB, never freed.  Now one per device, handed from stream to stream with a wait."""
    from clc_amd import ops

    ops.release_workspaces()
    p0, n0 = ops._group_ws()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        p1, n1 = ops._group_ws()
    p2, _ = ops._group_ws()
    assert p0 == p1 == p2 and n0 == n1 == ops._L().clc_conv2d_wgrad_group_workspace_bytes() // 4 * 4
    assert list(ops._S().group_ws) == [None]
    ops.release_workspaces()
