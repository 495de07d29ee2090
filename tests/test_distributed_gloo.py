"""world_size-2 gloo tests (CPU) of the N>1 logic: data-parallel sharding of the hot path and the
one-broadcast mask synchronisation that replaces the reference's per-step, per-tensor broadcasts
(sparse_core.py:404-407).  The arithmetic here is done by the oracle -- these tests pin the
DISTRIBUTED LOGIC (what is sharded, what is reduced, what is broadcast), not the kernels."""
import os
import socket
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _sync_masks(rank, world):
    from slak_amd.sparse_core import Masking
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=None, only_L=False, sparse_init="uniform", sparsity=0.4, distributed=True)
    mk = Masking(None, None, None, prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none", args=args)
    torch.manual_seed(100 + rank)                                  # per-rank RNG, as main.py:232
    mk.masks = {"a": (torch.rand(7, 1, 51, 5) < 0.6).float(), "b": (torch.rand(33, 17) < 0.6).float()}
    before = {k: v.clone() for k, v in mk.masks.items()}
    mk.synchronism_masks()                                         # ONE coalesced broadcast from rank 0
    after1 = {k: v.clone() for k, v in mk.masks.items()}
    mk.masks["a"][0, 0, 0, 0] = 1 - mk.masks["a"][0, 0, 0, 0]      # later local edits are NOT overwritten:
    mk.synchronism_masks()                                         # subsequent calls are free (no collective)
    return dict(before={k: v.numpy() for k, v in before.items()}, after={k: v.numpy() for k, v in after1.items()},
                edited=float(mk.masks["a"][0, 0, 0, 0]), synced=mk._synced_once)


def test_mask_sync_is_one_broadcast_rank0_wins():
    r0, r1 = _run(_sync_masks)
    for k in ("a", "b"):
        assert not np.array_equal(r0["before"][k], r1["before"][k])          # per-rank init differs
        np.testing.assert_array_equal(r0["after"][k], r0["before"][k])       # rank 0 keeps its masks
        np.testing.assert_array_equal(r1["after"][k], r0["before"][k])       # rank 1 adopts rank 0's
    assert r0["synced"] and r1["synced"]
    assert r1["edited"] == 1 - r0["before"]["a"][0, 0, 0, 0]                 # second call did not re-broadcast


def _dp_wgrad(rank, world):
    """bwd-filter is a sum over the batch: each rank's local dw, summed by the (DDP) all-reduce, equals the
    full-batch dw; fwd / bwd-data shard trivially over n."""
    import oracle
    rng = np.random.default_rng(0)
    N, C, H, W, kh, kw = 4, 3, 9, 9, 13, 5
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, 1, kh, kw)) * 0.1).astype(np.float32)
    sl = slice(rank * N // world, (rank + 1) * N // world)
    dw_local = torch.from_numpy(oracle.dwconv2d_bwd_filter(dy[sl], x[sl], kh, kw))
    dist.all_reduce(dw_local)                                      # what DDP's bucketed all-reduce does (sum; DDP then /world)
    y_local = oracle.dwconv2d_fwd(x[sl], w)
    return dict(dw=dw_local.numpy(), dw_full=oracle.dwconv2d_bwd_filter(dy, x, kh, kw), y_ok=np.allclose(y_local, oracle.dwconv2d_fwd(x, w)[sl]))


def test_data_parallel_sharding_of_the_path():
    r0, r1 = _run(_dp_wgrad)
    np.testing.assert_allclose(r0["dw"], r0["dw_full"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(r0["dw"], r1["dw"])              # every rank ends with identical grads
    assert r0["y_ok"] and r1["y_ok"]


def _identical_masks_without_collectives(rank, world):
    """After the one init broadcast, prune/regrow needs NO collective: identical (weights, all-reduced grads,
    masks, rate) on every rank give identical masks.  (Oracle arithmetic; the HIP kernels implement the same
    deterministic function -- tests/test_masking_gpu.py.)"""
    import oracle
    rng = np.random.default_rng(3)                                 # replicated state: same on every rank
    w = {"t": rng.standard_normal((6, 1, 13, 5)).astype(np.float32)}
    m = {"t": (rng.random((6, 1, 13, 5)) < 0.6).astype(np.float32)}
    w["t"] *= m["t"]
    g_local = torch.from_numpy(np.random.default_rng(10 + rank).standard_normal((6, 1, 13, 5)).astype(np.float32))
    dist.all_reduce(g_local); g_local /= world                     # DDP-averaged gradient
    _, nm, _ = oracle.truncate_weights(w, m, {"t": g_local.numpy()}, 0.3)
    return nm["t"]


def test_masks_agree_across_ranks_without_per_step_broadcast():
    r0, r1 = _run(_identical_masks_without_collectives)
    np.testing.assert_array_equal(r0, r1)


def _random_growth_resync(rank, world):
    """growth='random' draws from per-rank host generators (main.py:232), so the masks differ after the growth until they are
    re-broadcast from rank 0: the mirror marks them unsynchronised right after such a growth (the device part -- prune, apply --
    is stubbed here; this pins the DISTRIBUTED LOGIC on CPU tensors with the oracle doing the arithmetic)."""
    import math
    import oracle
    from slak_amd.sparse_core import Masking
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=None, only_L=False, sparse_init="uniform", sparsity=0.4, distributed=True)
    mk = Masking(None, None, None, prune_mode="magnitude", growth_mode="random", redistribution_mode="none", args=args)
    mk.init_growth_prune_and_redist()                              # 'random' is an accepted growth mode
    torch.manual_seed(5)                                           # same weights and masks on both ranks ...
    w = torch.randn(9, 1, 13, 5) * 0.05
    mk.masks = {"w": (torch.rand(w.shape) < 0.6).float()}
    mk._synced_once = True
    torch.manual_seed(100 + rank)                                  # ... but per-rank generators from here on
    m = mk.masks["w"].numpy()
    nz = float(m.sum()); pruned = oracle.magnitude_prune(m, (w * mk.masks["w"]).numpy(), 0.3, nz, m.size - nz)
    removed = nz - float(pruned.sum())
    grown = oracle.random_growth(pruned.astype(np.uint8), math.floor(removed), torch.rand(w.shape).numpy())
    mk.masks["w"].copy_(torch.from_numpy(grown))
    local = mk.masks["w"].clone()
    mk._synced_once = False                                        # what _truncate_weights_random_growth does after growing
    mk.synchronism_masks()
    return dict(local=local.numpy(), synced=mk.masks["w"].numpy())


def test_random_growth_masks_are_rebroadcast_from_rank0():
    r0, r1 = _run(_random_growth_resync)
    assert not np.array_equal(r0["local"], r1["local"])                     # ranks grew different weights
    np.testing.assert_array_equal(r0["synced"], r0["local"])                # rank 0's masks win (sparse_core.py:404-407)
    np.testing.assert_array_equal(r1["synced"], r0["local"])


def _checkpoint_main_process_only(rank, world):
    import tempfile
    from slak_amd import checkpoint
    d = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(d, src=0)
    args = types.SimpleNamespace(output_dir=d[0], save_ckpt_num=2, save_ckpt_freq=1, resume='', auto_resume=True)
    torch.manual_seed(rank)                                        # ranks hold DIFFERENT weights here: the file must be rank 0's
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    checkpoint.save_model(args, 0, net, net, opt, None)
    dist.barrier()
    files = sorted(os.listdir(d[0]))
    net2 = torch.nn.Linear(3, 2)
    checkpoint.auto_load_model(args, net2, net2, opt, None)
    return dict(files=files, mine=net.weight.detach().numpy().copy(), loaded=net2.weight.detach().numpy().copy())


def test_checkpoint_is_written_by_the_main_process():
    r0, r1 = _run(_checkpoint_main_process_only)
    assert r0["files"] == r1["files"] == ["checkpoint-0.pth"]
    np.testing.assert_array_equal(r0["loaded"], r0["mine"])                 # both ranks resume from rank 0's weights
    np.testing.assert_array_equal(r1["loaded"], r0["mine"])
    assert not np.array_equal(r1["mine"], r0["mine"])


def _rank_divergence_heals(rank, world):
    """A rank-local perturbation after the one broadcast (here: rank 1 flips mask bits) is caught by the checksum all-reduce that
    follows every prune-and-grow round, and healed the way the reference heals every step: rank 0's masks are re-broadcast
    (sparse_core.py:404-407).  The device checksum kernel is replaced by a host hash (no GPU here); the control flow is the product's."""
    from slak_amd.sparse_core import Masking
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=None, only_L=False, sparse_init="uniform", sparsity=0.4, distributed=True)
    mk = Masking(None, None, None, prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none", args=args)
    torch.manual_seed(7)
    mk.masks = {"a": (torch.rand(7, 1, 51, 5) < 0.6).float(), "b": (torch.rand(33, 17) < 0.6).float()}
    mk.synchronism_masks()
    mk.mask_checksum = lambda: int(sum(int((m.reshape(-1) * torch.arange(1, m.numel() + 1)).sum().item()) for m in mk.masks.values())) & ((1 << 63) - 1)
    mk._verify_ranks()                                             # ranks agree: nothing happens
    first = mk.rank_resyncs
    ref = {k: v.clone() for k, v in mk.masks.items()}
    if rank == 1:
        mk.masks["a"][3, 0, 10:20, :] = 1 - mk.masks["a"][3, 0, 10:20, :]
    mk._verify_ranks()                                             # mismatch: rank 0's masks come back
    healed = all(torch.equal(mk.masks[k], ref[k]) for k in ref)
    strict = Masking(None, None, None, prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none",
                     args=types.SimpleNamespace(**dict(vars(args), debug_mask_sync=True)))
    strict.masks = {k: v.clone() for k, v in ref.items()}
    strict._synced_once = True
    strict.mask_checksum = lambda: (5 + rank)
    try:
        strict._verify_ranks(); raised = False
    except RuntimeError:
        raised = True
    return dict(first=first, resyncs=mk.rank_resyncs, healed=healed, raised=raised)


def test_rank_divergence_is_detected_and_healed_by_default():
    r0, r1 = _run(_rank_divergence_heals)
    for r in (r0, r1):
        assert r["first"] == 0 and r["resyncs"] == 1 and r["healed"] and r["raised"]


def _bucket_view_grads(rank, world):
    """The mechanism of block_ops.adopt_grad_slots / _grad_destinations on torch's own reducer (round 6): an autograd node that writes a parameter
    gradient INTO the tensor a previous backward left in .grad (a DDP bucket view) and returns a new tensor object on that storage.  Pins what the
    C++ block runner relies on: AccumulateGrad installs the returned tensor without a copy, DistributedDataParallel(gradient_as_bucket_view=True)
    finds it aliasing its bucket view and leaves it alone, the averaged gradients and the weights equal the ordinary path's bit for bit, a
    parameter whose .grad is still set (gradient accumulation) gets no destination, and a bucket rebuild heals by itself."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from slak_amd import block_ops

    class Scale(torch.autograd.Function):                             # y = x * w (w: [F]); stands in for a block: dw goes where the runner would put it
        wrote_in_place = 0

        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            ctx.param = w
            return x * w

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            dst = block_ops._grad_destinations((ctx.param,))
            dw = (dy * x).sum(0)
            if dst and dst[0] is not None:
                dst[0].copy_(dw)                                        # "the kernel writes into the destination"
                Scale.wrote_in_place += 1
                dw = dst[0].detach()                                    # a NEW tensor object on the same storage (GradDst::take)
            return dy * w, dw

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w1 = torch.nn.Parameter(torch.linspace(0.5, 1.5, 12))
            self.lin = torch.nn.Linear(12, 12)
            self.w2 = torch.nn.Parameter(torch.linspace(1.0, 2.0, 12))

        def forward(self, x):
            return Scale.apply(self.lin(Scale.apply(x, self.w1)), self.w2)

    def train(slots, accumulate=False):
        torch.manual_seed(0)
        net = Net()
        ddp = torch.nn.parallel.DistributedDataParallel(net, gradient_as_bucket_view=True, broadcast_buffers=False)
        ddp.register_comm_hook(None, default_hooks.allreduce_hook)
        block_ops.drop_grad_slots()
        block_ops.grad_slots_enabled = slots
        Scale.wrote_in_place = 0
        opt = torch.optim.SGD(ddp.parameters(), lr=0.05)
        g = torch.Generator().manual_seed(7 + rank)
        aliased = []
        for it in range(6):
            x = torch.randn(5, 12, generator=g)
            if accumulate and it % 2 == 0:
                with ddp.no_sync():
                    ddp(x).square().mean().backward()                  # leaves .grad set: the next backward must ACCUMULATE, not overwrite
                x = torch.randn(5, 12, generator=g)
            before = Scale.wrote_in_place
            ddp(x).square().mean().backward()
            if slots:
                e = [block_ops._grad_slots.get(id(p)) for p in (net.w1, net.w2)]
                aliased.append((Scale.wrote_in_place - before,
                                all(q is not None and p.grad.data_ptr() == q[1].data_ptr() for p, q in zip((net.w1, net.w2), e))))
                block_ops.adopt_grad_slots(list(ddp.parameters()))      # what MaskedAdamW.step does
            opt.step()
            opt.zero_grad(set_to_none=True)
        block_ops.grad_slots_enabled = False
        block_ops.drop_grad_slots()
        return [p.detach().clone().numpy() for p in net.parameters()], aliased

    ref, _ = train(False)
    got, aliased = train(True)
    ref_acc, _ = train(False, accumulate=True)
    got_acc, aliased_acc = train(True, accumulate=True)
    return dict(ref=ref, got=got, aliased=aliased, ref_acc=ref_acc, got_acc=got_acc, aliased_acc=aliased_acc)


def test_gradients_written_into_ddp_bucket_views_match_the_copy_path():
    r0, r1 = _run(_bucket_view_grads)
    for r in (r0, r1):
        for a, b in zip(r["ref"], r["got"]):
            np.testing.assert_array_equal(a, b)                      # bit-identical weights after six steps
        for a, b in zip(r["ref_acc"], r["got_acc"]):
            np.testing.assert_array_equal(a, b)
        # iteration 0: nothing adopted yet; iteration 1: the views of the first bucket layout (DDP rebuilds its buckets once, so the reducer may
        # copy once more); from iteration 2 on both gradients are written in place and .grad IS the destination after the backward
        assert r["aliased"][0][0] == 0
        assert all(n == 2 and same for n, same in r["aliased"][2:]), r["aliased"]
        # gradient accumulation: the synchronising backward of an accumulating pair finds .grad set -> no destination handed out there
        assert all(n == 0 for n, _ in r["aliased_acc"][0::2]), r["aliased_acc"]
    for a, b in zip(r0["got"], r1["got"]):
        np.testing.assert_array_equal(a, b)                          # the ranks agree


def _exchange_from_cpp(rank, world):
    """The C++ block runner issues the SyncBatchNorm statistics all-reduce itself when it is handed the ProcessGroup (block_ops._runner_exchange):
    the pybind cast of torch's ProcessGroup and the SUM, synchronous and asynchronous, against the callable form."""
    from slak_amd import block_ops
    R = block_ops._runner()
    if R is None:
        return None
    pg = dist.group.WORLD
    a = torch.arange(8, dtype=torch.float64) + 10 * rank
    block_ops._pg_direct = True                                    # (opt-in: SLAK_PG_DIRECT=1)
    direct = R._exchange_probe(block_ops._runner_exchange(pg), a, False)
    block_ops._pg_direct = False
    assert callable(block_ops._runner_exchange(pg))
    b = torch.arange(8, dtype=torch.float32) * (rank + 1)
    direct2 = R._exchange_probe(pg, b, True)
    c = torch.ones(4) * (rank + 1)
    via_py = R._exchange_probe(lambda buf, async_op: block_ops._sync_bn_all_reduce(buf, pg, async_op=async_op), c, True)
    return dict(direct=direct, direct2=direct2, via_py=via_py, a=a.numpy(), b=b.numpy(), c=c.numpy())


def test_runner_issues_the_statistics_all_reduce_from_cpp():
    import pytest
    r0, r1 = _run(_exchange_from_cpp)
    if r0 is None:
        pytest.skip("the C++ block runner is not built")
    for r in (r0, r1):
        assert r["direct"] and r["direct2"] and not r["via_py"]
        np.testing.assert_array_equal(r["a"], 2 * np.arange(8) + 10.0)
        np.testing.assert_array_equal(r["b"], 3.0 * np.arange(8))
        np.testing.assert_array_equal(r["c"], 3.0 * np.ones(4))
