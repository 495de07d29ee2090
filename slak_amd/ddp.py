"""DistributedDataParallel set-up for the MI355X path (the reference builds the wrapper in main.py:374-376 and steps it in engine.py:80-88).

`wrap(model, device_ids)` = the reference's constructor call plus three results-neutral settings that remove per-parameter work from the step
(measured on one MI355X over RCCL, profiles/r06_*):

* broadcast_buffers=False -- the only buffers are the (Sync)BatchNorm running statistics and counters, rank-identical by construction;
* gradient_as_bucket_view=True + a per-BUCKET comm hook -- no division launch per parameter (312 per SLaK-T step);
* block_ops.enable_grad_slots_for(ddp) -- the blocks' weight-gradient launches write INTO the reducer's bucket views, so its per-parameter copy
  launch disappears as well (288 of 312 parameters).

`main.py:374-376`'s own call (`--ddp-reference-flags` in bench.py) keeps working: same results, ~1 ms per step more.
"""
import torch
import torch.distributed as dist


def avg_allreduce_hook(process_group, bucket):
    """DDP comm hook: ONE collective per bucket with the division inside it (ncclAvg) -- the stock allreduce_hook launches `buffer.div_(world)` in
    front of its all-reduce.  For a power-of-two world size the results are bit-identical to DDP's own (scaling by 2^-k commutes with every rounding
    of the ring sum)."""
    group = process_group if process_group is not None else dist.group.WORLD
    buf = bucket.buffer()
    fut = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])


def wrap(model, device_ids, reference_flags=False, **kw):
    """-> DistributedDataParallel(model, device_ids=device_ids, find_unused_parameters=False) as main.py:374-376 builds it; unless
    `reference_flags`, with the settings of the module docstring."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if reference_flags:
        return DDP(model, device_ids=device_ids, find_unused_parameters=False, **kw)
    ddp = DDP(model, device_ids=device_ids, find_unused_parameters=False, broadcast_buffers=False, gradient_as_bucket_view=True, **kw)
    world = dist.get_world_size()
    if dist.get_backend() == "nccl" and (world & (world - 1)) == 0:
        ddp.register_comm_hook(None, avg_allreduce_hook)
        ddp._slak_comm_hook = "avg_allreduce_hook (ncclAvg: division inside the collective)"
    else:                                                            # gloo has no AVG; an odd world size would not be bit-identical to DDP's arithmetic
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.allreduce_hook)
        ddp._slak_comm_hook = "allreduce_hook (division per bucket)"
    from . import block_ops
    block_ops.enable_grad_slots_for(ddp)                             # SLAK_GRAD_SLOTS=0: the A/B switch
    return ddp
