"""Data-parallel training: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI; "gloo" on CPU for tests.

The reference has no multi-GPU code; the contract here is
"G ranks x local batch B  ==  one rank on the concatenated batch G*B" for everything except
BatchNorm batch statistics, which stay per replica.  Mechanics:
  * every rank scales its loss gradient by 1/(G*B*T) in the fused softmax/CE kernel
    (``model.dp_world``), so a SUM all-reduce of the flat gradient arena IS the mean over
    the global batch -- no extra averaging pass over 70 MB;
  * the Embedding IndexedSlices norm (sum of squares of un-merged rows) is a
    scalar per rank and is summed with the same collective call pattern;
  * L2 terms and per-variable clipping are applied after the all-reduce by the optimizer
    kernels, identically on every rank, so replicas stay bit-identical.
"""
import torch
import torch.distributed as dist


def make_grad_sync(world, bucket_elems=None):
    """Returns grad_sync(model): all-reduce of model.arena.grad (+ the sparse-norm slots).
    ``bucket_elems`` splits the arena into several collectives (reverse order: the decoder
    gradients are produced first, the encoder kernel last)."""
    def sync(model):
        g = model.arena.grad
        if bucket_elems is None or bucket_elems >= g.numel():
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        else:
            n = g.numel()
            hi = n
            while hi > 0:
                lo = max(0, hi - bucket_elems)
                dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM)
                hi = lo
        dist.all_reduce(model.arena.sq_override, op=dist.ReduceOp.SUM)
    sync.world = world
    return sync


def attach(model, world=None, bucket_elems=None):
    """Make ``model`` data-parallel over the default process group."""
    world = dist.get_world_size() if world is None else world
    model.dp_world = world
    model.grad_sync = make_grad_sync(world, bucket_elems)
    broadcast_parameters(model)
    return model


def broadcast_parameters(model, src=0):
    """Start every replica from rank 0's parameters and moving statistics."""
    dist.broadcast(model.arena.theta, src)
    for t in model.state_tensors():
        dist.broadcast(t, src)


def allreduce_metrics(metrics):
    """Mean of the logged scalars over ranks (logging only)."""
    world = dist.get_world_size()
    keys = sorted(metrics)
    t = torch.stack([metrics[k].float() for k in keys])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: t[i] / world for i, k in enumerate(keys)}
