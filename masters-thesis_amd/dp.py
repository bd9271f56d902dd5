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


class PipelinedDenseSync:
    """DP schedule for the dense-encoder NIC (config 2), shaped for xGMI point-to-point links:

      graph A : forward + loss + vocabulary-head backward
         -> async all-reduce of the head gradients (10 MB), async all-gather of the betas X (5 MB/rank)
      graph B : BPTT, LSTM / embedding / BatchNorm gradients, dpre          (runs while A's collectives fly)
         -> async all-reduce of the middle gradient slice (18 MB) + sparse-norm scalar, all-gather of dpre
      graph C : encoder dW = X_all^T dpre_all on every rank (K = G*B), norms, clip + Adam

    The 41 MB encoder-kernel gradient -- 59 % of the arena and the LAST gradient backward produces --
    is never reduced: its operands are gathered instead (8x less traffic, and X's transfer hides under
    the whole step), and each rank computes the identical global-batch gradient.  Collectives are
    issued with async_op on the compute stream: RCCL runs them on its own stream after the work already
    queued, and ``wait()`` only makes the compute stream wait before graph C.
    """
    pipelined = True

    def __init__(self, world):
        self.world = world
        self._bufs = {}

    def _gather(self, out, t):
        if dist.get_backend() == "gloo":
            w = dist.all_gather(list(out.view(self.world, -1).unbind(0)), t.reshape(-1), async_op=True)
        else:
            w = dist.all_gather_into_tensor(out, t, async_op=True)
        return w

    def step(self, m, B, T):
        a = m.arena
        G = self.world
        key = (B, T)
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros(G * B, m.ldx, dtype=torch.float32, device=m.device),
                               torch.zeros(G * B, m.E, dtype=torch.float32, device=m.device))
        x_all, dpre_all = self._bufs[key]
        e = a.entries
        mid0 = e["dense_img/bias"].off
        head0 = e["time_distributed_softmax/kernel"].off
        m._run_captured(("dpA", B, T), lambda: (m._forward(B, T, True), m._loss_metrics(B, T, True), m._bwd_head(B, T)))
        x_used = m.xd if m.r_in > 0 else m.x
        works = [dist.all_reduce(a.grad[head0:], op=dist.ReduceOp.SUM, async_op=True), self._gather(x_all, x_used)]
        m._run_captured(("dpB", B, T), lambda: m._bwd_seq(B, T))
        works += [dist.all_reduce(a.grad[mid0:head0], op=dist.ReduceOp.SUM, async_op=True),
                  dist.all_reduce(a.sq_override, op=dist.ReduceOp.SUM, async_op=True), self._gather(dpre_all, m.dpre)]
        for w in works:
            w.wait()
        m._run_captured(("dpC", B, T), lambda: (m._bwd_enc(B, T, x_all, dpre_all), m._update_graph()))

    def __call__(self, model):          # generic fallback (models without a pipelined schedule)
        make_grad_sync(self.world)(model)


def attach(model, world=None, bucket_elems=None, pipelined=None):
    """Make ``model`` data-parallel over the default process group.  The dense-encoder NIC gets the
    pipelined schedule (PipelinedDenseSync) unless pipelined=False."""
    world = dist.get_world_size() if world is None else world
    model.dp_world = world
    from .nic import NIC as DenseNIC
    if pipelined is None:
        pipelined = isinstance(model, DenseNIC)
    model.grad_sync = PipelinedDenseSync(world) if (pipelined and isinstance(model, DenseNIC)) else make_grad_sync(world, bucket_elems)
    model._graphs = {}
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
