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
import warnings

import os

import torch
import torch.distributed as dist

warnings.filterwarnings("ignore", message=".*all_reduce_coalesced.*")


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
        dist.all_reduce(_sparse_norm_slot(model), op=dist.ReduceOp.SUM)
    sync.world = world
    return sync


def _whole_step_graph(sync, m, key, body):
    """ONE hipGraph for the whole data-parallel step, collectives included.  Under capture ProcessGroupNCCL enqueues
    each ``async_op`` collective on its own stream behind an event of the capturing stream -- a forked branch of the
    graph -- and ``work.wait()`` joins the branch, so the captured graph has exactly the overlap structure of the
    segmented schedule without its six launch boundaries (each graph launch costs 15-20 us of device idle time, a
    launch plan ~6 us of host time per kernel).  Tried once per (model, key): if the runtime refuses the capture the
    segmented schedule takes over for good.  OPT-IN (TNT_DP_ONE_GRAPH=1); gloo (CPU tests) always runs segmented.
    Measured at world size 1 over RCCL (tools/dp_rehearsal.py, profiles/r02_dp_onegraph_world1_trace.txt): the captured
    graph replays correctly (bit-identical weights for the dense model), but as soon as a collective puts a second
    stream into the graph (at world size 1: the all-gathers, which become copy kernels) every node behind the fork
    runs in the runtime's slower cross-stream dependency mode -- small kernels 4.7 -> 11 us each -- and the dense
    step is 0.815 ms against 0.763 for the segmented schedule and 0.641 for the single-GPU graph.  The attention
    schedule (all-reduces only, which launch nothing at world size 1) replays at 1.135 ms against 1.309 segmented
    and 1.110 single-GPU, but that says nothing about world sizes where the all-reduce is a real kernel."""
    if sync.one_graph is False or m.device.type != "cuda" or not m.use_graph or dist.get_backend() != "nccl":
        return False
    st = m._graphs.get(key)
    if st is None:                       # first call: eager warm-up through the segmented path (creates buffers, comms)
        m._graphs[key] = "pending"
        return False
    if st == "pending":
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
        except Exception as ex:          # noqa: BLE001 -- any capture failure means: not supported here
            sync.one_graph = False
            m._graphs.pop(key, None)
            sync.capture_error = repr(ex)
            warnings.warn(f"TNT_DP_ONE_GRAPH: capture of the whole data-parallel step failed ({sync.capture_error}); "
                          "the segmented schedule is used from here on")
            torch.cuda.synchronize()
            return False
        m._graphs[key] = g
        g.replay()
        return True
    st.replay()
    return True


class _IssueBehind:
    """Host-side issue order of the pipelined schedules.  A collective must start when the segment that PRODUCES its operand
    has finished, but the host should not sit in torch.distributed (20-40 us per call) while the GPU runs dry: the next
    compute segment is enqueued FIRST, then the collective is issued from a side stream that waits only for the producing
    segment's event (ProcessGroupNCCL orders its communication stream behind the CURRENT stream at call time).  Device-side
    order is unchanged; what disappears is the 10-30 us of device idle time at every segment boundary
    (tools/trace_step.py of the world-size-1 rehearsal: 131 us of gaps in a 729 us step)."""

    def __init__(self, m):
        self.on = m.device.type == "cuda"
        self.side = torch.cuda.Stream(device=m.device) if self.on else None
        self.events = [torch.cuda.Event() for _ in range(4)] if self.on else []

    def mark(self, i):
        """event i := everything enqueued on the compute stream so far"""
        if self.on:
            self.events[i].record()
        return i

    def behind(self, i, fn):
        """issue fn()'s collectives so that they start behind event i (not behind what was enqueued after it)"""
        if not self.on:
            return fn()
        self.side.wait_event(self.events[i])
        with torch.cuda.stream(self.side):
            return fn()


class PipelinedDenseSync:
    """DP schedule for the dense-encoder NIC (config 2), shaped for xGMI point-to-point links.
    Six launch segments (hipGraphs or recorded launch plans, see ``eager`` below); every collective is issued async
    right after the segment that produces its operand and is waited for only by the segment that consumes it:

      A  : forward + loss + vocabulary-head backward
             -> all-reduce head gradients (10 MB), all-gather the betas X (5 MB / rank)
      B1 : BPTT + LSTM kernel / recurrent-kernel / bias gradients
             -> all-reduce LSTM gradients (8 MB)
      B2 : dXin, embedding rows, BatchNorm / activation backward, encoder bias, dpre
             -> all-reduce {encoder bias, BN, embedding} (10 MB) + sparse-norm scalar, all-gather dpre
      C0 : (waits head)            step tick; norms + clip + Adam of the head -- runs while the all-gather of dpre,
                                   the one small collective on the critical path, is in flight
      C1 : (waits X, dpre)         encoder dW = X_all^T dpre_all on every rank (K = G*B);
           norms + clip + Adam of the encoder kernel  (59 % of the arena)
      C2 : (waits LSTM, embedding) norms + clip + Adam of the rest, L2 metric

    The 41 MB encoder-kernel gradient -- 59 % of the arena and the LAST gradient backward produces --
    is never reduced: its operands are gathered instead (8x less traffic, and X's transfer hides under
    the whole step), and each rank computes the identical global-batch gradient.  The head all-reduce
    hides under B1+B2, the LSTM one under B2+C1, and the last bucket under C1, so at 8 ranks only the
    tail of that last 10 MB bucket is exposed.  RCCL runs the collectives on its own stream after the
    work already queued on the compute stream; ``wait()`` makes the compute stream wait, not the host.
    """
    pipelined = True
    # Segments replayed as recorded launch plans (ModelBase._run_planned: plain kernel launches re-issued from
    # bound C-ABI calls) instead of hipGraphs.  Every hipGraphLaunch costs ~15-20 us of device idle time before
    # its first kernel, and a step has six segments; only B1, the chain of 15 dependent LSTM steps, keeps its
    # graph.  World-size-1 rehearsal (tools/host_overhead.py --dp): all graphs 0.905 ms/step, this choice 0.854
    # with the host at 0.52 ms/step, all plans 0.847 with the host at 0.69.  The single-GPU step stays ONE graph
    # (0.7195 ms vs 0.7142 as a plan: within noise, and the plan keeps the host 80 % busy).
    eager = frozenset(os.environ.get("TNT_DP_EAGER", "A,B2,C,C0,C1,C2,S1,S2").split(","))

    def __init__(self, world, shard_encoder=None, rank=None):
        self.world = world
        self._bufs = {}
        self._slices = None
        # row-sharded update of the encoder kernel (opt-in: dp.attach(shard_encoder=True) / TNT_DP_SHARD_ENC=1), see step()
        self.shard_encoder = (os.environ.get("TNT_DP_SHARD_ENC", "0") == "1") if shard_encoder is None else bool(shard_encoder)
        self.rank = rank
        self.dpre_ride = os.environ.get("TNT_DP_DPRE_RIDE", "1") == "1"
        self.split_update = os.environ.get("TNT_DP_SPLIT_UPDATE", "0") == "1"
        self.one_graph = os.environ.get("TNT_DP_ONE_GRAPH", "0") == "1"
        self.capture_error = None

    def _gather(self, out, t):
        if dist.get_backend() == "gloo":
            w = dist.all_gather(list(out.view(self.world, -1).unbind(0)), t.reshape(-1), async_op=True)
        else:
            w = dist.all_gather_into_tensor(out, t, async_op=True)
        return w

    @staticmethod
    def _ar(t):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def step(self, m, B, T):
        a = m.arena
        G = self.world
        key = (B, T)
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros(G * B, m.ldx, dtype=torch.float32, device=m.device),
                               torch.zeros(G * B, m.E, dtype=torch.float32, device=m.device))
        x_all, dpre_all = self._bufs[key]
        # The gathered dpre is the LAST operand the update waits for, behind the last gradient bucket on RCCL's stream: two
        # collectives in a row, each a launch latency of its own.  It rides in that bucket's all-reduce instead (dpre_ride): the
        # model writes its dpre straight into ITS block of dpre_all, the other blocks are zero, and a SUM over the ranks is
        # the gather, bit for bit (x + 0 + ... + 0); G B E extra floats in a 10 MB launch.  The blocks of the other ranks are
        # re-zeroed at the top of the next step (they hold this step's result until then).
        ride = self.dpre_ride and not self.split_update
        if ride:
            rank = dist.get_rank() if self.rank is None else self.rank
            mine = dpre_all[rank * B:(rank + 1) * B]
            if m.dpre.data_ptr() != mine.data_ptr():
                m.dpre = mine                     # (captured segments bake the pointer in: drop them once)
                m._graphs = {}
            if rank > 0:
                dpre_all[:rank * B].zero_()
            if rank < G - 1:
                dpre_all[(rank + 1) * B:].zero_()
        e = a.entries
        if self._slices is None:
            sg = {k: v.seg for k, v in e.items()}
            self._slices = (a.seg_slice(0, 1),                                                     # encoder kernel
                            a.seg_slice(sg["time_distributed_softmax/kernel"], a.nseg),            # head
                            a.seg_slice(1, sg["time_distributed_softmax/kernel"]))                 # the rest
        s_enc, s_head, s_mid = self._slices
        front0, lstm0, head0 = e["dense_img/bias"].off, e["lstm/kernel"].off, e["time_distributed_softmax/kernel"].off
        x_used = m.xd if m.r_in > 0 else m.x

        def body():
            m._forward(B, T, True); m._loss_metrics(B, T, True); m._bwd_head(B, T)
            w_head, w_x = self._ar(a.grad[head0:]), self._gather(x_all, x_used)
            m._bwd_seq_lstm(B, T); m.join()
            w_lstm = self._ar(a.grad[lstm0:head0])
            m._bwd_seq_front(B, T); m.join()
            w_front = dist.all_reduce_coalesced([a.grad[front0:lstm0], _sparse_norm_slot(m)], op=dist.ReduceOp.SUM, async_op=True)
            w_dpre = self._gather(dpre_all, m.dpre)
            w_head.wait()
            m._tick(); m._update_slice(s_head)
            w_x.wait(); w_dpre.wait()
            m._bwd_enc(B, T, x_all, dpre_all); m._update_slice(s_enc)
            w_lstm.wait(); w_front.wait()
            m._update_slice(s_mid); m.be.l2_total(a.wsq, a.seg_l2, a.nseg, m.met[2:3])
        if _whole_step_graph(self, m, ("dpall", B, T), body):
            return

        cap = lambda key, fn: (m._run_planned if key[0][2:] in self.eager else m._run_captured)(key, fn)
        ib = self.__dict__.get("_ib") or self.__dict__.setdefault("_ib", _IssueBehind(m))
        cap(("dpA", B, T), lambda: _segment_a(m, B, T, defer_totals=not self.split_update))
        eA = ib.mark(0)
        cap(("dpB1", B, T), lambda: (m._bwd_seq_lstm(B, T), m.join()))          # enqueued before the host turns to the collectives
        eB1 = ib.mark(1)
        w_head, w_x = ib.behind(eA, lambda: (self._ar(a.grad[head0:]), self._gather(x_all, x_used)))
        cap(("dpB2", B, T), lambda: (m._bwd_seq_front(B, T), m.join()))
        eB2 = ib.mark(2)
        w_lstm = ib.behind(eB1, lambda: self._ar(a.grad[lstm0:head0]))
        # the 40-byte sparse-norm vector rides in the same launch as the last gradient bucket
        if ride:
            w_front = ib.behind(eB2, lambda: dist.all_reduce_coalesced([a.grad[front0:lstm0], _sparse_norm_slot(m), dpre_all],
                                                                       op=dist.ReduceOp.SUM, async_op=True))
            w_dpre = w_front
        else:
            w_front, w_dpre = ib.behind(eB2, lambda: (
                dist.all_reduce_coalesced([a.grad[front0:lstm0], _sparse_norm_slot(m)], op=dist.ReduceOp.SUM, async_op=True),
                self._gather(dpre_all, m.dpre)))
        if self.split_update:
            # three update slices, each behind the collectives it needs (the head update runs while the small, latency-bound
            # all-gather of dpre -- the one collective on the critical path -- is in flight)
            w_head.wait()
            cap(("dpC0", B, T), lambda: (m._tick(), m._update_slice(s_head)))
            for w in (w_x, w_dpre):
                w.wait()
            cap(("dpC1", B, T), lambda: (m._bwd_enc(B, T, x_all, dpre_all), m._update_slice(s_enc)))
            for w in (w_lstm, w_front):
                w.wait()
            cap(("dpC2", B, T), lambda: (m._update_slice(s_mid), m.be.l2_total(a.wsq, a.seg_l2, a.nseg, m.met[2:3])))
            return
        if self._shard_ok(m):
            # Row-sharded encoder update (opt-in until a multi-GPU node has timed it against the default):
            # every rank forms N / G rows of dW from the gathered operands -- the single-process product's work instead of G
            # times it (segment C1 at world-8 operand sizes: 160 us against 54 fused, profiles/r03_dp_c1_at_world8_operand_sizes.txt)
            # --, the variable's clip norm is an 8-byte all-reduce of the shards' (sum g^2, sum theta^2) pairs, Adam runs on the
            # shard (1 / G of the optimizer traffic of 59 % of the parameters), and the updated rows are all-gathered: 41 MB x (G-1)/G
            # per rank that the NEXT step's first launch waits for -- the price, and the reason this is not the default unmeasured.
            # The moments of the other shards' rows are never touched on this rank (they live on their owner).
            G, rank = self.world, (dist.get_rank() if self.rank is None else self.rank)
            nr = m.N // G
            r0 = rank * nr
            for w in (w_x, w_dpre):
                w.wait()
            cap(("dpS1", B, T), lambda: m._enc_shard_grad(x_all, dpre_all, r0, nr))
            eS = ib.mark(3)
            w_n = ib.behind(eS, lambda: self._ar(a.partial[0:2]))
            for w in (w_head, w_lstm, w_front, w_n):
                w.wait()
            cap(("dpS2", B, T), lambda: (m._update_fused(m.met[2:3], skip_first=True), m._enc_shard_adam(r0, nr)))
            e0 = a.entries["dense_img/kernel"]
            whole, mine = a.theta[e0.off:e0.off + e0.size], a.theta[e0.off + r0 * m.E:e0.off + (r0 + nr) * m.E]
            if dist.get_backend() == "gloo":         # (its all_gather takes a list of outputs: the shard goes through a copy)
                send = m._enc_shard["send"]
                send.copy_(mine)
                self._gather(whole, send).wait()
            else:                                    # RCCL in place: rank r's input IS slot r of the output
                dist.all_gather_into_tensor(whole, mine, async_op=True).wait()
            return
        # default: ONE update behind all collectives -- the encoder gradient from the gathered operands, then the
        # single-process update sequence (span norms of every variable in one launch, one finalize launch with the tick and
        # the L2 metric, one clip + Adam launch): 5 launches instead of 14 (world-size-1 rehearsal: -28 us of kernels and
        # ~-40 us of host-paced gaps; what it gives up is the head update's overlap with the last two collectives)
        for w in (w_head, w_x, w_dpre, w_lstm, w_front):
            w.wait()
        cap(("dpC", B, T), lambda: (m._bwd_enc(B, T, x_all, dpre_all), m._update_fused(m.met[2:3])))

    def _shard_ok(self, m):
        """the row-sharded encoder update applies: opted in, equal 16-byte-aligned shards, Adam, no adaptive clipping"""
        opt = m.optimizer
        return bool(self.shard_encoder and not self.split_update and m.N % self.world == 0 and (m.N // self.world) % 4 == 0
                    and opt is not None and opt.kind == "adam" and not m.__dict__.get("agc")
                    and m.arena.entries["dense_img/kernel"].seg == 0 and hasattr(m.be, "step_finalize"))

    def __call__(self, model):          # generic fallback (models without a pipelined schedule)
        make_grad_sync(self.world)(model)


class PipelinedAttentionSync:
    """DP schedule for the region-wise encoder + attention decoder (lc_nic.NIC, BASELINE config 4; also the
    multi-subject model, whose S encoders simply widen the last bucket).  27 MB of gradients, all of them all-reduced
    (no tensor dominates the way the dense encoder kernel does), in the four buckets backward finishes them in, each
    issued ``async_op`` right behind the launch group that produces it and waited for only by the optimizer slice
    that consumes it (the arena keeps the variables in reverse production order, so every bucket is one range):

      A  : forward + loss + head backward             -> all-reduce head (nonlinear + softmax, 5.7 MB)
      B1 : T-step chain backward + LSTM gradients     -> all-reduce LSTM (8.7 MB)
      B2 : text branch, embedding scatter             -> all-reduce embedding (10.2 MB) + its sparse-norm scalar
      B3 : attention parameters, BatchNorm, encoder   -> all-reduce {encoder, BatchNorm, attention} (2.7 MB)
      C0 : (waits head, LSTM)        step tick; norms + clip + Adam of {LSTM, head}  (53 % of the arena)
      C1 : (waits embedding, front)  norms + clip + Adam of the rest, L2 metric

    The head bucket hides under the chain (B1, ~0.45 ms of a 1.12 ms step), the LSTM bucket under B2 + B3, the
    embedding bucket under B3 + C0; only the small front bucket and the tail of the embedding one are exposed.
    The generic schedule (one all-reduce of the whole arena between backward and update) exposes all 27 MB.
    A and B1 -- the two serial T-step chains -- are hipGraphs, the short segments recorded launch plans (see
    PipelinedDenseSync.eager; this model has ~270 launches a step, so the host cannot afford plans for the chains:
    world-size-1 rehearsal, tools/host_overhead.py --dp --attention: 1.32 ms/step with the host at 0.81, against
    1.35 / 0.85 with six graphs and 1.32 / 1.14 with A as a plan too; bench.py --force-dp --workload attention: 1.23 ms
    against 1.12 for the single-graph step)."""
    pipelined = True
    eager = frozenset(os.environ.get("TNT_DP_EAGER", "B2,B3,C,C0,C1").split(","))

    def __init__(self, world):
        self.world = world
        self._slices = None
        self.split_update = os.environ.get("TNT_DP_SPLIT_UPDATE", "0") == "1"
        self.one_graph = os.environ.get("TNT_DP_ONE_GRAPH", "0") == "1"
        self.capture_error = None

    def step(self, m, B, T):
        a = m.arena
        if self._slices is None:
            e = a.entries
            s_lstm = e["lstm/kernel"].seg
            self._slices = (a.seg_slice(s_lstm, a.nseg), a.seg_slice(0, s_lstm))
            self._offs = (e["emb_text/embeddings"].off, e["lstm/kernel"].off, e["time_distributed_nonlinear/kernel"].off)
        sl_tail, sl_front = self._slices
        emb0, lstm0, head0 = self._offs
        ar = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

        def body():
            m._forward(B, T, True); m._loss_metrics(B, T, True); m._bwd_head(B, T)
            w_head = ar(a.grad[head0:])
            m._bwd_chain(B, T)
            w_lstm = ar(a.grad[lstm0:head0])
            m._bwd_emb(B, T)
            w_emb = dist.all_reduce_coalesced([a.grad[emb0:lstm0], _sparse_norm_slot(m)], op=dist.ReduceOp.SUM, async_op=True)
            m._bwd_front(B, T)
            w_front = ar(a.grad[:emb0])
            w_head.wait(); w_lstm.wait()
            m._tick(); m._update_slice(sl_tail)
            w_emb.wait(); w_front.wait()
            m._update_slice(sl_front); m.be.l2_total(a.wsq, a.seg_l2, a.nseg, m.met[2:3])
        if _whole_step_graph(self, m, ("dpall", B, T), body):
            return
        cap = lambda key, fn: (m._run_planned if key[0][2:] in self.eager else m._run_captured)(key, fn)

        ib = self.__dict__.get("_ib") or self.__dict__.setdefault("_ib", _IssueBehind(m))
        cap(("dpA", B, T), lambda: _segment_a(m, B, T, defer_totals=not self.split_update))
        eA = ib.mark(0)
        cap(("dpB1", B, T), lambda: m._bwd_chain(B, T))               # every compute segment is enqueued before the host turns to
        eB1 = ib.mark(1)                                              # the collective behind its predecessor (_IssueBehind)
        w_head = ib.behind(eA, lambda: ar(a.grad[head0:]))
        cap(("dpB2", B, T), lambda: m._bwd_emb(B, T))
        eB2 = ib.mark(2)
        w_lstm = ib.behind(eB1, lambda: ar(a.grad[lstm0:head0]))
        cap(("dpB3", B, T), lambda: m._bwd_front(B, T))
        eB3 = ib.mark(3)
        w_emb = ib.behind(eB2, lambda: dist.all_reduce_coalesced([a.grad[emb0:lstm0], _sparse_norm_slot(m)], op=dist.ReduceOp.SUM,
                                                               async_op=True))
        w_front = ib.behind(eB3, lambda: ar(a.grad[:emb0]))
        if self.split_update:
            for w in (w_head, w_lstm):
                w.wait()
            cap(("dpC0", B, T), lambda: (m._tick(), m._update_slice(sl_tail)))
            for w in (w_emb, w_front):
                w.wait()
            cap(("dpC1", B, T), lambda: (m._update_slice(sl_front), m.be.l2_total(a.wsq, a.seg_l2, a.nseg, m.met[2:3])))
            return
        for w in (w_head, w_lstm, w_emb, w_front):      # default: ONE update behind all collectives (see PipelinedDenseSync)
            w.wait()
        cap(("dpC", B, T), lambda: m._update_fused(m.met[2:3]))

    def __call__(self, model):          # not used by lc_nic.train_step (it calls step); kept for the generic protocol
        make_grad_sync(self.world)(model)


def _segment_a(m, B, T, defer_totals):
    """Segment A of the pipelined schedules: forward + loss + head backward.  ``defer_totals`` (the default one-update form):
    the step's loss / accuracy (/ attention-metric) totals are not summed by a launch of their own behind the loss kernel but
    by the finalize work of the update launch at the end of the step (ModelBase._update_fused), as in the single-process step.
    Only this segment defers: the Embedding backward and the encoder update keep their data-parallel forms."""
    m._defer_sum2 = bool(defer_totals)
    try:
        m._forward(B, T, True); m._loss_metrics(B, T, True); m._bwd_head(B, T)
    finally:
        m._defer_sum2 = False


def _sparse_norm_slot(model):
    """The one slot of ``sq_override`` that carries data: the Embedding IndexedSlices squared norm (a per-rank partial
    sum).  The other slots hold the -1 "not overridden" sentinel and must not take part in the SUM all-reduce."""
    seg = model.emb_seg
    return model.arena.sq_override[seg:seg + 1]


def attach(model, world=None, bucket_elems=None, pipelined=None, rank=None, sync_bn=False, shard_encoder=None):
    """Make ``model`` data-parallel over the default process group.  The dense-encoder NIC and the attention NIC
    (incl. its multi-subject form) get their pipelined schedules unless pipelined=False; every other model the
    generic one (all-reduce of the whole arena between backward and update).
    Each replica draws its own dropout masks: the Philox key is derived from (seed, rank), so G replicas see G
    independent mask sets like G disjoint slices of one large batch would (rank 0 keeps the model's seed).  The
    "G ranks x local batch == one rank on the concatenated batch" contract is exact for the deterministic part of
    the step (tests/test_dp_gloo.py, dropout-free) and statistical for the masks.
    sync_bn=True: BatchNorm statistics over the GLOBAL batch (ModelBase._bn_fwd / _bn_bwd: an all-gather of the chunk
    partials in the forward, an all-reduce of two sums per channel in the backward), which extends that contract to
    BatchNorm encoders.  The collectives sit inside the forward / backward pass, so such a model runs the generic schedule
    eagerly (no captured segments); per-replica statistics with the pipelined schedules remain the default.
    shard_encoder=True (dense-encoder NIC; default: TNT_DP_SHARD_ENC): the encoder kernel's update is row-sharded over the
    ranks (PipelinedDenseSync.step).  Every rank keeps the complete, identical kernel; only the optimizer moments of a shard
    live on its owner alone."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    if model.__dict__.get("agc") and world > 1:
        # agc.py:25-30 clips the Embedding by the column norms of its UN-merged IndexedSlices rows; across replicas those
        # are per-rank partial sums that would have to be all-reduced before every rank scales the summed gradient, and
        # the pipelined schedules update slice by slice without an AGC pass.  Not built: refuse instead of training wrong.
        raise NotImplementedError("adaptive gradient clipping (enable_agc) is not supported under data parallel: "
                                  "call enable_agc(None) before dp.attach, or train on one device")
    model.dp_world = world
    model.seed = (int(model.seed) + 0x9E3779B1 * int(rank)) & 0x7FFFFFFFFFFFFFFF
    from .nic import NIC as DenseNIC
    from .lc_nic import NIC as AttentionNIC
    dense = type(model) is DenseNIC          # subclasses (fc mode) have other variables -> generic schedule
    att = isinstance(model, AttentionNIC) and hasattr(model, "_bwd_chain")
    model.sync_bn = bool(sync_bn)
    if sync_bn:
        pipelined = False
        model.use_graph = False
    if pipelined is None:
        pipelined = dense or att
    if pipelined and dense:
        model.grad_sync = PipelinedDenseSync(world, shard_encoder=shard_encoder, rank=rank)
    elif pipelined and att:
        model.grad_sync = PipelinedAttentionSync(world)
    else:
        model.grad_sync = make_grad_sync(world, bucket_elems)
    model._graphs = {}
    broadcast_parameters(model)
    return model


def broadcast_parameters(model, src=0):
    """Start every replica from rank 0's parameters and moving statistics."""
    dist.broadcast(model.arena.theta, src)
    for t in model.state_tensors():
        dist.broadcast(t, src)


def average_moving_stats(model):
    """BatchNorm moving mean / variance are per-replica state (each replica normalises over its own
    local batch, exactly the single-GPU behaviour); a checkpoint stores their mean over the replicas
    (SURVEY 8e).  Collective: every rank calls it (ModelCheckpoint does, before rank 0 writes)."""
    world = dist.get_world_size()
    for t in model.state_tensors():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(world)


def allreduce_metrics(metrics):
    """Mean of the logged scalars over ranks (logging only)."""
    world = dist.get_world_size()
    keys = sorted(metrics)
    t = torch.stack([metrics[k].float() for k in keys])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: t[i] / world for i, k in enumerate(keys)}
