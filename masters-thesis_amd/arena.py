"""Flat parameter arena: every trainable tensor of a model lives in ONE contiguous
float32 device buffer (theta), with parallel buffers for the gradient and the optimizer
slots.  One buffer means one RCCL all-reduce for data-parallel training and one
multi-tensor clip+Adam launch (reference: optimizer.apply_gradients over
model.trainable_variables, lc_NIC.py:386-389), while each variable keeps its own
clipnorm segment.
"""
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch

SPAN = 8192      # elements per optimizer workgroup span
ALIGN = 64       # segment alignment in floats (256 B)


@dataclass
class Spans:
    span_seg: torch.Tensor
    span_off: torch.Tensor
    span_len: torch.Tensor
    seg_first: torch.Tensor
    nspan: int
    first_host: tuple = ()


@dataclass
class SegSlice:
    """Spans of the contiguous segment range [seg0, seg1): views into the arena's span table with a
    slice-local seg_first, so norms + clip + Adam can run on part of the arena as soon as that part's
    gradients are final (pipelined data-parallel update)."""
    seg0: int
    nseg: int
    nspan: int
    span_seg: torch.Tensor
    span_off: torch.Tensor
    span_len: torch.Tensor
    seg_first: torch.Tensor
    partial: torch.Tensor
    sq: torch.Tensor
    wsq: torch.Tensor


def build_spans(offs, lens, device):
    """Cut segments (offset, length) into spans of <= SPAN elements."""
    seg, off, ln, first = [], [], [], [0]
    for s, (o, n) in enumerate(zip(offs, lens)):
        k = 0
        while k < n:
            m = min(SPAN, n - k)
            seg.append(s); off.append(o + k); ln.append(m)
            k += m
        first.append(len(seg))
    return Spans(torch.tensor(seg, dtype=torch.int32, device=device),
                 torch.tensor(off, dtype=torch.int64, device=device),
                 torch.tensor(ln, dtype=torch.int32, device=device),
                 torch.tensor(first, dtype=torch.int32, device=device), len(seg), tuple(first))


@dataclass
class Entry:
    name: str
    off: int
    shape: tuple          # storage shape inside the arena (may include padding)
    size: int             # prod(shape)
    l2: float
    seg: int


class ParamArena:
    def __init__(self, device):
        self.device = device
        self.entries = OrderedDict()
        self.total = 0
        self.theta = self.grad = None

    def add(self, name, shape, l2=0.0, align=ALIGN):
        """Register a trainable tensor.  ``align`` (floats, multiple of 4) pads the slot; with
        align=4 and sizes that are multiples of 4, consecutive adds are exactly contiguous
        (used for the per-region encoder kernels, which one kernel launch reads as a CSR
        concatenation while each stays its own clipnorm variable)."""
        assert self.theta is None, "arena already finalized"
        assert align % 4 == 0
        size = int(np.prod(shape))
        e = Entry(name, self.total, tuple(int(s) for s in shape), size, float(l2), len(self.entries))
        self.entries[name] = e
        self.total += (size + align - 1) // align * align
        return e

    def finalize(self):
        dev = self.device
        self.total = (self.total + ALIGN - 1) // ALIGN * ALIGN
        self.theta = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        es = list(self.entries.values())
        self.nseg = len(es)
        self.spans = build_spans([e.off for e in es], [e.size for e in es], dev)
        self.seg_l2 = torch.tensor([e.l2 for e in es], dtype=torch.float32, device=dev)
        self.sq = torch.zeros(self.nseg, dtype=torch.float32, device=dev)
        self.wsq = torch.zeros(self.nseg, dtype=torch.float32, device=dev)
        self.sq_override = torch.full((self.nseg,), -1.0, dtype=torch.float32, device=dev)
        self.partial = torch.zeros(2 * self.spans.nspan, dtype=torch.float32, device=dev)
        return self

    def seg_slice(self, seg0, seg1):
        sp, fh = self.spans, self.spans.first_host
        s0, s1 = fh[seg0], fh[seg1]
        local = torch.tensor([f - s0 for f in fh[seg0:seg1 + 1]], dtype=torch.int32, device=self.device)
        return SegSlice(seg0, seg1 - seg0, s1 - s0, sp.span_seg[s0:s1], sp.span_off[s0:s1], sp.span_len[s0:s1], local,
                        self.partial[2 * s0:2 * s1], self.sq[seg0:seg1], self.wsq[seg0:seg1])

    def p(self, name):
        e = self.entries[name]
        return self.theta[e.off:e.off + e.size].view(e.shape)

    def g(self, name):
        e = self.entries[name]
        return self.grad[e.off:e.off + e.size].view(e.shape)

    def slot(self, buf, name):
        e = self.entries[name]
        return buf[e.off:e.off + e.size].view(e.shape)


class AgcTable:
    """Work table of tnt_agc_f32 (adaptive gradient clipping, agc.py:20-38): every trainable cut into items of
    <= 64 columns x <= ROWS rows.  ``shapes``: name -> keras shape (rank 1: one unit; rank 2: one unit per output
    column; the storage may be wider than the keras shape -- zero padding never clips)."""
    ROWS = 256

    def __init__(self, arena, shapes, emb_name=None):
        dev = arena.device
        off, ld, lam, items, cb_first = [], [], [], [], [0]
        self.emb = None
        for v, (name, e) in enumerate(arena.entries.items()):
            ks = tuple(shapes[name])
            rows = int(ks[0]) if len(ks) >= 2 else e.size
            cols = e.size // rows if len(ks) >= 2 else 1
            assert rows * cols == e.size, (name, ks, e.shape)
            off.append(e.off); ld.append(cols); lam.append(e.l2)
            cb0 = len(cb_first) - 1
            for c0 in range(0, cols, 64):
                for r0 in range(0, rows, self.ROWS):
                    items.append((v, c0, min(64, cols - c0), r0, min(rows, r0 + self.ROWS), len(cb_first) - 1))
                cb_first.append(len(items))
            if name == emb_name:
                self.emb = (v, cb0, len(cb_first) - 1 - cb0)
        self.nitem = len(items)
        self.var_off = torch.tensor(off, dtype=torch.int64, device=dev)
        self.var_ld = torch.tensor(ld, dtype=torch.int32, device=dev)
        self.var_lam = torch.tensor(lam, dtype=torch.float32, device=dev)
        self.item = torch.tensor(items, dtype=torch.int32, device=dev).contiguous()
        self.cb_first = torch.tensor(cb_first, dtype=torch.int32, device=dev)
        self.partial = torch.zeros(self.nitem * 128, dtype=torch.float32, device=dev)
        self.sq_part = torch.zeros(max(1, self.emb[2] if self.emb else 1), dtype=torch.float32, device=dev)
