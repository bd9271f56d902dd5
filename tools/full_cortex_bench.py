"""the region-wise model fed at full-cortex width (N = 327 684 betas per sample, 84 MB per batch of 64).
Times train_step per batch, PCIe included, for: pageable numpy batches, PinnedPrefetcher (float32 / float16 on the
wire), and compact_groups (only the 62 756 referenced voxels cross the link), next to the HBM-resident step."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_batch
from masters_thesis_amd.data import PinnedPrefetcher
from masters_thesis_amd.lc_nic import NIC, synthetic_groups, compact_groups
from masters_thesis_amd.optimizers import Adam
N, NUSED, R, D, B, T, V, U = 327684, 62756, 360, 32, 64, 15, 5001, 512
rng = np.random.default_rng(12)
g0 = synthetic_groups(NUSED, R, D, seed=42)
perm = np.sort(rng.choice(N, NUSED, replace=False))
groups = ([perm[np.asarray(g)] for g in g0[0]], g0[1])
NB = 24
xs = [np.random.default_rng(i).standard_normal((B, N)).astype(np.float32) for i in range(4)]
d0, tgt0 = synth_batch(B, 4, T, V, U, rng)


class Gen:
    """stands for a loader that has the batches in host memory: pageable numpy arrays, or (pinned=True) arrays it
    reads straight into pinned memory; cols: a loader that keeps only the referenced voxel columns on disk"""
    def __init__(self, cols=None, pinned=False, half=False):
        self.x = [np.ascontiguousarray(x[:, cols]) if cols is not None else x for x in xs]
        if half: self.x = [x.astype(np.float16) for x in self.x]
        if pinned: self.x = [torch.as_tensor(x).pin_memory() for x in self.x]
    def __len__(self): return NB
    def on_epoch_end(self): pass
    def __getitem__(self, i):
        return ((self.x[i % 4], d0[1], d0[2], d0[3]), tgt0)


def mk(g):
    m = NIC(g, U, 512, 512, 32, V, T, 0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.01, 0.001, 3e-5, 1e-5, seed=5)
    m.compile(Adam(1e-4, beta_2=0.98, epsilon=1e-8, clipnorm=0.1))
    return m


def run(name, model, src):
    for i in range(4): model.train_step(src[i])
    if hasattr(src, "on_epoch_end"): src.on_epoch_end()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(NB): model.train_step(src[i])
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / NB
    print(f"{name:58s} {el * 1e3:7.3f} ms/batch  {B * T / el:10.0f} caption-tokens/s")


m = mk(groups)
dev_batch = tuple(torch.as_tensor(a).cuda() for a in Gen()[0][0]), torch.as_tensor(tgt0).cuda()
run("betas resident in HBM (327 684 wide)", m, [dev_batch] * NB)
run("pageable numpy batches, no prefetcher (84 MB per batch)", m, Gen())
run("PinnedPrefetcher, pageable source, float32", m, PinnedPrefetcher(Gen(), "cuda"))
run("PinnedPrefetcher, pinned source, float32 (84 MB DMA)", m, PinnedPrefetcher(Gen(pinned=True), "cuda"))
run("PinnedPrefetcher, pinned source, float16 (42 MB DMA)", m, PinnedPrefetcher(Gen(pinned=True, half=True), "cuda", betas_dtype="float16"))
cols, cg = compact_groups(groups)
mc = mk(cg)
run("compact_groups: HBM-resident (62 756 wide)", mc, [(tuple(torch.as_tensor(a).cuda() for a in Gen(cols)[0][0]), torch.as_tensor(tgt0).cuda())] * NB)
run("compact_groups + PinnedPrefetcher, pageable, float32 (16 MB)", mc, PinnedPrefetcher(Gen(cols), "cuda"))
run("compact_groups + PinnedPrefetcher, pinned, float32 (16 MB)", mc, PinnedPrefetcher(Gen(cols, pinned=True), "cuda"))
run("compact_groups + PinnedPrefetcher, pinned, float16 (8 MB)", mc, PinnedPrefetcher(Gen(cols, pinned=True, half=True), "cuda", betas_dtype="float16"))
