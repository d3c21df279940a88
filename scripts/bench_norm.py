"""RMSNorm backward (+ its partial-sum reduction) at the cfg-3 shape, per number of row-slab blocks
(ops.RMSNORM_BLOCKS): rows are processed one after the other inside a block with a block-wide reduction each,
so the blocks per CU decide how many rows are in flight.  usage: python scripts/bench_norm.py [rows] [cols]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from macaw_llm_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
NB = 24                                     # rotating operand sets: 24 x 113 MB, nothing stays in the caches
g = torch.Generator().manual_seed(0)
sets = []
for i in range(NB):
    dy = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev)
    h = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev)
    dres = torch.randn((rows, cols), generator=g).to(torch.bfloat16).to(dev)
    sets.append((dy, h, dres))
w = torch.randn((cols,), generator=g).to(torch.bfloat16).to(dev)
rstd = torch.rand((rows,), generator=g).to(dev) + 0.5
ref = None
for nb in (128, 192, 256, 320, 384, 512, 1024):
    ops.RMSNORM_BLOCKS = nb
    dx, dw = ops.rmsnorm_bwd(*sets[0][:2], w, rstd, dres=sets[0][2])
    if ref is None:
        ref = (dx.clone(), dw.float().clone())
    else:                                   # dx does not depend on the slab count; dw only in fp32 summation order
        assert torch.equal(dx, ref[0]), nb
        assert (dw.float() - ref[1]).abs().max().item() <= 2 ** -6 * ref[1].abs().max().item(), nb
    for _ in range(3):
        for s in sets:
            ops.rmsnorm_bwd(s[0], s[1], w, rstd, dres=s[2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for s in sets:
            ops.rmsnorm_bwd(s[0], s[1], w, rstd, dres=s[2])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * NB)
    gb = rows * cols * 2 * 4 / 1e9          # dy, h, dres read + dx written
    print(f"RMSNORM_BLOCKS {nb:5d}: {us:7.1f} us per rmsnorm_bwd + colsum_partials, {gb / us * 1e6 / 1e3:5.2f} TB/s of algorithmic bytes", flush=True)
