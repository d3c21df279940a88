import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
rows, cols = 4608, 4096
dy = torch.randn(rows, cols, device=dev).to(torch.bfloat16); h = torch.randn_like(dy); dres = torch.randn_like(dy)
w = torch.ones(cols, device=dev, dtype=torch.bfloat16); rstd = torch.rand(rows, device=dev) + 0.5
for nb in (256, 512, 1024, 2048, 4608):
    ops.NORM_BLOCKS = nb
    for _ in range(3): ops.rmsnorm_bwd(dy, h, w, rstd, dres)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.rmsnorm_bwd(dy, h, w, rstd, dres)
    e1.record(); torch.cuda.synchronize()
    print(nb, "blocks:", e0.elapsed_time(e1) / 50 * 1e3, "us per rmsnorm_bwd + colsum")
