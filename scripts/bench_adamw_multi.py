"""Multi-tensor AdamW stream (mk_adamw_multi, the N = 1 optimizer launch): 6 gate|up-sized bf16 tensors, 28 B/param.
MK_ADAMW_VAR=0 -> 8-byte 16-bit accesses (rounds 1-3), 2 -> 16-byte accesses through a wave-internal exchange
(default), 1 -> timing-only ablation without the 16-bit traffic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd.optim import FusedAdamW
dev = torch.device("cuda:0")
n = 22016 * 4096
ps = [torch.nn.Parameter(torch.randn(n, device=dev).bfloat16()) for _ in range(6)]
for p in ps:
    p.grad = torch.randn(n, device=dev).bfloat16()
opt = FusedAdamW(ps, lr=1e-4)
for _ in range(2):
    opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    opt.step()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
var = os.environ.get("MK_ADAMW_VAR", "2")
byts = (24 if var == "1" else 28) * n * len(ps)
print(f"MK_ADAMW_VAR={var}: {t * 1e3:.3f} ms per {len(ps)} x 90M params = {byts / t / 1e12:.2f} TB/s "
      f"({28 * n * len(ps) / t / 1e12:.2f} TB/s at 28 B/param)")
