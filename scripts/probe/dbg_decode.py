import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
hd, H, B, T = 128, 32, 1, 173
torch.manual_seed(3 * hd + T)
D, Tmax, p = H * hd, T + 5, T - 1
cache = torch.zeros((B, Tmax, 2 * D), dtype=torch.bfloat16, device=dev)
cache[:, :p] = torch.randn((B, p, 2 * D), device=dev).to(torch.bfloat16)
qkv = torch.randn((B, 3 * D), device=dev).to(torch.bfloat16)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
ang = torch.cat((torch.outer(torch.arange(Tmax).float(), inv),) * 2, dim=-1)
cos, sin = ang.cos().to(dev).to(torch.bfloat16), ang.sin().to(dev).to(torch.bfloat16)
t_dev = torch.tensor([p], dtype=torch.int32, device=dev)
ref_qkv = qkv.clone()
pos = torch.full((B,), p, dtype=torch.int32, device=dev)
ops.rope_(ref_qkv[:, :2 * D], cos, sin, pos, 2 * H, hd)
want = cache.clone(); want[:, p] = ref_qkv[:, D:]
out = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
ops.decode_step_attn(qkv, qkv, qkv, 3 * D, cos, sin, cache, t_dev, Tmax, B, H, hd, out, 1 / hd ** 0.5, k_off=D, v_off=2 * D)
d = (cache.float() - want.float()).abs()
idx = torch.nonzero(d > 0)
print("ndiff", idx.shape[0], "max", d.max().item())
print(idx[:12].tolist())
for i in idx[:6].tolist():
    print(i, cache[tuple(i)].item(), want[tuple(i)].item(), "k_in", qkv[i[0], D + i[2]].item() if i[2] < D else None)
