import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from macaw_llm_amd import ops
from oracle import restate
dev = torch.device("cuda:0")
hd, Bn, S, H = 8, 1, 3, 1
g = torch.Generator().manual_seed(0)
cos, sin = restate.rotary_tables(hd, 64)
cos, sin = cos.bfloat16(), sin.bfloat16()
q = torch.randn((Bn, S, H, hd), generator=g).bfloat16()
pos = torch.arange(S).repeat(Bn, 1)
qr, _ = restate.apply_rope(q.transpose(1, 2), q.transpose(1, 2), cos, sin, pos)
qd = q.reshape(Bn * S, H * hd).to(dev).clone()
ops.rope_(qd, cos.to(dev), sin.to(dev), pos.reshape(-1).int().to(dev), H, hd)
got = qd.view(Bn, S, H, hd).transpose(1, 2).cpu()
bad = (got != qr).nonzero()
print("nbad", len(bad))
for idx in bad[:6]:
    b, h, s, j = idx.tolist()
    half = hd // 2
    x = q[b, s, h].float()
    c, sn = cos[s].float(), sin[s].float()
    jj = j % half
    a_, b_ = x[jj].item(), x[jj + half].item()
    print(idx.tolist(), "x1", a_, "x2", b_, "cos", c[j].item(), "sin", sn[j].item(), "got", got[b, h, s, j].item(), "ref", qr[b, h, s, j].item())
    if j < half:
        p1 = torch.tensor(a_ * c[j].item()).bfloat16().float().item(); p2 = torch.tensor(-b_ * sn[j].item()).bfloat16().float().item()
    else:
        p1 = torch.tensor(b_ * c[j].item()).bfloat16().float().item(); p2 = torch.tensor(a_ * sn[j].item()).bfloat16().float().item()
    print("   p1", p1, "p2", p2, "sum", p1 + p2, "rounded", torch.tensor(p1 + p2).bfloat16().item(), " unrounded-products sum", torch.tensor((a_ if j < half else b_) * c[j].item() + ((-b_) if j < half else a_) * sn[j].item()).bfloat16().item())
