"""bench.py's cpu_baseline is the oracle PORT (oracle/restate.py): /root/reference does not exist on the GPU box.
Where it does exist (the build container), this script times the REFERENCE's own classes against the port on the
same inputs and host cores, for the component that is 90 % of the baseline's time (one LLaMA-7B decoder layer,
forward + backward, S = 144, B = 1) -- evidence that timing the port is timing the reference's arithmetic.

    python scripts/cpu_port_vs_reference.py [threads]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import configs, ref_loader, restate

threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(16, os.cpu_count() or 1)
torch.set_num_threads(threads)
if not ref_loader.reference_available():
    raise SystemExit("needs /root/reference")
mod = ref_loader.load_reference_modeling()
from transformers import LlamaConfig

ll = configs.get("real_7b")["llama"]
lcfg = LlamaConfig(**ll)
lcfg._attn_implementation = "eager"
torch.manual_seed(0)
layer = mod.LlamaDecoderLayer(lcfg)
D, H, S = ll["hidden_size"], ll["num_attention_heads"], 144
x = torch.randn(1, S, D, requires_grad=True)
mask = restate.decoder_mask(torch.ones(1, S, dtype=torch.long), 1, S, torch.float32, x.device)
pos = torch.arange(S)[None]
sd = {"l." + n: p for n, p in layer.named_parameters()}
cos, sin = restate.rotary_tables(D // H, 2048)


def ref_step():
    layer(x, attention_mask=mask, position_ids=pos)[0].sum().backward()


def port_step():
    restate.llama_layer(sd, "l.", x, mask, pos, H, ll["rms_norm_eps"], cos, sin).sum().backward()


def timeit(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


y_ref = layer(x, attention_mask=mask, position_ids=pos)[0]
y_port = restate.llama_layer(sd, "l.", x, mask, pos, H, ll["rms_norm_eps"], cos, sin)
print(f"outputs: max |reference - port| = {(y_ref - y_port).abs().max().item():.3e} of {y_ref.abs().max().item():.3f}")
a, b = [], []
for _ in range(2):                      # interleaved
    a.append(timeit(ref_step))
    b.append(timeit(port_step))
tr, tp = min(a), min(b)
print(f"{threads} threads, LLaMA-7B decoder layer fwd + bwd, S = {S}, B = 1: reference {tr * 1e3:.0f} ms, port {tp * 1e3:.0f} ms "
      f"(port / reference = {tp / tr:.3f})")
