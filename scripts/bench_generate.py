"""Greedy generate() throughput (KV-cache path), LLaMA-7B bf16, image + audio + 128-token prompt."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd.factory import baseline_config, build_model, synthetic_inputs
dev = torch.device("cuda:0")
cfg = baseline_config("real_7b")
model = build_model(cfg, dtype=torch.bfloat16, device=dev, seed=1).eval()
# warm-up (allocator pools, kernel attributes) so that the first measured batch size is not inflated
_w = synthetic_inputs(cfg, 1, 128, modalities=("images", "audios"), seed=2, device=dev)
with torch.no_grad():
    model.llm.generate(inputs_embeds=model.prepare_inputs_for_generation(_w)[0], max_new_tokens=4, eos_token_id=-1)
# usage: bench_generate.py [B ...]; BG_SKINNY32="19,22" repeats every batch size > 16 with each kernel for
# 17 ... 32 token rows (MK_GEMM_SKINNY32, read per call by csrc/gemm.hip: one process and one model serve all variants)
BATCHES = [int(a) for a in sys.argv[1:]] or [1, 8, 32]
SK = [v for v in os.environ.get("BG_SKINNY32", "").split(",") if v]
for B, sk in [(B, s) for B in BATCHES for s in (SK if (SK and B > 16) else [None])]:
    if sk is not None:
        os.environ["MK_GEMM_SKINNY32"] = sk
    var = None if sk is None else f"skinny32 {sk}"
    inp = synthetic_inputs(cfg, B, 128, modalities=("images", "audios"), seed=2, device=dev)
    with torch.no_grad():
        emb, am, _ = model.prepare_inputs_for_generation(inp)
        for new in (8, 72):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = model.llm.generate(inputs_embeds=emb, max_new_tokens=new, eos_token_id=-1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if new == 8: t8 = dt
        per_tok = (dt - t8) / 64
        print(("" if var is None else f"[{var}] ") + f"B={B:2d}: prompt S={emb.shape[1]}, prefill+8 tok {t8 * 1e3:7.1f} ms, decode {per_tok * 1e3:6.2f} ms/token "
              f"= {B / per_tok:7.0f} tokens/s (weights streamed once per token: {13.5 / per_tok / 1e3:4.2f} TB/s)")
