"""one generate() call (B from argv, 40 new tokens) for profiling the decode step"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd.factory import baseline_config, build_model, synthetic_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
cfg = baseline_config("real_7b")
model = build_model(cfg, dtype=torch.bfloat16, device=dev, seed=1).eval()
inp = synthetic_inputs(cfg, B, 128, modalities=("images", "audios"), seed=2, device=dev)
with torch.no_grad():
    emb, am, _ = model.prepare_inputs_for_generation(inp)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.llm.generate(inputs_embeds=emb, max_new_tokens=40, eos_token_id=-1)
        torch.cuda.synchronize(); print("generate", time.perf_counter() - t0, out.shape)
