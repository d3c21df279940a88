"""ONE accuracy experiment with the E8M0 32-element block scales that
v_mfma_scale_f32_32x32x64_f8f6f4 takes natively.  The random-init LLaMA-13B + CLIP-L/14 + Whisper of BASELINE cfg 5
through the fp32 ORACLE (oracle/restate.py on the GPU) with only the fp8 path's operand quantisation added at the q|k|v
and alignment K/V sites (tests/fp8_ref.py), per-row scales (what the kernels ship) against per-32-block E8M0 scales;
logits relative L2 error and loss against the plain fp32 oracle.  No kernel runs: this prices the FORMATS.

    python scripts/fp8_block_scale_experiment.py [real_13b|real_7b]        (on the GPU box)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from fp8_ref import fake_quant_oracle  # noqa: E402
from macaw_llm_amd.factory import baseline_config, build_model, synthetic_inputs  # noqa: E402
from oracle import restate  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "real_13b"
    dev = torch.device("cuda:0")
    cfg = baseline_config(name)
    model = build_model(cfg, dtype=torch.bfloat16, device=dev, seed=11, fuse=True).eval()
    inp = synthetic_inputs(cfg, 2, 128, modalities=("images", "audios"), seed=5, device=dev)
    sd = restate.hot_path_state({k: v.detach().float() for k, v in model.state_dict().items()})
    del model
    torch.cuda.empty_cache()
    fin = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
    out = {}
    with torch.no_grad():
        out["fp32"] = restate.mm_forward(sd, fin, cfg)
        for scheme in ("rows", "blocks32"):
            with fake_quant_oracle(("qkv", "align"), scheme=scheme):
                out[scheme] = restate.mm_forward(sd, fin, cfg)
        sd16 = {k: v.to(torch.bfloat16) for k, v in sd.items()}
        f16 = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
        out["eager bf16"] = restate.mm_forward(sd16, f16, cfg)
    ref = out["fp32"]["logits"].float()
    print(f"{name}: logits relative L2 error vs the fp32 oracle (q|k|v + alignment K/V operands quantised, forward only)")
    for k in ("rows", "blocks32", "eager bf16"):
        z = out[k]["logits"].float()
        print(f"  {k:12s} rel L2 {((z - ref).norm() / ref.norm()).item():.4f}   max|d| / max|z| "
              f"{((z - ref).abs().max() / ref.abs().max()).item():.4f}   loss {out[k]['loss'].item():.5f} (fp32 {out['fp32']['loss'].item():.5f})")


if __name__ == "__main__":
    main()
