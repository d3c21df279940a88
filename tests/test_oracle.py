"""CPU tests of the oracle (test infrastructure): the restatement in oracle/restate.py must
reproduce (a) the committed golden vectors, which are outputs of the reference itself, and
(b) when /root/reference is present (build container only), the live reference on fresh inputs."""
import pytest
import torch

from golden_util import load_case
from oracle import configs, inputs as oin, ref_loader, restate


@pytest.mark.parametrize("case", ["micro_all", "micro_image"])
@pytest.mark.parametrize("hoist", [True, False])
def test_restatement_matches_golden(case, hoist):
    fx = load_case(case)
    cfg = configs.get(fx["config_name"])
    with torch.no_grad():
        r = restate.mm_forward(fx["state"], fx["inputs"], cfg, hoist=hoist)
    # integer outputs: bit exact
    assert torch.equal(r["attention_mask"], fx["attention_mask"])
    assert torch.equal(r["labels"], fx["labels"])
    # fp32: same algorithm, different summation order inside torch ops only
    assert (r["inputs_embeds"] - fx["inputs_embeds"]).abs().max().item() < 1e-6
    assert (r["logits"] - fx["logits"]).abs().max().item() < 5e-6
    assert abs(r["loss"].item() - fx["loss"].item()) < 1e-6


def test_restated_gradients_match_golden():
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    sd = {k: v.clone().requires_grad_(("encoder" not in k) and v.is_floating_point() and "inv_freq" not in k)
          for k, v in fx["state"].items()}
    r = restate.mm_forward(sd, fx["inputs"], cfg)
    r["loss"].backward()
    for name, g in fx["grads"].items():
        got = sd[name].grad
        assert got is not None, name
        assert (got - g).abs().max().item() <= 1e-5 * max(1.0, g.abs().max().item()), name
    for name, n in fx["grad_norms"].items():
        if name in sd and sd[name].grad is not None:
            assert abs(sd[name].grad.norm().item() - n) <= 1e-4 * max(n, 1e-3), name


def test_restatement_matches_reference_at_real_7b_dimensions():
    """tests/golden/real7b_layer.pt = one LLaMA-7B decoder layer + final norm + 512 lm_head rows
    run by the REFERENCE's own classes (oracle/make_golden.py make_real7b_layer; weights from the
    seeded recipe in oracle/inputs.py): the restatement must reproduce it at D = 4096, FF = 11008,
    32 heads, with right padding -- forward and the gradient w.r.t. the layer input."""
    import os
    from golden_util import GOLDEN_DIR
    fx = torch.load(os.path.join(GOLDEN_DIR, "real7b_layer.pt"), weights_only=False)
    w = oin.real7b_layer_weights(fx["seed"], fx["head_rows"])
    x, am = oin.real7b_layer_inputs(fx["seed"], fx["B"], fx["S"])
    sd = {k.replace("layer.", "L."): v for k, v in w.items()}
    B, S = fx["B"], fx["S"]
    x = x.clone().requires_grad_(True)
    mask = restate.decoder_mask(am, B, S, torch.float32, x.device)
    cos, sin = restate.rotary_tables(128, 2048)
    h = restate.llama_layer(sd, "L.", x, mask, torch.arange(S).unsqueeze(0), 32, 1e-6, cos, sin)
    logits = torch.nn.functional.linear(restate.rms_norm(h, sd["norm.weight"], 1e-6), sd["lm_head.weight"])
    (logits * oin.real7b_layer_cotangent(fx["seed"], B, S, fx["head_rows"])).sum().backward()
    assert (h - fx["layer_out"]).abs().max().item() < 2e-5
    assert (logits - fx["logits"]).abs().max().item() < 2e-5
    assert (x.grad - fx["dx"]).abs().max().item() < 2e-5 * max(1.0, fx["dx"].abs().max().item())


def _fullsize_fixture(name):
    import os
    from golden_util import GOLDEN_DIR
    from oracle import hashweights as hw
    fx = torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=False)
    cfg = configs.get(fx["config_name"])
    cfg["llama"]["num_hidden_layers"] = fx["llama_layers"]
    inp = hw.make_inputs(cfg, 1, fx["text_len"], fx["modalities"], tag=fx["name"])
    return fx, cfg, hw.HashState(fx["shapes"]), inp


def _check_fullsize(fx, r, hidden, tol):
    """restatement vs the REFERENCE's stored outputs; returns the measured errors"""
    pos = fx["positions"]
    assert torch.equal(r["attention_mask"], fx["attention_mask"]) and torch.equal(r["labels"], fx["labels"])   # INT
    e = {"inputs_embeds": (r["inputs_embeds"] - fx["inputs_embeds"]).abs().max().item(),
         "logits": (r["logits"][0, pos] - fx["logits_at"]).abs().max().item(),
         "loss": abs(r["loss"].item() - fx["loss"].item())}
    for d, want in fx["hidden_at"].items():
        e[f"hidden_{d}"] = (hidden[d][0, pos] - want).abs().max().item() / want.abs().max().item()
    if "clip_last_hidden" in fx:
        e["image_aligned"] = (r["image_aligned"] - fx["image_aligned"]).abs().max().item()
    if "whisper_last_hidden" in fx:
        e["whisper"] = (r["audio_features"][0, fx["whisper_rows"]] - fx["whisper_last_hidden"]).abs().max().item()
        e["audio_aligned"] = (r["audio_aligned"] - fx["audio_aligned"]).abs().max().item()
    if "video_long_attn" in fx:
        e["video_long_attn"] = (r["video_features"][0, fx["video_rows"]] - fx["video_long_attn"]).abs().max().item()
        e["video_aligned"] = (r["video_aligned"] - fx["video_aligned"]).abs().max().item()
    print(fx["name"], {k: f"{v:.2e}" for k, v in e.items()})
    assert e["logits"] <= tol and e["loss"] <= tol and e["inputs_embeds"] <= tol, e
    assert all(v <= tol for k, v in e.items() if k.startswith("hidden_")), e
    assert all(e[k] <= tol for k in ("image_aligned", "audio_aligned", "video_aligned", "whisper", "video_long_attn")
               if k in e), e
    # greedy ids: the restatement's argmax at every position is the reference's, or a near-tie inside the tolerance
    am = r["logits"][0].argmax(-1)
    flips = (am != fx["argmax_ids"]).nonzero().flatten().tolist()
    for p in flips:
        z = r["logits"][0, p]
        assert (z.max() - z[fx["argmax_ids"][p]]).item() <= 2 * tol, (p, flips)
    return e


def test_restatement_matches_reference_at_cfg1_full_size():
    """BASELINE cfg 1 -- the one configuration that IS the reference: tests/golden/cfg1_full.pt holds the outputs of
    /root/reference/modeling.py's own MM_LLMs (CLIP-ViT-L/14 + alignment attention over the 32,007-row table +
    32-layer LLaMA-7B; image-only, B = 1, fp32, CPU) run by oracle/make_golden_cfg1.py on integer-hash weights that
    every box regenerates bit-identically (oracle/hashweights.py).  The restatement, streaming the same 7.4 B
    weights layer by layer, must reproduce: INT mask / labels bit-exact; CLIP last hidden state, aligned image
    features, inputs_embeds, the decoder's hidden states after 1 / 8 / 16 / 24 / 32 layers, the logits at 8
    positions x all 32,007 columns and the loss within 5e-5 (modeling.py:941-963,965-1048,1085-1093)."""
    fx, cfg, sd, inp = _fullsize_fixture("cfg1_full")
    assert fx["llama_layers"] == 32 and fx["modalities"] == ("images",) and len(fx["shapes"]) > 600
    hidden = {}
    with torch.no_grad():
        r = restate.mm_forward(sd, inp, cfg, hidden_states=hidden)
        clip = restate.clip_vision_forward(sd, "image_encoder.vision_model.", inp["images"], cfg["clip"]["vision_config"])
    assert r["logits"].shape == (1, 136, 32007)
    e_clip = (clip[0] - fx["clip_last_hidden"]).abs().max().item() / fx["clip_last_hidden"].abs().max().item()
    assert e_clip <= 5e-5, e_clip
    _check_fullsize(fx, r, hidden, 5e-5)


def test_restatement_matches_reference_with_real_whisper_and_video_encoders():
    """tests/golden/real_av_trunc.pt: the reference's MM_LLMs with the real Whisper-base encoder (conv stem, 1500
    positions, 6 layers), the 6-frame CLIP-L/14 video path (`encode_video_long`: positional-encoding quirk +
    video_long_self_attention over 1536 tokens) and both alignment attentions at V = 32,007 / D = 4096; LLaMA
    truncated to 2 layers (the stack is pinned by cfg1_full).  Same hash weights, same 5e-5."""
    fx, cfg, sd, inp = _fullsize_fixture("real_av_trunc")
    assert fx["modalities"] == ("audios", "videos") and fx["llama_layers"] == 2
    hidden = {}
    with torch.no_grad():
        r = restate.mm_forward(sd, inp, cfg, hidden_states=hidden)
    assert r["logits"].shape == (1, 189, 32007)
    _check_fullsize(fx, r, hidden, 5e-5)


class _LazyHashDict(dict):
    """state dict that materialises hash weights on first access and KEEPS them (autograd leaves): every tensor outside
    the frozen towers requires grad, as run_clm_llms.py:390-393 leaves it"""

    def __init__(self, shapes, device=None):
        super().__init__()
        self.shapes, self.device = shapes, device

    def __missing__(self, key):
        from oracle import hashweights as hw
        t = hw.hash_tensor(key, self.shapes[key], self.device).requires_grad_("encoder" not in key)
        self[key] = t
        return t


def check_reference_gradients(fx, grads, tol_rows, tol_norm):
    """`grads`: {name: full gradient tensor} of a run on fx's weights / inputs; compared with the REFERENCE's stored rows
    (max error relative to the reference gradient's largest magnitude) and with the L2 norms of ALL its gradients"""
    worst = {}
    for name, want in fx["grad_rows"].items():
        rows = fx["grad_row_index"][name]
        got = grads[name].detach().float().cpu()
        got = got if rows is None else got[rows]
        worst[name] = (got - want).abs().max().item() / fx["grad_absmax"][name]
        assert worst[name] <= tol_rows, (name, worst[name])
    for name, n in fx["grad_norms"].items():
        assert name in grads, name
        assert abs(grads[name].detach().double().norm().item() - n) <= tol_norm * n, (name, n)     # float64 on both sides
    return worst


def test_restated_backward_matches_the_references_gradients_at_real_dimensions():
    """tests/golden/real_grad_trunc.pt: forward + loss.backward() of the reference's own MM_LLMs at real dimensions
    (CLIP-L/14 + Whisper-base + both alignment attentions over the 32,007-row table + 2 LLaMA-7B layers + lm_head at
    V = 32,007, image + audio, B = 2, encoders frozen).  Autograd through the restatement on the same hash weights must
    reproduce its 41 gradients: stored rows within 2e-5 of the largest magnitude, every L2 norm within 1e-4; the set
    of parameters that receive a gradient is the reference's."""
    import os
    from golden_util import GOLDEN_DIR
    from oracle import hashweights as hw
    fx = torch.load(os.path.join(GOLDEN_DIR, "real_grad_trunc.pt"), weights_only=False)
    cfg = configs.get(fx["config_name"])
    cfg["llama"]["num_hidden_layers"] = fx["llama_layers"]
    inp = hw.make_inputs(cfg, fx["batch"], fx["text_len"], fx["modalities"], tag=fx["name"], n_prompt=fx["n_prompt"])
    sd = _LazyHashDict(fx["shapes"])
    r = restate.mm_forward(sd, inp, cfg)
    r["loss"].backward()
    pos = fx["positions"]
    assert torch.equal(r["attention_mask"], fx["attention_mask"]) and torch.equal(r["labels"], fx["labels"])
    assert (r["logits"][:, pos].detach() - fx["logits_at"]).abs().max().item() <= 5e-5
    assert abs(r["loss"].item() - fx["loss"].item()) <= 1e-5
    grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
    assert set(grads) == set(fx["grad_norms"])                      # who gets a gradient: as in the reference
    worst = check_reference_gradients(fx, grads, 2e-5, 1e-4)
    print("real_grad_trunc: restated gradients vs the reference (max err / max |g|):",
          {k.split(".", 1)[-1][-40:]: f"{v:.1e}" for k, v in worst.items()})


def test_restated_greedy_loop_matches_the_references_cached_decode_at_real_width():
    """real_grad_trunc.pt also holds 12 greedy ids per sample produced by the REFERENCE's forward with its own KV cache
    (oracle/make_golden.reference_greedy) from the multimodal prefix at D = 4096 / V = 32,007: the restated
    full-recompute loop emits the same ids (smallest top-1 / top-2 margin on the path: 8.5e-3)."""
    import os
    from golden_util import GOLDEN_DIR
    fx = torch.load(os.path.join(GOLDEN_DIR, "real_grad_trunc.pt"), weights_only=False)
    cfg = configs.get(fx["config_name"])
    cfg["llama"]["num_hidden_layers"] = fx["llama_layers"]
    from oracle import hashweights as hw
    sd = hw.HashState({k: v for k, v in fx["shapes"].items() if k.startswith("llm.")}, keep=[k for k in fx["shapes"] if k.startswith("llm.")])
    with torch.no_grad():
        ids = restate.greedy_generate(sd, fx["inputs_embeds"], cfg, max_new_tokens=12, eos=2, pad=cfg["tags"]["pad"])
    assert fx["generate_ids_source"].startswith("reference LlamaForCausalLM.forward + past_key_values")
    assert fx["generate_margin"].min().item() > 5e-3
    assert torch.equal(ids, fx["generate_ids"])


def test_hash_weights_are_a_pure_integer_function_of_name_and_index():
    """the fixtures above rest on every box regenerating the same tensors: known answers of the recipe (any device /
    torch version must reproduce them -- tests/test_fullsize_gpu.py repeats this on the GPU), bf16-exactness,
    independence of the chunking"""
    from oracle import hashweights as hw
    assert hw.name_seed("llm.lm_head.weight") == 0x659CFCE8                       # zlib.crc32
    assert hw.hash_levels(8, 12345).tolist() == [104, 101, 191, 173, 189, 92, 24, 71]
    assert hw.hash_levels(4, 12345, start=4).tolist() == [189, 92, 24, 71]
    q = "llm.model.layers.0.self_attn.q_proj.weight"
    assert hw.hash_tensor(q, (2, 4)).tolist() == [[0.009765625, -0.010986328125, 0.015869140625, -0.031005859375],
                                                  [0.02001953125, 0.018798828125, 0.0224609375, -0.01416015625]]
    assert hw.hash_tensor("llm.model.norm.weight", (6,)).tolist() == [0.875, 1.015625, 1.09375, 1.0625, 1.046875, 0.96875]
    assert hw.hash_ids("t", (1, 6), 3, 32000).tolist() == [[19972, 1612, 8524, 27058, 30541, 23662]]
    assert hw.hash_input("cfg1_full.images", (1, 1, 1, 4)).tolist() == [[[[-0.34375, 0.9375, 1.34375, -0.703125]]]]
    w = hw.hash_tensor(q, (64, 64))
    assert torch.equal(w, w.to(torch.bfloat16).float()) and w.abs().max().item() <= 128 / 4096
    assert abs(w.std().item() - 0.018) < 2e-3 and abs(w.mean().item()) < 2e-3
    n = hw.hash_tensor("llm.model.norm.weight", (4096,))
    assert torch.equal(n, n.to(torch.bfloat16).float()) and 0.875 <= n.min().item() and n.max().item() <= 1.11
    a, old = hw.hash_tensor("x.weight", (37, 91)), hw._CHUNK
    try:
        hw._CHUNK = 1000
        assert torch.equal(a, hw.hash_tensor("x.weight", (37, 91)))
    finally:
        hw._CHUNK = old
    sd = hw.HashState({q: (8, 8)})
    assert torch.equal(sd[q], hw.hash_tensor(q, (8, 8))) and list(sd) == [q] and len(sd) == 1


def test_positional_encoding_quirk():
    """exponent uses 2*i with i already stepping by 2 (non-textbook)."""
    a, b = restate.positional_encoding(16, 48), restate.positional_encoding_loop(16, 48)
    assert (a - b).abs().max().item() < 1e-6
    i = torch.arange(0, 48, 2, dtype=torch.float32)
    textbook = torch.sin(torch.arange(16.)[:, None] * torch.exp(-torch.log(torch.tensor(10000.0)) * i / 48))
    assert (a[:, 0::2] - textbook).abs().max().item() > 0.1


def test_greedy_generate_matches_golden():
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    with torch.no_grad():
        ids = restate.greedy_generate(fx["state"], fx["inputs_embeds"], cfg, max_new_tokens=8, eos=2,
                                      pad=cfg["tags"]["pad"])
    assert torch.equal(ids, fx["generate_ids"])


def test_generate_golden_is_the_reference_cached_decode():
    """the committed ids were produced by the REFERENCE's forward with its own KV cache and its
    `prepare_inputs_for_generation` (oracle/make_golden.reference_greedy); the restated loop above only
    reproduces them"""
    fx = load_case("micro_all")
    assert fx["generate_ids_source"].startswith("reference LlamaForCausalLM.forward + past_key_values")


needs_ref = pytest.mark.skipif(not ref_loader.reference_available(), reason="/root/reference absent")


@needs_ref
def test_reference_cached_decode_reproduces_the_committed_ids():
    """here, where /root/reference exists: drive the reference's cached forward again and compare with the fixture"""
    from oracle import make_golden
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = ref_loader.build_reference_model(cfg, seed=fx["seed"])
    with torch.no_grad():
        ids = make_golden.reference_greedy(model.llm, fx["inputs_embeds"], max_new_tokens=8, eos=2, pad=cfg["tags"]["pad"])
    assert torch.equal(ids, fx["generate_ids"])


@needs_ref
def test_reference_text_only_with_labels_raises():
    """Reference quirk: with no modality, `torch.tensor([-100]*0)` is a FLOAT tensor, the
    concatenated labels become float and CrossEntropyLoss raises (modeling.py:1043-1044).
    Our implementation keeps integer dtypes and handles the text-only case."""
    cfg = configs.get("micro")
    model = ref_loader.build_reference_model(cfg, seed=77)
    inp = oin.make_inputs(cfg, 2, 10, modalities=(), seed=5)
    with torch.no_grad(), pytest.raises(RuntimeError):
        model(inputs=inp)


@needs_ref
@pytest.mark.parametrize("mods", [("images", "audios", "videos"), ("images",), ("audios", "videos")])
def test_restatement_matches_live_reference(mods):
    cfg = configs.get("micro")
    model = ref_loader.build_reference_model(cfg, seed=77)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    inp = oin.make_inputs(cfg, 3, 10, modalities=mods, seed=5, pad_tail=2)
    with torch.no_grad():
        emb, am, lab = model.prepare_inputs_for_generation(inp)
        out = model(inputs=inp)
        r = restate.mm_forward(sd, inp, cfg)
    assert torch.equal(am, r["attention_mask"]) and torch.equal(lab, r["labels"])
    assert (emb - r["inputs_embeds"]).abs().max().item() < 1e-6
    assert (out.logits - r["logits"]).abs().max().item() < 5e-6
    assert abs(out.loss.item() - r["loss"].item()) < 1e-6


@needs_ref
def test_reference_positional_encoding():
    mod = ref_loader.load_reference_modeling()
    assert (mod.create_positional_encoding(12, 48) - restate.positional_encoding(12, 48)).abs().max().item() < 1e-6
    # the product's exported helper (same name as the reference's: root modeling.py shim) is the vectorised form
    from macaw_llm_amd import modeling as M
    assert (mod.create_positional_encoding(12, 48) - M.create_positional_encoding(12, 48)).abs().max().item() < 1e-6
    assert M.create_positional_encoding(5, 8).shape == (5, 8)
