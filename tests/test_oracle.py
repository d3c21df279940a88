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


needs_ref = pytest.mark.skipif(not ref_loader.reference_available(), reason="/root/reference absent")


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
