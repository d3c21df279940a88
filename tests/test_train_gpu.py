"""GPU: the training-step runtime (train.OverlappedStep + optim.FusedAdamW) through RCCL.
The box has one GPU, so the process group has ONE rank: the collectives are trivial but the
whole call path of the multi-GPU step (async reduce-scatter on RCCL's stream -> shard AdamW on
the side stream -> in-place all-gather, small-tensor coalescing, stream joins) executes for real
and must reproduce the collective-free step bit for bit."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from golden_util import load_case  # noqa: E402
from oracle import configs  # noqa: E402
from test_model_gpu import build_model, to_dev  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(dev, fx, cfg, steps, **kw):
    from macaw_llm_amd.optim import FusedAdamW
    from macaw_llm_amd.train import OverlappedStep
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()   # eval: no dropout RNG
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.01)
    rt = OverlappedStep(params, opt, small_threshold=4096, **kw)
    inp = to_dev(fx["inputs"], dev)
    losses = []
    for _ in range(steps):
        rt.begin()
        loss = model(inputs=inp).loss
        loss.backward()
        rt.finish()
        losses.append(loss.item())
    torch.cuda.synchronize()
    rt.remove()
    # parameters that are not in the golden state dict (unused towers) keep an unseeded random init
    return losses, {n: p.detach().clone() for n, p in model.named_parameters()
                    if p.requires_grad and n in fx["state"]}, rt


def test_sharded_and_allreduce_steps_match_plain_step_through_rccl(dev):
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    ref_losses, ref_params, _ = _run(dev, fx, cfg, 3)
    assert ref_losses[2] < ref_losses[0]                       # it trains
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        for shard in (True, False):
            losses, params, rt = _run(dev, fx, cfg, 3, force_collectives=True, shard_optimizer=shard)
            assert rt.collective and rt.shard == shard
            assert losses == ref_losses, (shard, losses, ref_losses)
            assert params.keys() == ref_params.keys()
            for n in params:
                assert torch.equal(params[n], ref_params[n]), (shard, n)
    finally:
        dist.destroy_process_group()
