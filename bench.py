#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): multimodal samples/s, image + 30 s audio + 128-token
instruction, forward + backward + AdamW step (+ gradient reduction for N > 1), CLIP-ViT-L/14 +
Whisper-base + LLaMA-7B in bf16 on synthetic data — BASELINE cfg 3 at 32 samples per GPU
(global 256 at 8 GPUs, weak scaling).

    python bench.py --gpus 1 --steps 5 --warmup 2                 # cfg 3 = the metric's config
    python bench.py --config 2|4|5 ...                            # the other BASELINE.json configs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--config: 2 = image + text, B = 16 (S = 136); 3 = image + audio + text, 32 per GPU (S = 144);
4 = video (6 frames) + audio + text at sequence 2048, 4 per GPU; 5 = LLaMA-13B backbone with the
fp8 (e4m3) MFMA forward of the q|k|v and alignment K/V GEMMs, 32 per GPU, activation
checkpointing on (one GPU holds the whole 13B training state: peak memory is printed).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     : dominant kernel = the bf16 MFMA GEMM (csrc/gemm.hip); achieved = algorithmic
                 GEMM FLOPs (2*M*N*K per launch) / kernel time measured live with HIP events
                 on the launch stream during the LAST timed step; peak = 2.5 PFLOP/s dense bf16.
  cpu_baseline : the CPU oracle (oracle/restate.py, a port of the reference's algorithm; the
                 reference itself is not present on the GPU box) timed on this box's host cores,
                 composed from real-dimension components on a bounded sample (see `sample`).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TEXT_LEN = 128
PER_GPU_BATCH = 32
MFMA_BF16_PEAK = 2.5e15   # dense, /opt/skills/guides/MI355X_MICROARCH.md:42
MFMA_FP8_PEAK = 5.0e15    # dense e4m3 (MX-scaled f8f6f4 MFMA), MI355X_MICROARCH.md:43
# BASELINE.json configs 2-5 (config 1 is the CPU plumbing check).  alg_tf = algorithmic fwd+bwd
# TFLOP per sample, minimal formulation, encoders frozen as the reference's driver always does
# (5.997 / 6.42 / 94.3 - 2 x the frozen towers' forward / 12.07).
CONFIGS = {
    2: dict(model="real_7b", modalities=("images",), batch=16, text_len=128, seq=136, alg_tf=5.997,
            fp8=False, ckpt=False,
            workload="BASELINE cfg 2: CLIP-ViT-L/14 -> alignment -> LLaMA-7B, image + 128-token text (S=136), "
                     "bf16, batch 16 per GPU"),
    3: dict(model="real_7b", modalities=("images", "audios"), batch=32, text_len=128, seq=144, alg_tf=6.42,
            fp8=False, ckpt=False,
            workload="BASELINE cfg 3: CLIP-ViT-L/14 + Whisper-base + LLaMA-7B, image + 30 s audio + 128-token "
                     "text (S=144)"),
    4: dict(model="real_7b", modalities=("audios", "videos"), batch=4, text_len=2048 - 61, seq=2048, alg_tf=92.2,
            fp8=False, ckpt=False,
            workload="BASELINE cfg 4: 6 video frames (per-frame CLIP-ViT-L/14) + 30 s audio (Whisper-base) + text, "
                     "LLaMA-7B at sequence length 2048, 4 samples per GPU"),
    5: dict(model="real_13b", modalities=("images", "audios"), batch=32, text_len=128, seq=144, alg_tf=12.07,
            fp8=True, ckpt=True,
            workload="BASELINE cfg 5: LLaMA-13B backbone (D=5120, 40 layers), image + 30 s audio + 128-token text "
                     "(S=144), fp8 (e4m3) MFMA forward + grad-input of the q|k|v and alignment K/V GEMMs, activation "
                     "checkpointing on, whole training state on one GPU"),
}


# BASELINE.json's metric, verbatim (the N of "at 1/2/4/8 MI355X" is this line's n_gpus)
METRIC = "multimodal samples/sec (img+audio+128 tok) fwd+bwd at 1/2/4/8 MI355X"


def pmc_gemm_traffic(config: int, mk_gemm_launches: int = 0):
    """(bytes, source file) -- HBM-side bytes per mk_gemm launch from the committed rocprofv3 --pmc
    passes of this same command FOR THIS CONFIGURATION (profiles/rNN_step_traffic_pmc[_cfgK].csv:
    FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE, separate passes, last
    step).  PMC cannot be collected inside a timed run, so the bench line carries the profiled
    figure; (None, None) when no PMC profile of this configuration has been committed -- a number
    measured on another configuration is never substituted."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    suffix = "" if config == 3 else f"_cfg{config}"
    path = None
    for rnd in ("r06", "r05", "r04h", "r04f", "r03f", "r03", "r02", "r01"):      # newest first (rNNf / rNNh = final binary of round NN)
        cand = os.path.join(here, f"{rnd}_step_traffic_pmc{suffix}.csv")
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None, None
    try:
        gb, n = 0.0, 0
        with open(path) as f:
            for line in f:
                c = line.strip().split(",")
                if "gemm_bf16" in c[0]:
                    # the template argument list contains commas: the numeric columns are the last four
                    n += int(c[-4])
                    gb += float(c[-3]) + float(c[-2])
        # per mk_gemm LAUNCH, like `achieved`: since round 5 one mk_gemm call may be two kernels (gemm_v9 for the whole
        # tiles + the sub-tile kernel for the spatial tail), so the step's total is divided by the step's mk_gemm count
        # when the caller knows it (the profiled step and the timed step are the same command); else by kernel launches
        n = mk_gemm_launches if mk_gemm_launches > 0 else n
        return (round(gb * 1e9 / n), os.path.basename(path)) if n else (None, None)
    except (OSError, ValueError, IndexError):
        return None, None


def cpu_baseline(threads: int, spec: dict) -> dict:
    """Reference algorithm on the host cores (oracle port), composed from real-dimension
    components exactly as BASELINE.md §2 prescribes; bounded to ~20-30 s.  The alignment leg is
    the reference's per-sample nn.MultiheadAttention over the full token table (forward + backward,
    timed once); the frozen towers run forward only (errs in the CPU's favour).  kind = "port":
    /root/reference does not exist on the GPU box (and this file needs the GPU), so the reference
    itself can never be the thing timed here; oracle/restate.py is pinned to it by tests/test_oracle.py."""
    import torch.nn.functional as F
    from oracle import restate
    from macaw_llm_amd.factory import baseline_config
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    mcfg = baseline_config(spec["model"])
    ll = mcfg["llama"]
    D, FF, H, V = ll["hidden_size"], ll["intermediate_size"], ll["num_attention_heads"], 32007
    L, S = ll["num_hidden_layers"], spec["seq"]
    mods = spec["modalities"]
    n_clip = (1 if "images" in mods else 0) + (mcfg["mm"]["n_frames"] if "videos" in mods else 0)
    n_wh = 1 if "audios" in mods else 0

    def rnd(*shape, s=0.02):
        return torch.randn(*shape, generator=g) * s

    def timeit(fn, reps=2):
        t0 = time.perf_counter()
        fn()                       # warm-up (allocator, thread pool); also the fallback sample
        first = time.perf_counter() - t0
        if first > 3.0:            # keep the whole baseline leg bounded (~10-30 s of CPU work)
            return first
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    # one LlamaDecoderLayer fwd+bwd at S=144, B=1 (modeling.py:234-299)
    p = "l."
    sd = {p + "self_attn.q_proj.weight": rnd(D, D), p + "self_attn.k_proj.weight": rnd(D, D),
          p + "self_attn.v_proj.weight": rnd(D, D), p + "self_attn.o_proj.weight": rnd(D, D),
          p + "mlp.gate_proj.weight": rnd(FF, D), p + "mlp.up_proj.weight": rnd(FF, D),
          p + "mlp.down_proj.weight": rnd(D, FF), p + "input_layernorm.weight": torch.ones(D),
          p + "post_attention_layernorm.weight": torch.ones(D)}
    for v in sd.values():
        v.requires_grad_(True)
    x = rnd(1, S, D, s=1.0).requires_grad_(True)
    cos, sin = restate.rotary_tables(D // H, 2048)
    mask = restate.decoder_mask(torch.ones(1, S, dtype=torch.long), 1, S, torch.float32, x.device)
    pos = torch.arange(S)[None]

    def layer():
        y = restate.llama_layer(sd, p, x, mask, pos, H, 1e-6, cos, sin)
        y.sum().backward()
    t_layer = timeit(layer)
    del sd
    # final norm + lm_head + CE
    Wlm = rnd(V, D).requires_grad_(True)
    nw = torch.ones(D, requires_grad=True)
    lab = torch.randint(0, V, (S,), generator=g)

    def head():
        F.cross_entropy(F.linear(restate.rms_norm(x[0], nw, 1e-6), Wlm), lab).backward()
    t_head = timeit(head)
    del Wlm
    # alignment attention EXACTLY as the reference formulates it (modeling.py:974-975, 986): per sample
    # and per modality the whole [V, D] table is the key/value input of nn.MultiheadAttention -- K/V
    # projection of all 32,007 rows, bias_k / zero rows, scores, softmax, PV, out-proj -- forward and
    # backward (the table and the in-proj weight receive gradients).  Timed ONCE on the full table
    # (no sampling, no scaling); the Lq = 6 pooled tokens of the image / audio prefix.
    E = rnd(V, D).requires_grad_(True)
    asd = {"a.in_proj_weight": rnd(3 * D, D).requires_grad_(True), "a.in_proj_bias": torch.zeros(3 * D),
           "a.bias_k": rnd(1, 1, D), "a.bias_v": rnd(1, 1, D), "a.out_proj.weight": rnd(D, D),
           "a.out_proj.bias": torch.zeros(D)}
    qin = rnd(6, 1, D, s=1.0)
    t0 = time.perf_counter()
    restate.mha_forward(asd, "a.", qin, E[:, None, :], E[:, None, :], 2 * mcfg["mm"]["attention_heads"]).sum().backward()
    t_align = time.perf_counter() - t0
    del E, asd
    # frozen encoders: forward only (run_clm_llms.py:390-393)
    vc, wc = mcfg["clip"]["vision_config"], mcfg["whisper"]
    csd = {}
    pv = "v."
    Ed, Fd = vc["hidden_size"], vc["intermediate_size"]
    csd[pv + "embeddings.patch_embedding.weight"] = rnd(Ed, 3, 14, 14)
    csd[pv + "embeddings.class_embedding"] = rnd(Ed)
    csd[pv + "embeddings.position_embedding.weight"] = rnd(257, Ed)
    for n in ("pre_layrnorm",):
        csd[pv + n + ".weight"], csd[pv + n + ".bias"] = torch.ones(Ed), torch.zeros(Ed)
    for i in range(vc["num_hidden_layers"]):
        lp = f"{pv}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            csd[lp + f"self_attn.{n}.weight"], csd[lp + f"self_attn.{n}.bias"] = rnd(Ed, Ed), torch.zeros(Ed)
        for n in ("layer_norm1", "layer_norm2"):
            csd[lp + n + ".weight"], csd[lp + n + ".bias"] = torch.ones(Ed), torch.zeros(Ed)
        csd[lp + "mlp.fc1.weight"], csd[lp + "mlp.fc1.bias"] = rnd(Fd, Ed), torch.zeros(Fd)
        csd[lp + "mlp.fc2.weight"], csd[lp + "mlp.fc2.bias"] = rnd(Ed, Fd), torch.zeros(Ed)
    img = torch.randn(1, 3, 224, 224, generator=g)
    with torch.no_grad():
        t_clip = timeit(lambda: restate.clip_vision_forward(csd, pv, img, vc), reps=1)
    del csd
    wsd = {}
    pw = "w."
    dm, ffn = wc["d_model"], wc["encoder_ffn_dim"]
    wsd[pw + "conv1.weight"], wsd[pw + "conv1.bias"] = rnd(dm, 80, 3), torch.zeros(dm)
    wsd[pw + "conv2.weight"], wsd[pw + "conv2.bias"] = rnd(dm, dm, 3), torch.zeros(dm)
    wsd[pw + "embed_positions.weight"] = rnd(1500, dm)
    wsd[pw + "layer_norm.weight"], wsd[pw + "layer_norm.bias"] = torch.ones(dm), torch.zeros(dm)
    for i in range(wc["encoder_layers"]):
        lp = f"{pw}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            wsd[lp + f"self_attn.{n}.weight"], wsd[lp + f"self_attn.{n}.bias"] = rnd(dm, dm), torch.zeros(dm)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            wsd[lp + n + ".weight"], wsd[lp + n + ".bias"] = torch.ones(dm), torch.zeros(dm)
        wsd[lp + "fc1.weight"], wsd[lp + "fc1.bias"] = rnd(ffn, dm), torch.zeros(ffn)
        wsd[lp + "fc2.weight"], wsd[lp + "fc2.bias"] = rnd(dm, ffn), torch.zeros(dm)
    mel = torch.randn(1, 80, 3000, generator=g)
    with torch.no_grad():
        t_wh = timeit(lambda: restate.whisper_encoder_forward(wsd, pw, mel, wc), reps=1)
    total = L * t_layer + t_head + n_clip * t_clip + n_wh * t_wh + len(mods) * t_align
    return dict(value=1.0 / total, unit="samples/s", cores=threads, kind="port",
                sample=("composed from real-dimension components, B=1, fp32, reference formulation: "
                        f"{L} x LlamaDecoderLayer f+b at S={S} ({t_layer:.3f}s each) + norm/lm_head/CE f+b "
                        f"({t_head:.3f}s) + {n_clip} x CLIP-L/14 fwd ({t_clip:.3f}s) + {n_wh} x Whisper-base fwd "
                        f"({t_wh:.3f}s) + {len(mods)} x per-sample alignment nn.MultiheadAttention f+b over the FULL "
                        f"32,007-row table incl. scores / softmax / PV / out-proj ({t_align:.3f}s each, timed once, "
                        "not sampled)"),
                seconds_per_sample=total)


def host_cores() -> int:
    """Cores this process may actually use: min(affinity, cgroup cpu.max quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MACAW_CPU_THREADS", "64"))))


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(n: int, argv: list, port: int | None = None) -> list:
    """the command a bare `python bench.py --gpus N ...` re-executes itself through: one process per GPU on this
    node, rendezvous on 127.0.0.1 (the container hostname may not resolve), same arguments"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.abspath(__file__), *argv]


# DESIGN.md section 6's prediction for the metric's configuration (cfg 3, 32 samples per GPU), so that the first
# measured 1/2/4/8 curve can be read against it from the line alone.  N = 2 is ONE xGMI link between the pair
# (2 x 6.7 GB per step over ~60 GB/s): expected to scale badly whatever the software does.
PREDICTED_CFG3 = {
    1: dict(step_ms=[220, 221], speedup=[1.0, 1.0], note="no collective; one fused AdamW launch behind the backward"),
    2: dict(step_ms=[305, 325], speedup=[1.35, 1.45], rs_ag_ms_per_bucket=[6.7, 6.7],
            note="link-bound: one xGMI link (7 links go to 7 different peers), the collectives outlast the backward"),
    4: dict(step_ms=[230, 240], speedup=[3.7, 3.8], rs_ag_ms_per_bucket=[3.6, 3.6], note="3 links per GPU in use"),
    8: dict(step_ms=[222, 270], speedup=[6.5, 7.9], rs_ag_ms_per_bucket=[2.0, 3.1],
            note="222-230 ms if the collectives hide behind the backward as in the 1-rank overlap experiment "
                 "(profiles/r05_overlap_1rank.txt), 270 ms if only half of them do"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (3 = the metric's configuration, the default)")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="override the configuration's per-GPU batch")
    ap.add_argument("--model", default=None, help="override the configuration's backbone (real_7b / real_13b)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="force the cfg 5 precision (q|k|v and alignment K/V "
                    "forward GEMMs on the fp8 MFMA path) on another configuration")
    ap.add_argument("--no-fp8", action="store_true", help="cfg 5 in pure bf16 (the yardstick of its fp8 speed-up)")
    ap.add_argument("--fp8-mlp", action="store_true", help="extend the fp8 path to the gate|up / down GEMMs (beyond "
                    "BASELINE cfg 5's wording; reported separately)")
    ap.add_argument("--checkpoint", action="store_true", help="force activation checkpointing of the decoder layers")
    ap.add_argument("--layers", type=int, default=None, help="debug only: truncates the LLaMA stack "
                    "(the printed line is then marked invalid)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N` (how the driver invokes N = 1): start the N ranks ourselves, exactly as
        # train.sh:13 does (torchrun --nnodes 1 --nproc_per_node N); rank 0's JSON line and the ranks' stderr pass
        # straight through, the exit code is the launcher's
        import signal
        proc = subprocess.Popen(self_launch_command(args.gpus, sys.argv[1:]))
        for sig in (signal.SIGTERM, signal.SIGINT):       # a driver that stops us stops the ranks (torchrun tears them down)
            signal.signal(sig, lambda s, _f: proc.send_signal(s))
        raise SystemExit(proc.wait())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # MACAW_SHARE_GPU=1 + MACAW_DIST_BACKEND=gloo: N ranks on ONE GPU with gloo collectives -- how the
    # N > 1 branch of this file is exercised on a 1-GPU box (tests/test_train_gpu.py); never a benchmark
    share_gpu = bool(os.environ.get("MACAW_SHARE_GPU"))
    backend = os.environ.get("MACAW_DIST_BACKEND", "nccl")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_coll = bool(os.environ.get("MACAW_FORCE_COLLECTIVES"))   # 1-rank RCCL group: call-path check
    if world > 1 or force_coll:
        os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")     # per-collective device time for `comm` (below)
        # each RCCL channel is a resident workgroup that holds a CU for the whole collective; the 256 x 256 GEMM
        # needs whole CUs (scripts/probe/cu_hold.cpp, profiles/r04_cu_hold.csv): cap the channels and let the
        # GEMMs plan for the CUs that remain (BucketedStep(comm_cus=...)).  16 channels over 7 xGMI links.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        kw = dict(device_id=dev) if backend == "nccl" else {}
        if force_coll and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group(backend=backend, rank=0, world_size=1, **kw)
        else:
            dist.init_process_group(backend=backend, **kw)

    from macaw_llm_amd import ops
    from macaw_llm_amd.bucketed import BucketedStep, default_comm_cus
    from macaw_llm_amd.factory import baseline_config, build_model, synthetic_inputs
    from macaw_llm_amd.optim import FusedAdamW

    spec = dict(CONFIGS[args.config])
    if args.model:
        spec["model"] = args.model
    if args.batch_per_gpu:
        spec["batch"] = args.batch_per_gpu
    spec["fp8"] = (spec["fp8"] or args.fp8 or args.fp8_mlp) and not args.no_fp8
    spec["ckpt"] = spec["ckpt"] or args.checkpoint
    cfg = baseline_config(spec["model"])
    if args.layers is not None:
        cfg["llama"]["num_hidden_layers"] = args.layers
    model = build_model(cfg, dtype=torch.bfloat16, device=dev, seed=1234).train()
    if spec["fp8"]:
        model.set_fp8(qkv=True, align=True, mlp=args.fp8_mlp)
    if spec["ckpt"]:
        model.llm.model.gradient_checkpointing = True     # modeling.py:474-489
    params = [p for p in model.parameters() if p.requires_grad]

    # ONE step runtime at every N (macaw_llm_amd/bucketed.py): flat buckets; N > 1 = ZeRO-1 (one
    # reduce-scatter + shard AdamW + one all-gather per ~768 MiB bucket behind the backward, fixed
    # rank-invariant order), N = 1 = the same buckets with one fused AdamW launch and no collectives.
    # MACAW_FORCE_COLLECTIVES=1 runs the collective path through a 1-rank RCCL group.
    def make_runtime(zero1=True):
        o = FusedAdamW(params, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        return BucketedStep(params, o, model=model, overlap=not os.environ.get("MACAW_NO_OVERLAP"),
                            force_collectives=force_coll, zero1=zero1,
                            comm_cus=default_comm_cus() if (world > 1 or force_coll) else 0,
                            bucket_bytes=int(os.environ.get("MACAW_BUCKET_MB", "768")) << 20)

    runtime = make_runtime()
    B = spec["batch"]
    inputs = synthetic_inputs(cfg, B, spec["text_len"], modalities=spec["modalities"], seed=1 + rank, device=dev)

    # MACAW_STEP_GRAPH=1 (N = 1): the whole step (zero grads, forward, backward, fused AdamW) is
    # captured ONCE in a hipGraph and replayed (macaw_llm_amd.train.GraphedStep: bit-identical to the
    # eager step, the optimizer scalars and the dropout seed offset live in device memory); the
    # setup step and the LAST timed step run eagerly, the latter because the per-launch HIP events
    # behind `roofline` cannot be recorded inside a replay.  Measured 249.1 vs 249.6 ms per step: the
    # 7 ms of inter-kernel gaps in the rocprofv3 traces are the tracer's, not the eager step's --
    # hence off by default.
    graphed = None
    if world == 1 and not force_coll and os.environ.get("MACAW_STEP_GRAPH"):
        from macaw_llm_amd.train import GraphedStep
        graphed = GraphedStep(model, lambda: model(inputs=inputs).loss, runtime)

    def step(eager=False):
        if graphed is not None:
            return graphed.eager_step() if eager else graphed.step()
        runtime.begin()                      # zero grads, advance Adam's step counter
        loss = model(inputs=inputs).loss
        loss.backward()                      # hooks: bucket collectives + shard AdamW behind the backward (N > 1)
        runtime.finish()                     # remaining buckets, the fused AdamW launch (N = 1), stream joins
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one untimed SETUP step: materialises the optimizer state (fp32 master / m / v, 84 GB at 7B),
    # the allocator pools and the frozen bucket order, like building the model.  The W warm-up steps
    # follow.  N > 1: the ZeRO-1 collectives are the one path a 1-GPU pool cannot run on real RCCL.
    # If the setup step raises (the realistic N > 1 failures are symmetric: an unsupported collective,
    # an out-of-memory at the same point, a bug on the path -- every rank raises in the same place),
    # ALL ranks agree on it (the verdict travels through a gloo side group, which does not depend on
    # the state of the RCCL communicator) and continue together on all-reduce + replicated AdamW
    # (zero1=False), and the line SAYS SO in `config.parallelism`: a degraded run must not pass as
    # ZeRO-1; a failure at N = 1 is an error.  (A rank that dies ALONE leaves the others inside a
    # collective; that ends at the process group's watchdog timeout, as in any RCCL job.)
    degraded = None
    side_group = None
    if world > 1 and dist.get_backend() != "gloo":
        try:
            side_group = dist.new_group(backend="gloo")
        except Exception as e:      # noqa: BLE001  (no usable TCP interface for gloo: agree over the default group)
            print(f"[bench] rank {rank}: no gloo side group ({e!r}); the setup verdict travels over RCCL", file=sys.stderr)
    err = None
    inject = os.environ.get("MACAW_BENCH_INJECT_FAIL")     # test hook: "all" fails every rank's ZeRO-1 setup step
    try:
        if inject is not None and (inject == "all" or inject == str(rank)):
            raise RuntimeError("injected setup-step failure (MACAW_BENCH_INJECT_FAIL)")
        step()
        torch.cuda.synchronize()
    except Exception as e:       # noqa: BLE001
        if world == 1:
            raise
        err = repr(e)[:300]
    if world > 1:
        on_dev = side_group is None and dist.get_backend() != "gloo"
        flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev if on_dev else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=side_group)
        if int(flag.item()):
            degraded = err or "the setup step failed on another rank"
            print(f"[bench] rank {rank}: ZeRO-1 setup step failed ({degraded}); all ranks continue on all-reduce + "
                  "replicated AdamW (recorded in config.parallelism)", file=sys.stderr, flush=True)
            runtime.remove()
            runtime = make_runtime(zero1=False)
            step()
    for _ in range(args.warmup):
        l0 = step()
        if os.environ.get("MACAW_BENCH_VERBOSE") and rank == 0:
            print(f"[warmup] loss {float(l0.detach()):.4f}", file=sys.stderr)
    fence()
    dw_side_used = False
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = i == args.steps - 1
        if last:
            # the per-kernel figures (`roofline`) come from THIS step: it updates behind the backward in one launch, so
            # that no kernel of another stream runs beside the GEMMs and a launch's duration is its own (the other
            # steps overlap the per-bucket AdamW launches with the backward on one rank: BucketedStep.local_overlap)
            runtime.serial_update = True
            from macaw_llm_amd import engine as _engine
            dw_side_used = bool(_engine.DW_SIDE["streams"])   # did the steps so far put grad-weight GEMMs on a second stream?
            _engine.DW_SIDE["on"] = False        # (grad-weight GEMMs back on the compute stream for this step, see engine.DW_SIDE)
            _engine.ENC_SIDE["on"] = False       # (and the audio tower behind the image tower: one stream)
            ops.prof_begin()
            runtime.profile_comm(True)
        loss = step(eager=last)
    fence()
    dt = time.perf_counter() - t0
    comm = runtime.comm_report()
    runtime.profile_comm(False)
    digest = None
    if os.environ.get("MACAW_BENCH_DIGEST") and rank == 0:
        # (after the timed region) sha1 over the trained LLaMA layer 0 / last layer / lm_head weights and the exact loss bits: two
        # runs that must be bit-identical (e.g. MACAW_DW_STREAM=0 against 1) are compared on this
        import hashlib
        hsh = hashlib.sha1()
        for n, p in model.named_parameters():
            if p.requires_grad and (".layers.0." in n or f".layers.{len(model.llm.model.layers) - 1}." in n or "lm_head" in n):
                hsh.update(n.encode())
                hsh.update(p.detach().view(torch.int16 if p.element_size() == 2 else torch.int32).cpu().numpy().tobytes())
        digest = {"params_sha1": hsh.hexdigest(), "loss_hex": float(loss.detach()).hex()}
    if os.environ.get("MACAW_GEMM_REPORT") and rank == 0:
        ops.prof_report(os.environ["MACAW_GEMM_REPORT"])
    att_f, att_b, gemm8 = ops.prof_sum(1), ops.prof_sum(2), ops.prof_sum(3)
    gemm_ms, gemm_flops, gemm_n = ops.prof_end()
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    value = world * B / (dt / args.steps)

    if rank == 0:
        S = spec["seq"]
        alg_tf = spec["alg_tf"]
        traffic, traffic_src = pmc_gemm_traffic(args.config, gemm_n + gemm8[2])
        # all mk_gemm launches: bf16 (kind 0) + fp8 (kind 3, cfg 5).  `frac` prices every launch against
        # ITS OWN dense peak (2.5 PFLOP/s bf16, 5 PFLOP/s e4m3 -- MI355X_MICROARCH.md:42-43): the time the
        # step's GEMM FLOPs would take at peak / the time they took.
        ms8, fl8, n8 = gemm8
        tot_ms, tot_fl = gemm_ms + ms8, gemm_flops + fl8
        achieved = tot_fl / (tot_ms * 1e-3) if tot_ms > 0 else 0.0
        gemm_frac = ((gemm_flops / MFMA_BF16_PEAK + fl8 / MFMA_FP8_PEAK) / (tot_ms * 1e-3)) if tot_ms > 0 else 0.0

        def rate(t):      # (ms, flops, launches) -> dict
            ms, fl, n = t
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"achieved": round(tf, 1), "frac": round(tf * 1e12 / MFMA_BF16_PEAK, 4), "ms_per_step": round(ms, 3),
                    "launches_per_step": n}
        line = {
            "metric": METRIC,
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not spec["fp8"] else
                     ("bf16 + fp8 (e4m3, per-row / per-channel scales) forward and grad-input GEMMs of q|k|v and the "
                      "alignment K/V projection" + (" and of gate|up / down" if args.fp8_mlp else "")
                      + "; grad-weight GEMMs, attention and everything else bf16"),
            "data": "synthetic",
            "config": {"workload": (spec["workload"] + "; fwd+bwd+fused AdamW, encoders frozen as "
                                    "run_clm_llms.py:390-393, alignment-attention dropout on"),
                       "baseline_config": args.config,
                       "global_batch": world * B, "per_gpu_batch": B, "seq_len": S,
                       "parallelism": f"dp{world}: " + runtime.describe()
                                      + (f": DEGRADED from ZeRO-1 (its setup step failed: {degraded})" if degraded else ""),
                       "step_launch": ("hipGraph replay of the whole step (train.GraphedStep); setup step and the "
                                       "last timed step eager (per-launch HIP events)") if graphed is not None
                                      else "eager, kernel by kernel",
                       "grad_weight_side_stream": ("on except in the instrumented last step (engine.DW_SIDE auto: the [M, D] "
                                                   "grad-input GEMMs leave CUs idle at this shape)") if dw_side_used
                                                  else "off (engine.DW_SIDE auto)",
                       "setup_steps": 1, "activation_checkpointing": bool(spec["ckpt"]),
                       "peak_mem_gib": round(peak_mem, 1),
                       "loss": round(float(loss.detach()), 4),
                       **({"digest": digest} if digest else {})},
            "roofline": {"bound": "mfma", "kernel": "all mk_gemm launches of the step: gemm_bf16_v7_kernel (256x256, 8 waves, "
                                                    "csrc/gemm_v7.hip; its <.., FP8> instantiation for e4m3 operands) + "
                                                    "gemm_bf16_v9_kernel (256x256, 4 waves, generated inline-asm K loop, "
                                                    "csrc/gemm_v9.hip: grad-weight and grad-input) + "
                                                    "gemm_bf16_v2_kernel (128x128, csrc/gemm.hip)",
                         "measured_on": ("the LAST timed step, HIP events around every launch on the compute stream; that "
                                         "step updates behind the backward in one AdamW launch, so no kernel of another "
                                         "stream runs beside the GEMMs"
                                         + (" (the other timed steps overlap the per-bucket AdamW launches with the "
                                            "backward on a high-priority side stream)"
                                            if getattr(runtime, "local_overlap", False) else "")),
                         "achieved": round(achieved / 1e12, 2), "peak": MFMA_BF16_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": round(gemm_frac, 4),
                         "traffic": traffic,
                         "traffic_note": (f"HBM-side bytes per mk_gemm launch of the cfg {args.config} step (rocprofv3 "
                                          f"PMC, profiles/{traffic_src}; includes Infinity-Cache hits)") if traffic
                                         else f"no PMC profile of cfg {args.config} committed under profiles/",
                         "launches_per_step": gemm_n + n8, "gemm_ms_per_step": round(tot_ms, 3),
                         "gemm_tflop_per_step": round(tot_fl / 1e12, 2),
                         **({"fp8_gemms": {"launches_per_step": n8, "ms_per_step": round(ms8, 3),
                                           "achieved": round(fl8 / (ms8 * 1e-3) / 1e12, 1) if ms8 > 0 else 0.0,
                                           "peak": MFMA_FP8_PEAK / 1e12,
                                           "frac": round(fl8 / (ms8 * 1e-3) / MFMA_FP8_PEAK, 4) if ms8 > 0 else 0.0},
                             "bf16_gemms": {"launches_per_step": gemm_n, "ms_per_step": round(gemm_ms, 3),
                                            "achieved": round(gemm_flops / (gemm_ms * 1e-3) / 1e12, 1) if gemm_ms > 0 else 0.0,
                                            "frac": round(gemm_flops / (gemm_ms * 1e-3) / MFMA_BF16_PEAK, 4) if gemm_ms > 0 else 0.0},
                             "frac_note": "frac = (bf16 FLOPs / 2.5 PF + fp8 FLOPs / 5 PF) / GEMM time: each launch "
                                          "priced against its own dense peak"} if n8 else {}),
                         # context, not the contract's peak: a register-only MFMA loop sustains 2475 TFLOP/s on all-zero
                         # operands but 1767-1781 on random bf16 data on this part (power-bound; profiles/r06_mfma_power.txt)
                         **({"real_data_mfma_ceiling": {"tflops": 1775.0, "frac_of_it": round(achieved / 1775.0e12, 4),
                                                        "source": "scripts/probe/mfma_power.hip, profiles/r06_mfma_power.txt"}}
                            if not n8 else {}),          # (bf16 runs only: the probe measured the bf16 pipe)
                         "whole_step_model_tflops": round(value / world * alg_tf, 1),
                         "whole_step_frac": round(value / world * alg_tf * 1e12 / MFMA_BF16_PEAK, 4),
                         # fused attention kernels (csrc/attention.hip), algorithmic FLOPs (causal =
                         # lower triangle), same live HIP-event timing
                         "attention_fwd": rate(att_f), "attention_bwd": rate(att_b)},
        }
        # how the step's communication went on rank 0 (last timed step): the un-overlapped tail behind the backward
        # and every bucket's reduce-scatter / all-gather duration and bytes -- see BucketedStep.comm_report()
        line["comm"] = comm
        if args.config == 3 and world in PREDICTED_CFG3 and B == CONFIGS[3]["batch"]:
            pred = dict(PREDICTED_CFG3[world], source="DESIGN.md section 6 (model: RCCL ring bus bandwidth 230-350 GB/s at "
                        "N = 8, ~170 at N = 4, ~60 at N = 2; 27 ms of step time per 100 ms of overlapped side work)")
            if isinstance(line["comm"], dict):
                line["comm"]["predicted"] = pred
                line["comm"]["predicted_ms"] = pred["step_ms"]
            else:
                line["comm"] = {"world": world, "collective": "none", "predicted": pred, "predicted_ms": pred["step_ms"]}
        if args.layers is not None:
            line["invalid"] = f"debug run with --layers {args.layers}"
        if share_gpu or backend != "nccl" or inject is not None:
            line["invalid"] = "call-path test (ranks sharing one GPU / gloo collectives / injected failure), not a benchmark"
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(host_cores(), spec)
                line["cpu_baseline"] = cb
                line["gpu_over_cpu"] = round(value / cb["value"], 1)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
