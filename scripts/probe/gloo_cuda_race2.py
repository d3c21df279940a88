"""VARIANT 2 (round 6): the input of every collective is produced on a THIRD stream (behind the matmul chain there) and the main
stream only waits for it through an event (`main.wait_stream(prod)`) right before the collective is issued -- the pattern
engine.DW_SIDE creates in a two-rank run (grad-weight GEMM on a side stream, stream-ordered join, gradient hook -> gloo).
Is gloo's staging of a CUDA tensor ordered behind an EVENT WAIT of the issuing stream, or only behind kernels of that stream?

Root-cause probe for the intermittent world-2 mismatch of tests/test_train_gpu.py (round 2: "one
weight differed twice in ~40 runs, replicas identical, only inside the full pytest process").

Two ranks share cuda:0 and talk through gloo, as the test does.  Each iteration produces a known
input LATE on the main stream (behind a long matmul chain), issues one collective asynchronously and
consumes the result IMMEDIATELY on a side stream after `work.wait()` -- the exact pattern of
BucketedStep._launch / _finish_bucket and OverlappedStep._reduce_scatter.  Values are chosen so
that every outcome is attributable:

    input of rank r at iteration i :  (i % 97) + 1 + 1000 * r      (exactly representable in bf16? no:
                                       fp32 tensors are used; the collectives are dtype agnostic)
    stale / unreduced / uninitialised outputs differ from the expected sum in recognisable ways.

usage: python scripts/probe/gloo_cuda_race.py [iters] [burner 0|1]
prints one line per primitive: mismatching iterations and what the wrong value looked like.
"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def burner(stop):
    torch.cuda.set_device(0)
    a = torch.randn(4096, 4096, device="cuda")
    while not stop.is_set():
        for _ in range(20):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()


def worker(rank, world, port, iters, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    side = torch.cuda.Stream(device=dev)
    prod = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    n = 1 << 20
    big = torch.randn(2048, 2048, device=dev)
    res = {}
    try:
        for prim in ("reduce_scatter_tensor", "all_reduce", "all_gather_into_tensor"):
            bad = []
            inp = torch.empty(n, dtype=torch.float32, device=dev)
            out = torch.full((n // world,), -7.0, dtype=torch.float32, device=dev)
            full = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
            for i in range(iters):
                val = float((i % 97) + 1 + 1000 * rank)
                want_sum = float(2 * ((i % 97) + 1) + 1000)
                prod.wait_stream(main)
                with torch.cuda.stream(prod):
                    t = big
                    for _ in range(6):           # the input becomes ready late on the PRODUCER stream
                        t = (t @ big).clamp_(-1, 1)
                    inp.fill_(val)
                    inp[0] += t[0, 0] * 0        # data dependence on the chain
                    if prim == "all_gather_into_tensor":
                        m_ = n // world
                        full.fill_(-7.0)
                        full[rank * m_:(rank + 1) * m_].fill_(val)
                main.wait_stream(prod)           # the ONLY ordering between the producer and what follows: an event wait
                if prim == "reduce_scatter_tensor":
                    out.fill_(-7.0)
                    h = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, async_op=True)
                    with torch.cuda.stream(side):
                        h.wait()
                        got = out.clone()
                    exp = want_sum
                elif prim == "all_reduce":
                    h = dist.all_reduce(inp, op=dist.ReduceOp.SUM, async_op=True)
                    with torch.cuda.stream(side):
                        h.wait()
                        got = inp.clone()
                    exp = want_sum
                else:
                    m = n // world
                    h = dist.all_gather_into_tensor(full, full[rank * m:(rank + 1) * m], async_op=True)
                    with torch.cuda.stream(side):
                        h.wait()
                        got = full.clone()
                    exp = None
                side.synchronize()
                torch.cuda.synchronize()
                if exp is not None:
                    wrong = got != exp
                else:
                    m = n // world
                    ref = torch.cat([torch.full((m,), float((i % 97) + 1 + 1000 * r), device=dev) for r in range(world)])
                    wrong = got != ref
                if bool(wrong.any()):
                    vals = got[wrong][:3].tolist()
                    bad.append((i, int(wrong.sum()), vals))
                    # is the result correct a moment later?  (= the consumer ran too early)
                    time.sleep(0.05)
                    torch.cuda.synchronize()
                    late = (out if prim == "reduce_scatter_tensor" else inp if prim == "all_reduce" else full)
                    late_ok = bool((late == exp).all()) if exp is not None else bool((late == ref).all())
                    bad[-1] = bad[-1] + ("correct after 50 ms" if late_ok else "still wrong",)
            res[prim] = bad
        q.put((rank, res))
    except Exception as e:      # noqa: BLE001
        q.put((rank, {"error": repr(e)}))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    burn = len(sys.argv) > 2 and sys.argv[2] == "1"
    ctx = mp.get_context("spawn")
    stop = ctx.Event()
    bp = None
    if burn:
        bp = ctx.Process(target=burner, args=(stop,))
        bp.start()
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, iters, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted((q.get(timeout=1200) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
    stop.set()
    if bp is not None:
        bp.join(timeout=30)
    for rank, res in out:
        for prim, bad in res.items():
            print(f"rank {rank} {prim}: {len(bad) if isinstance(bad, list) else bad} bad of {iters}"
                  + (f"  first: {bad[:3]}" if isinstance(bad, list) and bad else ""), flush=True)
