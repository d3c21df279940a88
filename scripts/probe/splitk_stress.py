"""Stress the in-launch split-K reduction of the 128x128 GEMM kernel (csrc/gemm_impl.inc, "stream-K
tail": write-through fp32 slabs, arrival counter, last arriver reads the slabs after ONE agent-scope
acquire) for the failure the guide warns about (cdna_hip_programming.md G16: rare stale reads UNDER
UNEVEN LOAD when a hand-off protocol is subtly wrong).  Candidate cause of round 2's intermittent
one-weight mismatch between two world-2 training steps (only ever seen with three processes on the
GPU): every small GEMM of the micro model takes this path.

Several worker processes share cuda:0 (as the test's ranks do), optionally beside a burner process;
each repeats a set of split-K shapes thousands of times with cache-polluting traffic in between and
compares every result BIT FOR BIT with its first one (the reduction order is fixed, so any
difference is a stale or torn slab read).

usage: python scripts/probe/splitk_stress.py [iters] [workers] [burner 0|1]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.multiprocessing as mp


def burner(stop):
    torch.cuda.set_device(0)
    a = torch.randn(4096, 4096, device="cuda")
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    while not stop.is_set():
        for _ in range(10):
            a = (a @ a).clamp_(-1, 1)
            big.add_(1)
        torch.cuda.synchronize()


def worker(rank, iters, q):
    from macaw_llm_amd import ops
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(1234)          # same data in every worker
    # (M, N, K, a_red, b_red): tiny-MN / long-K products like the micro model's dW and conv GEMMs,
    # and tile counts that leave a partial round (K-split tail of several tiles)
    shapes = [(128, 128, 8192, False, False), (96, 352, 4096, True, True), (200, 136, 2048, False, True),
              (128, 384, 1024, True, True), (384, 640, 1536, False, False)]
    cases = []
    for (M, N, K, ar, br) in shapes:
        A = (torch.randn((K, M) if ar else (M, K), generator=g)).to(torch.bfloat16).to(dev)
        B = (torch.randn((K, N) if br else (N, K), generator=g) * 0.1).to(torch.bfloat16).to(dev)
        cases.append((M, N, K, ar, br, A, B))
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    refs, bad = [], [0] * len(cases)
    worst = [0.0] * len(cases)
    for it in range(iters):
        for ci, (M, N, K, ar, br, A, B) in enumerate(cases):
            C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            ops.gemm_raw(A, B, C, M, N, K, A.stride(0), B.stride(0), N, a_red=ar, b_red=br)
            if it == 0:
                refs.append(C.clone())
            elif not torch.equal(C, refs[ci]):
                bad[ci] += 1
                worst[ci] = max(worst[ci], (C.float() - refs[ci].float()).abs().max().item())
        if it % 4 == 0:
            junk.add_(1)                                # evict / dirty the caches between rounds
    torch.cuda.synchronize()
    q.put((rank, bad, worst))


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    burn = len(sys.argv) > 3 and sys.argv[3] == "1"
    ctx = mp.get_context("spawn")
    stop = ctx.Event()
    bp = None
    if burn:
        bp = ctx.Process(target=burner, args=(stop,))
        bp.start()
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, iters, q)) for r in range(nw)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=1500) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    stop.set()
    if bp is not None:
        bp.join(timeout=30)
    for rank, bad, worst in out:
        print(f"worker {rank}: mismatching launches per shape {bad} of {iters}; worst |diff| {worst}", flush=True)
