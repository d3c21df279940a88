"""splitk_stress.py with TWO streams per worker: the in-launch split-K reduction (write-through slabs, arrival counter, one agent
acquire) while ANOTHER GEMM of the same process runs beside it on a second stream -- the situation engine.DW_SIDE creates.  Round 6:
with the side stream forced on, the world-2 test of the retired per-tensor runtime deviated in ONE 128-row tile of a micro-model
grad-weight GEMM (a split-K launch) in 2 of 5 full-suite runs.  Every result is compared BIT FOR BIT with the first one of its case.

usage: python scripts/probe/splitk_stress2.py [iters] [workers] [burner 0|1]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.multiprocessing as mp

from splitk_stress import burner


def worker(rank, iters, q):
    from macaw_llm_amd import ops
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(1234)
    # micro-model grad-weight / grad-input shapes (tokens ~ 100-300 as the reduction of dW) and the original tail shapes
    shapes = [(256, 64, 144, True, True), (192, 64, 272, True, True), (144, 64, 256, False, True), (272, 256, 64, False, True),
              (96, 352, 4096, True, True), (128, 384, 1024, True, True), (200, 136, 2048, False, True),
              (128, 128, 8192, False, False)]
    cases = []
    for (M, N, K, ar, br) in shapes:
        A = (torch.randn((K, M) if ar else (M, K), generator=g)).to(torch.bfloat16).to(dev)
        B = (torch.randn((K, N) if br else (N, K), generator=g) * 0.1).to(torch.bfloat16).to(dev)
        cases.append((M, N, K, ar, br, A, B))
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    refs = [None] * len(cases)
    bad, worst = [0] * len(cases), [0.0] * len(cases)
    n = len(cases)
    for it in range(iters):
        outs = [None] * n
        for ci in range(0, n, 2):
            # case ci on the main stream, case ci + 1 beside it on the side stream (submitted back to back)
            for cj, st in ((ci, main), (ci + 1, side)):
                if cj >= n:
                    continue
                M, N, K, ar, br, A, B = cases[(cj + it) % n] if False else cases[cj]
                with torch.cuda.stream(st):
                    C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
                    ops.gemm_raw(A, B, C, M, N, K, A.stride(0), B.stride(0), N, a_red=ar, b_red=br)
                outs[cj] = C
        main.wait_stream(side)
        for ci in range(n):
            if refs[ci] is None:
                refs[ci] = outs[ci].clone()
            elif not torch.equal(outs[ci], refs[ci]):
                bad[ci] += 1
                worst[ci] = max(worst[ci], (outs[ci].float() - refs[ci].float()).abs().max().item())
        if it % 4 == 0:
            junk.add_(1)
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
        print(f"worker {rank}: mismatches per case {bad} (of {iters - 1} repeats), worst |diff| {worst}")
    print("TOTAL MISMATCHES", sum(sum(b) for _, b, _ in out))
