"""run the two world-2 gloo-on-one-GPU steps several times and print the digest of one weight"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch.multiprocessing as mp
import test_train_gpu as T

def run(target, extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    return res

if __name__ == "__main__":
    ref = None
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for name, tgt, extra in (("per-tensor", T._worker_two_ranks_one_gpu, (True,)), ("bucketed", T._worker_bucketed_two_ranks, ())):
            res = run(tgt, extra)
            if res[0][1] == "error":
                print(name, "ERROR", res[0][2][:200]); continue
            d = res[0][2]
            if ref is None: ref = d
            bad = [n for n in d if d[n] != ref[n]]
            print(i, name, "replicas equal" if res[0][2] == res[1][2] else "REPLICAS DIFFER", "differs from first run in:", bad[:6], len(bad), flush=True)
