"""Reads a rocprofv3 kernel-trace CSV and reports the LAST training step of a run whose optimizer kernels are
NOT a single trailing launch (the collective path: shard AdamW on a side stream behind the backward, RCCL
kernels in between), i.e. where scripts/trace_gaps.py cannot find its step boundary.  A step starts at the
CLIP `patchify_kernel` (one per step, first kernel of the forward).

Prints: wall time of the step, union of busy time over all queues, per-queue busy time, RCCL / copy kernels,
and the top kernels by total duration -- next to profiles/rNN_cfg3_last_step.txt (the collective-free step) this
names where the extra time of the collective path goes."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
rows.sort()
starts = [i for i, r in enumerate(rows) if "patchify_kernel" in r[2]]
if len(starts) < 2:
    sys.exit("fewer than two steps in the trace")
a, b = starts[-2], starts[-1]          # the last COMPLETE step in the trace: [second-to-last patchify, last patchify)
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
busy = sum(r[1] - r[0] for r in step)
# union of the busy intervals (kernels on different queues overlap)
iv = sorted((r[0], r[1]) for r in step)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"step: {len(step)} kernels, wall {1e-6 * (t1 - t0):.2f} ms, sum of kernel durations {1e-6 * busy:.2f} ms, "
      f"union of busy time {1e-6 * union:.2f} ms (idle {1e-6 * (t1 - t0 - union):.2f} ms, overlapped "
      f"{1e-6 * (busy - union):.2f} ms)")
perq = {}
for s, e, n, q in step:
    perq[q] = perq.get(q, 0) + e - s
print("busy per queue: " + ", ".join(f"queue {q}: {1e-6 * t:.2f} ms" for q, t in sorted(perq.items(), key=lambda kv: -kv[1])))


def short(n):
    key = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return key.split("(")[0][-70:]


agg = {}
for s, e, n, q in step:
    a_ = agg.setdefault(short(n), [0, 0])
    a_[0] += 1
    a_[1] += e - s
comm = {k: v for k, v in agg.items() if any(t in k.lower() for t in ("nccl", "rccl", "copybuffer", "fillbuffer"))}
print(f"collective / copy kernels: {1e-6 * sum(v[1] for v in comm.values()):.3f} ms in {sum(v[0] for v in comm.values())} launches")
for key, (c, t) in sorted(comm.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"   {1e-6 * t:9.3f} ms {c:5d}  {key}")
print("top kernels:")
for key, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{1e-6 * t:9.3f} ms {c:5d}  {key}")
