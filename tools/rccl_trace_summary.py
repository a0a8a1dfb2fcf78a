#!/usr/bin/env python3
"""Where do the RCCL kernels of the one-rank rehearsal run?   python3 tools/rccl_trace_summary.py <kernel_trace.csv>

For the LAST complete training step in the trace: every RCCL kernel (name contains nccl / rccl) with its start and end relative
to the step's first kernel, its duration, the compute kernels that were executing when it started and when it ended, and the
gap between the end of the compute kernel that produced the bucket's last gradient (the one enqueued just before it on the
compute queue = the last compute kernel that STARTED before the collective did) and the collective's start."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else [k for k in rows[0] if "name" in k.lower()][0]
s_k = [k for k in rows[0] if k.lower().startswith("start")][0]
e_k = [k for k in rows[0] if k.lower().startswith("end")][0]
q_k = next((k for k in rows[0] if k.lower() in ("queue_id", "stream_id")), None)
ev = sorted(((int(r[s_k]), int(r[e_k]), r[name_k], r.get(q_k, "")) for r in rows), key=lambda t: t[0])
is_rccl = lambda n: any(k in n.lower() for k in ("nccl", "rccl", "onerankreduce"))   # noqa: E731  (one rank: RCCL launches its oneRankReduce kernel)
# step boundaries: the optimizer kernel ends a step
ends = [i for i, (_, _, n, _) in enumerate(ev) if "adam_pack_kernel" in n]
if len(ends) < 3:
    print("fewer than three optimizer launches in the trace")
    sys.exit(1)
lo, hi = ends[-2] + 1, ends[-1]
step = ev[lo:hi + 1]
t0 = step[0][0]
comp = [(s, e, n, q) for s, e, n, q in step if not is_rccl(n)]
coll = [(s, e, n, q) for s, e, n, q in step if is_rccl(n)]
print(f"columns: {list(rows[0].keys())}")
print(f"last complete step: {len(step)} kernels, {(step[-1][1] - t0) / 1e6:.3f} ms from its first kernel's start to the optimizer's end; "
      f"{len(coll)} RCCL kernels, queues seen: {sorted({q for *_, q in step})}")
busy = sum(e - s for s, e, *_ in comp) / 1e6
print(f"sum of compute kernel durations {busy:.3f} ms; sum of RCCL kernel durations {sum(e - s for s, e, *_ in coll) / 1e6:.3f} ms")

# gaps on the compute queue (a collective may or may not explain them)
gaps = []
for a, b in zip(comp, comp[1:]):
    g = b[0] - a[1]
    if g > 3000:
        inside = [n.replace("unetdc::", "").replace("void ", "")[:30] for s_, e_, n, _ in coll if s_ < b[0] and e_ > a[1]]
        gaps.append((g / 1e3, a[2].replace("unetdc::", "").replace("void ", "")[:40], b[2].replace("unetdc::", "").replace("void ", "")[:40], inside))
print(f"compute-queue gaps > 3 us: {len(gaps)}, total {sum(g[0] for g in gaps):.1f} us")


def running_at(t):
    return [n for s, e, n, _ in comp if s <= t < e]


def short(n):
    n = n.replace("unetdc::", "").replace("void ", "")
    return n[:60]


print(f"{'RCCL kernel':40s} {'start ms':>9s} {'dur us':>8s} {'queue':>6s}  running at start | running at end | wait behind producer")
for s, e, n, q in coll:
    prod = [c for c in comp if c[0] <= s]
    last = max(prod, key=lambda c: c[0]) if prod else None
    wait = (s - last[1]) / 1e3 if last else float("nan")
    print(f"{short(n):40s} {(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {q:>6s}  {[short(x)[:34] for x in running_at(s)]} | "
          f"{[short(x)[:34] for x in running_at(e - 1)]} | producer {short(last[2])[:34] if last else '-'} ended {wait:+.1f} us before the start")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:7.1f} us between {g[1]} and {g[2]}  RCCL kernels inside: {g[3]}")
