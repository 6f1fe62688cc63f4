#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time, and how much kernels of the batched decode step overlap
in time (two graph branches: does one branch's GEMM chain really run beside the other's attention?).
usage: trace_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = defaultdict(int)
cnt = defaultdict(int)
for s, e, n, q in rows:
    busy[n] += e - s
    cnt[n] += 1
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.2f} ms; queues: {sorted(set(r[3] for r in rows))}")
for n, b in sorted(busy.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {b / 1e6:9.3f} ms  {cnt[n]:7d} x {b / cnt[n] / 1e3:8.2f} us  {n}")
# union of busy intervals and time with >= 2 kernels in flight
ev = []
for s, e, n, q in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, t_any, t_multi = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1:
        t_any += t - last
    if depth >= 2:
        t_multi += t - last
    depth += d
    last = t
print(f"time with >=1 kernel in flight {t_any / 1e6:.2f} ms, with >=2 in flight {t_multi / 1e6:.2f} ms ({100.0 * t_multi / max(t_any, 1):.1f} %), idle {(t1 - t0 - t_any) / 1e6:.2f} ms")
# attention x cgemm overlap specifically
att = [(s, e) for s, e, n, q in rows if "decode_attention" in n]
gem = [(s, e) for s, e, n, q in rows if "decode_cgemm" in n]
j = ov = 0
for s, e in gem:
    while j < len(att) and att[j][1] <= s:
        j += 1
    k = j
    while k < len(att) and att[k][0] < e:
        ov += min(e, att[k][1]) - max(s, att[k][0])
        k += 1
tg = sum(e - s for s, e in gem)
print(f"clip-block GEMM time {tg / 1e6:.2f} ms, of which beside an attention kernel {ov / 1e6:.2f} ms ({100.0 * ov / max(tg, 1):.1f} %)")
