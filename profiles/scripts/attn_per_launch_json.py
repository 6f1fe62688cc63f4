#!/usr/bin/env python3
"""profiles/rNN_b64_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --batch 64 --steps 1 --warmup 1 --no-extras`)
-> profiles/r06_b64_attn_per_launch.json: the 32-clip cross-attention launch of the 64-clip step as the trace sees it (launches
serialised by the profiler), which bench.py quotes next to the overlapped-replay figure (config.other_configs).
usage: attn_per_launch_json.py <kernel_stats.csv> <out.json> [clips per launch 32]"""
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
clips = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows = [r for r in csv.DictReader(open(src)) if "decode_attention_kernel<1" in r["Name"] or "decode_attention_kernel<true" in r["Name"]]
assert rows, "no fused cross-attention launches in " + src
r = max(rows, key=lambda r: int(r["Calls"]))
avg_us = float(r["AverageNs"]) / 1e3
# K and V of 1500 keys x 768 dims x 12 layers... per LAUNCH: one layer, `clips` clips, h16
nbytes = clips * 2 * 1500 * 768 * 2
out = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_launch_us": round(avg_us, 3), "clips_per_launch": clips,
       "bytes_per_launch": nbytes, "achieved_GBs": round(nbytes / (avg_us * 1e-6) / 1e9, 1),
       "frac": round(nbytes / (avg_us * 1e-6) / 1e9 / 8000.0, 4),
       "what": "cross-attention launch of one 32-clip branch, average duration under rocprofv3 --kernel-trace (branches serialised), "
               "algorithmic K/V bytes of the launch over it, against the 8 TB/s HBM peak"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
