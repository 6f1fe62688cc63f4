#!/usr/bin/env python3
"""Copy what collect_r02.sh left under gpurun_out/prof/ into profiles/ as the round's judged evidence (r02_*)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
merged = {}
for part in ("b1", "b64"):
    p = os.path.join(src, f"r02_pmc_traffic_{part}.json")
    if os.path.exists(p):
        merged.update({k: v for k, v in json.load(open(p)).items() if k.endswith(part)})
json.dump(merged, open(os.path.join(dst, "r02_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for a, b in (("b1_kernel_stats.csv", "r02_b1_kernel_stats.csv"), ("b64_kernel_stats.csv", "r02_b64_kernel_stats.csv"),
             ("persist_phases_summary.txt", "r02_persist_phases_summary.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for f in ("bench_b1", "bench_b1_fp16", "bench_turbo_fp16_b1", "bench_b1_graph", "bench_b64", "bench_turbo_fp16_b16"):
    if os.path.exists(os.path.join(src, f + ".json")):
        shutil.copy(os.path.join(src, f + ".json"), os.path.join(dst, "r02_" + f + ".json"))
        j = json.load(open(os.path.join(src, f + ".json")))
        print(f, j["dtype"], j["value"], "clips/s", j["ms_per_step"], "ms/step rtf", j["rtf"], j["stage_ms"], "frac", j["roofline"]["frac"],
              "traffic", j["roofline"].get("traffic"))
        if "batch64" in j:
            b = j["batch64"]
            print("   batch64:", b["value"], b["ms_per_step"], b["stage_ms"], b["roofline"]["frac"], b["roofline"]["decode_step"])
            print("   cpu:", j["cpu_baseline"]["value"], j["cpu_baseline"]["single_thread"]["value"], j["cpu_baseline"]["frontend_only"], "host_pcm", j["host_pcm"]["rtf"])
for k, v in merged.items():
    for kern in ("decode_persistent_kernel", "decode_attention_kernel", "decode_gemm_kernel", "gemv_kernel"):
        if kern in v:
            print(k, kern, v[kern]["hbm_bytes_per_launch"], "B/launch over", v[kern]["launches_sampled"])
