"""decode_attention_kernel on one time axis while the graph's branches run side by side.

One replay of the PRODUCTION step graph (every launch, every branch) at decode offset t whose attention launches stamp
their own {first workgroup start, last workgroup end} (100 MHz wall clock; Engine::bench "attn_stamp", a separate template
instantiation of the kernel). Prints the table and the summary that profiles/r04_attn_stamps_*.txt hold:
K/V bytes of the step / length of the UNION of the attention intervals = what the launches achieve in production.

usage: attn_stamp.py [clips 64] [model small] [fp16] [t 224]     (AX_WHISPER_DECODE_BRANCHES=1 for the one-branch form)"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = sys.argv[2] if len(sys.argv) > 2 else "small"
fp16 = len(sys.argv) > 3 and sys.argv[3] == "fp16"
t = int(sys.argv[4]) if len(sys.argv) > 4 else 224
mdir = os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models") + ("_f16" if fp16 else "")
if not os.path.exists(os.path.join(mdir, model, model + ".safetensors")):
    modelgen.write_model_dir(mdir, model, seed=0, dtype="F16" if fp16 else "BF16")
path = os.environ.setdefault("AX_WHISPER_ATTN_STAMP", "/tmp/attn_stamps.csv")
e = wa.Whisper(model, mdir, "zh", device=0, max_batch=B)
e.run_tokens_batch([modelgen.synth_clip(i, 480000) for i in range(B)], max_new=2)  # cross K/V of real clips in every slot
e.bench("decode_step", B, t, 20)
ms_attn = e.bench("decode_attn", B, t, 200) / 200
ms_step = e.bench("decode_step", B, t, 200) / 200
for warm, title in ((0, "ONE replay on an idle device"), (6, "the 7th of 7 replays back to back (a step of the decode loop: its head overlaps its predecessor's tail)")):
    unions = [e.bench("attn_stamp", B, t, 100 + warm) for _ in range(5)]
    rows = [l for l in open(path).read().splitlines()]
    data = [l.split(",") for l in rows if l and l[0].isdigit()]
    tot = sum(float(r[8]) for r in data)
    un = float(rows[-1].split(":")[1].split()[0])
    print(f"## {title}")
    print("\n".join(rows))
    for kind in ("cross", "self"):
        d = [float(r[7]) for r in data if r[1] == kind]
        by = [float(r[8]) for r in data if r[1] == kind]
        print(f"# {kind}: {len(d)} launches, duration mean {sum(d) / len(d):.2f} us (min {min(d):.2f}, max {max(d):.2f}), "
              f"per-launch rate {sum(by) / sum(d) / 1e6:.3f} TB/s = {sum(by) / sum(d) / 1e6 / 8:.4f} of the 8 TB/s peak")
    span = max(float(r[6]) for r in data) - min(float(r[5]) for r in data)
    print(f"# K/V bytes of the step {tot / 1e9:.3f} GB / union of the attention intervals {un:.2f} us = {tot / un / 1e6:.3f} TB/s = {tot / un / 1e6 / 8:.4f} of the HBM peak; "
          f"first start to last end {span:.1f} us (branches: {e.L.AX_WHISPER_GetConfigInt(e.h, b'decode_branches')}; unions of 5 runs: {', '.join('%.1f' % (u * 1e3) for u in unions)} us)")
print(f"# the same step as hipEvent replays (200 back to back): whole step {ms_step * 1e3:.1f} us, attention launches only {ms_attn * 1e3:.1f} us "
      f"-> {tot / (ms_attn * 1e-3) / 1e12:.3f} TB/s = {tot / (ms_attn * 1e-3) / 1e12 / 8:.4f} (bench.py's roofline figure: no GEMM launch between the attention launches)")
e.close()
