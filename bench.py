#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Whisper hot path (BASELINE.json: RTF on a 30 s clip + clips/s).

A "step" is one pass of the whole hot path over one batch of synthetic 30 s clips that are already
resident in HBM: log-mel front-end -> encoder -> greedy decode (4 SOT steps + up to 444 tokens; with
synthetic weights eot practically never wins, so every clip runs the full 448-step context) -> ids.
Default workload = BASELINE.json configs[1]: Whisper-small bf16, 1 GPU, batch 1.

    python bench.py                                  # N=1, batch 1
    python bench.py --batch 64                       # configs[2]
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8   # one rank per GPU, weak scaling

Prints ONE JSON line (rank 0). `roofline` is for the dominant kernel family of the decode loop,
`cpu_baseline` is the CPU oracle ("port") timed on this box's host cores on one clip of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "whisper.axera_amd", "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)


def decode_step_bytes(dims, batch, step, s=2):
    """weights once per step + cross-KV + self-KV per clip, element size s (bf16)."""
    d, L, nv = dims["d"], dims["dec_layers"], dims["n_vocab"]
    weights = s * (L * 14 * d * d + nv * d)
    cross = batch * s * 2 * L * 1500 * d
    self_kv = batch * s * 2 * L * (step + 1) * d
    return weights, cross, self_kv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="clips per GPU (weak scaling)")
    ap.add_argument("--model", default="small")
    ap.add_argument("--max-new", type=int, default=0, help="0 = until eot or context (444 ids)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model-dir", default=os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import modelgen
    import whisper_axera_amd as wa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # AXW_BENCH_FORCE_DIST=1: run the RCCL group, the barriers and the result gather even with one rank (a rehearsal of
    # the N>1 code path on a one-GPU box)
    use_dist = world > 1 or os.environ.get("AXW_BENCH_FORCE_DIST") == "1"
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its communicator is created; stdout must carry the ONE JSON
        # line only, so everything until then goes to stderr (file-descriptor level: the banner comes from C code)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    # ---- synthetic-weight model directory (no weights exist in the reference or this image)
    dims = modelgen.DIMS[args.model]
    mdir = os.path.join(args.model_dir, args.model)
    if local_rank == 0 and not os.path.exists(os.path.join(mdir, f"{args.model}.safetensors")):
        modelgen.write_model_dir(args.model_dir, args.model, dims, seed=0,
                                 tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
    barrier()

    B = args.batch
    eng = wa.Whisper(args.model, args.model_dir, "zh", device=local_rank, max_batch=B)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)

    n_samp = 480000
    clips = np.stack([modelgen.synth_clip(rank * B + i, n_samp) for i in range(B)])
    d_pcm = torch.from_numpy(clips).to(dev)
    torch.cuda.synchronize(dev)

    from whisper_axera_amd import dp

    def one_step():
        ids = eng.run_device_tokens(d_pcm.data_ptr(), n_samp, [n_samp] * B, max_new=args.max_new)
        if use_dist:  # the ONE collective of the path: result gather over RCCL/xGMI
            dp.gather_ids(ids, B, device=dev)
        return ids

    for _ in range(args.warmup):
        ids = one_step()
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    stage = {"frontend_ms": 0.0, "encoder_ms": 0.0, "decode_ms": 0.0, "steps": 0}
    for _ in range(args.steps):
        ids = one_step()
        tm = eng.timings()
        for k in stage:
            stage[k] += tm[k]
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms_per_step = dt / args.steps * 1e3
    clips_per_s = world * B * args.steps / dt
    n_tok = float(np.mean([len(r) for r in ids]))
    dec_steps = stage["steps"] / args.steps

    # ---- roofline of the dominant kernel of the timed region.
    iters = 50
    w_bytes, c_bytes, s_bytes = decode_step_bytes(dims, B, 224)
    small_batch = B <= 4
    persistent = B == 1 and eng.L.AX_WHISPER_GetConfigInt(eng.h, b"persistent_decode") == 1
    n_launch = (dims["dec_layers"] * 6 + 1) * ((B + 3) // 4 if small_batch else (B + 63) // 64)
    ms_fam = eng.bench("decode_gemv", B, 224, iters)
    per_launch_s = ms_fam * 1e-3 / (iters * n_launch)
    fam_gbs = (w_bytes / n_launch) / per_launch_s / 1e9
    ms_step = eng.bench("decode_step", B, 224, iters) / iters
    ms_attn = eng.bench("decode_attn", B, 224, iters) / iters
    step_gbs = (w_bytes + c_bytes + s_bytes) / (ms_step * 1e-3) / 1e9
    attn_gbs = (c_bytes + s_bytes) / (ms_attn * 1e-3) / 1e9
    fam = "gemv_kernel" if small_batch else "decode_gemm_kernel"  # PMC family: clip-block GEMMs + vocabulary projection
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    pmc = json.load(open(pmc_path)).get(f"{args.model}_b{B}", {}) if os.path.exists(pmc_path) else {}

    def pmc_traffic(name):  # HBM bytes per launch from rocprofv3 --pmc passes (recipe in profiles/README.md)
        rec = pmc.get(name)
        return rec["hbm_bytes_per_launch"] if rec else None

    graph_path = {"kernel": f"{fam}{'/gemv1_kernel' if small_batch else ''} (decode linear layers, all shapes)",
                  "achieved": round(fam_gbs, 1), "frac": round(fam_gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(fam),
                  "launches_per_decode_step": n_launch, "avg_launch_us": round(per_launch_s * 1e6, 3),
                  "bytes_per_launch": int(w_bytes / n_launch),
                  "decode_step": {"ms": round(ms_step, 4), "algorithmic_GBs": round(step_gbs, 1),
                                  "frac_of_hbm_peak": round(step_gbs / HBM_PEAK_GBS, 4)},
                  "decode_attention": {"ms_per_step": round(ms_attn, 4), "algorithmic_GBs": round(attn_gbs, 1),
                                       "frac_of_hbm_peak": round(attn_gbs / HBM_PEAK_GBS, 4)}}
    if persistent:
        # Batch 1: the whole greedy loop of a clip is ONE launch of decode_persistent_kernel (99 % of the timed region).
        # Algorithmic bytes per launch (SURVEY §8d per step, summed over the steps the launch ran): decoder-layer
        # weights + cross-K/V every step, the vocabulary projection on the steps whose logits are used (all but the
        # 3 SOT steps), self-K/V of the t+1 cached keys. Duration: hipEvents on the engine's stream around the
        # launch (Engine::run_tokens, stage "decode"), averaged over the timed clips.
        d_, L_, nv_ = dims["d"], dims["dec_layers"], dims["n_vocab"]
        n_steps = int(round(dec_steps))
        launch_bytes = n_steps * (2 * L_ * 14 * d_ * d_ + 2 * 2 * L_ * 1500 * d_) + max(n_steps - 3, 0) * 2 * nv_ * d_ \
            + sum(2 * 2 * L_ * (t + 1) * d_ for t in range(n_steps))
        launch_s = stage["decode_ms"] / args.steps * 1e-3
        gbs = launch_bytes / launch_s / 1e9
        roofline = {"kernel": "decode_persistent_kernel (the whole greedy loop of one clip: %d decoder steps in one launch)" % n_steps,
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic("decode_persistent_kernel"), "launch_ms": round(launch_s * 1e3, 3),
                    "bytes_per_launch": int(launch_bytes), "us_per_decode_step": round(launch_s * 1e6 / max(n_steps, 1), 2),
                    "launch_per_phase_path": graph_path}
    elif small_batch:
        roofline = dict({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}, **graph_path)
    else:
        # 5+ clips: the step is dominated by decode_attention_kernel (cross + self attention, one workgroup per
        # (clip, head)), which streams every clip's K/V once per step: algorithmic bytes = B * (55.3 MB cross +
        # (t+1) * 36.9 KB self) at t = 224 (the mid-utterance step this leg replays), over 2 launches per layer.
        n_attn = 2 * dims["dec_layers"]
        roofline = {"kernel": "decode_attention_kernel (self + cross attention of one decoder step, %d launches)" % n_attn,
                    "bound": "hbm", "achieved": round(attn_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(attn_gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("decode_attention_kernel"),
                    "avg_launch_us": round(ms_attn * 1e3 / n_attn, 3), "bytes_per_launch": int((c_bytes + s_bytes) / n_attn),
                    "share_of_decode_step": round(ms_attn / ms_step, 3),
                    "decode_step": graph_path["decode_step"],
                    "linear_layers": {k: graph_path[k] for k in ("kernel", "achieved", "frac", "traffic", "launches_per_decode_step",
                                                                "avg_launch_us", "bytes_per_launch")}}

    # ---- the other two stages against their own rooflines: encoder = MFMA-bound, front-end = HBM-bound
    enc_flop = {"tiny": 40.48e9, "small": 386.63e9, "turbo": 2313.09e9}.get(args.model)
    ms_enc = eng.bench("encoder", B, 0, 5) / 5
    ms_fe = eng.bench("frontend", B, 0, 10) / 10
    stages = {"frontend": {"ms": round(ms_fe, 4), "GBs": round(B * (4 * n_samp + 2 * dims["n_mels"] * 3000) / (ms_fe * 1e-3) / 1e9, 1),
                           "bound": "hbm"}}
    if enc_flop:
        tf = enc_flop * B / (ms_enc * 1e-3) / 1e12
        stages["encoder"] = {"ms": round(ms_enc, 3), "TFLOPs": round(tf, 1), "bound": "mfma", "peak": MFMA_BF16_PEAK_TF,
                             "frac": round(tf / MFMA_BF16_PEAK_TF, 4)}

    out = {
        "metric": "clips_per_sec (30 s clips, greedy decode, whisper-%s)" % args.model,
        "value": round(clips_per_s, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"whisper-{args.model} bf16, batch {B}/GPU, 30 s synthetic 16 kHz clips resident in HBM, "
                               f"greedy decode {n_tok:.0f} ids/clip ({dec_steps:.0f} decoder steps)",
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}", "weights": "seeded synthetic"},
        "rtf": round(dt / args.steps / (B * 30.0), 6),
        "stage_ms": {k: round(v / args.steps, 3) for k, v in stage.items() if k != "steps"},
        "roofline": roofline,
        "stage_rooflines": stages,
    }

    # ---- CPU baseline: the oracle ("port") on this box's host cores, one clip of the same workload
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle

        threads = min(os.cpu_count() or 1, 32)
        weights = modelgen.read_safetensors(os.path.join(mdir, f"{args.model}.safetensors"))
        orc = oracle.Oracle(modelgen.make_config(args.model, dims), weights, bf16_policy=True, threads=threads)
        max_new = args.max_new if args.max_new > 0 else 444
        cpu_new = min(max_new, 96)  # bounded sample: full front-end + encoder, 4 + 96 decoder steps
        t1 = time.perf_counter()
        cpu_ids = orc.transcribe(clips[0], "zh", max_new=cpu_new)
        t_cpu = time.perf_counter() - t1
        # scale the decode part to the GPU's step count: measure encoder and decode separately
        t2 = time.perf_counter()
        mel, _, _ = oracle.log_mel(clips[0], dims["n_mels"])
        ck, cv = orc.encoder(mel)
        t_enc = time.perf_counter() - t2
        t_dec_step = max(t_cpu - t_enc, 1e-9) / (4 + len(cpu_ids))
        t_full = t_enc + t_dec_step * dec_steps
        agree = 0
        for a, b in zip(cpu_ids, ids[0]):
            if a != b:
                break
            agree += 1
        out["cpu_baseline"] = {"value": round(1.0 / t_full, 5), "unit": "clips/s", "cores": threads, "kind": "port",
                               "sample": f"clip 0: front-end + encoder measured ({t_enc:.2f} s) + {4 + len(cpu_ids)} decoder "
                                         f"steps measured ({t_dec_step * 1e3:.1f} ms/step) scaled to {dec_steps:.0f} steps; "
                                         f"CPU oracle (bf16 policy), {threads} OpenMP threads of {os.cpu_count()}",
                               "ids_agree_prefix": f"{agree}/{len(cpu_ids)}"}
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    if saved_stdout is not None:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
