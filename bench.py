#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X Whisper hot path (BASELINE.json: RTF on a 30 s clip + clips/s at batch).

A "step" is one pass of the whole hot path over one batch of synthetic 30 s clips that are already
resident in HBM: log-mel front-end -> encoder -> greedy decode (4 SOT steps + up to 444 tokens; with
synthetic weights eot practically never wins, so every clip runs the full 448-step context) -> ids.

    python bench.py                      # N=1: BASELINE configs[1] (Whisper-small, batch 1) as the headline value,
                                         #      plus a "batch64" object = configs[2] timed in the same run (and "batch256": the same at 256 clips)
    python bench.py --batch 64           # configs[2] as the headline value
    python bench.py --gpus 8             # configs[4]: spawns 8 ranks (one per GPU, RCCL), 64 clips per GPU
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8     # the same, launched by the driver

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts `python -m torch.distributed.run` as a CHILD
process before anything in this process has touched a GPU, relays its single JSON line and exits with its code.

Prints ONE JSON line (rank 0). `roofline` is for the dominant kernel of the timed region, `cpu_baseline` is the CPU
oracle ("port") timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "whisper.axera_amd", "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16/fp16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)
N_SAMP = 480000  # 30 s at 16 kHz
PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")
ROCPROF_ATTN_FILE = os.path.join("profiles", "r06_b64_attn_per_launch.json")  # {"frac": .., "avg_launch_us": .., "bytes_per_launch": ..}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (weak scaling); 0 = 1 at N=1, 64 at N>1")
    ap.add_argument("--model", default="small")
    ap.add_argument("--max-new", type=int, default=0, help="0 = until eot or context (444 ids)")
    ap.add_argument("--dtype", default="", choices=["", "bf16", "fp16"],
                    help="16-bit storage / MFMA operand type; default: bf16, fp16 for --model turbo (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch64", action="store_true", help="N=1, batch 1 only: skip the batch-64 leg")
    ap.add_argument("--no-batch256", action="store_true", help="N=1, batch 1 only: skip the batch-256 leg (not a BASELINE config: what one GPU's HBM allows)")
    ap.add_argument("--no-turbo", action="store_true", help="N=1, batch 1 only: skip the turbo fp16 batch-16 leg (configs[3])")
    ap.add_argument("--no-realistic", action="store_true",
                    help="N=1, batch 1 only: skip the realistic-length legs (batch64_len100: 64 clips with budgets of 60-150 ids; stream64: 384 clips through 64 refilled slots)")
    ap.add_argument("--no-config0", action="store_true", help="N=1, batch 1 only: skip the tiny / demo.wav CPU-vs-GPU leg (configs[0])")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region and its roofline: no host-PCM / forced-length / demo.wav / batch-64 / turbo / "
                         "config0 / CPU legs (profiling runs: every launch in the trace belongs to the headline workload)")
    ap.add_argument("--model-dir", default=os.environ.get("AXW_BENCH_MODEL_DIR", "/tmp/axw_bench_models"))
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------- N > 1: spawn the ranks
def spawn_ranks(args) -> int:
    """Start one rank per GPU as children of this process (which has not touched a GPU) and relay rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # A rank that dies mid-step leaves the others inside a collective: torch.distributed.run notices the dead worker and
    # tears the rest down (non-zero exit). Should that ever not happen, this process does not wait forever either: after
    # AXW_BENCH_TIMEOUT_S (default 50 min) it kills the process group it started — that exact group, nothing by pattern.
    timeout_s = float(os.environ.get("AXW_BENCH_TIMEOUT_S", "3000"))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    try:
        stdout, _ = p.communicate(timeout=timeout_s)
        code = p.returncode
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        stdout, _ = p.communicate()
        print(f"bench.py: the {args.gpus} ranks did not finish within {timeout_s:.0f} s; killed", file=sys.stderr)
        code = 124
    lines = [l for l in stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines and code == 0:
        print(lines[-1], flush=True)
    return code if code != 0 or lines else 1


# ------------------------------------------------------------------------------------------- algorithmic work
def decode_step_bytes(dims, batch, step, s=2):
    """weights once per step + cross-KV + self-KV per clip, element size s."""
    d, L, nv = dims["d"], dims["dec_layers"], dims["n_vocab"]
    weights = s * (L * 14 * d * d + nv * d)
    cross = batch * s * 2 * L * 1500 * d
    self_kv = batch * s * 2 * L * (step + 1) * d
    return weights, cross, self_kv


def load_pmc(model, B):
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return {}, None
    return json.load(open(path)).get(f"{model}_b{B}", {}), PMC_FILE


class RehearsalEngine:
    """AXW_BENCH_REHEARSAL=1 (CPU tests of the N>1 plumbing only): no GPU, no engine — ids are a function of the clip."""

    def run(self, clips, max_new):
        return [[int(abs(float(c[:16].sum())) * 1e3) % 50000 + i for i in range(5 + (k % 3))] for k, c in enumerate(clips)]


def timed_steps(one_step, steps, warmup, sync, barrier):
    for _ in range(warmup):
        one_step()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ids = one_step()
    sync()
    barrier()
    return time.perf_counter() - t0, ids


def roofline_for(eng, dims, model, B, dec_steps, decode_ms_per_step, dtype_bytes=2):
    """Roofline of the dominant kernel of a B-clip step, measured live with hipEvents on the engine's stream."""
    iters = 50
    w_bytes, c_bytes, s_bytes = decode_step_bytes(dims, B, 224, dtype_bytes)
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
    fam = "gemv_kernel" if small_batch else "decode_gemm_kernel"  # PMC family: linear layers + vocabulary projection
    pmc, pmc_src = load_pmc(model, B)

    def traffic(name):  # HBM bytes per launch from rocprofv3 --pmc passes of this command (recipe: profiles/README.md)
        rec = pmc.get(name)
        return rec["hbm_bytes_per_launch"] if rec else None

    graph_path = {"kernel": f"{fam}{'/gemv1_kernel' if small_batch else ''} (decode linear layers, all shapes)",
                  "achieved": round(fam_gbs, 1), "frac": round(fam_gbs / HBM_PEAK_GBS, 4), "traffic": traffic(fam),
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
        d_, L_, nv_, s_ = dims["d"], dims["dec_layers"], dims["n_vocab"], dtype_bytes
        n_steps = int(round(dec_steps))
        launch_bytes = n_steps * (s_ * L_ * 14 * d_ * d_ + s_ * 2 * L_ * 1500 * d_) + max(n_steps - 3, 0) * s_ * nv_ * d_ \
            + sum(s_ * 2 * L_ * (t + 1) * d_ for t in range(n_steps))
        launch_s = decode_ms_per_step * 1e-3
        gbs = launch_bytes / launch_s / 1e9
        roof = {"kernel": "decode_persistent_kernel (the whole greedy loop of one clip: %d decoder steps in one launch)" % n_steps,
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic("decode_persistent_kernel"), "launch_ms": round(launch_s * 1e3, 3),
                "bytes_per_launch": int(launch_bytes), "us_per_decode_step": round(launch_s * 1e6 / max(n_steps, 1), 2),
                "launch_per_phase_path": graph_path}
    elif small_batch:
        roof = dict({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}, **graph_path)
    else:
        # 5+ clips: the step is dominated by decode_attention_kernel (cross + self attention, one workgroup per
        # (clip, head)), which streams every clip's K/V once per step: algorithmic bytes = B * (55.3 MB cross +
        # (t+1) * 36.9 KB self) at t = 224 (the mid-utterance step this leg replays), over 2 launches per layer.
        branches = max(1, eng.L.AX_WHISPER_GetConfigInt(eng.h, b"decode_branches"))  # clip blocks run as parallel graph branches
        n_attn = 2 * dims["dec_layers"] * branches
        roof = {"kernel": "decode_attention_kernel (self + cross attention of one decoder step, %d launches in %d parallel graph branches)" % (n_attn, branches),
                "bound": "hbm", "achieved": round(attn_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(attn_gbs / HBM_PEAK_GBS, 4), "traffic": traffic("decode_attention_kernel"),
                "avg_launch_us": round(ms_attn * 1e3 / n_attn, 3), "bytes_per_launch": int((c_bytes + s_bytes) / n_attn),
                "share_of_decode_step": round(ms_attn / ms_step, 3),
                "decode_step": graph_path["decode_step"],
                "linear_layers": {k: graph_path[k] for k in ("kernel", "achieved", "frac", "traffic", "launches_per_decode_step",
                                                            "avg_launch_us", "bytes_per_launch")}}
    roof["traffic_source"] = (f"{pmc_src}: static, rocprofv3 --pmc passes of this command on the same build (not collected in "
                              "this run)") if pmc_src and roof.get("traffic") is not None else None
    return roof


def step_chain_roofline(dims, B, n_steps, decode_ms, s=2):
    """4+ clips per call without per-kernel replays: the decoder steps of one call as a whole against the HBM roofline.
    Algorithmic bytes: layer weights + every clip's cross K/V on every step, the vocabulary projection on
    the steps whose logits are used (all but the 3 SOT steps), self K/V of the t + 1 cached keys per clip. Duration: the
    engine's hipEvents around the decode stage of the call (stage "decode"), averaged over the timed calls."""
    d, L, nv = dims["d"], dims["dec_layers"], dims["n_vocab"]
    total = n_steps * (s * L * 14 * d * d + B * s * 2 * L * 1500 * d) + max(n_steps - 3, 0) * s * nv * d \
        + B * sum(s * 2 * L * (t + 1) * d for t in range(n_steps))
    gbs = total / (decode_ms * 1e-3) / 1e9
    return {"kernel": "one decoder step = one replay of the captured step graph (7 dependent launches per layer + vocabulary "
                      "projection + token feedback), %d steps per call" % n_steps,
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": None, "bytes_per_step": int(total / n_steps), "us_per_decode_step": round(decode_ms * 1e3 / n_steps, 2)}


def stage_rooflines(eng, dims, model, B):
    """The other two stages against their own rooflines: encoder = MFMA-bound; the front-end evaluates the
    400-point DFT of every frame as an exact-fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: the fp32 matrix peak
    equals the fp32 vector peak, 157 TFLOP/s; rounds 1-3 ran a scalar FMA loop at 0.18 of it), its bytes are tiny."""
    enc_flop = {"tiny": 40.48e9, "small": 386.63e9, "turbo": 2313.09e9}.get(model)
    ms_enc = eng.bench("encoder", B, 0, 5) / 5
    ms_fe = eng.bench("frontend", B, 0, 10) / 10
    fe_flop = B * 3001 * (2 * 400 * 201 * 2 + 2 * 201 * dims["n_mels"])
    stages = {"frontend": {"ms": round(ms_fe, 4), "GBs": round(B * (4 * N_SAMP + 2 * dims["n_mels"] * 3000) / (ms_fe * 1e-3) / 1e9, 1),
                           "fp32_TFLOPs": round(fe_flop / (ms_fe * 1e-3) / 1e12, 2), "peak_fp32_matrix_TFLOPs": 157.3,
                           "frac": round(fe_flop / (ms_fe * 1e-3) / 1e12 / 157.3, 4),
                           # SURVEY §8(d) bounds this stage by HBM (4 B per sample in, 2 B per mel value out): next to the matrix
                           # figure, because the DFT-as-GEMM does ~37x the arithmetic of a 400-point FFT
                           "frac_of_hbm_bound": round(B * (4 * N_SAMP + 2 * dims["n_mels"] * 3000) / (ms_fe * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "bound": "fp32 MFMA (DFT of 201 bins as a GEMM + mel projection) at batch, launch latency at one clip"}}
    if enc_flop:
        tf = enc_flop * B / (ms_enc * 1e-3) / 1e12
        stages["encoder"] = {"ms": round(ms_enc, 3), "TFLOPs": round(tf, 1), "bound": "mfma", "peak": MFMA_BF16_PEAK_TF,
                             "frac": round(tf / MFMA_BF16_PEAK_TF, 4)}
    return stages


def other_configs_summary(out):
    """The other BASELINE configs' numbers of this run inside `config` (the driver's record keeps `config`, `roofline` and
    `cpu_baseline` whole and truncates the rest of the line). Every field is named for what it is; absent legs are absent."""
    oc = {}

    def get(path, d=out):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    def put(name, v):
        if v is not None:
            oc[name] = v

    put("batch64_clips_per_s", get(("batch64", "value")))  # configs[2]
    put("batch64_ms_per_call", get(("batch64", "ms_per_step")))
    put("batch64_step_frac_hbm", get(("batch64", "roofline", "decode_step", "frac_of_hbm_peak")))  # whole decoder step, hipEvents
    put("batch64_step_ms", get(("batch64", "roofline", "decode_step", "ms")))
    # attention-only replays with both graph branches overlapping (hipEvents) — NOT the per-launch figure of a serialised trace
    put("batch64_attn_frac_overlapped_replays", get(("batch64", "roofline", "frac")))
    prof = load_rocprof_attn()
    if prof:  # static: from the rocprofv3 kernel-trace summary of this build committed under profiles/
        put("batch64_attn_frac_per_launch_rocprof", prof.get("frac"))
        put("batch64_attn_rocprof_source", prof.get("source"))
    put("batch64_encoder_frac_mfma", get(("batch64", "stage_rooflines", "encoder", "frac")))
    put("batch64_encoder_ms", get(("batch64", "stage_rooflines", "encoder", "ms")))
    put("turbo_fp16_b16_clips_per_s", get(("turbo_fp16_b16", "value")))  # configs[3]
    put("turbo_fp16_b16_step_frac_hbm", get(("turbo_fp16_b16", "roofline", "decode_step", "frac_of_hbm_peak")))
    put("turbo_fp16_b16_encoder_frac_mfma", get(("turbo_fp16_b16", "stage_rooflines", "encoder", "frac")))
    put("host_pcm_clips_per_s", get(("host_pcm", "clips_per_s")))  # H2D + D2H inside the timed call
    put("host_pcm_rtf", get(("host_pcm", "rtf")))
    put("config0_ids_agree", get(("config0", "ids_agree_prefix")))  # configs[0]: tiny / demo.wav, GPU vs CPU oracle
    put("config0_gpu_rtf_true_duration", get(("config0", "gpu", "rtf_true_duration")))
    put("low_load_clips_per_s", get(("low_load_clips_per_s",)))
    for nb in (2, 3, 4, 8):
        put("batch%d_frac_hbm" % nb, get(("batch%d" % nb, "roofline", "frac")))
    put("batch64_len100_clips_per_s", get(("batch64_len100", "value")))
    put("stream64_clips_per_s", get(("stream64", "value")))
    put("stream64_steady_clips_per_s", get(("stream64", "steady_state_clips_per_s")))
    put("stream128_clips_per_s", get(("stream128", "value")))
    put("batch256_clips_per_s", get(("batch256", "value")))
    put("encoder_frac_mfma_this_batch", get(("stage_rooflines", "encoder", "frac")))
    put("single_gpu_same_batch_clips_per_s", get(("single_gpu_same_batch", "value")))  # N > 1 lines
    return oc


def load_rocprof_attn():
    path = os.path.join(ROOT, ROCPROF_ATTN_FILE)
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    d["source"] = ROCPROF_ATTN_FILE
    return d


def ensure_model_dir(modelgen, model_dir, model, dims, dtype):
    """Seeded synthetic-weight model directory (no weights exist in the reference or this image); the weights file's
    dtype (BF16 / F16) selects the engine build. Returns (model_root, model directory)."""
    model_root = model_dir + ("_f16" if dtype == "fp16" else "")
    mdir = os.path.join(model_root, model)
    if not os.path.exists(os.path.join(mdir, f"{model}.safetensors")):
        modelgen.write_model_dir(model_root, model, dims, seed=0, dtype="F16" if dtype == "fp16" else "BF16",
                                 tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
    return model_root, mdir


def realistic_budgets(n, seed=20260104):
    """Per-clip id budgets of the realistic-length legs: real 30 s utterances end after ~60-150 ids; synthetic weights never
    emit eot, so every clip gets its own budget, drawn ONCE from U{60..150} (numpy PCG64, fixed seed: the same on every box)."""
    import numpy as np

    return [int(x) for x in np.random.Generator(np.random.PCG64(seed)).integers(60, 151, size=n)]


def batch_leg(torch, dev, dev_index, sync, model, dtype, B, steps, max_new, model_dir, rooflines=True, budgets=None):
    """One more BASELINE config timed in the same run on the same GPU: B synthetic 30 s clips resident in HBM through
    the whole hot path, `steps` timed passes after one warm-up pass. budgets: per-clip id budgets (a ragged batch)."""
    import numpy as np

    import modelgen
    import whisper_axera_amd as wa

    dims = modelgen.DIMS[model]
    model_root, _ = ensure_model_dir(modelgen, model_dir, model, dims, dtype)
    clips = np.stack([modelgen.synth_clip(i, N_SAMP) for i in range(B)])
    d_pcm = torch.from_numpy(clips).to(dev)
    eng = wa.Whisper(model, model_root, "zh", device=dev_index, max_batch=B)
    assert eng.L.AX_WHISPER_GetConfigInt(eng.h, b"fp16") == (1 if dtype == "fp16" else 0)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    st = {"frontend_ms": 0.0, "encoder_ms": 0.0, "decode_ms": 0.0, "steps": 0}

    def step():
        r = eng.run_device_tokens(d_pcm.data_ptr(), N_SAMP, [N_SAMP] * B, max_new=max_new, max_new_clip=budgets)
        tm = eng.timings()
        for k in st:
            st[k] += tm[k]
        return r

    step()
    for k in st:
        st[k] = 0
    dt, ids = timed_steps(step, steps, 0, sync, lambda: None)
    if budgets is not None:
        assert [len(r) for r in ids] == [min(b, max_new) if max_new > 0 else b for b in budgets]
    leg = {
        "value": round(B * steps / dt, 3), "unit": "clips/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2),
        "rtf": round(dt / steps / (B * 30.0), 7), "dtype": dtype,
        "config": {"workload": f"whisper-{model} {dtype}, batch {B}, 30 s synthetic clips resident in HBM, greedy decode "
                               f"{float(np.mean([len(r) for r in ids])):.0f} ids/clip ({st['steps'] / steps:.0f} decoder steps)"},
        "stage_ms": {k: round(v / steps, 3) for k, v in st.items() if k != "steps"}}
    if rooflines:
        leg["roofline"] = roofline_for(eng, dims, model, B, st["steps"] / steps, st["decode_ms"] / steps)
        leg["stage_rooflines"] = stage_rooflines(eng, dims, model, B)
    eng.close()
    del d_pcm
    return leg, ids


def stream_leg(torch, dev, dev_index, sync, model, dtype, n_slots, n_clips, model_dir, policies=(1, 8), group_sizes=True):
    """The serving path on the realistic-length workload: n_clips 30 s clips (host PCM, as requests arrive) through n_slots
    utterance slots that are refilled while the others decode (AX_WHISPER_Stream*: one batched front-end + encoder pass per
    group of freed slots, cross K/V scattered into the idle slots, per-slot decode offsets). Budgets as in the ragged batch
    leg. Also: what an admission pass of k clips costs per clip (the encoder runs at a fraction of its batched rate on few
    clips; at 60-150 ids per utterance it is 20-27 % of the GPU time of this workload)."""
    import numpy as np

    import modelgen
    import whisper_axera_amd as wa

    dims = modelgen.DIMS[model]
    model_root, _ = ensure_model_dir(modelgen, model_dir, model, dims, dtype)
    distinct = [modelgen.synth_clip(i, N_SAMP) for i in range(min(n_slots, 64))]
    clips = [distinct[i % len(distinct)] for i in range(n_clips)]
    budgets = realistic_budgets(n_clips, seed=20260105)
    eng = wa.Whisper(model, model_root, "zh", device=dev_index, max_batch=n_slots)
    res = {}
    for min_admit in policies:
        eng.run_stream(clips[: 2 * n_slots], n_slots, max_new=budgets[: 2 * n_slots], min_admit=min_admit)  # warm: graphs, buffers
        sync()
        t0 = time.perf_counter()
        stamps = []
        got, calls = eng.run_stream(clips, n_slots, max_new=budgets, min_admit=min_admit, stamps=stamps)
        dt = time.perf_counter() - t0
        assert [len(g) for g in got] == budgets
        # the burst as a whole (first fill and last drain included) and its middle: completions 25 % .. 75 % of the clips
        q1, q3 = n_clips // 4, (3 * n_clips) // 4
        res[f"min_admit_{min_admit}"] = {"clips_per_s": round(n_clips / dt, 2), "wall_s": round(dt, 4), "step_calls": calls,
                                          "steady_state_clips_per_s": round((q3 - q1) / (stamps[q3] - stamps[q1]), 2)}
    enc_flop = {"tiny": 40.48e9, "small": 386.63e9, "turbo": 2313.09e9}.get(model)
    groups = {}
    for k in (1, 2, 4, 8, 16, 32, 64):
        if k > n_slots or not group_sizes:
            break
        ms = eng.bench("encoder", k, 0, 5) / 5
        groups[str(k)] = {"ms": round(ms, 3), "ms_per_clip": round(ms / k, 4), "TFLOPs": round(enc_flop * k / (ms * 1e-3) / 1e12, 1) if enc_flop else None}
    eng.close()
    best = max(res.values(), key=lambda r: r["clips_per_s"])
    return {"value": best["clips_per_s"], "unit": "clips/s", "dtype": dtype,
            "config": {"workload": f"whisper-{model} {dtype}, {n_clips} 30 s synthetic clips (host PCM) through {n_slots} refilled slots, "
                                   f"greedy decode, per-clip budgets U{{60..150}} ids (mean {float(np.mean(budgets)):.0f})"},
            "steady_state_clips_per_s": best["steady_state_clips_per_s"],
            "by_admission_policy": res,
            "what": "value: the whole burst, first fill and last drain included; steady_state: completions 25 % .. 75 % of the clips over the "
                    "time between them. min_admit_k: an admission pass waits for k free slots while anything is still decoding (whisper_srv --min-admit)",
            "encoder_pass_by_group_size": groups}


def config0_leg(torch, dev, dev_index, sync, model_dir, n_ids=32):
    """BASELINE configs[0]: Whisper-tiny on demo.wav, greedy, batch 1 — the loop of the reference's CPU/ONNX path
    (model_convert/generate_data.py:179-250) stood in for by the fp32 CPU oracle, timed on this box's host cores
    (1 thread and all cores) beside the GPU engine at tiny dims on the same file and the same seeded weights. RTF over
    the clip's TRUE duration (4.2039 s), as whisper_cli.cpp:93-103 and the README compute it. Synthetic weights never
    emit eot, so both sides decode a fixed number of ids."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wave

    import numpy as np

    import modelgen
    import oracle
    import whisper_axera_amd as wa

    dims = modelgen.DIMS["tiny"]
    model_root, mdir = ensure_model_dir(modelgen, model_dir, "tiny", dims, "bf16")
    w = wave.open(os.path.join(ROOT, "tests", "golden", "demo.wav"))
    demo = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / np.float32(32768.0)
    audio_s = len(demo) / 16000.0
    eng = wa.Whisper("tiny", model_root, "zh", device=dev_index, max_batch=1)
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    gpu_ids = eng.run_tokens_batch([demo], max_new=n_ids)[0]
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.run_tokens_batch([demo], max_new=n_ids)
    tg = (time.perf_counter() - t0) / 5
    eng.close()
    weights = modelgen.read_safetensors(os.path.join(mdir, "tiny.safetensors"))
    cfg = modelgen.make_config("tiny", dims)
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    res = {}
    many = f"{threads}_threads"  # the oracle's decoder stops scaling beyond ~32 threads: named for what it uses, not "all cores"
    for name, th in ((many, threads), ("single_thread", 1)):
        orc = oracle.Oracle(cfg, weights, bf16_policy=False, threads=th)
        t1 = time.perf_counter()
        cpu_ids = orc.transcribe(demo, "zh", max_new=n_ids)
        tc = time.perf_counter() - t1
        res[name] = {"seconds": round(tc, 4), "rtf_true_duration": round(tc / audio_s, 5), "cores": th}
    orc.L.orc_set_threads(min(ncpu, 16))
    agree = 0
    for a, b in zip(cpu_ids, gpu_ids):
        if a != b:
            break
        agree += 1
    return {"workload": f"whisper-tiny (seeded synthetic weights), tests/golden/demo.wav ({audio_s:.4f} s, zh), greedy, batch 1, "
                        f"{n_ids} ids (synthetic weights never emit eot)",
            "cpu": dict(res, kind="port", what="fp32 CPU oracle (the restatement of export_onnx.py's graph + Whisper.cpp's loop; "
                                               "the reference's own fp32 path needs onnxruntime + openai-whisper, absent here)",
                        cpu_model=cpu_model_string(), host_cores=ncpu),
            "gpu": {"ms": round(tg * 1e3, 3), "rtf_true_duration": round(tg / audio_s, 6), "dtype": "bf16",
                    "what": "AX_WHISPER_RunPCMBatchTokens (host PCM, H2D + D2H inside)"},
            "ids_agree_prefix": f"{agree}/{len(cpu_ids)}",
            f"speedup_vs_{many}": round(res[many]["seconds"] / tg, 1)}


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, dims, mdir, clip, gpu_ids, dec_steps, dtype="bf16"):
    """The CPU oracle ("port") on this box's host cores on ONE clip of the headline workload, measured in full (round 4:
    rounds 1-3 scaled 100 decoder steps to 448 and extrapolated the single-thread encoder from 1- and 2-layer runs):
    front-end + encoder + the whole greedy loop with all cores (the reported value, ~7 s) and with one thread (~20 s),
    + the front-end alone. A clip whose full single-thread run would exceed ~30 s (turbo dims) keeps the
    bounded form: measured steps scaled to the GPU's step count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import modelgen
    import oracle

    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    weights = modelgen.read_safetensors(os.path.join(mdir, f"{args.model}.safetensors"))
    cfg = modelgen.make_config(args.model, dims)
    policy = 2 if dtype == "fp16" else True
    max_new = args.max_new if args.max_new > 0 else 444
    full = dims["d"] <= 768  # Whisper-small and below: the whole clip fits the 10-30 s budget even on one thread

    def run(n_threads, n_new):
        orc = oracle.Oracle(cfg, weights, bf16_policy=policy, threads=n_threads)
        t0 = time.perf_counter()
        mel, _, _ = oracle.log_mel(clip, dims["n_mels"])
        t_fe = time.perf_counter() - t0
        ck, cv = orc.encoder(mel)
        t_enc = time.perf_counter() - t0
        ids = orc.greedy(ck, cv, "zh", max_new=n_new)
        t_all = time.perf_counter() - t0
        return ids, t_fe, t_enc, t_all

    n_all = max_new if full else min(max_new, 96)
    cpu_ids, t_fe_all, t_enc, t_cpu = run(threads, n_all)
    steps_all = 4 + len(cpu_ids)
    t_full = t_cpu if steps_all >= dec_steps else t_enc + (t_cpu - t_enc) / steps_all * dec_steps
    agree = 0
    for a, b in zip(cpu_ids, gpu_ids):
        if a != b:
            break
        agree += 1
    how = "measured in full" if steps_all >= dec_steps else f"{steps_all} decoder steps measured, scaled to {dec_steps:.0f}"
    out = {"value": round(1.0 / t_full, 5), "unit": "clips/s", "cores": threads, "kind": "port",
           "cpu_model": cpu_model_string(), "host_cores": ncpu,
           "sample": f"clip 0, {how}: front-end + encoder {t_enc:.2f} s + {steps_all} decoder steps "
                     f"{(t_cpu - t_enc) / steps_all * 1e3:.1f} ms/step = {t_cpu:.2f} s; CPU oracle ({dtype} policy), {threads} OpenMP threads of {ncpu}",
           "ids_agree_prefix": f"{agree}/{len(cpu_ids)}"}
    n_one = max_new if full else 4
    ids1, t_fe1, t_enc1, t_one = run(1, n_one)
    steps_one = 4 + len(ids1)
    t_full1 = t_one if steps_one >= dec_steps else t_enc1 + (t_one - t_enc1) / steps_one * dec_steps
    how1 = "measured in full" if steps_one >= dec_steps else f"{steps_one} decoder steps measured, scaled to {dec_steps:.0f}"
    out["single_thread"] = {"value": round(1.0 / t_full1, 5), "unit": "clips/s", "cores": 1,
                            "sample": f"clip 0, {how1}: front-end {t_fe1 * 1e3:.1f} ms, encoder {t_enc1 - t_fe1:.2f} s, {steps_one} decoder steps "
                                      f"{(t_one - t_enc1) / steps_one * 1e3:.1f} ms/step = {t_one:.2f} s",
                            f"ids_equal_{threads}_threads": ids1 == cpu_ids[:len(ids1)]}
    fe = {f"port_{threads}_threads_ms": round(t_fe_all * 1e3, 2), "port_1_thread_ms": round(t_fe1 * 1e3, 2)}
    if oracle.ref_lib() is not None:  # the reference's own librosa.h front-end, compiled by oracle/Makefile `ref`
        t6 = time.perf_counter()
        oracle.log_mel(clip, dims["n_mels"], use_ref=True)
        fe["reference_1_thread_ms"] = round((time.perf_counter() - t6) * 1e3, 2)
    out["frontend_only"] = fe
    oracle.lib().orc_set_threads(min(ncpu, 16))
    return out


# ------------------------------------------------------------------------------------------- one rank
def run_rank(args) -> int:
    import numpy as np

    import modelgen

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
              f"(python bench.py --gpus {args.gpus} does it by itself)", file=sys.stderr)
        return 2
    rehearsal = os.environ.get("AXW_BENCH_REHEARSAL") == "1"  # CPU test of the N>1 plumbing: gloo, no GPU, no engine
    backend = "gloo" if rehearsal else os.environ.get("AXW_BENCH_BACKEND", "nccl")
    # AXW_BENCH_ONE_DEVICE=1: every rank uses GPU 0 (a rehearsal of N ranks with real engines on a one-GPU box; gloo)
    dev_index = 0 if os.environ.get("AXW_BENCH_ONE_DEVICE") == "1" else local_rank
    # AXW_BENCH_FORCE_DIST=1: run the process group, the barriers and the result gather even with one rank
    use_dist = world > 1 or os.environ.get("AXW_BENCH_FORCE_DIST") == "1"
    B = args.batch if args.batch > 0 else (1 if world == 1 else 64)
    dims = modelgen.DIMS[args.model]
    dtype = args.dtype or ("fp16" if args.model == "turbo" else "bf16")

    import torch
    import torch.distributed as dist

    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its communicator is created; stdout must carry the ONE JSON
        # line only, so everything until then goes to stderr (file-descriptor level: the banner comes from C code)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = None
    if not rehearsal:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else None  # where the gathered rows live

    def barrier():
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    def sync():
        if dev is not None:
            torch.cuda.synchronize(dev)

    from whisper_axera_amd import dp

    clips = np.stack([modelgen.synth_clip(rank * B + i, N_SAMP) for i in range(B)])
    out = {}
    rc = 0
    if rehearsal:
        eng = RehearsalEngine()

        fail_rank = int(os.environ.get("AXW_BENCH_FAIL_RANK", "-1"))  # test hook: this rank dies inside its second step
        calls = [0]

        def one_step():
            calls[0] += 1
            if rank == fail_rank and calls[0] == 2:
                raise RuntimeError(f"rank {rank}: injected failure (AXW_BENCH_FAIL_RANK)")
            ids = eng.run(clips, args.max_new)
            return dp.gather_ids(ids, B, device=None) if use_dist else ids

        dt, ids = timed_steps(one_step, args.steps, args.warmup, sync, barrier)
        stage, dec_steps, n_tok = {}, 0, float(np.mean([len(r) for r in ids]))
    else:
        import whisper_axera_amd as wa

        # ---- synthetic-weight model directory (no weights exist in the reference or this image); the weights file's
        #      dtype (BF16 / F16) selects the engine build
        model_root = args.model_dir + ("_f16" if dtype == "fp16" else "")
        mdir = os.path.join(model_root, args.model)
        if local_rank == 0:
            ensure_model_dir(modelgen, args.model_dir, args.model, dims, dtype)
        barrier()
        eng = wa.Whisper(args.model, model_root, "zh", device=dev_index, max_batch=B)
        assert eng.L.AX_WHISPER_GetConfigInt(eng.h, b"fp16") == (1 if dtype == "fp16" else 0)
        stream = torch.cuda.current_stream(dev)
        eng.set_stream(stream.cuda_stream)
        d_pcm = torch.from_numpy(clips).to(dev)
        sync()
        stage = {"frontend_ms": 0.0, "encoder_ms": 0.0, "decode_ms": 0.0, "steps": 0}
        gathered = {}

        def one_step():
            ids = eng.run_device_tokens(d_pcm.data_ptr(), N_SAMP, [N_SAMP] * B, max_new=args.max_new)
            tm = eng.timings()
            for k in stage:
                stage[k] += tm[k]
            if use_dist:  # the ONE collective of the path: result gather over RCCL/xGMI
                gathered["ids"] = dp.gather_ids(ids, B, device=coll_dev)
            return ids

        for _ in range(args.warmup):
            one_step()
        for k in stage:
            stage[k] = 0
        dt, ids = timed_steps(one_step, args.steps, 0, sync, barrier)
        if use_dist and len(gathered["ids"]) != world * B:
            print(f"bench.py: gathered {len(gathered['ids'])} id rows, expected {world * B}", file=sys.stderr)
            rc = 3
        n_tok = float(np.mean([len(r) for r in ids]))
        dec_steps = stage["steps"] / args.steps

    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    clips_per_s = world * B * args.steps / dt

    out = {
        "metric": "clips_per_sec (30 s clips, greedy decode, whisper-%s)" % args.model,
        "value": round(clips_per_s, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"whisper-{args.model} {dtype}, batch {B}/GPU, 30 s synthetic 16 kHz clips resident in HBM, "
                               f"greedy decode {n_tok:.0f} ids/clip ({dec_steps:.0f} decoder steps)",
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}", "weights": "seeded synthetic",
                   "collective": (f"{backend} all_gather of int32 [count, ids] rows, once per step" if use_dist else "none")},
        "rtf": round(dt / args.steps / (B * 30.0), 6),
    }
    if os.environ.get("AXW_BENCH_ONE_DEVICE") == "1" and world > 1:
        out["data"] = "synthetic; REHEARSAL on ONE device (AXW_BENCH_ONE_DEVICE=1): all ranks share GPU 0 — not a multi-GPU measurement"
    def single_gpu_reference(step_fn):
        """N > 1: the same per-GPU workload on ONE GPU in the same run — rank 0 alone times a few more steps (no collective
        inside) while the other ranks wait at the second barrier — the reference point of the weak-scaling ratio
        value / (N * this): N=1 runs of this script default to batch 1, so the ratio of two driver lines says nothing."""
        if not (use_dist and world > 1):
            return
        barrier()
        if rank == 0:
            n1 = max(1, min(args.steps, 5))
            t0 = time.perf_counter()
            for _ in range(n1):
                step_fn()
            sync()
            t1 = (time.perf_counter() - t0) / n1
            out["single_gpu_same_batch"] = {"value": round(B / t1, 3), "unit": "clips/s", "ms_per_step": round(t1 * 1e3, 3), "steps": n1,
                                            "what": f"rank 0 alone, {B} clips, the other {world - 1} ranks idle at a barrier "
                                                    "(the one-GPU point of this workload; the driver computes efficiency from its own per-N lines)"}
        barrier()

    if rehearsal:
        out["data"] = "REHEARSAL (AXW_BENCH_REHEARSAL=1): no GPU and no engine ran — plumbing test of the N>1 path only"
        single_gpu_reference(lambda: eng.run(clips, args.max_new))
        out["value"] = None
        out["gathered_rows"] = len(ids)
    else:
        out["stage_ms"] = {k: round(v / args.steps, 3) for k, v in stage.items() if k != "steps"}
        single_gpu_reference(lambda: eng.run_device_tokens(d_pcm.data_ptr(), N_SAMP, [N_SAMP] * B, max_new=args.max_new))
        if rank == 0 and args.no_extras:
            out["roofline"] = None  # --no-extras: a profiling run; the roofline legs would put other launches into the trace
        elif rank == 0:
            out["roofline"] = roofline_for(eng, dims, args.model, B, dec_steps, stage["decode_ms"] / args.steps)
            out["stage_rooflines"] = stage_rooflines(eng, dims, args.model, B)
            # SURVEY §8(d) defines RTF on the host-pointer call (H2D of the PCM and D2H of the ids included): same
            # engine, same clips, through AX_WHISPER_RunPCMBatchTokens. Never `value`.
            hsteps = max(1, min(args.steps, 5))
            host_clips = [clips[i] for i in range(B)]
            eng.run_tokens_batch(host_clips, max_new=args.max_new)
            sync()
            t0 = time.perf_counter()
            for _ in range(hsteps):
                eng.run_tokens_batch(host_clips, max_new=args.max_new)
            th = (time.perf_counter() - t0) / hsteps
            out["host_pcm"] = {"rtf": round(th / (B * 30.0), 6), "clips_per_s": round(B / th, 4), "ms_per_step": round(th * 1e3, 3),
                               "what": "AX_WHISPER_RunPCMBatchTokens: PCM in host memory, H2D + D2H inside the timed call"}
            # forced decode lengths beside the full context (real utterances end after ~15-150 ids; synthetic
            # weights never emit eot), and demo.wav with its TRUE duration as the RTF denominator (whisper_cli's definition,
            # whisper_cli.cpp:93-103; the README figures are quoted on that 4.2 s clip)
            if B == 1 and args.max_new == 0:
                forced = {}
                for n_new in (32, 128):
                    eng.run_device_tokens(d_pcm.data_ptr(), N_SAMP, [N_SAMP], max_new=n_new)
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        eng.run_device_tokens(d_pcm.data_ptr(), N_SAMP, [N_SAMP], max_new=n_new)
                    tf = (time.perf_counter() - t0) / 5
                    forced[str(n_new)] = {"ms": round(tf * 1e3, 3), "rtf": round(tf / 30.0, 6), "clips_per_s": round(1.0 / tf, 2)}
                out["forced_decode_lengths"] = forced
                import wave
                w = wave.open(os.path.join(ROOT, "tests", "golden", "demo.wav"))
                demo = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / np.float32(32768.0)
                eng.run_tokens_batch([demo], max_new=32)
                t0 = time.perf_counter()
                for _ in range(5):
                    eng.run_tokens_batch([demo], max_new=32)
                td = (time.perf_counter() - t0) / 5
                out["demo_wav"] = {"ms": round(td * 1e3, 3), "audio_s": round(len(demo) / 16000.0, 4), "ids": 32,
                                   "rtf_true_duration": round(td / (len(demo) / 16000.0), 5),
                                   "what": "tests/golden/demo.wav through AX_WHISPER_RunPCMBatchTokens, 32 ids forced (synthetic weights), "
                                           "RTF over the clip's true 4.2 s as whisper_cli computes it"}
        eng.close()
        sync()
        # ---- the other configs of BASELINE.json in the same run, so that the driver's line carries every headline number:
        #      configs[2] (small, batch 64: "clips/s at batch"), configs[3] (turbo fp16, batch 16), configs[0] (tiny, demo.wav,
        #      the CPU path timed beside the GPU)
        if rank == 0 and world == 1 and B == 1 and args.model == "small":
            n2 = max(1, min(args.steps, 5))
            if not args.no_batch64:
                leg, ids2 = batch_leg(torch, dev, dev_index, sync, "small", dtype, 64, n2, args.max_new, args.model_dir)
                leg["clip0_ids_equal_batch1"] = ids2[0] == ids[0]
                out["batch64"] = leg
            if not args.no_batch64 and args.max_new == 0:
                # two / three clips per call = their greedy loops in ONE persistent launch (decode_persistent2.hip): the low-load end
                # of a server. Algorithmic bytes of that launch: the decoder-layer weights and the vocabulary projection stream once
                # per step for ALL its clips, cross K/V once per clip; the later clips' self-attention caches are read from global
                # memory (clip 0's lives in LDS). Duration: the engine's hipEvents around the launch (stage "decode").
                for nb in (2, 3):
                    leg, idsp = batch_leg(torch, dev, dev_index, sync, "small", dtype, nb, n2, 0, args.model_dir, rooflines=False)
                    leg["clip0_ids_equal_batch1"] = idsp[0] == ids[0]
                    d_, L_, nv_, s_ = dims["d"], dims["dec_layers"], dims["n_vocab"], 2
                    n_st = int(round(len(idsp[0]) + 4))
                    bts = n_st * (s_ * L_ * 14 * d_ * d_ + nb * s_ * 2 * L_ * 1500 * d_) + max(n_st - 3, 0) * s_ * nv_ * d_ \
                        + (nb - 1) * sum(s_ * 2 * L_ * (t + 1) * d_ for t in range(n_st))
                    l_s = leg["stage_ms"]["decode_ms"] * 1e-3
                    pmcn, _ = load_pmc("small", nb)
                    leg["roofline"] = {"kernel": "decode_persistent_kernel<..., %d> (the greedy loops of %d clips: %d decoder steps in one launch)" % (nb, nb, n_st),
                                       "bound": "hbm", "achieved": round(bts / l_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(bts / l_s / 1e9 / HBM_PEAK_GBS, 4),
                                       "traffic": (pmcn.get("decode_persistent_kernel") or {}).get("hbm_bytes_per_launch"),
                                       "launch_ms": round(l_s * 1e3, 3), "bytes_per_launch": int(bts),
                                       "us_per_decode_step": round(l_s * 1e6 / n_st, 2)}
                    out["batch%d" % nb] = leg
                # four and eight clips per call: the clip-block step sequence (one step = ~87 dependent launches whatever the clip
                # count). The low-load ladder of a server in one place: clips/s at 1, 2, 3, 4, 8 clips per call
                for nb in (4, 8):
                    leg, idsp = batch_leg(torch, dev, dev_index, sync, "small", dtype, nb, min(n2, 3), 0, args.model_dir, rooflines=False)
                    leg["clip0_ids_equal_batch1"] = idsp[0] == ids[0]
                    leg["roofline"] = step_chain_roofline(dims, nb, len(idsp[0]) + 4, leg["stage_ms"]["decode_ms"])
                    out["batch%d" % nb] = leg
                out["low_load_clips_per_s"] = {"1": out["value"], **{str(nb): out["batch%d" % nb]["value"] for nb in (2, 3, 4, 8)}}
            if not args.no_realistic and args.max_new == 0:
                # realistic utterance lengths (SURVEY §8d asks for them beside the full context): every clip of the 64 leaves the
                # loop at its own budget of 60-150 ids; then the same workload through the slot scheduler (384 clips, 64 slots)
                leg, _ = batch_leg(torch, dev, dev_index, sync, "small", dtype, 64, n2, 0, args.model_dir, rooflines=False,
                                   budgets=realistic_budgets(64))
                out["batch64_len100"] = leg
                out["stream64"] = stream_leg(torch, dev, dev_index, sync, "small", dtype, 64, 384, args.model_dir)
                # the slot count is a deployment knob (a slot costs 57 MB of cross K/V + 16.5 MB of self K/V): the step's chain of small
                # GEMMs is paid once whatever the slot count, so twice the slots serve more clips per second
                out["stream128"] = stream_leg(torch, dev, dev_index, sync, "small", dtype, 128, 768, args.model_dir, policies=(1,), group_sizes=False)
            if not args.no_batch256 and args.max_new == 0:
                # not a BASELINE config: the same workload at the batch 288 GB of HBM invite (19 GB of K/V caches). A decoder
                # step's chain of small GEMMs is paid once whatever the batch: 17.9 us per clip and step at 64, 13.1 at 256.
                leg, ids3 = batch_leg(torch, dev, dev_index, sync, "small", dtype, 256, min(n2, 2), 0, args.model_dir)
                leg["clip0_ids_equal_batch1"] = ids3[0] == ids[0]
                out["batch256"] = leg
            if not args.no_turbo and args.max_new == 0 and dtype == "bf16":
                leg, _ = batch_leg(torch, dev, dev_index, sync, "turbo", "fp16", 16, min(n2, 3), 0, args.model_dir)
                out["turbo_fp16_b16"] = leg
            if not args.no_config0 and args.max_new == 0 and dtype == "bf16":
                out["config0"] = config0_leg(torch, dev, dev_index, sync, args.model_dir)
        # ---- CPU baseline: the oracle ("port") on this box's host cores, one clip of the same workload
        if rank == 0 and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, dims, mdir, clips[0], ids[0], dec_steps, dtype)

    if rank == 0 and not rehearsal:
        out["config"]["other_configs"] = other_configs_summary(out)

    if use_dist:
        barrier()  # rank 0 has run its roofline legs meanwhile: leave together
        dist.destroy_process_group()
    sys.stdout.flush()
    if saved_stdout is not None:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    return rc


def main():
    args = parse_args()
    if args.no_extras:
        args.no_batch64 = args.no_batch256 = args.no_turbo = args.no_config0 = args.no_cpu_baseline = args.no_realistic = True
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)  # nothing in this process has imported torch or touched a GPU
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
