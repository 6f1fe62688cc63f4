"""Model-directory tooling for the MI355X Whisper engine (host side, no GPU needed).

The reference loads ``{model_path}/{model_type}/{model_type}-encoder.axmodel``,
``-decoder.axmodel``, ``-tokens.txt`` and ``_config.json`` (cpp/src/Whisper.cpp:86-90).
The two ``.axmodel`` NPU blobs are replaced by one ``{model_type}.safetensors`` holding the
openai-whisper state_dict names (``encoder.conv1.weight`` ... ``decoder.ln.bias``) in bf16 or
fp32; the tokens and config files keep the reference's formats (model_convert/
export_onnx.py:391-417 and :592-629).

No weights exist in the reference or this image, so benchmarks and parity tests run on seeded
synthetic weights produced here (numpy PCG64: identical on every machine).
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict

import numpy as np

# [upstream openai-whisper dims; the reference reads them from {type}_config.json at run time]
DIMS = {
    #            n_mels d     heads enc_l dec_l n_vocab  n_langs
    "tiny":   dict(n_mels=80,  d=384,  heads=6,  enc_layers=4,  dec_layers=4,  n_vocab=51865, n_langs=99),
    "base":   dict(n_mels=80,  d=512,  heads=8,  enc_layers=6,  dec_layers=6,  n_vocab=51865, n_langs=99),
    "small":  dict(n_mels=80,  d=768,  heads=12, enc_layers=12, dec_layers=12, n_vocab=51865, n_langs=99),
    "turbo":  dict(n_mels=128, d=1280, heads=20, enc_layers=32, dec_layers=4,  n_vocab=51866, n_langs=100),
    # reduced dims for fast parity tests (same graph, head_dim 64)
    "micro":  dict(n_mels=80,  d=128,  heads=2,  enc_layers=2,  dec_layers=2,  n_vocab=51865, n_langs=99),
    "mini":   dict(n_mels=80,  d=256,  heads=4,  enc_layers=2,  dec_layers=3,  n_vocab=51865, n_langs=99),
    # turbo-shaped front half at reduced width: 128 mels, 100 languages (adds yue), n_vocab 51866, shallow decoder
    "miniturbo": dict(n_mels=128, d=256, heads=4, enc_layers=3, dec_layers=2, n_vocab=51866, n_langs=100),
    # full widths of base / medium / large at reduced depth: every width-dependent kernel instantiation, cheap oracle
    "w512":   dict(n_mels=80,  d=512,  heads=8,  enc_layers=1,  dec_layers=2,  n_vocab=51865, n_langs=99),
    "w1024":  dict(n_mels=80,  d=1024, heads=16, enc_layers=1,  dec_layers=2,  n_vocab=51865, n_langs=99),
    "w1280":  dict(n_mels=128, d=1280, heads=20, enc_layers=1,  dec_layers=2,  n_vocab=51866, n_langs=100),
}

N_AUDIO_CTX = 1500
N_TEXT_CTX = 448

LANGUAGE_CODES = (
    "en,zh,de,es,ru,ko,fr,ja,pt,tr,pl,ca,nl,ar,sv,it,id,hi,fi,vi,he,uk,el,ms,cs,ro,da,hu,ta,no,"
    "th,ur,hr,bg,lt,la,mi,ml,cy,sk,te,fa,lv,bn,sr,az,sl,kn,et,mk,br,eu,is,hy,ne,mn,bs,kk,sq,sw,"
    "gl,mr,pa,si,km,sn,yo,so,af,oc,ka,be,tg,sd,gu,am,yi,lo,uz,fo,ht,ps,tk,nn,mt,sa,lb,my,bo,tl,"
    "mg,as,tt,haw,ln,ha,ba,jw,su,yue"
).split(",")


def make_config(model_type: str, dims: dict) -> dict:
    """The ``{type}_config.json`` the reference's exporter writes (export_onnx.py:592-629).

    Special ids follow the tokenizer layout (python/whisper_tokenizer.py:340-355): 50257 base
    ranks, then eot, sot, languages, translate, transcribe, startoflm, startofprev, nospeech,
    notimestamps, timestamps.
    """
    n_langs = dims["n_langs"]
    base = 50257
    eot, sot = base, base + 1
    lang_tokens = [sot + 1 + i for i in range(n_langs)]
    translate = sot + 1 + n_langs
    transcribe = translate + 1
    sot_lm, sot_prev, no_speech, no_timestamps = transcribe + 1, transcribe + 2, transcribe + 3, transcribe + 4
    return {
        "model_type": f"whisper-{model_type}",
        "version": "1",
        "maintainer": "k2-fsa",
        "n_mels": dims["n_mels"],
        "n_audio_ctx": N_AUDIO_CTX,
        "n_audio_state": dims["d"],
        "n_audio_head": dims["heads"],
        "n_audio_layer": dims["enc_layers"],
        "n_vocab": dims["n_vocab"],
        "n_text_ctx": N_TEXT_CTX,
        "n_text_state": dims["d"],
        "n_text_head": dims["heads"],
        "n_text_layer": dims["dec_layers"],
        "sot_sequence": f"{sot},{lang_tokens[0]},{transcribe}",
        "all_language_tokens": ",".join(map(str, lang_tokens)),
        "all_language_codes": ",".join(LANGUAGE_CODES[:n_langs]),
        "sot": sot,
        "sot_index": 0,
        "eot": eot,
        "blank_id": 220,
        "is_multilingual": 1,
        "no_speech": no_speech,
        "non_speech_tokens": "",
        "transcribe": transcribe,
        "translate": translate,
        "sot_prev": sot_prev,
        "sot_lm": sot_lm,
        "no_timestamps": no_timestamps,
    }


def sinusoids(length: int, channels: int) -> np.ndarray:
    """upstream whisper/model.py sinusoids() — the encoder's fixed positional embedding."""
    half = channels // 2
    inc = np.float32(np.log(10000.0)) / np.float32(half - 1)
    inv = np.exp(-inc * np.arange(half, dtype=np.float32)).astype(np.float32)
    st = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16 -> fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32).reshape(x.shape)


def tensor_names(dims: dict):
    """(name, shape, kind) for every tensor of the openai-whisper state_dict."""
    d, nm, nv = dims["d"], dims["n_mels"], dims["n_vocab"]
    out = [
        ("encoder.conv1.weight", (d, nm, 3), "w"), ("encoder.conv1.bias", (d,), "b"),
        ("encoder.conv2.weight", (d, d, 3), "w"), ("encoder.conv2.bias", (d,), "b"),
        ("encoder.positional_embedding", (N_AUDIO_CTX, d), "sin"),
    ]

    def block(prefix, cross):
        t = [(f"{prefix}.attn_ln.weight", (d,), "g"), (f"{prefix}.attn_ln.bias", (d,), "b")]
        for a in (["attn", "cross_attn"] if cross else ["attn"]):
            if a == "cross_attn":
                t += [(f"{prefix}.cross_attn_ln.weight", (d,), "g"), (f"{prefix}.cross_attn_ln.bias", (d,), "b")]
            t += [
                (f"{prefix}.{a}.query.weight", (d, d), "w"), (f"{prefix}.{a}.query.bias", (d,), "b"),
                (f"{prefix}.{a}.key.weight", (d, d), "w"),
                (f"{prefix}.{a}.value.weight", (d, d), "w"), (f"{prefix}.{a}.value.bias", (d,), "b"),
                (f"{prefix}.{a}.out.weight", (d, d), "w"), (f"{prefix}.{a}.out.bias", (d,), "b"),
            ]
        t += [
            (f"{prefix}.mlp_ln.weight", (d,), "g"), (f"{prefix}.mlp_ln.bias", (d,), "b"),
            (f"{prefix}.mlp.0.weight", (4 * d, d), "w"), (f"{prefix}.mlp.0.bias", (4 * d,), "b"),
            (f"{prefix}.mlp.2.weight", (d, 4 * d), "w"), (f"{prefix}.mlp.2.bias", (d,), "b"),
        ]
        return t

    for i in range(dims["enc_layers"]):
        out += block(f"encoder.blocks.{i}", False)
    out += [("encoder.ln_post.weight", (d,), "g"), ("encoder.ln_post.bias", (d,), "b")]
    out += [("decoder.token_embedding.weight", (nv, d), "emb"),
            ("decoder.positional_embedding", (N_TEXT_CTX, d), "pos")]
    for i in range(dims["dec_layers"]):
        out += block(f"decoder.blocks.{i}", True)
    out += [("decoder.ln.weight", (d,), "g"), ("decoder.ln.bias", (d,), "b")]
    return out


def synth_weights(dims: dict, seed: int = 0, bf16: bool = True) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (fan-in scaled normals), optionally bf16-representable."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape, kind in tensor_names(dims):
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.5 / np.sqrt(fan_in))
        elif kind == "b":
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        elif kind == "g":
            w = np.float32(1.0) + rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "emb":
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        elif kind == "pos":
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.01)
        elif kind == "sin":
            w = sinusoids(*shape)
        else:
            raise ValueError(kind)
        out[name] = bf16_round(w) if bf16 else w.astype(np.float32)
    return out


def _round16(x: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "F16":
        return x.astype(np.float16).astype(np.float32)
    if dtype == "BF16":
        return bf16_round(x)
    return x.astype(np.float32)


def realistic_weights(dims: dict, seed: int = 0, dtype: str = "BF16", outlier_rel: float = 14.0, hidden_peak: float = 6000.0,
                      logit_gain: float = 1.0, temps=(0.2, 2.0, 4.0, 8.0), out_scale: float = 1.0,
                      gain_centre: float = 4.0, v_damp: float = 2.0) -> Dict[str, np.ndarray]:
    """Seeded weights whose ACTIVATIONS look like a trained Whisper's instead of N(0, 0.02) everywhere (no checkpoint exists
    in the reference or this image; tests/golden/probe_realistic_stats.py prints what these produce):

      * a few residual channels carry |x| of 10^2..10^3 in both stacks (a bias and a scaled row of an early mlp.2, the decoder's
        positional embedding), so every later LayerNorm sees one channel that owns most of the variance;
      * LayerNorm gains spread over 0.2..30 (log-normal, re-amplifying what the outlier channel shrank; tiny on the outlier
        channels themselves, as trained models do), biases up to +-2;
      * attention heads cycle through four temperatures: flat over all keys (scores ~1e-2), ordinary, peaked, and saturated
        (one key >> the rest) — query and key rows of a head scaled together (export_onnx.py:116-126 scales both);
      * a few FFN hidden units run at `hidden_peak` (thousands: a third of the half range when it is 6000; GELU is the
        identity there) and feed back through small mlp.2 columns;
      * token-embedding rows with log-normal norms and a final LayerNorm gain that put the logit std at >= 5, top-2 margins
        from ~1e-3 to ~10.
    Coordinate 0 of the residual stream is left ordinary (tests/eot_case.py reserves it). Values are rounded to `dtype`
    (both sides of a parity test then hold identical numbers)."""
    w = dict(synth_weights(dims, seed, bf16=False))
    rng = np.random.Generator(np.random.PCG64(seed * 1000003 + 7919))
    d, H = dims["d"], dims["heads"]
    f32 = np.float32
    oc = rng.choice(np.arange(1, d), size=4, replace=False)  # outlier channels: two per stack
    outlier = outlier_rel * np.sqrt(d)                       # 158 (d = 128) .. 388 (768) .. 500 (1280)
    shrink = outlier_rel                                     # what a LayerNorm behind an outlier of that size divides by

    def gains(n, centre, lo_ch=()):
        g = centre * np.exp(rng.standard_normal(n) * 0.6)
        hot = rng.choice(n, size=max(1, n // 96), replace=False)
        g[hot] = rng.uniform(12.0, 30.0, size=len(hot))
        g = np.clip(g, 0.05, 30.0)
        for c in lo_ch:
            g[c] = rng.uniform(0.02, 0.1)
        return g.astype(f32)

    def ln_bias(n):
        b = rng.standard_normal(n) * 0.02
        hot = rng.choice(n, size=max(1, n // 128), replace=False)
        b[hot] = rng.uniform(-2.0, 2.0, size=len(hot))
        return b.astype(f32)

    # temps: factor on a head's query AND key rows: scores go with its square

    def heads(prefix, first):
        for h in range(H):
            t = f32(temps[(h + 3 * first) % 4])
            for n in ("query.weight", "query.bias", "key.weight"):
                w[f"{prefix}.{n}"][h * 64:(h + 1) * 64] *= t
            if t > 1.0 and v_damp:
                # a head whose scores have std s turns a relative perturbation of q or k into s times that much on its
                # output; with every head writing O(1) into the stream a 12-layer stack amplifies rounding 1e4-fold (fp32
                # itself is then 0.1 logits from fp64, bfloat16 tens). Trained stacks are not chaotic: hot heads write
                # proportionally less, so that sensitivity x weight is the same for every temperature
                for n in ("value.weight", "value.bias"):
                    w[f"{prefix}.{n}"][h * 64:(h + 1) * 64] /= t ** v_damp

    def stack(side, n_layers, c_bias, c_row, cross):
        for l in range(n_layers):
            p = f"{side}.blocks.{l}"
            behind = l > 0 or side == "decoder"   # an outlier channel is already in the stream
            centre = min(shrink, gain_centre) if behind else 1.0
            lo = (c_bias, c_row) if behind else ()
            for ln in (["attn_ln", "cross_attn_ln", "mlp_ln"] if cross else ["attn_ln", "mlp_ln"]):
                w[f"{p}.{ln}.weight"] = gains(d, centre, lo)
                w[f"{p}.{ln}.bias"] = ln_bias(d)
            heads(f"{p}.attn", l)
            if cross:
                heads(f"{p}.cross_attn", l + 2)
            # a few hidden units far out: mlp.0 rows scaled up with a positive bias, their mlp.2 columns small
            hot = rng.choice(4 * d, size=3, replace=False)
            for j in hot:
                w[f"{p}.mlp.0.weight"][j] *= f32(40.0)
                w[f"{p}.mlp.0.bias"][j] = f32(hidden_peak * rng.uniform(0.5, 1.0))
                w[f"{p}.mlp.2.weight"][:, j] *= f32(1e-4)
            # out_scale > 1: what a block adds outgrows the stream. Kept at 1: every saturated head multiplies a relative
            # perturbation by its score magnitude, and with larger block outputs fp32 itself drifts from fp64 tenfold
            # per layer (probe_realistic_stats.py prints fp32-vs-fp64 next to the statistics)
            for n in (["attn.out", "cross_attn.out", "mlp.2"] if cross else ["attn.out", "mlp.2"]):
                w[f"{p}.{n}.weight"] *= f32(out_scale)
        p0 = f"{side}.blocks.0"
        w[f"{p0}.mlp.2.bias"][c_bias] = f32(outlier if side == "encoder" else -outlier)
        w[f"{p0}.mlp.2.weight"][c_row] *= f32(20.0)

    stack("encoder", dims["enc_layers"], oc[0], oc[1], False)
    stack("decoder", dims["dec_layers"], oc[2], oc[3], True)
    w["encoder.ln_post.weight"] = gains(d, min(shrink, gain_centre), (oc[0], oc[1]))
    w["encoder.ln_post.bias"] = ln_bias(d)
    # decoder: the outlier is there from the first LayerNorm on (positional embedding), varying along the context
    pe = w["decoder.positional_embedding"] * f32(10.0)
    t = np.arange(pe.shape[0], dtype=np.float64)
    pe[:, oc[2]] += (0.75 * outlier * (1.0 + 0.2 * np.sin(t / 7.0))).astype(f32)
    w["decoder.positional_embedding"] = pe
    # ... and is taken out again by the last block (mean of the two injections), so that the final LayerNorm sees
    # only its variation along the context: trained stacks remove their massive activations before the output
    w[f"decoder.blocks.{dims['dec_layers'] - 1}.mlp.2.bias"][oc[2]] += f32(0.25 * outlier)
    emb = w["decoder.token_embedding.weight"] * f32(12.0)
    emb *= np.exp(rng.standard_normal(emb.shape[0]) * 0.3).astype(f32)[:, None]
    emb[:, oc[2]] *= f32(0.1)   # the outlier channel reaches the logits through a tiny gain only
    w["decoder.token_embedding.weight"] = emb
    near_dup = True
    # the stream's non-outlier part grows with depth, and the logit std with it: ~8-12 at every depth with this
    g = gains(d, float(np.clip(80.0 / np.sqrt(d), 1.5, 8.0)) * logit_gain / (dims["dec_layers"] / 2.0), (oc[2], oc[3]))
    w["decoder.ln.weight"] = g
    w["decoder.ln.bias"] = ln_bias(d)
    out = {k: _round16(v, dtype) for k, v in w.items()}
    if near_dup:
        # the top-2 margin spectrum of a trained vocabulary (near-synonyms, spelling variants): every other text row is its
        # left neighbour with k of its entries moved to the next 16-bit value, k = 0 (an exact tie: the LOWER id must win,
        # Whisper.cpp:42-45), 1, 2, 8, 32, all — margins from 0 over ~1e-3 to several units whenever such a pair leads
        e = out["decoder.token_embedding.weight"]
        ks = (0, 1, 2, 8, 32, d)
        for v in range(1, 50257, 2):
            if (v // 2) % 2:   # every other pair stays two independent rows (margins of units)
                continue
            k = ks[(v // 4) % len(ks)]
            row = e[v - 1].copy()
            if k:
                idx = rng.choice(d, size=k, replace=False) if k < d else np.arange(d)
                row[idx] = _next16(row[idx], dtype)
            e[v] = row
    return out


def _next16(x: np.ndarray, dtype: str) -> np.ndarray:
    """The next representable value of `dtype` away from zero (one unit in the last place)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dtype == "F16":
        h = x.astype(np.float16)
        return (h.view(np.uint16) + np.uint16(1)).view(np.float16).astype(np.float32)
    step = np.uint32(0x10000 if dtype == "BF16" else 0x100)
    return (x.view(np.uint32) + step).view(np.float32)


# ----------------------------------------------------------------------------- safetensors
def write_safetensors(path: str, tensors: Dict[str, np.ndarray], dtype: str = "BF16") -> None:
    """Minimal safetensors writer (8-byte LE header length, JSON header, raw little-endian data)."""
    header, blobs, off = {}, [], 0
    for name, arr in tensors.items():
        a = np.ascontiguousarray(arr, dtype=np.float32)
        if dtype == "BF16":
            raw = (bf16_round(a).view(np.uint32) >> np.uint32(16)).astype(np.uint16).tobytes()
        elif dtype == "F16":  # the dtype openai's and HF's Whisper checkpoints are published in
            raw = a.astype(np.float16).tobytes()
        elif dtype == "F32":
            raw = a.tobytes()
        else:
            raise ValueError(dtype)
        header[name] = {"dtype": dtype, "shape": list(a.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for b in blobs:
            f.write(b)


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """Read BF16/F16/F32 tensors back as fp32 numpy arrays."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n))
        data = f.read()
    out = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        s, e = meta["data_offsets"]
        raw = data[s:e]
        if meta["dtype"] == "BF16":
            a = (np.frombuffer(raw, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)
        elif meta["dtype"] == "F16":
            a = np.frombuffer(raw, dtype=np.float16).astype(np.float32)
        elif meta["dtype"] == "F32":
            a = np.frombuffer(raw, dtype=np.float32)
        else:
            raise ValueError(meta["dtype"])
        out[name] = a.reshape(meta["shape"]).copy()
    return out


def write_tokens(path: str, tiktoken_path: str | None) -> None:
    """``{type}-tokens.txt``: lines ``<base64> <rank>`` (export_onnx.py:391-417). ``tiktoken_path=None`` writes a
    synthetic vocabulary (rank i -> the bytes of ``t{i} ``) — for synthetic-weight models only; a path that does not
    exist is an error (a real checkpoint with a made-up vocabulary would transcribe garbage without a warning)."""
    import base64

    if tiktoken_path is not None and not os.path.exists(tiktoken_path):
        raise FileNotFoundError(f"vocabulary file {tiktoken_path} does not exist")
    if tiktoken_path:
        with open(tiktoken_path) as src, open(path, "w") as dst:
            for line in src:
                if line.strip():
                    tok, rank = line.split()
                    dst.write(f"{tok} {rank}\n")
        return
    with open(path, "w") as dst:
        for i in range(50257):
            dst.write(base64.b64encode(f"t{i} ".encode()).decode() + f" {i}\n")


def write_model_dir(root: str, model_type: str, dims: dict | None = None, seed: int = 0,
                    tiktoken_path: str | None = None, dtype: str = "BF16",
                    weights: Dict[str, np.ndarray] | None = None) -> str:
    """Create ``{root}/{model_type}/`` with config, tokens and weights; returns the directory."""
    dims = dims or DIMS[model_type]
    d = os.path.join(root, model_type)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"{model_type}_config.json"), "w") as f:
        json.dump(make_config(model_type, dims), f, indent=4)
    write_tokens(os.path.join(d, f"{model_type}-tokens.txt"), tiktoken_path)
    if weights is None:
        weights = synth_weights(dims, seed, bf16=(dtype == "BF16"))
        if dtype == "F16":  # synthetic fp16 model: weights exactly representable in fp16 (both sides hold equal values)
            weights = {k: v.astype(np.float16).astype(np.float32) for k, v in weights.items()}
    write_safetensors(os.path.join(d, f"{model_type}.safetensors"), weights, dtype)
    return d


def synth_clip(i: int, n_samples: int = 480000) -> np.ndarray:
    """BASELINE.md synthetic clip i: 0.1*N(0,1) + 0.2*sin(2*pi*220*(1 + i%7)*t), clipped to
    [-1, 1], PCG64 seed 1234+i, 16 kHz mono f32."""
    rng = np.random.Generator(np.random.PCG64(1234 + i))
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    x = 0.1 * rng.standard_normal(n_samples) + 0.2 * np.sin(2 * np.pi * 220.0 * (1 + i % 7) * t)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def synth_long_clip(seconds: int = 75, loud_at: int = 70) -> np.ndarray:
    """A clip longer than the 30 s window AND longer than the engine's 60 s staging row whose loudest second lies
    behind both: synthetic clip 8 at 1/20 of its level, with second `loud_at` at full level. The reference takes
    the log-mel maximum over ALL frames of the input before it keeps the first 3000 (Whisper.cpp:158-172), so the
    clamp floor of the kept frames is set by audio that is itself cut away."""
    x = synth_clip(8, seconds * 16000).astype(np.float32)
    g = np.full(len(x), np.float32(0.05))
    g[loud_at * 16000:(loud_at + 1) * 16000] = np.float32(1.0)
    return (x * g).astype(np.float32)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="write a synthetic-weight model directory")
    ap.add_argument("--model_type", "-t", default="small")
    ap.add_argument("--model_path", "-p", required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tiktoken", default=None)
    a = ap.parse_args()
    print(write_model_dir(a.model_path, a.model_type, seed=a.seed, tiktoken_path=a.tiktoken))
