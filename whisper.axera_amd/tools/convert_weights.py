"""Weight converters: real checkpoints -> the model directory libax_whisper.so loads.

    python convert_weights.py --openai small.pt       --model_type small --model_path ./models --tiktoken multilingual.tiktoken
    python convert_weights.py --hf openai/whisper-small-dir --model_type small --model_path ./models --tiktoken multilingual.tiktoken

The reference obtains its weights through ``whisper.load_model(name)`` inside model_convert/export_onnx.py:508 and
bakes them into NPU blobs; here the same tensors are written once as ``{type}.safetensors`` (openai-whisper
state_dict names) next to the reference's ``{type}_config.json`` / ``{type}-tokens.txt``. No checkpoint exists in
this environment, so the HF path is exercised in tests with a randomly initialised HF model.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import modelgen  # noqa: E402

_ATTN = (("q_proj", "query"), ("k_proj", "key"), ("v_proj", "value"), ("out_proj", "out"))


def hf_to_openai_names(sd: dict) -> dict:
    """transformers WhisperForConditionalGeneration state_dict -> openai-whisper names (numpy fp32)."""
    def arr(t):
        return t.detach().float().cpu().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float32)

    out = {}
    g = lambda k: arr(sd[k if k in sd else k.replace("model.", "", 1)])
    out["encoder.conv1.weight"], out["encoder.conv1.bias"] = g("model.encoder.conv1.weight"), g("model.encoder.conv1.bias")
    out["encoder.conv2.weight"], out["encoder.conv2.bias"] = g("model.encoder.conv2.weight"), g("model.encoder.conv2.bias")
    out["encoder.positional_embedding"] = g("model.encoder.embed_positions.weight")
    out["encoder.ln_post.weight"], out["encoder.ln_post.bias"] = g("model.encoder.layer_norm.weight"), g("model.encoder.layer_norm.bias")
    out["decoder.token_embedding.weight"] = g("model.decoder.embed_tokens.weight")
    out["decoder.positional_embedding"] = g("model.decoder.embed_positions.weight")
    out["decoder.ln.weight"], out["decoder.ln.bias"] = g("model.decoder.layer_norm.weight"), g("model.decoder.layer_norm.bias")
    for side in ("encoder", "decoder"):
        i = 0
        while any(k.endswith(f"{side}.layers.{i}.fc1.weight") for k in sd):
            s, t = f"model.{side}.layers.{i}", f"{side}.blocks.{i}"
            for hf, oa in _ATTN:
                out[f"{t}.attn.{oa}.weight"] = g(f"{s}.self_attn.{hf}.weight")
                if oa != "key":
                    out[f"{t}.attn.{oa}.bias"] = g(f"{s}.self_attn.{hf}.bias")
            out[f"{t}.attn_ln.weight"], out[f"{t}.attn_ln.bias"] = g(f"{s}.self_attn_layer_norm.weight"), g(f"{s}.self_attn_layer_norm.bias")
            if side == "decoder":
                for hf, oa in _ATTN:
                    out[f"{t}.cross_attn.{oa}.weight"] = g(f"{s}.encoder_attn.{hf}.weight")
                    if oa != "key":
                        out[f"{t}.cross_attn.{oa}.bias"] = g(f"{s}.encoder_attn.{hf}.bias")
                out[f"{t}.cross_attn_ln.weight"] = g(f"{s}.encoder_attn_layer_norm.weight")
                out[f"{t}.cross_attn_ln.bias"] = g(f"{s}.encoder_attn_layer_norm.bias")
            out[f"{t}.mlp.0.weight"], out[f"{t}.mlp.0.bias"] = g(f"{s}.fc1.weight"), g(f"{s}.fc1.bias")
            out[f"{t}.mlp.2.weight"], out[f"{t}.mlp.2.bias"] = g(f"{s}.fc2.weight"), g(f"{s}.fc2.bias")
            out[f"{t}.mlp_ln.weight"], out[f"{t}.mlp_ln.bias"] = g(f"{s}.final_layer_norm.weight"), g(f"{s}.final_layer_norm.bias")
            i += 1
    return out


def dims_from_weights(w: dict) -> dict:
    d = w["encoder.conv1.weight"].shape[0]
    n_enc = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("encoder.blocks."))
    n_dec = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("decoder.blocks."))
    n_vocab = w["decoder.token_embedding.weight"].shape[0]
    return dict(n_mels=w["encoder.conv1.weight"].shape[1], d=d, heads=d // 64, enc_layers=n_enc, dec_layers=n_dec,
                n_vocab=n_vocab, n_langs=n_vocab - 51765 - 1)  # 51865 -> 99 languages, 51866 -> 100


def read_hf_checkpoint(path: str) -> dict:
    """A transformers checkpoint: one .safetensors file, a directory with model.safetensors, or a directory with a
    sharded checkpoint (model.safetensors.index.json -> weight_map of shard files)."""
    import json

    if path.endswith(".safetensors"):
        return modelgen.read_safetensors(path)
    single, index = os.path.join(path, "model.safetensors"), os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(single):
        return modelgen.read_safetensors(single)
    if os.path.exists(index):
        shards = sorted(set(json.load(open(index))["weight_map"].values()))
        sd = {}
        for sh in shards:
            sd.update(modelgen.read_safetensors(os.path.join(path, sh)))
        return sd
    raise FileNotFoundError(f"{path}: neither model.safetensors nor model.safetensors.index.json")


def write_model(weights: dict, model_type: str, model_path: str, dtype: str = "BF16", tiktoken_path: str | None = None) -> str:
    """tiktoken_path is REQUIRED for a real checkpoint: {type}-tokens.txt must carry the vocabulary the model was
    trained with (the reference's exporter writes it from the tokenizer, export_onnx.py:391-417)."""
    if not tiktoken_path:
        raise ValueError("convert_weights needs the vocabulary: pass --tiktoken multilingual.tiktoken "
                         "(the reference ships it as python/assets/multilingual.tiktoken)")
    if not os.path.exists(tiktoken_path):
        raise FileNotFoundError(f"vocabulary file {tiktoken_path} does not exist")
    dims = dims_from_weights(weights)
    expect = {n: s for n, s, _ in modelgen.tensor_names(dims)}
    missing = [n for n in expect if n not in weights and n != "encoder.positional_embedding"]
    assert not missing, f"missing tensors: {missing[:5]}"
    for n, s in expect.items():
        if n in weights:
            assert tuple(weights[n].shape) == tuple(s), (n, weights[n].shape, s)
    return modelgen.write_model_dir(model_path, model_type, dims, weights={k: v for k, v in weights.items() if k in expect},
                                    dtype=dtype, tiktoken_path=tiktoken_path)


def main():
    ap = argparse.ArgumentParser()
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--openai", help="openai-whisper .pt checkpoint (dict with model_state_dict)")
    src.add_argument("--hf", help="directory or .safetensors of a transformers Whisper model")
    ap.add_argument("--model_type", "-t", required=True)
    ap.add_argument("--model_path", "-p", required=True)
    ap.add_argument("--dtype", default="BF16", choices=["BF16", "F16", "F32"])
    ap.add_argument("--tiktoken", required=True, help="multilingual.tiktoken vocabulary for {type}-tokens.txt")
    a = ap.parse_args()
    if a.openai:
        import torch

        ck = torch.load(a.openai, map_location="cpu")
        sd = ck.get("model_state_dict", ck)
        w = {k: v.float().numpy() for k, v in sd.items()}
    else:
        w = hf_to_openai_names(read_hf_checkpoint(a.hf))
    print(write_model(w, a.model_type, a.model_path, a.dtype, a.tiktoken))


if __name__ == "__main__":
    main()
