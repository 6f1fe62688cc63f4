"""CPU: model-directory formats shared with the reference (config JSON schema, tokens file) and
the weights container that replaces the .axmodel blobs."""
import base64
import json
import os

import numpy as np

from conftest import GOLDEN


def test_config_schema_matches_exporter_keys(tmp_path):
    """Keys written by model_convert/export_onnx.py:592-625 and read by Whisper.cpp:99-100,130-137."""
    import modelgen

    d = modelgen.write_model_dir(str(tmp_path), "micro", seed=1)
    cfg = json.load(open(os.path.join(d, "micro_config.json")))
    for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab", "n_text_ctx",
              "n_text_state", "n_text_head", "n_text_layer", "sot", "eot", "transcribe", "translate", "no_timestamps",
              "all_language_tokens", "all_language_codes", "sot_sequence", "no_speech", "sot_prev", "sot_lm"):
        assert k in cfg, k
    assert isinstance(cfg["all_language_tokens"], str) and isinstance(cfg["all_language_codes"], str)
    # SURVEY A.3 special ids, 99-language models
    assert (cfg["eot"], cfg["sot"], cfg["transcribe"], cfg["no_timestamps"]) == (50257, 50258, 50359, 50363)
    codes, toks = cfg["all_language_codes"].split(","), cfg["all_language_tokens"].split(",")
    assert len(codes) == len(toks) == 99 and codes[1] == "zh" and toks[1] == "50260"
    turbo = modelgen.make_config("turbo", modelgen.DIMS["turbo"])
    assert (turbo["transcribe"], turbo["no_timestamps"], turbo["n_vocab"]) == (50360, 50364, 51866)
    assert turbo["all_language_codes"].split(",")[-1] == "yue"


def test_safetensors_round_trip_bf16(tmp_path):
    import modelgen

    w = modelgen.synth_weights(modelgen.DIMS["micro"], 3)
    p = str(tmp_path / "w.safetensors")
    modelgen.write_safetensors(p, w, "BF16")
    r = modelgen.read_safetensors(p)
    assert set(r) == set(w)
    for k in w:
        assert r[k].shape == w[k].shape and np.array_equal(r[k], w[k]), k  # synthetic weights are bf16-representable
    modelgen.write_safetensors(p, w, "F32")
    r = modelgen.read_safetensors(p)
    assert all(np.array_equal(r[k], w[k]) for k in w)


def test_tokens_file_format(tmp_path):
    """'<base64> <rank>' lines, rank == line index (export_onnx.py:404-417; Whisper.cpp:115-127)."""
    import modelgen

    p = str(tmp_path / "t.txt")
    modelgen.write_tokens(p, os.path.join(GOLDEN, "multilingual.tiktoken"))
    lines = open(p).read().splitlines()
    assert len(lines) == 50257
    for i in (0, 1, 255, 50255, 50256):
        tok, rank = lines[i].split(" ")
        assert int(rank) == i
        base64.b64decode(tok)
    assert max(len(base64.b64decode(l.split(" ")[0])) for l in lines) == 33  # overflows the reference's char[32]


def test_synth_clip_is_deterministic():
    import modelgen

    a, b = modelgen.synth_clip(3, 1000), modelgen.synth_clip(3, 1000)
    assert np.array_equal(a, b) and a.dtype == np.float32 and np.abs(a).max() <= 1.0
    assert not np.array_equal(a, modelgen.synth_clip(4, 1000))
