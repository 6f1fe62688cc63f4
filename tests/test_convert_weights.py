"""The HF -> model-directory converter, checked live against transformers (importable on both boxes):
CPU: converted weights through the oracle == HF forward; GPU: converted weights through libax_whisper.so ~ HF forward."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def _hf_model(seed=5):
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    torch.manual_seed(seed)
    cfg = WhisperConfig(vocab_size=51865, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2,
                        decoder_layers=2, decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448, activation_function="gelu", dropout=0.0,
                        attention_dropout=0.0, activation_dropout=0.0, attn_implementation="eager")
    m = WhisperForConditionalGeneration(cfg).eval().float()
    with torch.no_grad():  # HF initialises most biases to zero: give every tensor a non-trivial value
        for n, p in m.named_parameters():
            if p.ndim == 1 and "layer_norm" not in n:
                p.add_(0.02 * torch.randn_like(p))
    return m


@torch.no_grad()
def _hf_reference(m, mel, toks):
    enc = m.model.encoder(input_features=torch.from_numpy(mel)[None]).last_hidden_state
    logits = []
    for n in range(4, len(toks) + 1):
        out = m.model.decoder(input_ids=torch.tensor([toks[:n]]), encoder_hidden_states=enc).last_hidden_state
        logits.append(m.proj_out(out[:, -1])[0].numpy())
    ck = torch.stack([l.encoder_attn.k_proj(enc)[0] for l in m.model.decoder.layers]).numpy()
    return ck, np.stack(logits)


def _convert(m, tmp_path, dtype):
    import convert_weights

    w = convert_weights.hf_to_openai_names(m.state_dict())
    d = convert_weights.write_model(w, "hfmicro", str(tmp_path), dtype=dtype, tiktoken_path=os.path.join(GOLDEN, "multilingual.tiktoken"))
    return w, d


def test_converter_through_oracle_matches_transformers(oracle_mod, tmp_path):
    import modelgen
    from make_model_goldens_inputs import demo_mel

    m = _hf_model()
    w, d = _convert(m, tmp_path, "F32")
    back = modelgen.read_safetensors(os.path.join(d, "hfmicro.safetensors"))
    assert all(np.array_equal(back[k], w[k]) for k in back)
    cfg = modelgen.make_config("hfmicro", dict(n_mels=80, d=128, heads=2, enc_layers=2, dec_layers=2, n_vocab=51865, n_langs=99))
    o = oracle_mod.Oracle(cfg, back)
    mel = demo_mel(80)
    ck, cv = o.encoder(mel)
    ids, lg = o.greedy(ck, cv, "zh", max_new=6, want_logits=True)
    hf_ck, hf_lg = _hf_reference(m, mel, o.sot_seq("zh") + ids)
    assert np.abs(ck - hf_ck).max() < 2e-5
    assert np.abs(lg - hf_lg).max() < 2e-5
    assert [int(r.argmax()) for r in hf_lg[:-1]] == ids


@pytest.mark.gpu
def test_converter_through_gpu_matches_transformers(built_lib, tmp_path):
    from make_model_goldens_inputs import demo_mel

    m = _hf_model()
    _convert(m, tmp_path, "BF16")  # bf16 storage: the HF reference is run on the same rounded weights
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    e = built_lib.Whisper("hfmicro", str(tmp_path), "zh", device=0)
    mel = demo_mel(80)
    e.encode_mel(mel)
    ids = e.decode_greedy(1, max_new=6)[0]
    logits, _ = e.decode_forced(1, np.array([ids]))
    hf_ck, hf_lg = _hf_reference(m, mel, e.sot_seq + ids)
    k, _ = e.get_cross_kv(0)
    assert np.abs(k - hf_ck).max() < 6e-2
    err = np.abs(logits[0] - hf_lg).max()
    print("GPU vs live transformers logits err", err)
    assert err < 5e-2
    e.close()


def test_converter_refuses_to_invent_a_vocabulary(tmp_path):
    """A real checkpoint converted without its vocabulary would transcribe garbage silently."""
    import convert_weights

    m = _hf_model()
    w = convert_weights.hf_to_openai_names(m.state_dict())
    with pytest.raises(ValueError, match="tiktoken"):
        convert_weights.write_model(w, "hfmicro", str(tmp_path), tiktoken_path=None)
    with pytest.raises(FileNotFoundError):
        convert_weights.write_model(w, "hfmicro", str(tmp_path), tiktoken_path=str(tmp_path / "nope.tiktoken"))


def test_converter_reads_sharded_hf_checkpoints(tmp_path):
    import json

    import convert_weights
    import modelgen

    m = _hf_model()
    sd = {k: v.float().numpy() for k, v in m.state_dict().items()}
    names = sorted(sd)
    half = len(names) // 2
    shards = {"model-00001-of-00002.safetensors": names[:half], "model-00002-of-00002.safetensors": names[half:]}
    for fn, ks in shards.items():
        modelgen.write_safetensors(str(tmp_path / fn), {k: sd[k] for k in ks}, dtype="F32")
    json.dump({"weight_map": {k: fn for fn, ks in shards.items() for k in ks}}, open(tmp_path / "model.safetensors.index.json", "w"))
    got = convert_weights.read_hf_checkpoint(str(tmp_path))
    assert sorted(got) == names and all(np.array_equal(got[k], sd[k]) for k in names)
    a = convert_weights.hf_to_openai_names(got)
    b = convert_weights.hf_to_openai_names(m.state_dict())
    assert all(np.array_equal(a[k], b[k]) for k in b)
    with pytest.raises(FileNotFoundError):
        convert_weights.read_hf_checkpoint(str(tmp_path / "empty"))


def test_f16_safetensors_round_trip(tmp_path):
    import modelgen

    x = {"a": np.linspace(-3, 3, 77, dtype=np.float32).reshape(7, 11)}
    modelgen.write_safetensors(str(tmp_path / "t.safetensors"), x, dtype="F16")
    back = modelgen.read_safetensors(str(tmp_path / "t.safetensors"))
    assert np.array_equal(back["a"], x["a"].astype(np.float16).astype(np.float32))
