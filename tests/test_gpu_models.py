"""GPU: other model shapes (mini: d=256/4 heads/3 decoder layers; tiny dims: d=384/6 heads/4 layers) against the
oracle and the transformers goldens; the CLI's stdout contract."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ModelCase, load_demo_pcm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_type,golden,seed", [("mini", "mini_synth", 13), ("tiny", "tiny_demo", 14)])
def test_model_vs_oracle_and_transformers(built_lib, oracle_mod, tmp_path, model_type, golden, seed):
    from make_model_goldens_inputs import golden_mel

    case = ModelCase(tmp_path, model_type, seed)
    g = np.load(os.path.join(GOLDEN, f"model_{golden}.npz"))
    mel = golden_mel(golden)
    e = built_lib.Whisper(model_type, case.root, "zh", device=0)
    e.encode_mel(mel)
    k, v = e.get_cross_kv(0)
    kb, vb = case.oracle_bf16.encoder(mel)
    print(model_type, "cross kv err vs bf16 policy", np.abs(k - kb).max(), np.abs(v - vb).max())
    assert np.abs(k - kb).max() < 4e-2 and np.abs(v - vb).max() < 4e-2
    assert np.abs(k[:, ::53, ::7] - g["cross_k_sub"]).max() < 8e-2  # HF fp32 golden
    n_new = int(g["n_new"])
    hf_ids = [int(x) for x in g["ids"][:n_new]]
    logits, am = e.decode_forced(1, np.array([hf_ids]))
    top = np.take_along_axis(logits[0], g["top_ids"][: n_new + 1], axis=1)
    err = np.abs(top - g["top_vals"][: n_new + 1]).max()
    print(model_type, "logits err vs transformers golden", err)
    assert err < 6e-2
    ids_o, lg_o = case.oracle_bf16.greedy(kb, vb, "zh", max_new=n_new, forced=hf_ids, want_logits=True)
    assert np.abs(logits[0] - lg_o).max() < 2.5e-2
    e.close()


def test_cli_stdout_contract(built_lib, micro_case):
    """whisper_cli.cpp:63-66,90,102-103: the printed lines and the RTF definition."""
    cli = os.path.join(os.path.dirname(built_lib.LIB_PATH), "whisper_cli")
    wav = os.path.join(GOLDEN, "demo.wav")
    r = subprocess.run([cli, "-w", wav, "-t", "micro", "-p", micro_case.root, "--language", "zh"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0] == f"wav_file: {wav}" and out[1] == f"model_path: {micro_case.root}"
    assert out[2] == "model_type: micro" and out[3] == "language: zh"
    assert out[4].startswith("Init whisper success, take ") and out[4].endswith("seconds")
    assert out[5].startswith("Result: ") and out[-1].startswith("RTF: ")
    assert float(out[-1].split()[1]) > 0
    r = subprocess.run([cli, "-t", "micro"], capture_output=True, text=True)
    assert r.returncode != 0 and "need option: --wav" in r.stderr
