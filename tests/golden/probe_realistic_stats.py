"""What `modelgen.realistic_weights` does to the ACTIVATIONS, measured with an independent implementation (transformers'
Whisper, fp32, hooks) — run in the build container only; prints one table per model:

    python tests/golden/probe_realistic_stats.py [model_type ...]

Per stack: max |x| of the residual stream by layer, LayerNorm gain range, max |FFN hidden|, per-head attention peak
(max probability of a row, averaged over rows: 1/keys = flat, 1 = saturated), and for the decoder loop: logit std and
the top-2 margins over the decoded steps. tests/test_gpu_realistic.py runs the engine on exactly these weights."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
sys.path.insert(0, HERE)
import modelgen  # noqa: E402
from make_model_goldens import hf_from_weights  # noqa: E402
from make_model_goldens_inputs import demo_mel, synth_mel  # noqa: E402


@torch.no_grad()
def probe(model_type, seed, mel, n_new=24, kind="realistic", **kw):
    dims = modelgen.DIMS[model_type]
    w = modelgen.realistic_weights(dims, seed, **kw) if kind == "realistic" else modelgen.synth_weights(dims, seed)
    cfg = modelgen.make_config(model_type, dims)
    m = hf_from_weights(dims, w)
    ffn = {}

    def hook(name):
        def f(mod, inp, out):
            ffn[name] = max(ffn.get(name, 0.0), float(torch.nn.functional.gelu(out).abs().max()))
        return f

    for side, layers in (("enc", m.model.encoder.layers), ("dec", m.model.decoder.layers)):
        for i, l in enumerate(layers):
            l.fc1.register_forward_hook(hook(f"{side}{i}"))
    eo = m.model.encoder(input_features=torch.from_numpy(mel)[None], output_hidden_states=True, output_attentions=True)
    enc = eo.last_hidden_state
    print(f"== {model_type} seed {seed} ({kind})")
    print("  encoder residual max|x| by layer:", [round(float(h.abs().max()), 1) for h in eo.hidden_states])
    print("  encoder attention peak by (layer, head):",
          [[round(float(a[0, h].max(dim=-1).values.mean()), 3) for h in range(a.shape[1])] for a in eo.attentions])
    print("  encoder FFN hidden max:", [round(ffn[k], 1) for k in sorted(ffn) if k.startswith("enc")])
    print("  encoder output max|x|:", round(float(enc.abs().max()), 2))
    ck = torch.stack([l.encoder_attn.k_proj(enc)[0] for l in m.model.decoder.layers])
    cv = torch.stack([l.encoder_attn.v_proj(enc)[0] for l in m.model.decoder.layers])
    print("  cross K / V max|x|:", round(float(ck.abs().max()), 2), round(float(cv.abs().max()), 2))
    toks = [cfg["sot"], int(cfg["all_language_tokens"].split(",")[1]), cfg["transcribe"], cfg["no_timestamps"]]
    margins, stds, ids = [], [], []
    for step in range(n_new + 1):
        o = m.model.decoder(input_ids=torch.tensor([toks]), encoder_hidden_states=enc, output_hidden_states=True,
                            output_attentions=True)
        logits = m.proj_out(o.last_hidden_state[:, -1])[0].numpy()
        srt = np.sort(logits)
        margins.append(float(srt[-1] - srt[-2]))
        stds.append(float(logits.std()))
        nxt = int(logits.argmax())
        ids.append(nxt)
        toks.append(nxt)
    print("  decoder residual max|x| by layer:", [round(float(h.abs().max()), 1) for h in o.hidden_states])
    print("  decoder self-attention peak (last row) by (layer, head):",
          [[round(float(a[0, h, -1].max()), 3) for h in range(a.shape[1])] for a in o.attentions])
    print("  decoder cross-attention peak (last row) by (layer, head):",
          [[round(float(a[0, h, -1].max()), 3) for h in range(a.shape[1])] for a in o.cross_attentions])
    print("  decoder FFN hidden max:", [round(ffn[k], 1) for k in sorted(ffn) if k.startswith("dec")])
    print(f"  logits: std {np.mean(stds):.2f}, max {float(np.abs(logits).max()):.1f}; top-2 margin min {min(margins):.2e} median "
          f"{np.median(margins):.2e} max {max(margins):.2e}")
    print("  ids:", ids[:12], "eot" if cfg["eot"] in ids else "")
    gains = np.concatenate([v.ravel() for k, v in w.items() if k.endswith("_ln.weight") or k.endswith("ln.weight") or k.endswith("ln_post.weight")])
    print(f"  LayerNorm gains: min {gains.min():.3f} median {np.median(gains):.2f} max {gains.max():.1f}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    types = sys.argv[1:] or ["micro", "mini"]
    for mt in types:
        n_mels = modelgen.DIMS[mt]["n_mels"]
        probe(mt, 41, demo_mel(n_mels) if n_mels == 80 else synth_mel(9, 128, 2500), kind="benign")
        probe(mt, 41, demo_mel(n_mels) if n_mels == 80 else synth_mel(9, 128, 2500))
