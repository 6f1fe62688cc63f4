"""Does cross K/V that is resident in the 256 MB memory-side cache (MALL) make decode_attention_kernel faster?

The attention-only replay of a decoder step (Engine::bench "decode_attn") on Whisper-small dims with ONE, TWO and THREE decoder
layers at 32 clips: one layer's cross K/V of 32 clips is 147 MB (fits the MALL and is re-read by every replay), two layers'
295 MB (does not fit: every replay streams from HBM).  Per-layer time of the replay tells whether a prefetch of the next
layer's K/V into the MALL during the GEMM phases could pay.

    python profiles/scripts/mall_probe.py [clips]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper.axera_amd", "tools"))
import modelgen  # noqa: E402
import whisper_axera_amd as wa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for layers in (1, 2, 3, 12):
    dims = dict(modelgen.DIMS["small"], dec_layers=layers, enc_layers=1)
    root = f"/tmp/axw_mall_probe_{layers}"
    if not os.path.exists(os.path.join(root, "small", "small.safetensors")):
        modelgen.write_model_dir(root, "small", dims, seed=0, tiktoken_path=os.path.join(ROOT, "tests", "golden", "multilingual.tiktoken"))
    os.environ["AX_WHISPER_DECODE_BRANCHES"] = "1"
    e = wa.Whisper("small", root, "zh", device=0, max_batch=B)
    e.bench("encoder", B, 0, 1)
    it = 40
    ms = e.bench("decode_attn", B, 224, it) / it
    kv = B * layers * 2 * 1500 * 768 * 2 / 1e6
    print(f"{layers:2d} decoder layers, {B} clips: attention launches of a step {ms * 1e3:8.1f} us = {ms * 1e3 / layers:6.1f} us per layer; "
          f"cross K/V {kv:7.1f} MB per replay -> {kv / ms / 1e3:5.2f} TB/s", flush=True)
    e.close()
