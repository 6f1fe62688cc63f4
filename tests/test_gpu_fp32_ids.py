"""GPU: greedy token ids against the PURE-fp32 oracle (no bf16 policy) — the north-star bar "token-ids bit-exact for
greedy decode; logits within stated fp tolerance" — including BASELINE configs[0]: Whisper-tiny dims on demo.wav.

Tolerance for logits vs the pure-fp32 oracle: 5e-3 abs (bf16 operand rounding in the encoder and the bf16 K/V caches;
logit std is 0.2-0.55). A token may differ only where the fp32 oracle's own top-2 margin is below the measured logit
error (none does on these inputs: 100 % of ids agree, including 440-token full-context runs)."""
import numpy as np
import pytest
import torch  # noqa: F401

from conftest import ModelCase, load_demo_pcm

pytestmark = pytest.mark.gpu


def _compare(e, case, oracle_mod, clips, n_new, n_mels=80):
    got = e.run_tokens_batch(clips, max_new=n_new) if len(clips) > 1 else [e.run_tokens(clips[0], max_new=n_new)]
    total = agree = 0
    for b, pcm in enumerate(clips):
        mel, _, _ = oracle_mod.log_mel(pcm, n_mels)
        ck, cv = case.oracle_fp32.encoder(mel)
        ids, lg = case.oracle_fp32.greedy(ck, cv, "zh", max_new=n_new, want_logits=True)
        e.encode_mel(e.compute_mel(pcm))
        logits, _ = e.decode_forced(1, np.array([ids]))
        err = np.abs(logits[0] - lg).max(axis=1)
        assert err.max() < 5e-3, err.max()
        total += len(ids)
        if got[b] == ids:
            agree += len(ids)
        else:
            i = next(i for i in range(min(len(ids), len(got[b]))) if ids[i] != got[b][i])
            srt = np.sort(lg[i])
            assert srt[-1] - srt[-2] < 2 * err[i] + 1e-4, (b, i, srt[-1] - srt[-2], err[i])
            agree += i
    return agree, total


def test_config0_tiny_demo_wav_ids_vs_fp32_oracle(built_lib, oracle_mod, tmp_path):
    case = ModelCase(tmp_path, "tiny", 14)
    e = built_lib.Whisper("tiny", case.root, "zh", device=0)
    agree, total = _compare(e, case, oracle_mod, [load_demo_pcm()], 120)
    print(f"tiny/demo.wav: {agree}/{total} ids equal to the fp32 oracle")
    assert agree == total
    e.close()


def test_full_context_ids_vs_fp32_oracle(built_lib, oracle_mod, micro_case):
    import modelgen

    e = built_lib.Whisper("micro", micro_case.root, "zh", device=0, max_batch=3)
    clips = [load_demo_pcm(), modelgen.synth_clip(1, 480000), modelgen.synth_clip(2, 150000)]
    agree, total = _compare(e, micro_case, oracle_mod, clips, 444)
    print(f"micro, 444-token runs: {agree}/{total} ids equal to the fp32 oracle")
    assert total == 3 * 444 and agree >= total - 3
    e.close()
