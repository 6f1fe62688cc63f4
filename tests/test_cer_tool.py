"""CPU: the CER text functions (methodology of python/test_wer.py:209-246 in the reference)."""
import os

import pytest

import cer


def test_edit_distance_known_values():
    assert cer.edit_distance("kitten", "sitting") == 3
    assert cer.edit_distance("", "abc") == 3 and cer.edit_distance("abc", "") == 3
    assert cer.edit_distance("abc", "abc") == 0
    assert cer.edit_distance("flaw", "lawn") == 2
    # the two transcripts the reference's READMEs print for demo.wav differ in two characters (README_EN.md:104,185)
    assert cer.edit_distance("甚至出现交易几乎停止的情况", "擅职出现交易几乎停止的情况") == 2


def test_strip_punctuation_keeps_words_and_spaces():
    assert cer.strip_punctuation("你好，世界！ a_b-c.") == "你好世界 abc"
    assert cer.strip_punctuation("no punctuation 123") == "no punctuation 123"


def test_total_is_ratio_of_sums():
    total, rows = cer.character_error_rate([("abcd", "abcf"), ("ab", "ab!")])
    assert abs(total - 100.0 * 1 / 6) < 1e-9
    assert [round(r[2], 3) for r in rows] == [25.0, 0.0]


@pytest.mark.gpu
def test_cer_main_runs_end_to_end_on_a_manifest(built_lib, oracle_mod, tmp_path, monkeypatch, capsys):
    """tools/cer.py main() EXECUTED (python/test_wer.py:249-303: per utterance `(n) file  gt: ...  predict: ...`, then
    `Total WER: x%`): a manifest of 6 WAV files, a seeded model whose greedy loop ends on eot after a different number of ids
    per clip (tests/eot_case.py), hypotheses from AX_WHISPER_RunFile. Expected: the ORACLE's transcripts of the same int16
    audio, detokenised here with python's base64, scored by a separate edit-distance routine."""
    import base64
    import wave

    import numpy as np
    import torch  # noqa: F401

    from conftest import GOLDEN
    from eot_case import EotCase

    case = EotCase("micro", 11)
    import modelgen

    root = str(tmp_path / "m")   # with the real vocabulary file (EotCase.write leaves the placeholder table)
    modelgen.write_model_dir(root, "micro", case.dims, weights=case.weights, dtype=case.dtype, tiktoken_path=os.path.join(GOLDEN, "multilingual.tiktoken"))
    table = [base64.b64decode(ln.split(b" ")[0]) for ln in open(os.path.join(GOLDEN, "multilingual.tiktoken"), "rb").read().split(b"\n") if ln]
    table[188] = b""   # the reference's strcpy semantics (tests/test_byte_paths.py)
    monkeypatch.setenv("AX_WHISPER_OPENCC_DIR", str(tmp_path))   # no t2s.json there: the zh post-pass stays out of this test
    _, clips, _ = case.select(6)
    lines, want_pairs = [], []
    for i, c in enumerate(clips):
        q = np.clip(np.round(c * 32767.0), -32768, 32767).astype(np.int16)
        path = str(tmp_path / f"utt{i}.wav")
        with wave.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(q.tobytes())
        ids = case.oracle.transcribe(q.astype(np.float32) / np.float32(32768.0), "zh", max_new=444)   # until eot, as RunFile does
        hyp = b"".join(table[t] for t in ids if t < len(table)).decode("utf-8", errors="replace")
        # a reference transcript that differs from the hypothesis: four characters the model never said, then the hypothesis
        # with every 5th character dropped (random ids decode to few word characters: the prefix keeps the denominator honest)
        ref = "甲乙丙丁" + "".join(ch for k, ch in enumerate(hyp) if k % 5 != 4)
        ref = ref.replace("\t", " ").replace("\n", " ").replace("\r", " ")
        lines.append(f"{path}\t{ref}")
        want_pairs.append((ref, hyp))
    manifest = tmp_path / "pairs.tsv"
    manifest.write_text("\n".join(lines) + "\n", encoding="utf-8")

    def lev(a, b):   # full-matrix Levenshtein, written separately from tools/cer.py's rolling rows
        d = np.zeros((len(a) + 1, len(b) + 1), dtype=np.int64)
        d[:, 0] = np.arange(len(a) + 1)
        d[0, :] = np.arange(len(b) + 1)
        for i in range(1, len(a) + 1):
            for j in range(1, len(b) + 1):
                d[i, j] = min(d[i - 1, j] + 1, d[i, j - 1] + 1, d[i - 1, j - 1] + (a[i - 1] != b[j - 1]))
        return int(d[-1, -1])

    import re

    strip = lambda t: re.sub(r"[^\w\s]|_", "", t)
    err = sum(lev(strip(r), strip(h)) for r, h in want_pairs)
    n = sum(len(strip(r)) for r, _ in want_pairs)
    monkeypatch.setattr("sys.argv", ["cer.py", "--manifest", str(manifest), "-t", "micro", "-p", root, "--language", "zh"])
    cer.main()
    out = capsys.readouterr().out
    for i, (ref, hyp) in enumerate(want_pairs):   # (a hypothesis may hold line breaks: whole-record match, not line by line)
        assert f"({i + 1}) utt{i}.wav  gt: {ref}  predict: {hyp}\n" in out, (i, ref, hyp)
    assert any(0 < len(h) for _, h in want_pairs) and len({len(h) for _, h in want_pairs}) >= 3
    total = float(re.search(r"Total WER: ([0-9.eE+-]+)%", out).group(1))
    assert n > 20 and abs(total - 100.0 * err / n) < 1e-9, (total, err, n)
    print(f"cer.py main(): 6 utterances, total {total:.2f} % = the oracle's transcripts scored independently")
