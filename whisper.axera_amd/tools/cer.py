"""Character-error-rate harness (the reference's accuracy methodology, python/test_wer.py:209-299).

    python cer.py --manifest pairs.tsv -t small -p ./models --language zh [--max_num N]

manifest: one ``<wav path>\\t<reference text>`` per line (AIShell transcripts / CommonVoice TSVs reduce to this).
Per utterance: strip punctuation (everything that is not a word character or whitespace, and ``_``), Levenshtein
distance between reference and hypothesis characters, error rate = distance / len(reference); the total is the ratio
of the sums, as in the reference. Needs real weights and a dataset, neither of which exists in this environment; the
text functions are unit-tested.
"""
from __future__ import annotations

import argparse
import os
import re
import sys

_PUNCT = re.compile(r"[^\w\s]|_")


def strip_punctuation(text: str) -> str:
    return _PUNCT.sub("", text)


def edit_distance(a: str, b: str) -> int:
    """Levenshtein distance (unit costs), two rolling rows."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = prev[j - 1] if ca == cb else 1 + min(prev[j - 1], prev[j], cur[j - 1])
        prev = cur
    return prev[-1]


def character_error_rate(pairs):
    """pairs: iterable of (reference, hypothesis) -> (total CER in percent, per-utterance list)."""
    err = n = 0
    rows = []
    for ref, hyp in pairs:
        ref, hyp = strip_punctuation(ref), strip_punctuation(hyp)
        e = edit_distance(ref, hyp)
        err += e
        n += len(ref)
        rows.append((ref, hyp, 100.0 * e / max(len(ref), 1)))
    return 100.0 * err / max(n, 1), rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest", required=True)
    ap.add_argument("--model_type", "-t", default="small")
    ap.add_argument("--model_path", "-p", required=True)
    ap.add_argument("--language", "-l", default="zh")
    ap.add_argument("--max_num", type=int, default=0)
    a = ap.parse_args()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import whisper_axera_amd as wa

    model = wa.Whisper(a.model_type, a.model_path, a.language)
    pairs = []
    with open(a.manifest, encoding="utf-8") as f:
        for n, line in enumerate(f):
            if a.max_num and n >= a.max_num:
                break
            path, ref = line.rstrip("\n").split("\t", 1)
            hyp = model.run(path)
            pairs.append((ref, hyp))
            print(f"({n + 1}) {os.path.basename(path)}  gt: {ref}  predict: {hyp}")
    total, _ = character_error_rate(pairs)
    print(f"Total WER: {total}%")


if __name__ == "__main__":
    main()
