"""CPU: the CER text functions (methodology of python/test_wer.py:209-246 in the reference)."""
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
