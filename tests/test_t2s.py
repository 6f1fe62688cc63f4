"""CPU: the Traditional -> Simplified post-pass of zh transcripts (cpp/src/Whisper.cpp:231-236, OpenCC t2s.json) read
from the reference's own dictionaries (tests/golden/opencc: TSPhrases.ocd2, TSCharacters.ocd2; t2s.json is written here) by
whisper.axera_amd/csrc/t2s.hpp through the C ABI (host-only entry point, no GPU needed).

Expected strings are standard OpenCC t2s behaviour: single characters through TSCharacters, and the phrase exceptions
of TSPhrases (keys listed in the dictionary itself) taking precedence over the character table."""
import os

import pytest

from conftest import GOLDEN

OCC = os.path.join(GOLDEN, "opencc")


def opencc_t2s_dir(dst):
    """A directory laid out like the reference's deployment: t2s.json (the standard OpenCC Traditional -> Simplified
    configuration: mmseg over TSPhrases, then the group [TSPhrases, TSCharacters]) next to the two dictionaries."""
    import json
    import shutil

    os.makedirs(dst, exist_ok=True)
    for f in ("TSPhrases.ocd2", "TSCharacters.ocd2"):
        shutil.copy(os.path.join(OCC, f), os.path.join(dst, f))
    ocd2 = lambda f: {"type": "ocd2", "file": f}
    cfg = {"name": "Traditional Chinese to Simplified Chinese",
           "segmentation": {"type": "mmseg", "dict": ocd2("TSPhrases.ocd2")},
           "conversion_chain": [{"dict": {"type": "group", "dicts": [ocd2("TSPhrases.ocd2"), ocd2("TSCharacters.ocd2")]}}]}
    with open(os.path.join(dst, "t2s.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    return os.path.join(dst, "t2s.json")


@pytest.fixture(scope="module")
def CFG(tmp_path_factory):
    return opencc_t2s_dir(str(tmp_path_factory.mktemp("opencc")))


def test_characters_and_phrases(built_lib, CFG):
    t2s = lambda s: built_lib.convert_t2s(CFG, s)
    # the reference README's demo transcript is already Simplified: unchanged
    assert t2s("甚至出现交易几乎停止的情况") == "甚至出现交易几乎停止的情况"
    # character table
    assert t2s("甚至出現交易幾乎停止的情況") == "甚至出现交易几乎停止的情况"
    assert t2s("體國說這個時們來為學發後") == "体国说这个时们来为学发后"
    assert t2s("長阪骯肅") == "长阪肮肃"  # 阪 has two values (阪, 坂): the first one is used
    # phrase exceptions win over the character table: 乾 alone becomes 干, but 乾坤 / 乾隆-style phrases keep 乾
    assert t2s("乾") == "干"
    assert t2s("乾坤大挪移") == "乾坤大挪移"
    assert t2s("袖裏乾坤") == "袖里乾坤"
    assert t2s("計畫") == "计划" and t2s("憑藉") == "凭借" and t2s("老態龍鍾") == "老态龙钟"
    # longest phrase match inside running text, characters around it converted one by one
    assert t2s("我們的計畫是憑藉經驗") == "我们的计划是凭借经验"
    # non-Chinese text, punctuation, 4-byte characters and the empty string pass through
    assert t2s("Hello, 世界! 123") == "Hello, 世界! 123"
    assert t2s("") == ""
    assert t2s("𢶫") == "𢫞"


def test_dictionary_is_read_completely(built_lib, CFG):
    """Every key of TSCharacters is one character and converts to something non-empty; idempotence on the result for a
    sample (Simplified text has no Traditional keys left, except characters that are both)."""
    t2s = lambda s: built_lib.convert_t2s(CFG, s)
    sample = "萬與醜專業叢東絲丟兩嚴喪個豐臨為麗舉麼義烏樂喬習鄉書買亂爭於虧雲亙亞產畝親褻嚲億僅從侖倉儀們價眾優會傴傘偉傳傷倀倫傖偽佇體餘傭僉俠侶僥偵側僑儈儕儂"
    out = t2s(sample)
    assert len(out) == len(sample) and out != sample
    assert out.startswith("万与丑专业丛东丝丢两严丧个丰临为丽举么义乌乐乔习乡书买乱争于亏云亘亚产亩亲亵")
    assert t2s(out) == out


def test_every_key_of_both_dictionaries_against_an_independent_reader(built_lib, CFG):
    """EXHAUSTIVE: all 4113 keys of TSCharacters.ocd2 and all 278 of TSPhrases.ocd2, enumerated and paired with their value
    lists by tests/ocd2_reader.py (a separate reader: forward walk of the trie, sequential read of the value table), go
    through AX_WHISPER_ConvertT2S and must come out as their FIRST value (OpenCC's rule; `阪` has two, `彷` has two). Keys are
    sent '|'-separated in one text, so no phrase can form across two of them; a phrase key is its own longest match."""
    import ocd2_reader

    t2s = lambda s: built_lib.convert_t2s(CFG, s)
    chars = ocd2_reader.read_ocd2(os.path.join(OCC, "TSCharacters.ocd2"))
    phrases = ocd2_reader.read_ocd2(os.path.join(OCC, "TSPhrases.ocd2"))
    assert len(chars) == 4113 and len(phrases) == 278
    assert all(len(k.decode("utf-8")) == 1 for k, _ in chars)          # one code point per key
    assert sum(1 for _, v in chars if len(v) > 1) == 56 and sum(1 for _, v in phrases if len(v) > 1) == 4
    for table in (chars, phrases):
        keys = [k.decode("utf-8") for k, _ in table]
        want = [v[0].decode("utf-8") for _, v in table]
        got = t2s("|".join(keys)).split("|")
        assert len(got) == len(keys)
        bad = [(k, w, g) for k, w, g in zip(keys, want, got) if w != g]
        assert not bad, bad[:10]
    # a phrase key wins over the characters it is made of wherever the two disagree, also inside running text
    tab = {k.decode("utf-8"): v[0].decode("utf-8") for k, v in chars}
    differ = [(k.decode("utf-8"), v[0].decode("utf-8")) for k, v in phrases if "".join(tab.get(c, c) for c in k.decode("utf-8")) != v[0].decode("utf-8")]
    assert len(differ) >= 50
    for k, v in differ:
        assert t2s("我" + k + "。") == "我" + v + "。", k


def test_bad_files_fail_cleanly(built_lib, tmp_path, CFG):
    with pytest.raises(RuntimeError):
        built_lib.convert_t2s(str(tmp_path / "missing.json"), "x")
    bad = tmp_path / "t2s.json"
    bad.write_text(open(CFG).read())
    (tmp_path / "TSPhrases.ocd2").write_bytes(b"OPENCC_MARISA_0.2.5We love Marisa.\x00" + b"\x10" * 40)
    (tmp_path / "TSCharacters.ocd2").write_bytes(open(os.path.join(OCC, "TSCharacters.ocd2"), "rb").read()[:1000])
    with pytest.raises(RuntimeError):
        built_lib.convert_t2s(str(bad), "體")
