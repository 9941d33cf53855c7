"""Mint tests/golden/text_frontend.json by running the *imported, unmodified* reference text front-end.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference):

    python -m oracle.make_golden_text

What is imported: chattts_plus/commons/text_utils.py (num_to_english, get_lang, num2text, remove_brackets, split_text,
split_text_by_punctuation) and chattts_plus/commons/norm.py (Normalizer).  Their import-time third-party dependencies are absent
from this image and are replaced by stand-ins that carry no arithmetic of their own:
  * ``numba.jit``  -> identity decorator (the jitted loops of norm.py run as plain Python);
  * ``zh_normalization.TextNormalizer`` -> ``normalize(text) -> [text]`` (the PaddleSpeech number / date reader is NOT available, so
    the Chinese branch of ``split_text`` is pinned for its control flow only -- "parity unpinned" for the Chinese number reading;
    the product takes that reader as a pluggable callable for the same reason);
  * ``nemo_text_processing`` is absent, which is the reference's own fallback case: English goes through ``num2text``.
The fixture holds inputs and the reference's outputs, nothing else.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import REFERENCE_ROOT, reference_available  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "text_frontend.json")

HOMOPHONES = {"粘": "年", "吗": "嘛", "呗": "贝", "嗯": "恩", "哦": "喔"}          # made-up map in the format of homophones_map.json

NUMBERS = [0, 1, 7, 10, 11, 12, 19, 20, 21, 30, 45, 99, 100, 101, 110, 111, 115, 120, 199, 200, 999, 1000, 1001, 1010, 1100, 1234, 2000,
           9999, 10000, 10001, 12345, 100000, 100100, 999999, 1000000, 1000001, 1002003, 20000000, 123456789, 1000000000, 1000000001,
           987654321012, 1000000000000, 1234567890123456, "007", "0", "00", "000123", "1000"]

LANG = ["hello world", "你好", "，。！", "hello，世界", "", "123", "abc。", "「引用」", "…—", "日本語のテキスト", "mixed 中 text"]

NUM2TEXT = [
    "I have 2 apples", "1,234 items", "3+4", "10 - 3", "6 x 7", "6*7", "1/2 of it", "3.5/2", "2.50 dollars", "50% off", "12.5% more",
    "call 911 now", "year 2024", "a7b", "7", "777", "x=5", "1+1=2", "12345678901234567 is long", "1234567890123456 fits", "0.001", "00.10",
    "no digits here", "5 -3", "100,000,000", "3 . 5", "version 1.2.3", "10%", "1 000", "3x", "8 * 9 = 72", "it's 7:30", "(42)", "2/3 + 1/4",
    "99 bottles, 98 bottles", "1.5 and 1.5 again", "15 and 115", "a1b2c3", "", "1/0",
]

BRACKETS = [
    "hello [uv_break] world", "a[laugh]b", "[lbreak]", "x [break] y", "[other] tag", "no tags", "wow！ok：{}｛a｝", "[UV_BREAK] upper",
    "end [laugh]", "[laugh] start", "a [uv_break][laugh] b", "nested [[uv_break]]", "laugh without brackets", "say laugh now",
    " ".join(["[uv_break]"] * 30), "x uv_break", "tab\t[laugh]\tend", "[laugh]\nnewline",
]

LONG_EN = ("This is a fairly long sentence, with several commas, semicolons; and other marks: it keeps going on and on. " * 4).strip()
LONG_ZH = "今天天气很好，我们一起去公园散步吧。路上看到了很多花，还有小鸟在唱歌；大家都很开心！" * 6
LONG_NUM = ("The value is 3.14159, and then 2.71828, which matters. " * 5).strip()
NO_PUNCT = "word " * 80
PUNCT_SPLIT = [LONG_EN, LONG_ZH, LONG_NUM, NO_PUNCT, "short.", "", "a" * 151 + "." + "b" * 10, "a" * 150 + "." + "b" * 10, "x" * 151 + "。" + "y" * 151 + "！" + "z",
               "1.5" * 60 + ". end", "q" * 160 + ".5 rest"]

SPLIT_TEXT = [
    ["Hello world.", "I have 2 cats"],
    ["你好世界", "今天是个好日子。"],
    ["mixed 中文 and English 42"],
    [LONG_EN],
    [LONG_ZH],
    ["[laugh] that is 50% funny！", "价格：100元"],
    [],
    ["a [uv_break] b", LONG_NUM],
]

NORM_CASES = [
    # text, do_text_normalization, do_homophone_replacement, lang
    ("hello world", True, True, None),
    ("你好，世界", True, True, None),
    ("粘贴一下吗", True, True, None),
    ("粘贴一下吗", True, False, None),
    ("粘贴一下吗", False, True, None),
    ("Hello, (world)! How are you?", True, True, None),
    ("Hello, (world)! How are you?", False, False, None),
    ("你好(世界)!真的吗?", True, True, None),
    ("你好(世界)!真的吗?", True, True, "en"),
    ("hello world", True, True, "zh"),
    ("数字123和symbols#@", True, True, None),
    ("keep [uv_break] and [laugh] tags [lbreak]", True, True, None),
    ("bad [break] tag", True, True, None),
    ("a-b;c:d<e>f", True, True, None),
    ("中文；标点：测试！（括号）【方】『书』「引」《名》－完", True, True, None),
    ("", True, True, None),
    ("emoji 😀 here", True, True, None),
    ("半角,句号.问号?", True, True, None),
    ("English words beat 中 文", True, True, None),
    ("中 文 字 beat en", True, True, None),
    ("tie 中 a", True, True, None),
    ("哦，嗯。呗", True, True, None),
    ("it's \"quoted\" ~ok~", True, True, "zh"),
    ("under_score and [brackets]", True, True, "zh"),
]


def _install_stand_ins():
    os.environ.setdefault("CHATTTS_PLUS_LOG_DIR", "/tmp/ctts_ref_logs")
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.jit = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)
        nb._ctts_stand_in = True
        sys.modules["numba"] = nb
    if "zh_normalization" not in sys.modules:
        zh = types.ModuleType("zh_normalization")

        class TextNormalizer:                                  # absent third-party reader: passes the text through
            def normalize(self, text):
                return [text]

        zh.TextNormalizer = TextNormalizer
        zh._ctts_stand_in = True
        sys.modules["zh_normalization"] = zh
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for pkg, sub in (("chattts_plus", ""), ("chattts_plus.commons", "commons")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REFERENCE_ROOT, "chattts_plus", sub)]
            sys.modules[pkg] = m


def remove_stand_ins():
    """Take the stand-in modules out of sys.modules again (tests: later imports must see the packages as absent)."""
    for name in ("numba", "zh_normalization"):
        m = sys.modules.get(name)
        if m is not None and getattr(m, "_ctts_stand_in", False):
            del sys.modules[name]


def load_text_reference():
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stand_ins()
    tu = importlib.import_module("chattts_plus.commons.text_utils")
    nm = importlib.import_module("chattts_plus.commons.norm")
    return tu, nm


def _run(f, *a):
    """The reference's result, or {"raises": type name} (e.g. its number reader has no word for 10: IndexError, text_utils.py:55)."""
    try:
        return f(*a)
    except Exception as e:  # noqa: BLE001
        return {"raises": type(e).__name__}


def main():
    import contextlib
    import io
    tu, nm = load_text_reference()
    out = {"homophones": HOMOPHONES}
    out["num_to_english"] = [[n, _run(tu.num_to_english, n)] for n in NUMBERS]
    out["get_lang"] = [[t, tu.get_lang(t)] for t in LANG]
    out["num2text"] = [[t, _run(tu.num2text, t)] for t in NUM2TEXT]
    out["remove_brackets"] = [[t, tu.remove_brackets(t)] for t in BRACKETS]
    out["split_text_by_punctuation"] = [[t, tu.split_text_by_punctuation(t)] for t in PUNCT_SPLIT]
    with contextlib.redirect_stdout(io.StringIO()):            # split_text prints its nemo fallback message
        out["split_text"] = [[t, _run(tu.split_text, list(t))] for t in SPLIT_TEXT]
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "homophones_map.json")
        with open(mp, "w", encoding="utf-8") as f:
            json.dump(HOMOPHONES, f, ensure_ascii=False)
        n = nm.Normalizer(mp)
        out["normalizer"] = [[list(c), n(*c)] for c in NORM_CASES]
        # registered per-language normalizers (norm.py:160-177): the callable runs before the width / character maps
        ok_en = n.register("en", lambda s: s.upper())
        ok_zh = n.register("zh", lambda s: s.replace("世界", "地球"))
        dup = n.register("en", lambda s: s)
        bad = n.register("fr", lambda s: 3)
        out["register"] = [ok_en, ok_zh, dup, bad]
        out["normalizer_registered"] = [[list(c), n(*c)] for c in NORM_CASES[:12]]
        n.unregister("en")
        out["normalizer_unregistered_en"] = [[list(c), n(*c)] for c in NORM_CASES[:3]]
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", OUT, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
