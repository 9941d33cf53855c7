"""CPU text front-end in front of the hot path (last item): what `ChatTTSPlusPipeline.infer()` does to the
input strings before they are tokenised -- it decides which utterances, and therefore which batch, the GPU path sees.

Mirrors, with the same names, argument meaning and results,
  * reference `chattts_plus/commons/text_utils.py`: `num_to_english` (:8-67), `get_lang` (:70-76), `num2text` (:87-114),
    `remove_brackets` (:117-124), `split_text` (:127-157), `split_text_by_punctuation` (:160-189);
  * reference `chattts_plus/commons/norm.py`: `Normalizer` (:36-209) -- language detection, per-language registered normalisers,
    half-width -> full-width punctuation for Chinese, the character simplifier + reject filter, homophone replacement.
Pinned by `tests/golden/text_frontend.json`, minted from the imported reference (`oracle/make_golden_text.py`).

The reference's quirks are part of its output (they change the tokens, hence the audio) and are kept:
  * the number reader says "One thousand,  and two hundred and thirty four" (scale word, comma, two blanks), reads 0 as "" and
    capitalises the first word only;
  * `num2text` substitutes each number by a global string replace in the order of appearance ("15 and 115" -> "Fifteen and  one Fifteen"),
    says "a over b" for fractions, " the pronunciation of  N" for percentages and spells 7 as "seven" without blanks in its final
    digit sweep;
  * `remove_brackets` passes its regex flags in the `count` position of `re.sub`: at most 26 control tags per string are protected and the
    match is case sensitive.
Two crashes of the reference are NOT kept (they would abort the request): its reader has no word for a group ending in 10
(`IndexError`, text_utils.py:55 -- "10 apples" raises) and no scale word above trillion (16-digit numbers).  Here: "ten", and
"quadrillion" / "quintillion".

Third-party readers the reference calls are absent from this image and are plug-ins here: `zh_normalization.TextNormalizer` for Chinese
numbers / dates (used when importable; else `zh_numbers.py`, the build's own minimal numeral reader) and `nemo_text_processing` for English (the reference
itself falls back to `num2text` when it is missing; so does this module unless `en_reader` is given).
"""
from __future__ import annotations

import json
import logging
import re
from typing import Callable, Dict, Iterable, List, Optional

_ONES = ("zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine")
_TEENS = {11: "eleven", 12: "twelve", 13: "thirteen", 14: "fourteen", 15: "fifteen", 16: "sixteen", 17: "seventeen", 18: "eighteen", 19: "nineteen"}
_TENS = ("", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety")
_SCALES = ("", "thousand", "million", "billion", "trillion", "quadrillion", "quintillion", "sextillion")


def _read_group(grp: str) -> str:
    """Up to three digits -> words, the way text_utils.py:27-55 reads them ("two hundred and thirty four"; 0 -> "")."""
    value = int(grp)
    hundreds = value // 100 if len(grp) == 3 else 0
    rest = value % 100 if len(grp) >= 2 else value
    words = ""
    if hundreds:
        words = _ONES[hundreds] + " hundred" + (" and " if rest else "")
    if rest in _TEENS:
        words += _TEENS[rest]
    elif rest >= 20:
        words += _TENS[rest // 10] + (" " + _ONES[rest % 10] if rest % 10 else "")
    elif rest == 10:
        words += "ten"                                   # the reference raises IndexError here (no entry for 10)
    elif rest:
        words += _ONES[rest]
    return words


def num_to_english(num) -> str:
    """Integer (or digit string; leading zeros allowed) -> English words, reference text_utils.py:8-67."""
    digits = str(num)
    groups = [digits[max(0, end - 3):end] for end in range(len(digits), 0, -3)][::-1]
    last = len(groups) - 1
    if last >= len(_SCALES):
        raise ValueError(f"num_to_english: {len(digits)} digits")
    out, first, seen_nonzero = "", True, False
    for i, grp in enumerate(groups):
        value = int(grp)
        if value == 0 and i < last:
            continue                                     # an all-zero group is silent (and does not end the "first group" state)
        words = _read_group(grp)
        if words and not first and seen_nonzero:
            out += " and "
        out += words
        if i < last and value:
            out += " " + _SCALES[last - i] + ", "
        first = False
        seen_nonzero = seen_nonzero or value != 0
    return out.capitalize()


_ZH_PUNCT = re.compile("[。？！，、；：‘’“”（）《》【】…—　]")
_CJK = re.compile("[一-鿿]")


def get_lang(text: str) -> str:
    """"zh" if a CJK ideograph remains once Chinese punctuation is removed, else "en" (text_utils.py:70-76)."""
    return "zh" if _CJK.search(_ZH_PUNCT.sub("", text)) else "en"


_SPACED = tuple(f" {w} " for w in _ONES)
# final sweep over whatever digits are left (text_utils.py:110-114): every digit padded with blanks except 7
_DIGIT_SWEEP = str.maketrans({str(d): (_ONES[d] if d == 7 else _SPACED[d]) for d in range(10)} | {"=": " equals "})
_NUM_TOKEN = re.compile(r"((\d+)(?:\.(\d+))?%?)")
_ARITH = ((re.compile(r"(\d)\,(\d)"), r"\1\2"), (re.compile(r"(\d+)\s*\+"), r"\1 plus "), (re.compile(r"(\d+)\s*\-"), r"\1 minus "),
          (re.compile(r"(\d+)\s*[\*x]"), r"\1 times "), (re.compile(r"((?:\d+\.)?\d+)\s*/\s*(\d+)"), r"\1 over \2"))


def num2text(text: str) -> str:
    """Spell the numbers of an English sentence (text_utils.py:87-114): thousands separators dropped, + - * x / read as words,
    then every number token replaced -- by a global string replace, in order of appearance -- with its reading."""
    for pat, repl in _ARITH:
        text = pat.sub(repl, text)
    for token, integer, fraction in _NUM_TOKEN.findall(text):
        if len(integer) > 16:
            continue
        words = num_to_english(integer)
        if fraction:
            words += " point " + "".join(_SPACED[int(c)] for c in fraction)
        if token.endswith("%"):
            words = " the pronunciation of  " + words
        text = text.replace(token, words)
    return text.translate(_DIGIT_SWEEP)


_TAG = re.compile(r"\[(uv_break|laugh|lbreak|break)\]")
_STRIP = re.compile(r"\[|\]|！|：|｛|｝")
_BARE_TAG = re.compile(r"\s(uv_break|laugh|lbreak|break)(?=\s|$)")
_TAG_LIMIT = int(re.I | re.S | re.M)                      # 26: the reference hands its flags to re.sub's `count` (text_utils.py:119)


def remove_brackets(text: str) -> str:
    """Drop square brackets and a few full-width marks but keep the control tags [uv_break] [laugh] [lbreak] [break]
    (text_utils.py:117-124): tags are unwrapped first, everything bracket-like is deleted, bare tag words are wrapped again."""
    text = _TAG.sub(r" \1 ", text, count=_TAG_LIMIT)
    return _BARE_TAG.sub(r" [\1] ", _STRIP.sub("", text))


_CUT_MARKS = frozenset("。？！，、；：”’》」』）】…—" + ".?!,:;)}…")
_DIGIT = re.compile(r"\d")
_MIN_PIECE = 150


def split_text_by_punctuation(text: str) -> List[str]:
    """Cut a long line after a punctuation mark whenever more than 150 characters have accumulated (text_utils.py:160-189); a '.'
    followed by a digit is a decimal point, not a cut."""
    pieces, start, n = [], 0, len(text)
    for i, ch in enumerate(text):
        if ch not in _CUT_MARKS:
            continue
        if ch == "." and i + 1 < n and _DIGIT.match(text[i + 1]):
            continue
        if i - start > _MIN_PIECE:
            pieces.append(text[start:i + 1])
            start = i + 1
    if start < n:
        pieces.append(text[start:])
    return pieces


def _default_zh_reader() -> Optional[Callable[[str], str]]:
    try:
        from zh_normalization import TextNormalizer     # PaddleSpeech-derived package the reference imports (text_utils.py:3); optional here
    except Exception:  # noqa: BLE001
        return None
    tn = TextNormalizer()
    return lambda s: "".join(tn.normalize(s))


def _default_en_reader() -> Callable[[str], str]:
    """nemo_text_processing when it is installed and constructs (text_utils.py:139-150), else -- as in the reference -- `num2text`."""
    try:
        from nemo_text_processing.text_normalization.normalize import Normalizer as NemoNormalizer
        nemo = NemoNormalizer(input_case="cased", lang="en")
    except Exception:  # noqa: BLE001
        return num2text
    state = {"failed": False}

    def read(line: str) -> str:
        if not state["failed"]:
            try:
                return nemo.normalize(line, verbose=False, punct_post_process=True)
            except Exception:  # noqa: BLE001
                state["failed"] = True                       # the reference latches its fallback after the first failure too
        return num2text(line)

    return read


_MAX_LINE = 200


def split_text(text_list: Iterable[str], zh_reader: Optional[Callable[[str], str]] = None, en_reader: Optional[Callable[[str], str]] = None) -> List[str]:
    """Per input line (text_utils.py:127-157): protect the control tags, spell numbers (Chinese lines through `zh_reader` -- by default
    `zh_normalization.TextNormalizer` when that package is installed, else the built-in `zh_numbers.read_numbers_zh`; English lines through `en_reader`, by default
    nemo_text_processing when installed, else `num2text`, the reference's own fallback), then cut lines longer than 200 characters."""
    if zh_reader is None:
        zh_reader = _default_zh_reader()
        if zh_reader is None:
            from .zh_numbers import read_numbers_zh     # the build's own minimal reader (not a restatement of zh_normalization)
            zh_reader = read_numbers_zh
    if en_reader is None:
        en_reader = _default_en_reader()
    out: List[str] = []
    for line in text_list:
        line = remove_brackets(line)
        line = zh_reader(line) if get_lang(line) == "zh" else en_reader(line)
        if len(line) > _MAX_LINE:
            out.extend(split_text_by_punctuation(line))
        else:
            out.append(line)
    return out


# ---- Normalizer (norm.py:36-209) ---------------------------------------------------------------------------------------------------

_SIMPLIFY = str.maketrans({"：": "，", "；": "，", "！": "。", "（": "，", "）": "，", "【": "，", "】": "，", "『": "，", "』": "，", "「": "，", "」": "，",
                           "《": "，", "》": "，", "－": "，", ":": ",", ";": ",", "!": ".", "(": ",", ")": ",", ">": ",", "<": ",", "-": ","})
# ASCII punctuation -> full-width forms for Chinese text; [ ] _ stay (control tags), quotes become opening quotes
_FULLWIDTH = str.maketrans(dict(zip("!\"'#$%&(),-*+./:;<=>?@\\^`{|}~", "！“‘＃＄％＆（），－＊＋。／：；＜＝＞？＠＼＾｀｛｜｝～")))
_REJECT = re.compile(r"[^一-鿿A-Za-z，。、,\. ]")
_KEEP_TAGS = re.compile(r"\[uv_break\]|\[laugh\]|\[lbreak\]")
_EN_WORD = re.compile(r"\b[A-Za-z]+\b")


class Normalizer:
    """`normalizer(text, do_text_normalization=True, do_homophone_replacement=True, lang=None) -> str`, the callable the pipeline
    applies to every utterance (pipeline:379-388).  `map_file_path`: homophones_map.json ({wrong character: right character}); None
    or a dict is accepted too.  Only characters of the Basic Multilingual Plane can be keys (the reference compares UTF-16 code units,
    norm.py:21-33)."""

    def __init__(self, map_file_path=None, logger=None):
        self.logger = logger or logging.getLogger(self.__class__.__name__)
        self.normalizers: Dict[str, Callable[[str], str]] = {}
        self.homophones_map = self._load_homophones_map(map_file_path)

    @staticmethod
    def _load_homophones_map(src) -> Dict[int, int]:
        if src is None:
            return {}
        if not isinstance(src, dict):
            with open(src, "r", encoding="utf-8") as f:
                src = json.load(f)
        return {ord(k): ord(v) for k, v in src.items() if len(k) == 1 and len(v) == 1 and ord(k) < 0x10000 and ord(v) < 0x10000}

    def __call__(self, text: str, do_text_normalization=True, do_homophone_replacement=True, lang: Optional[str] = None) -> str:
        if do_text_normalization:
            language = lang if lang is not None else self._detect_language(text)
            if language in self.normalizers:
                text = self.normalizers[language](text)
            if language == "zh":
                text = text.translate(_FULLWIDTH)
        invalid = self._count_invalid_characters(text)
        if invalid:
            self.logger.debug("found invalid characters: %s", invalid)
            text = text.translate(_SIMPLIFY)
        if do_homophone_replacement and self.homophones_map:
            replaced = text.translate(self.homophones_map)
            if replaced != text:
                self.logger.debug("replace homophones: %s", ", ".join(f"{a}->{b}" for a, b in zip(text, replaced) if a != b))
                text = replaced
        if invalid:
            text = _REJECT.sub("", text)
        return text

    def register(self, name: str, normalizer: Callable[[str], str]) -> bool:
        """Per-language normaliser ("zh" / "en"), run before the width / character maps.  False (nothing registered) if the name is taken
        or the callable does not map str -> str (norm.py:160-173)."""
        if name in self.normalizers:
            self.logger.warning("name %s has been registered", name)
            return False
        try:
            probe = normalizer("test string 测试字符串")
        except Exception as e:  # noqa: BLE001
            self.logger.warning(e)
            return False
        if not isinstance(probe, str):
            self.logger.warning("normalizer must have caller type (str) -> str")
            return False
        self.normalizers[name] = normalizer
        return True

    def unregister(self, name: str):
        self.normalizers.pop(name, None)

    def destroy(self):
        self.homophones_map = {}

    @staticmethod
    def _count_invalid_characters(s: str) -> set:
        return set(_REJECT.findall(_KEEP_TAGS.sub("", s)))

    @staticmethod
    def _detect_language(sentence: str) -> str:
        return "zh" if len(_CJK.findall(sentence)) > len(_EN_WORD.findall(sentence)) else "en"
