"""Built-in reader for Arabic numerals inside Chinese text -- the fallback `zh_reader` of `text_frontend.split_text`.

The reference hands Chinese lines to the third-party package `zh_normalization` (`TextNormalizer().normalize`, reference
`commons/text_utils.py:3,128-134`; PyPI `zh-normalization`, unpinned in requirements.txt, derived from PaddleSpeech's t2s front end).  That
package is absent from this image and cannot be pinned here; when it is installed `text_frontend` uses it, exactly like the reference.
Without it the reference cannot even import its text module, so there is no reference behaviour to reproduce: this file is the build's
OWN minimal reader (PARITY UNPINNED against zh_normalization, and not claiming it) so that "价格100元" reaches the tokenizer as
"价格一百元" instead of losing its digits to the Normalizer's reject filter (norm.py:148-157 drops every character outside CJK / Latin
letters / ， 。 、 , . and blank).

Covered, in this order (each pattern consumes what it matches): dates (2024年3月5日, 2024-03-05, 2024/3/5), clock times (12:30, 8:05:09),
temperatures (-3℃, 25°C, 37.5度), percentages, fractions, ranges (3-5, 10~20), mobile / long digit strings (digit by digit, 1 read as 幺),
signed decimals and integers (cardinals up to 10^16, beyond that digit by digit).
"""
from __future__ import annotations

import re

_D = "零一二三四五六七八九"


def read_digits(digits: str, yao: bool = False) -> str:
    """Digit by digit ("2024" -> 二零二四); `yao`: 1 is read 幺 (phone numbers)."""
    out = "".join(_D[int(c)] for c in digits)
    return out.replace("一", "幺") if yao else out


def _read_section(n: int) -> str:
    """0 < n < 10000 -> 千百十 reading with interior zeros ("一千零一十"); the caller handles a leading 一十 -> 十."""
    out, zero_pending = "", False
    for unit, name in ((1000, "千"), (100, "百"), (10, "十"), (1, "")):
        d, n = divmod(n, unit)
        if d == 0:
            zero_pending = bool(out)
            continue
        if zero_pending:
            out += "零"
            zero_pending = False
        out += _D[d] + name
    return out


def read_cardinal(digits: str) -> str:
    """Non-negative integer (digit string) -> Chinese cardinal: 10 十, 11 十一, 110 一百一十, 1001 一千零一, 100000 十万, 10^8 一亿."""
    digits = digits.lstrip("0")
    if not digits:
        return "零"
    if len(digits) > 16:
        return read_digits(digits)
    n = int(digits)
    out, sections = "", []
    while n:
        n, sec = divmod(n, 10000)
        sections.append(sec)                                   # little-endian groups of 4 digits: 个, 万, 亿, 万亿
    names = ("", "万", "亿", "万亿")
    need_zero = False
    for i in range(len(sections) - 1, -1, -1):
        sec = sections[i]
        if sec == 0:
            need_zero = bool(out)
            continue
        if out and (need_zero or sec < 1000):
            out += "零"
        out += _read_section(sec) + names[i]
        need_zero = False
    return out[1:] if out.startswith("一十") else out


def read_number(text: str) -> str:
    """[-]digits[.digits] -> 负 + cardinal + 点 + digits."""
    neg = text.startswith("-")
    body = text[1:] if neg else text
    whole, _, frac = body.partition(".")
    out = read_cardinal(whole or "0")
    if frac:
        out += "点" + read_digits(frac)
    return ("负" if neg else "") + out


_NUM = r"\d+(?:\.\d+)?"
_RE_DATE = re.compile(r"(?<![\d.])(\d{4}|\d{2})年(?:(0?[1-9]|1[0-2])月)?(?:(0?[1-9]|[12]\d|3[01])([日号]))?")
_RE_DATE_SEP = re.compile(r"(?<!\d)(\d{4})([-/.])(0?[1-9]|1[0-2])\2(0?[1-9]|[12]\d|3[01])(?!\d)")
_RE_TIME = re.compile(r"(?<![\d:])([01]?\d|2[0-3]):([0-5]\d)(?::([0-5]\d))?(?![\d:])")
_RE_TEMP = re.compile(r"(-?)(" + _NUM + r")\s*(°C|℃|度|摄氏度)")
_RE_PERCENT = re.compile(r"(-?)(" + _NUM + r")\s*[%％]")
_RE_FRACTION = re.compile(r"(?<![\d.])(-?)(\d+)\s*/\s*(\d+)(?![\d.])")
_RE_RANGE = re.compile(r"(?<![\d.])(" + _NUM + r")\s*[-~～—]\s*(" + _NUM + r")(?![\d.])")
_RE_MOBILE = re.compile(r"(?<!\d)(?:\+?86 ?)?(1[3-9]\d{9})(?!\d)")
_RE_LONG = re.compile(r"(?<![\d.])\d{9,}(?![\d.])")
_RE_NUMBER = re.compile(r"(?<![\d.])-?" + _NUM)


def _date(m: re.Match) -> str:
    year, month, day, suffix = m.groups()
    # two- and four-digit years are read digit by digit, like the package this stands in for does ("98年" = 九八年; a bare "20年" meaning
    # "twenty years" is therefore read 二零年 there too); longer numbers never reach this rule (digit lookbehind)
    out = read_digits(year) + "年"
    if month:
        out += read_cardinal(month) + "月"
    if day:
        out += read_cardinal(day) + suffix
    return out


def _clock(m: re.Match) -> str:
    hour, minute, second = m.groups()
    out = read_cardinal(hour) + "点"
    if int(minute) == 30 and not second:
        return out + "半"
    if int(minute) or second:
        out += ("零" if minute[0] == "0" and int(minute) else "") + read_cardinal(minute) + "分"
    if second:
        out += ("零" if second[0] == "0" and int(second) else "") + read_cardinal(second) + "秒"
    return out


def read_numbers_zh(text: str) -> str:
    """Every Arabic-numeral expression of a Chinese sentence spelled in Chinese characters (see the module header for the order)."""
    text = _RE_DATE.sub(_date, text)
    text = _RE_DATE_SEP.sub(lambda m: read_digits(m.group(1)) + "年" + read_cardinal(m.group(3)) + "月" + read_cardinal(m.group(4)) + "日", text)
    text = _RE_TIME.sub(_clock, text)
    text = _RE_TEMP.sub(lambda m: ("零下" if m.group(1) else "") + read_number(m.group(2)) + ("度" if m.group(3) == "度" else "摄氏度"), text)
    text = _RE_PERCENT.sub(lambda m: ("负" if m.group(1) else "") + "百分之" + read_number(m.group(2)), text)
    text = _RE_FRACTION.sub(lambda m: ("负" if m.group(1) else "") + read_cardinal(m.group(3)) + "分之" + read_cardinal(m.group(2)), text)
    text = _RE_RANGE.sub(lambda m: read_number(m.group(1)) + "到" + read_number(m.group(2)), text)
    text = _RE_MOBILE.sub(lambda m: read_digits(m.group(1), yao=True), text)
    text = _RE_LONG.sub(lambda m: read_digits(m.group(0), yao=True), text)
    return _RE_NUMBER.sub(lambda m: read_number(m.group(0)), text)
