"""The build's own Chinese numeral reader (chatttsplus_amd/zh_numbers.py): the fallback `zh_reader` when the third-party zh_normalization
package the reference uses (text_utils.py:3) is not installed.  Hand-checked expectations -- not a parity test (see the module header)."""
import pytest

from chatttsplus_amd import text_frontend as tf
from chatttsplus_amd.zh_numbers import read_cardinal, read_digits, read_number, read_numbers_zh


@pytest.mark.parametrize("digits,want", [
    ("0", "零"), ("7", "七"), ("10", "十"), ("11", "十一"), ("20", "二十"), ("99", "九十九"), ("100", "一百"), ("101", "一百零一"), ("110", "一百一十"),
    ("1000", "一千"), ("1001", "一千零一"), ("1010", "一千零一十"), ("1100", "一千一百"), ("9999", "九千九百九十九"), ("10000", "一万"), ("10001", "一万零一"),
    ("10100", "一万零一百"), ("11000", "一万一千"), ("100000", "十万"), ("110000", "十一万"), ("1010000", "一百零一万"), ("100000000", "一亿"),
    ("100000005", "一亿零五"), ("123456789", "一亿二千三百四十五万六千七百八十九"), ("1000000000000", "一万亿"), ("007", "七"),
    ("12345678901234567", "一二三四五六七八九零一二三四五六七"),
])
def test_cardinals(digits, want):
    assert read_cardinal(digits) == want


def test_digits_and_numbers():
    assert read_digits("2024") == "二零二四" and read_digits("110", yao=True) == "幺幺零"
    assert read_number("3.14") == "三点一四" and read_number("-0.5") == "负零点五" and read_number("12") == "十二"


@pytest.mark.parametrize("text,want", [
    ("价格100元", "价格一百元"),
    ("今天是2024年3月5日", "今天是二零二四年三月五日"),
    ("日期2024-03-05到了", "日期二零二四年三月五日到了"),
    ("98年出生", "九八年出生"),
    ("会议12:30开始", "会议十二点半开始"),
    ("8:05:09出发", "八点零五分零九秒出发"),
    ("现在9:00", "现在九点"),
    ("气温-3℃", "气温零下三摄氏度"),
    ("体温37.5度", "体温三十七点五度"),
    ("增长了12.5%", "增长了百分之十二点五"),
    ("占1/3", "占三分之一"),
    ("3-5天", "三到五天"),
    ("10~20个", "十到二十个"),
    ("电话13812345678", "电话幺三八幺二三四五六七八"),
    ("编号1234567890", "编号幺二三四五六七八九零"),
    ("圆周率3.14159", "圆周率三点一四一五九"),
    ("没有数字", "没有数字"),
    ("共10001人", "共一万零一人"),
])
def test_sentences(text, want):
    assert read_numbers_zh(text) == want


def test_split_text_uses_the_reader_when_zh_normalization_is_absent():
    try:
        import zh_normalization  # noqa: F401
        pytest.skip("zh_normalization is installed: split_text delegates to it, like the reference")
    except ImportError:
        pass
    out = tf.split_text(["价格100元，涨了5%", "I have 2 cats"])
    assert out == ["价格一百元，涨了百分之五", "I have Two cats"]
    # what reaches the tokenizer keeps its numbers (the Normalizer would have dropped the digits)
    assert tf.Normalizer()(out[0]) == "价格一百元，涨了百分之五"


def test_year_like_durations_and_longer_numbers():
    """without a digit lookbehind "100年" was read as "1" + "零零年".  Two-digit years stay digit-wise (zh_normalization's own rule)."""
    from chatttsplus_amd.zh_numbers import read_numbers_zh
    assert read_numbers_zh("100年") == "一百年"
    assert read_numbers_zh("过了20年") == "过了二零年"
    assert read_numbers_zh("98年3月") == "九八年三月"
    assert read_numbers_zh("2024年") == "二零二四年"
    assert read_numbers_zh("12345年") == "一万二千三百四十五年"
