"""chatttsplus_amd.tokenizer.Tokenizer.encode against the reference's own Tokenizer.encode (chattts_plus/models/tokenizer.py:49-137), run live
where /root/reference exists, and against the fixture minted from it (tests/golden/tokenizer_encode.json: inputs + the reference's ids /
masks for a tiny BERT vocabulary) everywhere else.  The audio-prompt string is decoded by this repo's codec on both sides (pybase16384 is
absent from the image; the codec is pinned separately on the reference's bundled speaker files, tests/test_codec.py)."""
import json
import os

import pytest
import torch

from chatttsplus_amd import codec
from chatttsplus_amd.tokenizer import Tokenizer

FIX = os.path.join(os.path.dirname(__file__), "golden", "tokenizer_encode.json")
VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]", "[Ebreak]",
         "[speed_5]", "[laugh]", "a", "b", "c", "d", "你", "好", "，"]
CASES = [
    dict(text=["[Stts][spk_emb][speed_5]a b c d [uv_break][Ptts]", "[Stts][empty_spk]c a[Ptts]"], prompt=False),
    dict(text=["a"], prompt=False),
    dict(text=["a b", "a b c d a b c d a b", "d", "你 好 ， a [laugh] zzz"], prompt=False),
    dict(text=["[Stts][spk_emb]b b[Ptts]", "[Stts][spk_emb]你 好 ， a b c[Ptts]"], prompt=True),
]


def _bert(tmp_path):
    from transformers import BertTokenizerFast
    (tmp_path / "vocab.txt").write_text("\n".join(VOCAB), encoding="utf-8")
    bt = BertTokenizerFast(vocab_file=str(tmp_path / "vocab.txt"), do_lower_case=False)
    bt.add_special_tokens({"additional_special_tokens": [v for v in VOCAB if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
    return bt


def _prompt_string():
    codes = (torch.arange(4 * 7).reshape(4, 7) * 13 % 626).to(torch.int32)          # [num_vq, n] audio codes
    return codec.encode_prompt(codes)


def _run(tok, case):
    ids, att, tm = tok.encode(list(case["text"]), 4, _prompt_string() if case["prompt"] else None, "cpu")
    return dict(ids=ids.tolist(), attention_mask=att.tolist(), text_mask=tm.to(torch.int64).tolist())


def test_encode_matches_the_fixture(tmp_path):
    gold = json.load(open(FIX, encoding="utf-8"))
    assert gold["vocab"] == VOCAB
    tok = Tokenizer(tokenizer=_bert(tmp_path))
    for case, want in zip(CASES, gold["cases"]):
        assert _run(tok, case) == want
    assert [tok.spk_emb_ids, tok.break_0_ids, tok.eos_token, tok.len] == gold["special"]


def test_encode_matches_the_imported_reference(tmp_path):
    from oracle.ref_import import load_reference, reference_available
    if not reference_available():
        pytest.skip("reference tree not present")
    import importlib
    load_reference()
    ref_mod = importlib.import_module("chattts_plus.models.tokenizer")
    bt = _bert(tmp_path)
    ref_tok = object.__new__(ref_mod.Tokenizer)               # __init__ unpickles tokenizer.pt, which is not part of the tree
    added = not hasattr(type(bt), "encode_plus")               # transformers 5 dropped the 4.x alias the reference calls (tokenizer.py:70):
    if added:
        type(bt).encode_plus = lambda self, *a, **k: self(*a, **k)  # for one text it was the same call as __call__
    ref_tok._tokenizer = bt
    ref_tok._decode_prompt = codec.decode_prompt                # pybase16384 is absent (see the module docstring)
    mine = Tokenizer(tokenizer=bt)
    out = []
    try:
        for case in CASES:
            ids, att, tm = ref_tok.encode(list(case["text"]), 4, _prompt_string() if case["prompt"] else None, "cpu")
            want = dict(ids=ids.tolist(), attention_mask=att.tolist(), text_mask=tm.to(torch.int64).tolist())
            assert _run(mine, case) == want
            out.append(want)
    finally:
        if added:
            del type(bt).encode_plus
    if os.environ.get("CTTS_MINT_TOKENIZER_FIXTURE"):
        json.dump(dict(vocab=VOCAB, cases=out, special=[mine.spk_emb_ids, mine.break_0_ids, mine.eos_token, mine.len]), open(FIX, "w", encoding="utf-8"), ensure_ascii=False)
