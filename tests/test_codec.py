"""Speaker / prompt wire formats (chatttsplus_amd/codec.py) against the reference's own data files
(assets/speakers/2222.pt is what tests/test_pipelines.py:66-120 of the reference feeds the TRT pipeline)."""
import os

import numpy as np
import torch

from chatttsplus_amd import codec
from tests.helpers import GOLDEN


def test_real_speaker_files_decode_and_reencode():
    for f in ("2222.pt", "zh_man_54.pt"):
        s = torch.load(os.path.join(GOLDEN, "speakers", f), weights_only=True)
        assert isinstance(s, str)
        v = codec.decode_spk_emb(s)
        assert v.shape == (768,) and v.dtype == np.float16
        assert 3.0 < float(v.astype(np.float32).std()) < 7.0
        assert codec.encode_spk_emb(v) == s                     # byte-identical wire format
        assert codec.b14_encode(codec.b14_decode(s)) == s


def test_b14_roundtrip_all_remainders():
    rng = np.random.default_rng(0)
    for n in range(0, 64):
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert codec.b14_decode(codec.b14_encode(d)) == d


def test_prompt_and_coef_roundtrip():
    p = torch.randint(0, 626, (4, 37))
    assert torch.equal(codec.decode_prompt(codec.encode_prompt(p)), p)
    c = np.random.default_rng(1).random(100).astype(np.float32)
    assert np.array_equal(codec.coef_from_string(codec.coef_to_string(c)), c)


def test_apply_spk_emb_matches_oracle():
    from oracle import ref_cpu
    s = torch.load(os.path.join(GOLDEN, "speakers", "2222.pt"), weights_only=True)
    ids = torch.randint(0, 100, (2, 6, 4)); ids[:, 1, :] = 7
    emb = torch.randn(2, 6, 768)
    a = codec.apply_spk_emb(emb.clone(), s, ids, 7)
    b = ref_cpu.OracleGPT.apply_spk_emb(emb.clone(), torch.from_numpy(codec.decode_spk_emb(s)), ids, 7)
    assert torch.equal(a, b)
    assert abs(float(a[0, 1].norm()) - 1.0) < 1e-5 and torch.equal(a[:, 0], emb[:, 0])
