"""The device noise stream (noise="device": the default for batches the reference cannot run and for the refine-text pass) pinned on the CPU:
its generator is the published Philox4x32-10 (Random123 known-answer vectors), its output is i.i.d. Exp(1) -- what torch.multinomial draws
inside the reference (gpt.py:480-481) -- and streams of different utterances / codebooks / steps / attempts are different streams.
The GPU side (tests/test_gpu_sampler.py) checks that the kernels draw exactly this stream, keyed as documented."""
import numpy as np
import pytest
from scipy import stats

from oracle.device_noise import exp_noise, philox4x32_10


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x) for x in philox4x32_10(ctr, key))
        assert got == want, (ctr, key, [hex(g) for g in got])
    # vectorised over the first counter word = what exp_noise does
    a = philox4x32_10((np.arange(4, dtype=np.uint64), 0, 0, 0), (0, 0))
    assert int(a[0][0]) == 0x6627e8d5 and len({int(x) for x in a[0]}) == 4


@pytest.mark.parametrize("key", [(0, 0, 0, 0, 0), (2 ** 62 + 12345, 2 ** 40 + 7, 3, 2047, 5), (77, 99, 4, 0, 0)])
def test_stream_is_exp1(key):
    seed, uid, stream, step, attempt = key
    q = exp_noise(seed, uid, stream, step, attempt, 21178 if stream == 4 else 626)
    assert q.dtype == np.float32 and np.all(q > 0) and np.all(np.isfinite(q))
    # 40 rows of one utterance (steps step .. step + 39): Kolmogorov-Smirnov against Exp(1)
    big = np.concatenate([exp_noise(seed, uid, stream, step + s, attempt, 626) for s in range(40)])
    assert stats.kstest(big, "expon").pvalue > 1e-3
    assert abs(float(big.mean()) - 1.0) < 0.03 and abs(float(big.var()) - 1.0) < 0.08


def test_streams_of_different_keys_are_different_and_uncorrelated():
    base = exp_noise(11, 5, 1, 9, 0, 626)
    others = {"seed": exp_noise(12, 5, 1, 9, 0, 626), "utterance": exp_noise(11, 6, 1, 9, 0, 626), "utterance_hi": exp_noise(11, 5 + 2 ** 32, 1, 9, 0, 626),
              "codebook": exp_noise(11, 5, 2, 9, 0, 626), "step": exp_noise(11, 5, 1, 10, 0, 626), "attempt": exp_noise(11, 5, 1, 9, 1, 626)}
    for name, o in others.items():
        assert not np.array_equal(base, o), name
        assert abs(float(np.corrcoef(base, o)[0, 1])) < 0.15, name
    # and the same key is the same stream: nothing else enters (no batch row, no batch size, no draw counter)
    assert np.array_equal(base, exp_noise(11, 5, 1, 9, 0, 626))
    assert np.array_equal(base[:100], exp_noise(11, 5, 1, 9, 0, 100))
