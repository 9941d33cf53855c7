"""GPU parity of the zero-shot encode branch through the C ABI: waveform -> log-mel -> DVAE encoder -> GFSQ codes.
Pinned part: the conv stack, against features minted by the reference's own downsample_conv + encoder modules
(tests/golden/dvae_encode_real.npz `feat`).  Unpinned (third-party absent offline): mel extractor and GFSQ -- checked against the
oracle restatement frozen in the same fixture (both GroupedResidualFSQ variants)."""
import os

import numpy as np
import pytest
import torch

from chatttsplus_amd import codec, synth
from oracle import ref_cpu
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _rms(x):
    return float(np.sqrt(np.mean(np.square(x))))


def _encoder(pre_bound=True):
    from chatttsplus_amd.hip_models import DVAEEncoder
    e = DVAEEncoder(dim=512, max_seconds=8.0, pre_bound=pre_bound)
    return e.load_state_dict(synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, 1234))


def test_encode_golden():
    z = np.load(os.path.join(GOLDEN, "dvae_encode_real.npz"))
    sd = synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, int(z["weight_seed"]))
    wav = torch.from_numpy(synth.speaker_wave(int(z["wave_seed"]), int(z["n_samples"])))
    e = _encoder(True)
    ids, mel, feat = e.encode(wav, return_debug=True)
    mel = mel.cpu().numpy() * sd["coef"].reshape(-1, 1)                  # the hook returns log-mel / coef
    assert mel.shape == z["mel"].shape and np.abs(mel - z["mel"]).max() <= 2e-3, np.abs(mel - z["mel"]).max()
    feat = feat.cpu().numpy().T
    assert feat.shape == z["feat"].shape
    assert _rms(feat - z["feat"]) <= 1e-3 * _rms(z["feat"]), _rms(feat - z["feat"]) / _rms(z["feat"])       # north_star tolerance
    assert np.array_equal(ids.cpu().numpy(), z["ids_pre_bound"])
    ids2 = _encoder(False).encode(wav)
    assert np.array_equal(ids2.cpu().numpy(), z["ids"])
    assert int(ids.max()) < 625 and int(ids.min()) >= 0


@pytest.mark.parametrize("n", [600, 12800, 50001, 191999])
def test_encode_vs_oracle_lengths(n):
    sd = synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, 1234)
    wav = torch.from_numpy(synth.speaker_wave(100 + n, n))
    ref = ref_cpu.dvae_encode(sd, wav).numpy()
    got = _encoder(True).encode(wav).cpu().numpy()
    assert got.shape == ref.shape == (4, ((1 + n // 256) - 2) // 2 + 1)
    # a code flips only if tanh(z) * 2.002 lands within float rounding of a half-integer
    assert (got != ref).mean() <= 0.005, f"n={n}: {(got != ref).sum()} of {ref.size} codes differ"


def test_capacity_and_errors():
    from chatttsplus_amd import _lib
    e = _encoder(True)
    with pytest.raises(_lib.HipBackendError):
        e.encode(torch.zeros(400))                                       # reflect padding needs more than n_fft / 2 samples
    with pytest.raises(_lib.HipBackendError):
        e.encode(torch.zeros(8 * 24000 + 1))                             # longer than max_seconds
    with pytest.raises(_lib.HipBackendError):
        e(torch.zeros(1, 4000), "decode")


def test_prompt_string_round_trip():
    wav = torch.from_numpy(synth.speaker_wave(3, 30000))
    ids = _encoder(True)(wav[None].cuda(), "encode")[0].cpu()
    s = codec.encode_prompt(ids)
    assert torch.equal(codec.decode_prompt(s), ids.to(torch.int64))
