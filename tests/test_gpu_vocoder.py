"""GPU parity of the DVAE decoder and Vocos kernels through the C ABI.
DVAE: against the golden mel minted from the imported reference and against the oracle (tolerance 1e-3 RMS
relative to the signal RMS, north_star; we assert a much tighter 2e-4 max-abs on O(1) mels).
Vocos: parity-unpinned upstream; checked against the oracle restatement only."""
import os

import numpy as np
import pytest
import torch

from chatttsplus_amd import synth
from oracle import ref_cpu
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def voc():
    from chatttsplus_amd.hip_models import DVAE, Synth, Vocos
    s = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2048)
    d = DVAE(dict(idim=384, odim=384, hidden=512, n_layer=12, bn_dim=128), dim=384, synth=s)
    v = Vocos(s)
    d.load_state_dict(synth.dvae_state_dict(synth.DVAE_REAL, 1234))
    v.load_state_dict(synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
    return s, d, v


def _rms(x):
    return float(np.sqrt(np.mean(np.square(x))))


def test_dvae_golden(voc):
    s, d, v = voc
    z = np.load(os.path.join(GOLDEN, "dvae_real.npz"))
    n = int(z["n"])
    hid = np.random.Generator(np.random.Philox(key=int(z["hidden_seed"]))).standard_normal((n, 768)).astype(np.float32)
    mel = d(torch.from_numpy(hid).permute(1, 0)[None].cuda())[0].cpu().numpy()
    assert mel.shape == z["mel"].shape
    err = np.abs(mel - z["mel"]).max()
    assert _rms(mel - z["mel"]) <= 1e-3 * _rms(z["mel"]) and err <= 2e-4, f"max err {err}, rms {_rms(mel - z['mel'])}"


@pytest.mark.parametrize("n", [1, 5, 64, 333, 1000])
def test_dvae_vs_oracle(voc, n):
    s, d, v = voc
    hid = np.random.Generator(np.random.Philox(key=100 + n)).standard_normal((n, 768)).astype(np.float32)
    ref = ref_cpu.dvae_decode(synth.dvae_state_dict(synth.DVAE_REAL, 1234), torch.from_numpy(hid)).numpy()
    mel = s.dvae_decode(torch.from_numpy(hid).cuda()).cpu().numpy()
    assert mel.shape == ref.shape == (100, 2 * n)
    assert _rms(mel - ref) <= 1e-3 * _rms(ref) and np.abs(mel - ref).max() <= 5e-4, f"n={n}: max err {np.abs(mel - ref).max()}"


@pytest.mark.parametrize("F", [2, 9, 74, 513])
def test_vocos_vs_oracle(voc, F):
    s, d, v = voc
    mel = np.random.Generator(np.random.Philox(key=200 + F)).standard_normal((100, F)).astype(np.float32)
    ref = ref_cpu.vocos_decode(synth.vocos_state_dict(synth.VOCOS_REAL, 1234), torch.from_numpy(mel)).numpy()
    wav = v.decode(torch.from_numpy(mel)[None].cuda())[0].cpu().numpy()
    assert wav.shape == ref.shape == (256 * (F - 1),)
    assert _rms(wav - ref) <= 1e-3 * _rms(ref), f"F={F}: rms err {_rms(wav - ref)} vs signal {_rms(ref)}; max {np.abs(wav - ref).max()}"


def test_hidden_to_wav_chain(voc):
    """P1 (_decode_to_wavs, pipeline:286-305): hidden -> mel -> wav; chained error still within 1e-3 RMS."""
    s, d, v = voc
    n = 96
    hid = np.random.Generator(np.random.Philox(key=77)).standard_normal((n, 768)).astype(np.float32)
    mel_ref = ref_cpu.dvae_decode(synth.dvae_state_dict(synth.DVAE_REAL, 1234), torch.from_numpy(hid))
    wav_ref = ref_cpu.vocos_decode(synth.vocos_state_dict(synth.VOCOS_REAL, 1234), mel_ref).numpy()
    wav = v.decode(d(torch.from_numpy(hid).permute(1, 0)[None].cuda()))[0].cpu().numpy()
    assert wav.shape == (256 * (2 * n - 1),)
    assert _rms(wav - wav_ref) <= 1e-3 * _rms(wav_ref)


def test_batched_synthesis_matches_per_utterance(voc):
    """Synth.decode_batch (ctts_synth_batch: ragged lengths, an empty utterance, more utterances than max_batch)
    == per-utterance dvae_decode + vocos_decode, bit for bit."""
    from chatttsplus_amd.hip_models import Synth
    s, d, v = voc
    pool = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2048, max_batch=4)
    pool.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234))
    pool.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
    rng = np.random.Generator(np.random.Philox(key=5))
    hs = [torch.from_numpy(rng.standard_normal((n, 768)).astype(np.float32)).cuda() for n in (40, 7, 0, 300, 41, 99, 1, 64, 17)]
    outs = pool.decode_batch(hs)
    torch.cuda.synchronize()
    for h, w in zip(hs, outs):
        if h.shape[0] == 0:
            assert w.numel() == 0
            continue
        ref = s.vocos_decode(s.dvae_decode(h))
        assert torch.equal(w, ref), "batched result differs from the per-utterance result"


def test_decode_window_is_a_slice_of_the_prefix_waveform(voc):
    """Streaming: a window vocoded from the tokens inside its receptive field is bit-identical to the same
    samples of the full-prefix waveform -- at the start, in the interior, at the end, for ragged batches and empty windows."""
    s, d, v = voc
    rng = np.random.Generator(np.random.Philox(key=77))
    hs = [torch.from_numpy(rng.standard_normal((n, 768)).astype(np.float32)).cuda() for n in (400, 37, 130)]
    full = s.decode_batch(hs)
    total = [int(w.shape[0]) for w in full]
    cases = [(0, 6000), (6000, 12000), (50000, 62000), (100000, 100001), (total[0] - 5000, total[0]), (0, total[0]), (70000, 70000)]
    for s0, s1 in cases:
        got = s.decode_window(hs, [s0] * 3, [s1] * 3)
        for u in range(3):
            ref = full[u][min(s0, total[u]):min(s1, total[u])]
            assert got[u].shape == ref.shape, (s0, s1, u, got[u].shape, ref.shape)
            assert torch.equal(got[u], ref), f"window [{s0},{s1}) of utterance {u}: max diff {float((got[u] - ref).abs().max())}"
    # per-utterance windows
    got = s.decode_window(hs, [1000, 0, 60000], [3000, total[1], 66000])
    assert torch.equal(got[0], full[0][1000:3000]) and torch.equal(got[1], full[1]) and torch.equal(got[2], full[2][60000:66000])


def test_batched_synthesis_equals_single_utterance_path(voc):
    """ctts_synth_batch over a ragged batch reproduces, bit for bit, what each utterance gives alone: an output element's
    arithmetic does not depend on the rows (or utterances) around it."""
    s, d, v = voc
    if s.max_batch < 8:
        from chatttsplus_amd.hip_models import Synth
        s = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2048, max_batch=8)
        s.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234))
        s.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
    rng = np.random.Generator(np.random.Philox(key=78))
    hs = [torch.from_numpy(rng.standard_normal((n, 768)).astype(np.float32)).cuda() for n in (400, 380, 399, 17, 256, 400, 311, 400)]
    together = s.decode_batch(hs)
    for u, h in enumerate(hs):
        alone = s.decode_batch([h])[0]
        assert torch.equal(together[u], alone), f"utterance {u}: max diff {float((together[u] - alone).abs().max())}"


@pytest.mark.parametrize("F", [9, 74])
def test_vocos_vs_second_independent_restatement(voc, F):
    """HIP Vocos against the second, separately written float64 restatement (tests/vocos_independent.py) -- Vocos is parity-unpinned
    (third-party, absent offline); two independent readings of the upstream module list and the kernels all agree within 1e-3."""
    from tests.vocos_independent import vocos_decode_f64
    s, d, v = voc
    mel = np.random.Generator(np.random.Philox(key=200 + F)).standard_normal((100, F)).astype(np.float32)
    ref = vocos_decode_f64(synth.vocos_state_dict(synth.VOCOS_REAL, 1234), mel)
    wav = v.decode(torch.from_numpy(mel)[None].cuda())[0].cpu().numpy()
    assert wav.shape == ref.shape
    assert _rms(wav - ref) <= 1e-3 * _rms(ref), f"F={F}: rms err {_rms(wav - ref)} vs signal {_rms(ref)}"


def test_dvae_full_decode_codes_golden_and_oracle():
    """use_decoder=False (pipeline:292): ids -> GFSQ._embed -> DVAE_full decoder stack -> mel.  Golden = the reference's own DVAE module on the
    latent our GFSQ restatement builds (tests/golden/dvae_full_decode_real.npz); lengths 1 / 41 / 200 tokens; then the batched
    codes -> waveform entry point against the oracle chain."""
    from chatttsplus_amd.hip_models import Synth
    cfg = synth.DVAE_FULL_DEC
    sd = synth.dvae_full_decoder_state_dict(cfg, 1234)
    vsd = synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    s = Synth(dict(cfg), dict(synth.VOCOS_REAL), max_frames=1024, max_batch=4, vq_cfg=dict(dim=1024, levels=[5, 5, 5, 5], G=2, R=2))
    s.load("dvae.", sd)
    s.load("vocos.", vsd)
    z = np.load(os.path.join(GOLDEN, "dvae_full_decode_real.npz"))
    for n in (int(x) for x in z["lengths"]):
        ids = z[f"ids_{n}"].astype(np.int64)
        mel = s.dvae_decode_codes(torch.from_numpy(ids).cuda()).cpu().numpy()
        gold = z[f"mel_{n}"]
        assert mel.shape == gold.shape == (100, 2 * n)
        assert _rms(mel - gold) <= 1e-3 * _rms(gold) and np.abs(mel - gold).max() <= 5e-4, f"n={n}: max err {np.abs(mel - gold).max()}"
    batch = [torch.from_numpy(z[f"ids_{n}"].astype(np.int64)) for n in (41, 1, 200)]
    wavs = s.decode_batch([b.cuda() for b in batch])
    for b, w in zip(batch, wavs):
        ref = ref_cpu.vocos_decode(vsd, ref_cpu.dvae_decode_codes(sd, b)).numpy()
        assert w.shape[0] == ref.shape[0] == 256 * (2 * b.shape[0] - 1)
        assert _rms(w.cpu().numpy() - ref) <= 1e-3 * _rms(ref), f"n={b.shape[0]}"
    plain = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=64)
    with pytest.raises(Exception, match="quantiser|not loaded"):
        plain.dvae_decode_codes(batch[1].cuda())
