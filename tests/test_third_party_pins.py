"""Third-party arithmetic of the path whose own packages are absent offline (vocos, vector_quantize_pytorch) pinned on PORTS of that
very code that DO ship in this image: transformers' xcodec2 model carries

  * `Xcodec2FiniteScalarQuantization` -- lucidrains' vector-quantize-pytorch FSQ (finite_scalar_quantization.py at commit 353d460), the layer
    `GroupedResidualFSQ` (dvae.py:72-77) stacks; its `Xcodec2Quantizer.forward` also reproduces ResidualFSQ's order for the original
    checkpoints: the projected input is bounded ONCE before the quantizer bounds it again -- the `pre_bound=True` order of the oracle;
  * `Xcodec2ISTFTHead` -- Vocos' ISTFTHead (Linear -> exp / clamp 1e2 -> polar -> irfft -> window -> overlap-add / window envelope) with
    vocos' "same" padding (spectral_ops.py at commit c859e3b), i.e. the "center" ISTFT the reference configures (pipeline:93-111) plus
    (n_fft - hop) / 2 - ... extra samples per side.

The oracle's restatements (oracle/ref_cpu.py: fsq_bound, gfsq_quantize, gfsq_latent_from_indices, vocos_head_spec / vocos_decode's ISTFT) are
compared with them; the residual / group wrapper of GroupedResidualFSQ and the Vocos backbone stay restatements."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu

xc = pytest.importorskip("transformers.models.xcodec2.modeling_xcodec2")


def _fsq(levels):
    return xc.Xcodec2FiniteScalarQuantization(types.SimpleNamespace(quantization_levels=list(levels)))


@pytest.mark.parametrize("levels", [(5, 5, 5, 5), (8, 5, 5, 5), (8, 6, 5)])
def test_fsq_layer_matches_the_port(levels):
    fsq = _fsq(levels)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(4000, len(levels), generator=g) * 3.0
    lv = torch.tensor(levels, dtype=torch.float32)
    assert torch.equal(ref_cpu.fsq_bound(z, lv), fsq.bound(z))
    codes_p, idx_p = fsq(z)                                                  # bound -> round -> / half_width ; indices
    hw = torch.tensor([l // 2 for l in levels], dtype=torch.float32)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1]), dtype=torch.float32), 0)
    codes_o = torch.round(ref_cpu.fsq_bound(z, lv)) / hw                    # the lines of gfsq_quantize
    idx_o = ((codes_o * hw + hw) * basis).sum(-1).to(torch.int32)
    assert torch.equal(codes_o, codes_p) and torch.equal(idx_o, idx_p)
    assert int(idx_p.min()) >= 0 and int(idx_p.max()) < int(np.prod(levels))
    # index -> code: the table get_output_from_indices / GFSQ._embed uses
    assert torch.equal(fsq._indices_to_codes(idx_p.long()), codes_p)
    assert torch.equal(fsq.codebook[idx_p.long()], codes_p)


@pytest.mark.parametrize("pre_bound", [True, False])
def test_grouped_residual_fsq_restatement_on_the_ported_layer(pre_bound):
    """gfsq_quantize / gfsq_latent_from_indices (G = 2 groups, R = 2 residual layers, levels 5^4, scale (L - 1)^-r: the reference's vq_config)
    against the same residual loop written around the PORTED FSQ layer; identity project_in so that only the quantiser is compared."""
    levels, G, R = (5, 5, 5, 5), 2, 2
    fsq = _fsq(levels)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3000, G * 4, generator=g) * 2.5
    sd = {}
    for gi in range(G):
        sd[f"vq_layer.quantizer.rvqs.{gi}.project_in.weight"] = torch.eye(4)
        sd[f"vq_layer.quantizer.rvqs.{gi}.project_in.bias"] = torch.zeros(4)
    idx_o, lat_o = ref_cpu.gfsq_quantize(x, sd, levels, G, R, pre_bound)
    lv = torch.tensor(levels, dtype=torch.float32)
    rows, lats = [], []
    for gi in range(G):
        z = x[:, gi * 4:(gi + 1) * 4]
        residual = fsq.bound(z) if pre_bound else z                         # Xcodec2Quantizer.forward: "for consistency with original checkpoint"
        acc = torch.zeros_like(z)
        for r in range(R):
            scale = (lv - 1) ** (-r)
            codes, idx = fsq(residual / scale)
            residual = residual - codes * scale
            acc = acc + codes * scale
            rows.append(idx)
        lats.append(acc)
    assert torch.equal(idx_o, torch.stack(rows, 0))
    assert torch.equal(lat_o, torch.stack(lats, 0))
    # decode side: indices alone -> the latent (GFSQ._embed before project_out)
    back = ref_cpu.gfsq_latent_from_indices(idx_o, levels, G, R)
    assert torch.allclose(back, lat_o, atol=1e-6)
    want = torch.stack([sum(fsq._indices_to_codes(idx_o[gi * R + r].long()) * (lv - 1) ** (-r) for r in range(R)) for gi in range(G)], 0)
    assert torch.allclose(back, want, atol=1e-6)


def test_istft_head_matches_the_ported_vocos_head():
    """oracle head (vocos_head_spec + the centred torch.istft of vocos_decode, and the direct fp64 definition istft_direct) against the ported
    Vocos ISTFTHead: identical Linear weights, random backbone features.  The port trims (n_fft - hop) / 2 = 384 samples per side ("same"
    padding), the reference's "center" padding n_fft / 2 = 512: the centred waveform is the port's output without 128 samples per side."""
    n_fft, hop, H, Fr = 1024, 256, 512, 37
    head = xc.Xcodec2ISTFTHead(types.SimpleNamespace(hidden_size=H, n_fft=n_fft, hop_length=hop))
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        head.linear.weight.copy_(torch.randn(n_fft + 2, H, generator=g) * 0.04)
        head.linear.bias.copy_(torch.randn(n_fft + 2, generator=g) * 0.5)
        head.linear.bias[:8] += 6.0                                            # some magnitudes beyond the 1e2 clamp
    feats = torch.randn(Fr, H, generator=g)
    with torch.no_grad():
        same = head(feats[None])[0, 0]                                         # [hop * Fr]
    assert same.shape[0] == hop * Fr
    sd = {"head.out.weight": head.linear.weight.detach(), "head.out.bias": head.linear.bias.detach(), "head.istft.window": torch.hann_window(n_fft)}
    re, im = ref_cpu.vocos_head_spec(sd, feats)
    assert float(torch.sqrt(re * re + im * im).max()) == pytest.approx(100.0, rel=1e-5)      # the clamp is exercised
    centre = torch.istft(torch.complex(re, im)[None], n_fft, hop, n_fft, sd["head.istft.window"], center=True)[0]
    assert centre.shape[0] == hop * (Fr - 1)
    ref = same[128:-128]
    scale = float(ref.abs().max())
    assert float((centre - ref).abs().max()) <= 2e-5 * scale
    direct = ref_cpu.istft_direct(re, im, sd["head.istft.window"], n_fft, hop)
    assert float((direct - ref).abs().max()) <= 2e-5 * scale


def test_mel_filterbank_matches_the_torchaudio_adapted_one():
    """The zero-shot encoder's mel features (dvae.py:184-191: torchaudio MelSpectrogram defaults = htk scale, no area normalisation, triangles in
    Hz) use oracle.mel_filterbank; transformers.audio_utils.mel_filter_bank is "adapted from torchaudio and librosa" and builds the same bank."""
    from transformers.audio_utils import mel_filter_bank
    ours = ref_cpu.mel_filterbank(513, 100, 24000).numpy()
    theirs = mel_filter_bank(513, 100, 0.0, 12000.0, 24000, norm=None, mel_scale="htk", triangularize_in_mel_space=False)
    assert ours.shape == theirs.shape == (513, 100)
    assert float(np.abs(ours - theirs).max()) <= 2e-5
    slaney = mel_filter_bank(513, 100, 0.0, 12000.0, 24000, norm="slaney", mel_scale="slaney")
    assert float(np.abs(ours - slaney).max()) > 1e-2              # (the librosa-style bank is a different one: the comparison can fail)
