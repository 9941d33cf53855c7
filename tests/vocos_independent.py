"""A SECOND, independent restatement of the third-party Vocos decoder (test infrastructure; float64 numpy, loops and einsum only --
shares no code with oracle/ref_cpu.py).  Written from the upstream module list recorded in SURVEY.md section 8(c) (vocos 0.1.0, absent
offline -> parity unpinned):

  VocosBackbone : embed = Conv1d(100, 512, kernel 7, padding 3);  norm = LayerNorm(512, eps 1e-6) over channels;
                  8 x ConvNeXtBlock { dwconv = Conv1d(512, 512, 7, padding 3, groups 512); LayerNorm(eps 1e-6); pwconv1 = Linear(512, 1536);
                                      GELU (exact, erf); pwconv2 = Linear(1536, 512); * gamma[512]; + residual };
                  final_layer_norm = LayerNorm(512, eps 1e-6)
  ISTFTHead     : out = Linear(512, 1026) -> (mag, phase) = two halves of 513;  mag = clip(exp(mag), max=1e2);
                  S = mag * (cos(phase) + i sin(phase));  torch.istft(S, n_fft 1024, hop 256, win_length 1024, hann window, center=True)
                  = per frame irfft (1/N normalisation) * window, overlap-add at hop, divided by the overlap-added squared window,
                    with n_fft/2 samples trimmed from both ends.
Two restatements written separately from the same published description agreeing to 1e-6 does not pin upstream, but it removes
transcription slips from the list of things that can be wrong."""
import math

import numpy as np


def _layer_norm(x, w, b, eps=1e-6):            # x [T, C]
    mu = x.mean(axis=1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _gelu_erf(x):
    return 0.5 * x * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))


def vocos_decode_f64(sd, mel):
    """sd: state dict (numpy arrays, upstream key names); mel [100, F] -> waveform [256 * (F - 1)] float64."""
    g = lambda k: np.asarray(sd[k], dtype=np.float64)
    mel = np.asarray(mel, dtype=np.float64)
    C_in, F = mel.shape
    # embed: full convolution, kernel 7, zero padding 3
    W, b = g("backbone.embed.weight"), g("backbone.embed.bias")             # [512, 100, 7]
    xp = np.pad(mel, ((0, 0), (3, 3)))
    x = np.stack([np.einsum("oik,ik->o", W, xp[:, t:t + 7]) for t in range(F)], axis=0) + b      # [F, 512]
    x = _layer_norm(x, g("backbone.norm.weight"), g("backbone.norm.bias"))
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("backbone.convnext."))
    for i in range(n_layers):
        p = f"backbone.convnext.{i}."
        dw, db = g(p + "dwconv.weight")[:, 0, :], g(p + "dwconv.bias")    # [512, 7]
        xpad = np.pad(x, ((3, 3), (0, 0)))
        y = np.zeros_like(x)
        for k in range(7):                                                  # depthwise: every channel its own 7 taps, no dilation
            y += xpad[k:k + F, :] * dw[:, k]
        y = y + db
        y = _layer_norm(y, g(p + "norm.weight"), g(p + "norm.bias"))
        y = y @ g(p + "pwconv1.weight").T + g(p + "pwconv1.bias")
        y = _gelu_erf(y)
        y = y @ g(p + "pwconv2.weight").T + g(p + "pwconv2.bias")
        x = x + g(p + "gamma") * y
    x = _layer_norm(x, g("backbone.final_layer_norm.weight"), g("backbone.final_layer_norm.bias"))
    # head
    h = x @ g("head.out.weight").T + g("head.out.bias")                     # [F, 1026]
    nb = h.shape[1] // 2
    mag = np.minimum(np.exp(h[:, :nb]), 1e2)
    ph = h[:, nb:]
    S = mag * (np.cos(ph) + 1j * np.sin(ph))                                # [F, 513]
    n_fft, hop = 2 * (nb - 1), 256
    win = g("head.istft.window")
    out = np.zeros(n_fft + hop * (F - 1))
    env = np.zeros_like(out)
    for t in range(F):
        frame = np.fft.irfft(S[t], n=n_fft) * win
        out[t * hop:t * hop + n_fft] += frame
        env[t * hop:t * hop + n_fft] += win * win
    half = n_fft // 2
    out, env = out[half:len(out) - half], env[half:len(env) - half]
    return out / env
