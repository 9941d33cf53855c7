"""Import the *unmodified* reference modules from /root/reference (build container only).

TEST INFRASTRUCTURE -- not part of the product.  This file is only used by
``oracle/make_golden.py`` (to mint the committed fixtures under ``tests/golden/``)
and by CPU tests that are skipped automatically when ``/root/reference`` is
absent (it does not exist on the GPU box).  Nothing from the reference is
copied; we only *import* it, after installing the small compatibility shims
SURVEY.md section 8(c) lists (the vendored llama.py targets transformers 4.41,
this image has 5.x; tokenizer/dvae import-time deps are absent).

Shims (none of them touches the arithmetic of the hot path):
  1. ``transformers.LogitsWarper`` -- annotation-only import (gpt.py:14).
  2. ``LlamaConfig.rope_theta`` / ``rope_scaling`` attributes (llama.py:252,268).
  3. ``DynamicCache.from_legacy_cache`` / ``to_legacy_cache`` (llama.py:942,1010).
  4. stub modules ``pybase16384``, ``vector_quantize_pytorch``, ``torchaudio``
     (dvae.py:6,10,11 import them; the decode path never calls them).
  5. ``CHATTTS_PLUS_LOG_DIR`` -> /tmp (logger.py:91-98 mkdirs under the read-only tree).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CTTS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "chattts_plus", "models"))


_loaded = {}


def load_reference():
    """Returns a namespace with .gpt, .llama, .processors, .dvae reference modules."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    os.environ.setdefault("CHATTTS_PLUS_LOG_DIR", "/tmp/ctts_ref_logs")
    import torch  # noqa: F401
    import transformers
    from transformers import LlamaConfig
    from transformers.cache_utils import DynamicCache

    if not hasattr(transformers, "LogitsWarper"):
        transformers.LogitsWarper = object
    if not hasattr(LlamaConfig, "rope_theta"):
        LlamaConfig.rope_theta = 10000.0
    # transformers 5.x aliases rope_scaling -> rope_parameters (a dict without "type");
    # the reference config has rope_scaling=None (llama.py:267-273 takes the plain RoPE branch).
    LlamaConfig.rope_scaling = property(lambda self: None, lambda self, v: None)

    if not hasattr(DynamicCache, "from_legacy_cache"):
        @classmethod
        def from_legacy_cache(cls, past_key_values=None):
            cache = cls()
            if past_key_values is not None:
                for i, (k, v) in enumerate(past_key_values):
                    cache.update(k, v, i)
            return cache

        def to_legacy_cache(self):
            return tuple((l.keys, l.values) for l in self.layers)

        DynamicCache.from_legacy_cache = from_legacy_cache
        DynamicCache.to_legacy_cache = to_legacy_cache

    for name in ("pybase16384", "torchaudio", "vector_quantize_pytorch"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["vector_quantize_pytorch"], "GroupedResidualFSQ"):
        sys.modules["vector_quantize_pytorch"].GroupedResidualFSQ = object
    if not hasattr(sys.modules["torchaudio"], "transforms"):
        # DVAE(encoder_config=...) constructs MelSpectrogramFeatures (dvae.py:185-192); torchaudio is absent, and the mel
        # extractor is NOT what the encoder golden pins (only downsample_conv + encoder are run): a parameter-free placeholder
        import torch

        class _AbsentMelSpectrogram(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

            def forward(self, x):
                raise RuntimeError("torchaudio is not installed: the mel extractor is restated in oracle/ref_cpu.mel_features")

        sys.modules["torchaudio"].transforms = types.SimpleNamespace(MelSpectrogram=_AbsentMelSpectrogram)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # bypass the package __init__ chains (they import tokenizer/pipeline deps)
    for pkg, sub in (("chattts_plus", ""), ("chattts_plus.models", "models")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REFERENCE_ROOT, "chattts_plus", sub)]
            sys.modules[pkg] = m
    for short in ("llama", "processors", "gpt", "dvae"):
        _loaded[short] = importlib.import_module("chattts_plus.models." + short)
    return types.SimpleNamespace(**_loaded)
