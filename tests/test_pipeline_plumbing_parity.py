"""Argument plumbing of the pipeline mirror against the IMPORTED reference pipeline (chattts_plus/pipelines/chattts_plus_pipeline.py):
`_infer_code` (:157-235), `_refine_text` (:237-277) and the text preamble + slicing of `_infer` (:349-416).  Both pipelines are built
without models (`object.__new__`) around the same recording stand-in for the GPT and the same tiny BERT vocabulary; what each hands to
`tokenizer.encode`, `gpt(...)` and `gpt.generate(...)` must be identical.  Runs only where /root/reference exists (CPU build container);
the import needs stand-ins for packages the image lacks (tensorrt-backed `trt_models`, `onnx2trt`, numba, zh_normalization) -- none of them
is executed on these paths."""
import importlib
import os
import sys
import types

import pytest
import torch

from chatttsplus_amd import codec, text_frontend
from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams, RefineTextParams
from chatttsplus_amd.tokenizer import Tokenizer
from oracle.ref_import import REFERENCE_ROOT, load_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]", "[Ebreak]",
         "[speed_5]", "[speed_3]", "[laugh]", "[Sbreak]", "[Pbreak]", "[oral_2]", "a", "b", "c", "d", "two", "and", "你", "好", "，", "。"]


class RecordingGPT:
    """Stands in for models_dict["gpt"] on both sides: remembers every call."""
    num_vq = 4
    max_batch = 4

    def __init__(self):
        self.emb_code = [types.SimpleNamespace(num_embeddings=626)]
        self.calls = []

    def __call__(self, input_ids, text_mask, **kw):
        self.calls.append(("embed", input_ids.clone(), text_mask.clone()))
        return torch.zeros(input_ids.shape[0], input_ids.shape[1], 768)

    def generate(self, emb, inputs_ids, **kw):
        rec = dict(kw)
        rec["temperature"] = [round(float(t), 6) for t in kw["temperature"]]
        rec["attention_mask"] = kw["attention_mask"].tolist()
        for key in ("logits_warpers", "logits_processors"):
            rec[key] = [sorted((a, round(float(getattr(o, a)), 6)) for a in ("top_p", "top_k", "min_tokens_to_keep", "penalty", "past_window", "max_input_ids")
                               if hasattr(o, a)) for o in kw[key]]
        rec.pop("context", None)
        self.calls.append(("generate", tuple(emb.shape), inputs_ids.clone(), rec))
        B = emb.shape[0]
        out = types.SimpleNamespace(ids=[torch.tensor([18, 19, 10, 20]) for _ in range(B)], hiddens=[torch.zeros(1, 768)] * B, attentions=[])
        return iter([out])


def _bert(tmp_path):
    from transformers import BertTokenizerFast
    (tmp_path / "vocab.txt").write_text("\n".join(VOCAB), encoding="utf-8")
    bt = BertTokenizerFast(vocab_file=str(tmp_path / "vocab.txt"), do_lower_case=False)
    bt.add_special_tokens({"additional_special_tokens": [v for v in VOCAB if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
    return bt


@pytest.fixture()
def both(tmp_path):
    from oracle import make_golden_text as mg
    load_reference()
    mg._install_stand_ins()
    added = []
    for name, attrs in (("chattts_plus.trt_models", dict(__path__=[])), ("chattts_plus.commons.onnx2trt", dict(convert_onnx_to_trt=lambda *a, **k: None)),
                        ("chattts_plus.pipelines", dict(__path__=[os.path.join(REFERENCE_ROOT, "chattts_plus", "pipelines")]))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            added.append(name)
    ref_pipe_mod = importlib.import_module("chattts_plus.pipelines.chattts_plus_pipeline")
    ref_tok_mod = importlib.import_module("chattts_plus.models.tokenizer")
    ref_norm_mod = importlib.import_module("chattts_plus.commons.norm")
    bt = _bert(tmp_path)
    patched = not hasattr(type(bt), "encode_plus")                 # transformers 5 dropped the 4.x alias the reference calls (tokenizer.py:70)
    if patched:
        type(bt).encode_plus = lambda self, *a, **k: self(*a, **k)
    ref_tok = object.__new__(ref_tok_mod.Tokenizer)
    ref_tok._tokenizer, ref_tok._decode_prompt = bt, codec.decode_prompt
    ref_tok.spk_emb_ids, ref_tok.break_0_ids, ref_tok.eos_token = (bt.convert_tokens_to_ids(t) for t in ("[spk_emb]", "[break_0]", "[Ebreak]"))
    ref_tok.len = len(bt)
    ref_tok.decode = bt.batch_decode
    ref_tok.apply_spk_emb = lambda emb, spk, ids, dev: emb       # the speaker overwrite is a kernel on the HIP side (ctts_gpt_embed); not plumbing
    hmap = tmp_path / "homophones_map.json"
    hmap.write_text('{"粘": "年"}', encoding="utf-8")
    ref = object.__new__(ref_pipe_mod.ChatTTSPlusPipeline)
    ref.logger = __import__("logging").getLogger("ref")
    ref.device, ref.dtype, ref.infer_type = torch.device("cpu"), torch.float32, "pytorch"
    ref.models_dict = {"gpt": RecordingGPT(), "tokenizer": ref_tok}
    ref.normalizer = ref_norm_mod.Normalizer(str(hmap))
    ref._decode_to_wavs = lambda hiddens, use_decoder: [torch.zeros(256)] * len(hiddens)
    mine = object.__new__(ChatTTSPlusPipeline)
    mine.logger = ref.logger
    mine.device = torch.device("cpu")
    mine.models_dict = {"gpt": RecordingGPT(), "tokenizer": Tokenizer(tokenizer=bt)}
    mine.normalizer = text_frontend.Normalizer(str(hmap))
    mine.text_splitter = lambda lines: text_frontend.split_text(lines, zh_reader=lambda s: s)   # the minting stand-in of zh_normalization passes text through
    mine._gpt_for_lora = lambda path: mine.models_dict["gpt"]
    mine._decode_to_wavs = ref._decode_to_wavs
    yield ref, mine
    if patched:
        del type(bt).encode_plus
    for name in added:
        sys.modules.pop(name, None)
    mg.remove_stand_ins()


def _same_calls(a, b):
    assert len(a) == len(b) and [c[0] for c in a] == [c[0] for c in b]
    assert sum(c[0] == "generate" for c in a) >= 1
    for x, y in zip(a, b):
        if x[0] == "embed":
            assert torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
        else:
            assert x[1] == y[1] and torch.equal(x[2], y[2])
            assert x[3] == y[3], (x[3], y[3])


PROMPT = codec.encode_prompt((torch.arange(4 * 5).reshape(4, 5) * 7 % 626).to(torch.int32))


@pytest.mark.parametrize("params", [
    dict(),
    dict(spk_emb="anything", temperature=[0.2, 0.3, 0.4, 0.5], top_P=0.9, top_K=8, repetition_penalty=1.2, max_new_token=77, min_new_token=3),
    dict(prompt="[speed_3]", txt_smp="a b", spk_smp=PROMPT, ensure_non_empty=False, stream_batch=5, show_tqdm=False),
    dict(prompt="", spk_emb="x", temperature=0.0003),
])
def test_infer_code_hands_the_generator_the_same_arguments(both, params):
    ref, mine = both
    text = ["a b c [uv_break]", "[Stts][spk_emb]d d[Ptts] [uv_break]", " 你 好 ， a"]
    ref._infer_code(list(text), False, True, ref_params(ref, "InferCodeParams", params))
    mine._infer_code(list(text), False, True, InferCodeParams(**params))
    _same_calls(ref.models_dict["gpt"].calls, mine.models_dict["gpt"].calls)


def ref_params(ref, cls, kw):
    utils = importlib.import_module("chattts_plus.commons.utils")
    return getattr(utils, cls)(**kw)


@pytest.mark.parametrize("params", [dict(), dict(prompt="[oral_2]", temperature=0.5, top_P=0.8, top_K=10, max_new_token=33, min_new_token=1, show_tqdm=False)])
def test_refine_text_hands_the_generator_the_same_arguments(both, params):
    ref, mine = both
    text = ["a b c", "你 好 ， d"]
    ref._refine_text(list(text), ref_params(ref, "RefineTextParams", params))
    mine._refine_text(list(text), RefineTextParams(**params))
    _same_calls(ref.models_dict["gpt"].calls, mine.models_dict["gpt"].calls)


@pytest.mark.parametrize("flags", [
    dict(),
    dict(do_text_optimization=False),
    dict(do_text_normalization=False, do_homophone_replacement=False),
    dict(lang="zh"),
    dict(skip_refine_text=False),
    dict(skip_refine_text=False, refine_text_only=True),
])
def test_infer_preamble_and_slicing_match(both, flags):
    """Text optimisation (split, number spelling, short-sentence merge), Normalizer, slices of 4, the refine-text round trip through the
    tokenizer, the '[uv_break]' suffix: the sequence of embed / generate calls is identical."""
    ref, mine = both
    long_line = ("c d a b, " * 30).strip()
    text = ["a b 2 c\nd and (b)!", "粘 好 ， 你", long_line, "a", "b [laugh] c", "d d d d d d d d d d d d d d d d d d d d d d d d d d d d d d"]
    kw = dict(skip_refine_text=True)
    kw.update(flags)
    out_ref = list(ref._infer(list(text), params_refine_text=ref_params(ref, "RefineTextParams", dict(show_tqdm=False)),
                              params_infer_code=ref_params(ref, "InferCodeParams", dict(show_tqdm=False)), **kw))
    out_mine = list(mine._infer(list(text), params_refine_text=RefineTextParams(show_tqdm=False), params_infer_code=InferCodeParams(show_tqdm=False),
                                slice_size=4, **kw))
    _same_calls(ref.models_dict["gpt"].calls, mine.models_dict["gpt"].calls)
    assert len(out_ref) == len(out_mine)
    if flags.get("refine_text_only"):
        assert out_ref == out_mine


def test_reference_pipeline_end_to_end_equals_the_oracle_chain(both, tmp_path):
    """The reference pipeline's own `_infer` (text wrapping -> tokenizer -> GPT.generate -> per-utterance DVAE -> Vocos) with the reference's
    real GPT and DVAE modules on synthetic weights, against the chain the GPU test compares the HIP pipeline with
    (tests/test_gpu_pipeline.py::test_pipeline_infer_matches_oracle_chain): tokenizer.encode -> OracleGPT -> ref_cpu.dvae_decode ->
    ref_cpu.vocos_decode.  Vocos is absent from the image: both sides vocode with the oracle's restatement, so the waveform comparison pins
    everything up to the mel plus the per-utterance loop, not Vocos itself."""
    import numpy as np
    from chatttsplus_amd import synth
    from oracle import make_golden as mgold
    from oracle import ref_cpu
    ref, _ = both
    refmods = load_reference()
    gsd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    dsd = synth.dvae_state_dict(synth.DVAE_REAL, 1234)
    vsd = synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    cfg = synth.DVAE_REAL
    dvae = refmods.dvae.DVAE(decoder_config=dict(idim=cfg["idim"], odim=cfg["odim"], hidden=cfg["hidden"], n_layer=cfg["n_layer"], bn_dim=cfg["bn_dim"]),
                             dim=cfg["dim"]).eval()
    dvae.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()}, strict=True)

    class OracleVocos:                                         # stand-in for the absent third-party package (pipeline:93-111,303)
        def parameters(self):
            return iter([torch.zeros(1)])

        def decode(self, mel):
            return ref_cpu.vocos_decode(vsd, mel[0])[None]

    tok = ref.models_dict["tokenizer"]
    tok._decode_spk_emb = codec.decode_spk_emb                 # pybase16384 is absent; the codec is pinned on the bundled speaker files
    del tok.apply_spk_emb                                      # the fixture's pass-through: here the reference's own overwrite must run
    ref.models_dict = {"gpt": mgold.build_ref_gpt(refmods, synth.GPT_REAL, gsd), "tokenizer": tok, "dvae_decode": dvae, "vocos": OracleVocos()}
    del ref._decode_to_wavs                                    # ... and the reference's own per-utterance loop (pipeline:286-305)
    spk = torch.load(os.path.join(os.path.dirname(__file__), "golden", "speakers", "2222.pt"), weights_only=True)
    params = ref_params(ref, "InferCodeParams", dict(prompt="[speed_5]", spk_emb=spk, max_new_token=10, min_new_token=2, show_tqdm=False))
    texts = ["a b c d a b", "c a"]
    torch.manual_seed(11)
    outs = list(ref._infer(list(texts), False, None, True, False, True, True, False, True, ref_params(ref, "RefineTextParams", {}), params))
    assert len(outs) == 1 and len(outs[0]) == 2
    # oracle chain on the same inputs
    mine_tok = Tokenizer(tokenizer=tok._tokenizer)
    wrapped = [f"[Stts][spk_emb][speed_5]{t} [uv_break][Ptts]" for t in texts]
    ids, att, tm = mine_tok.encode(wrapped, 4)
    o = ref_cpu.OracleGPT(gsd, 12)
    emb = o.apply_spk_emb(o.embed(ids, tm), torch.from_numpy(codec.decode_spk_emb(spk)), ids, mine_tok.spk_emb_ids)
    torch.manual_seed(11)
    gen = o.generate(emb, ids, ref_cpu.SamplerParams(min_new_token=2), attention_mask=att, max_new_token=10)
    for b in range(2):
        n = gen.ids[b].shape[0]
        wav_ref = outs[0][b].numpy()
        assert wav_ref.shape[0] == 256 * (2 * n - 1)
        wav = ref_cpu.vocos_decode(vsd, ref_cpu.dvae_decode(dsd, gen.hiddens[b])).numpy()
        rms = float(np.sqrt(np.mean((wav - wav_ref) ** 2))) / float(np.sqrt(np.mean(wav_ref ** 2)))
        assert rms <= 5e-4, (b, rms)                           # measured 2e-4: fp32 hiddens differ by ~5e-6, Vocos exponentiates magnitudes
