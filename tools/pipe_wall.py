"""Host-inclusive wall clock of ChatTTSPlusPipeline.infer() (tokenize -> embed -> generate -> DVAE/Vocos), synthetic
checkpoints and a toy vocabulary.  python tools/pipe_wall.py [--n 1] [--tokens 512] [--dtype fp32|fp16]
  --rows R          decode rows (slice_size; default = n)      --ragged   per-utterance lengths U{tokens/4 .. tokens}
  --continuous [throughput]   infer(continuous=...): also reports when the first list of waveforms arrived"""
import argparse, json, os, sys, tempfile, time, pathlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams, load_config

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]",
         "[Ebreak]", "[speed_5]", "a", "b", "c", "d"]

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1, help="texts per infer() call")
ap.add_argument("--tokens", type=int, default=512)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--ragged", action="store_true")
ap.add_argument("--continuous", nargs="?", const=True, default=False)
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = pathlib.Path(tempfile.mkdtemp())
from transformers import BertTokenizerFast
from chatttsplus_amd.tokenizer import Tokenizer
(tmp / "vocab.txt").write_text("\n".join(VOCAB))
bt = BertTokenizerFast(vocab_file=str(tmp / "vocab.txt"), do_lower_case=False)
bt.add_special_tokens({"additional_special_tokens": [v for v in VOCAB if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
tok = Tokenizer(tokenizer=bt)
cfg = load_config(os.path.join(root, "configs", "infer", "chattts_plus_hip.yaml"))
cfg["MODELS"]["gpt"]["kwargs"].update(weight_dtype=a.dtype, max_batch=max(a.rows or a.n, 1), max_seq_len=a.tokens + 128)
os.makedirs(tmp / "asset")
for name, sd in (("GPT.pt", synth.gpt_state_dict(synth.GPT_REAL, 1234)), ("Decoder.pt", synth.dvae_state_dict(synth.DVAE_REAL, 1234)),
                 ("Vocos.pt", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))):
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, tmp / "asset" / name)
pipe = ChatTTSPlusPipeline(cfg, device="cuda", tokenizer=tok, checkpoint_dir=str(tmp))
spk = torch.load(os.path.join(root, "tests", "golden", "speakers", "2222.pt"), weights_only=True)
params = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=a.tokens, min_new_token=a.tokens, show_tqdm=False)
texts = [" ".join("abcd"[(i + j) % 4] for j in range(40)) for i in range(a.n)]
import numpy as np
kw = {}
if a.ragged:
    rng = np.random.Generator(np.random.Philox(key=5))
    kw["max_new_tokens_per_utterance"] = [int(x) for x in rng.integers(max(a.tokens // 4, 1), a.tokens + 1, size=a.n)]
if a.rows:
    kw["slice_size"] = a.rows
if a.continuous:
    kw["continuous"] = a.continuous
best, samples, first_ms, lists = 1e9, 0, 0.0, 0
for r in range(a.reps + 1):
    torch.manual_seed(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wavs, t_first, n_lists = [], None, 0
    for out in pipe.infer(list(texts), skip_refine_text=True, do_text_optimization=False, params_infer_code=params, noise=("device" if a.continuous else "auto"), noise_seed=11, **kw):
        wavs += [w.cpu() for w in out]              # (the copy waits for the vocoder: the waveforms are really there)
        n_lists += 1
        if t_first is None:
            t_first = time.perf_counter() - t0
    dt = time.perf_counter() - t0
    samples = sum(int(w.shape[0]) for w in wavs)
    if r and dt < best:
        best, first_ms, lists = dt, t_first * 1e3, n_lists
print(json.dumps({"texts": a.n, "tokens_each": a.tokens if not a.ragged else f"U{{{max(a.tokens // 4, 1)}..{a.tokens}}}", "dtype": a.dtype, "rows": a.rows or a.n,
                  "continuous": a.continuous, "wall_ms": round(best * 1e3, 1), "first_waveforms_ms": round(first_ms, 1), "lists": lists,
                  "audio_s": round(samples / 24000.0, 2), "rtf_x_realtime": round(samples / 24000.0 / best, 1)}))
