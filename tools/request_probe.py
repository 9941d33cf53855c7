"""Where the wall clock of the 256-utterance sharded request goes (bench.sharded_request_leg's workload on one GPU): decode steps launched against the ideal
count, admissions, compactions, and the GPT / vocoder / host shares.  usage: python tools/request_probe.py [rows] [utterances] [compact_chunk] [admit_min]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT, Synth
from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NU = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = GPT(bench.LLAMA, max_batch=rows, max_seq_len=48 + 96 + 512 + 32, weight_dtype="fp32", device=str(dev))
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
if len(sys.argv) > 3:
    g.compact_chunk = int(sys.argv[3])
FIFO = os.environ.get("CTTS_SCHEDULE") == "fifo"           # arrival order instead of longest-first
syn = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2 * 512 + 64, device=str(dev), max_batch=32)
syn.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234)); syn.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
texts, limits, spk_index = bench._request_256(NU)
table = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)]))
params = InferCodeParams(prompt="[speed_5]", max_new_token=512, min_new_token=512, show_tqdm=False)
kw = {}
if len(sys.argv) > 4:
    kw["admit_min"] = int(sys.argv[4])
with tempfile.TemporaryDirectory() as td:
    pipe = ChatTTSPlusPipeline.from_components(g, syn, synth.toy_tokenizer(td), dev)
    if FIFO:
        pipe.throughput_order = "input"
    shares = {"vocoder_s": 0.0}
    orig = pipe._decode_to_wavs
    def timed(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = orig(*a, **k); torch.cuda.synchronize(); shares["vocoder_s"] += time.perf_counter() - t; return r
    pipe._decode_to_wavs = timed
    for rep in range(3):
        shares["vocoder_s"] = 0.0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mine, wavs, lens = pipe.infer_sharded(list(texts), speaker_index=spk_index, speaker_table=table, params_infer_code=params, noise_seed=4242, slice_size=rows,
                                              continuous=True, max_new_tokens_per_utterance=limits, **kw)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        useful = sum(limits)
        launched = g.admissions[-1][0] if g.admissions else 0
        last = max([x[0] for x in g.admissions] + [x[0] for x in g.compactions] + [0])
        print(json.dumps({"rep": rep, "rows": rows, "utterances": NU, "wall_s": round(wall, 4), "useful_tokens": useful, "useful_tok_s": round(useful / wall, 1), "ideal_steps": useful // rows,
                          "admissions": len(g.admissions), "admitted_rows": sum(k for _, k in g.admissions), "last_admission_at_step": launched, "compactions": g.compactions[-6:], "last_event_step": last,
                          "vocoder_s": round(shares["vocoder_s"], 4), "chunk": min(g.chunk_steps, g.compact_chunk)}), flush=True)
