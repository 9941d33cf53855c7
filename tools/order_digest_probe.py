"""Does an utterance keep its tokens whatever order / batch it is served in?  The 256-utterance request of bench.sharded_request_leg on one GPU: longest-first (default)
vs arrival order, 32 vs 16 decode rows (different batch sizes = different kernels along the way: persistent launch / 16-row groups / 32-row blocks) -- the `ids_digest`
must be the same (the noise of an utterance is keyed by its id; rows are independent).  usage: python tools/order_digest_probe.py"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT, Synth
from chatttsplus_amd.pipeline import ChatTTSPlusPipeline

dev = torch.device("cuda:0")
g = GPT(bench.LLAMA, max_batch=32, max_seq_len=48 + 96 + 512 + 32, weight_dtype="fp32", device=str(dev))
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
syn = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2 * 512 + 64, device=str(dev), max_batch=32)
syn.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234)); syn.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
from chatttsplus_amd.pipeline import InferCodeParams
import numpy as np
texts, limits, spk_index = bench._request_256(256)
table = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)]))
params = InferCodeParams(prompt="[speed_5]", max_new_token=512, min_new_token=512, show_tqdm=False)
with tempfile.TemporaryDirectory() as td:
    pipe = ChatTTSPlusPipeline.from_components(g, syn, synth.toy_tokenizer(td), dev)
    runs = {}
    for name, order, rows in (("longest_first_32_rows", "longest_first", 32), ("arrival_order_32_rows", "input", 32), ("longest_first_16_rows", "longest_first", 16), ("longest_first_8_rows", "longest_first", 8)):
        pipe.throughput_order = order
        ids = []
        mine, wavs, lens = pipe.infer_sharded(list(texts), speaker_index=spk_index, speaker_table=table, params_infer_code=params, noise_seed=4242, slice_size=rows,
                                              continuous=True, max_new_tokens_per_utterance=limits, ids_out=ids)
        runs[name] = [t.cpu() for t in ids]
    ref = runs["longest_first_32_rows"]
    for name, ids in runs.items():
        same = [bool(torch.equal(a, b)) for a, b in zip(ref, ids)]
        first = [int((a != b).any(-1).nonzero()[0]) if not s_ else -1 for a, b, s_ in zip(ref, ids, same)]
        div = sorted(f for f in first if f >= 0)
        print(json.dumps({"run": name, "utterances": len(ids), "identical_to_longest_first_32_rows": sum(same), "first_differing_step_of_the_others": div[:12],
                          "tokens_before_divergence_total": int(sum(limits[i] if s_ else first[i] for i, s_ in enumerate(same))), "tokens_total": int(sum(limits))}), flush=True)
