"""Small decode batches at long contexts: persistent launch vs launch chain (round 5: the key shares' tails stream 4 steps per round trip; no context limit any more).
python tools/long_ctx_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chatttsplus_amd import synth  # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT  # noqa: E402

dev = torch.device("cuda", 0)
g = GPT(bench.LLAMA, max_batch=4, max_seq_len=2048 + 64, weight_dtype="fp32", device=str(dev))
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
leg = bench.Leg(g, dev, 0, 1)
for B, P in ((1, 1900), (2, 1200), (2, 1900), (3, 700), (3, 1200), (3, 1900), (4, 700), (4, 1200), (4, 1900)):
    out = {"B": B, "prompt": P}
    for name, rows in (("launch_chain", 0), ("persistent", 4)):
        g.set_option("persistent_rows", rows)
        r = leg.run(B, P, 32, 8, spk=spk, gen_tokens=0)
        out[name + "_ms_per_step"] = bench.summarize(r, 1)["step_ms_hip_events"]
    print(json.dumps(out), flush=True)
