"""Decode step time vs batch size on one engine (same window as bench.py: steps around the middle of a 512-token generation).
usage: python tools/tb_curve.py fp32|fp16 [B ...]"""
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

wd = sys.argv[1] if len(sys.argv) > 1 else "fp32"
Bs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 5, 8, 12, 16, 20, 24, 28, 32]
dev = torch.device("cuda", 0)
g = GPT(bench.LLAMA, max_batch=max(Bs), max_seq_len=48 + 16 + 512 + 16, weight_dtype=wd, device=str(dev))
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
leg = bench.Leg(g, dev, 0, 1)
for B in Bs:
    r = leg.run(B, 48, 64, 8, spk=spk)
    s = bench.summarize(r, 1)
    print(json.dumps({"dtype": wd, "B": B, "ms_per_step": s["step_ms_hip_events"], "tokens_per_s": s["tokens_per_s"], "frac": s["frac_of_8TBps"]}), flush=True)
