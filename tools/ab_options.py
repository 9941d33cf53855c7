"""A/B of engine options (ctts_gpt_set_option) on ONE engine, interleaved rounds (cdna_hip_programming.md 5.4 rule 24): decode step time in the
bench window for each setting, median and min over the rounds.
usage: python tools/ab_options.py fp32 "valu_rows=0,4" --batches 1 2 4 [--rounds 3] [--steps 64]"""
import argparse
import json
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chatttsplus_amd import synth  # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dtype")
ap.add_argument("sweep", help="option=v0,v1,...")
ap.add_argument("--batches", type=int, nargs="+", default=[1])
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--fixed", default="", help="other options held fixed: a=1,b=2")
ap.add_argument("--prompt", type=int, default=48)
ap.add_argument("--gen-tokens", type=int, default=512)
ap.add_argument("--adapters", action="store_true", help="every row carries one of two rank-8 LoRA adapters")
a = ap.parse_args()
name, vals = a.sweep.split("=")
vals = [int(v) for v in vals.split(",")]
dev = torch.device("cuda", 0)
g = GPT(bench.LLAMA, max_batch=max(a.batches), max_seq_len=a.prompt + 16 + 512 + 16, weight_dtype=a.dtype, device=str(dev))
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
for kv in [x for x in a.fixed.split(",") if x]:
    k, v = kv.split("=")
    g.set_option(k, int(v))
if a.adapters:
    rl = np.random.Generator(np.random.Philox(key=31))
    for slot in range(2):
        g.load_adapter(slot, [(l, t, (rl.standard_normal((8, 768)) * 0.02).astype(np.float32), (rl.standard_normal((768, 8)) * 0.02).astype(np.float32), 2.0)
                              for l in range(20) for t in ("q_proj", "k_proj", "v_proj", "o_proj")])
spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
leg = bench.Leg(g, dev, 0, 1)
for B in a.batches:
    times = {v: [] for v in vals}
    if a.adapters:
        g.set_row_adapters([b % 2 for b in range(B)])
    for rnd in range(a.rounds + 1):
        for v in vals:
            g.set_option(name, v)
            r = leg.run(B, a.prompt, a.steps, 8, spk=spk, gen_tokens=a.gen_tokens)
            if rnd:                      # round 0 captures the graphs
                times[v].append(r["ev_ms"] / r["K"])
    print(json.dumps({"dtype": a.dtype, "B": B, "prompt": a.prompt, "gen_tokens": a.gen_tokens, "option": name, "fixed": a.fixed,
                      "ms_per_step": {str(v): {"median": round(statistics.median(t), 5), "min": round(min(t), 5)} for v, t in times.items()}}), flush=True)
