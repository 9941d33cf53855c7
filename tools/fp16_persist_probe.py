"""fp16 engines on the persistent launch (round 5: half weights + half K / V in the per-workgroup image, fp32 activations) vs the fp16 launch chain and vs the
fp32 engine: token agreement, hidden-state error against the fp32 run, step time.  python tools/fp16_persist_probe.py"""
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

LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
dev = torch.device("cuda", 0)
sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
g16 = GPT(bench.LLAMA, max_batch=4, max_seq_len=48 + 16 + 512 + 16, weight_dtype="fp16", device=str(dev)); g16.load_state_dict(sd)
g32 = GPT(bench.LLAMA, max_batch=4, max_seq_len=48 + 16 + 512 + 16, weight_dtype="fp32", device=str(dev)); g32.load_state_dict(sd)
print(json.dumps({"fp16_persistent_rows": g16.get_option("persistent_rows"), "fp32_persistent_rows": g32.get_option("persistent_rows")}), flush=True)


def gen(g, B, P, N, forced=None):
    ids, mask = synth.prompt_ids(B, P, 21178, 4321, pad_left=[(5 * b) % 7 for b in range(B)])
    emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
    res = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                          logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=7))[-1]
    return res.ids, res.hiddens


for B in (1, 2, 3, 4):
    r_ids, r_h = gen(g32, B, 40, 48)
    out = {"B": B}
    for name, rows in (("fp16_launch_chain", 0), ("fp16_persistent", 4)):
        g16.set_option("persistent_rows", rows)
        ids, hid = gen(g16, B, 40, 48)
        agree = [int(((a == b).all(-1)).to(torch.int32).cumprod(0).sum()) for a, b in zip(ids, r_ids)]      # leading steps with the fp32 run's tokens
        e0 = max(float((a[0] - b[0]).pow(2).mean().sqrt() / b[0].pow(2).mean().sqrt()) for a, b in zip(hid, r_h))
        out[name] = {"steps_agreeing_with_fp32_of_48": agree, "first_step_hidden_rel_rms_vs_fp32": round(e0, 6)}
    print(json.dumps(out), flush=True)
spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
leg = bench.Leg(g16, dev, 0, 1)
for B in (1, 2, 3, 4):
    out = {"B": B}
    for name, rows in (("fp16_launch_chain", 0), ("fp16_persistent", 4)):
        g16.set_option("persistent_rows", rows)
        r = leg.run(B, 48, 64, 8, spk=spk)
        out[name + "_ms_per_step"] = bench.summarize(r, 1)["step_ms_hip_events"]
    r = bench.Leg(g32, dev, 0, 1).run(B, 48, 64, 8, spk=spk)
    out["fp32_persistent_ms_per_step"] = bench.summarize(r, 1)["step_ms_hip_events"]
    print(json.dumps(out), flush=True)
