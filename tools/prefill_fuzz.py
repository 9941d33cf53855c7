"""Fuzz of the prompt pass's block shapes (prefill_split.hip sp_launch): random (B, P, paddings); the forced shapes (128 x 128 on both rings, 64 x 64, counter-phased NT 4 / 3) must
agree bit for bit with K slicing off, the default policy (K slicing on) within 2e-5 with equal tokens.  python tools/prefill_fuzz.py [n_cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT

LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=40, max_seq_len=420, weight_dtype="fp32")
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
base = {k: g.get_option(k) for k in ("prefill_pp_blocks", "prefill_splitk_rows", "prefill_ring4_blocks", "prefill_small_blocks")}


def gen(B, P, pad):
    ids, mask = synth.prompt_ids(B, P, 21178, 4321, pad_left=pad)
    emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
    r = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=2, min_new_token=2,
                        logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=5))[-1]
    return r.ids, r.hiddens


cases = [(1, 65), (1, 66), (2, 33), (1, 129), (16, 128), (17, 128), (5, 410 - 20), (40, 52), (33, 63)]
while len(cases) < n_cases:
    B = int(rng.integers(1, 41)); P = int(rng.integers(max(2, 66 // B + 1), 400))
    if B * P > 16000: continue
    cases.append((B, P))
bad = 0
for B, P in cases[:n_cases]:
    pad = [int(x) for x in rng.integers(0, max(1, P // 3), size=B)] if rng.random() < 0.7 else None
    try:
        for k, v in dict(prefill_splitk_rows=0, prefill_pp_blocks=0, prefill_ring4_blocks=0, prefill_small_blocks=0).items(): g.set_option(k, v)
        ref_ids, ref_h = gen(B, P, pad)
        worst = {}
        for name, opts in (("ring4", dict(prefill_ring4_blocks=1 << 20)), ("64x64", dict(prefill_small_blocks=1 << 20)), ("pp4", dict(prefill_pp_blocks=-4)), ("pp3", dict(prefill_pp_blocks=-3)),
                           ("default", base)):
            for k, v in dict(prefill_splitk_rows=0, prefill_pp_blocks=0, prefill_ring4_blocks=0, prefill_small_blocks=0).items(): g.set_option(k, v)
            for k, v in opts.items(): g.set_option(k, v)
            ids, hid = gen(B, P, pad)
            d = max(float((hid[b] - ref_h[b]).abs().max()) for b in range(B))
            same = all(torch.equal(ids[b], ref_ids[b]) for b in range(B))
            worst[name] = d
            ok = same and (d <= 2e-5 if name == "default" else d == 0.0)
            if not ok:
                bad += 1
                print(f"MISMATCH B={B} P={P} rows={B * P} {name}: ids equal {same}, max hidden diff {d:.3e}", flush=True)
        print(f"B={B:2d} P={P:3d} rows={B * P:5d} pad={'y' if pad else 'n'}  " + " ".join(f"{k}={v:.1e}" for k, v in worst.items()), flush=True)
    finally:
        for k, v in base.items(): g.set_option(k, v)
print("FUZZ", "FAILED" if bad else "OK", f"({len(cases[:n_cases])} cases)")
g.close()
