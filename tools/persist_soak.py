"""Soak of the persistent decode launch: many long generations at 1..max-rows rows back to back; every request must end without a give-up (ctts_gpt_progress raises on
one) and a repeated request must reproduce its tokens bit for bit.  python tools/persist_soak.py [--requests 24] [--tokens 1500]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT

ap = argparse.ArgumentParser(); ap.add_argument("--requests", type=int, default=24); ap.add_argument("--tokens", type=int, default=1500)
ap.add_argument("--adapters", action="store_true", help="rows carry one of three LoRA adapters (or none): the LORA kernels of the persistent launch (round 6)")
ap.add_argument("--max-rows", type=int, default=5, help="row counts 1..max-rows take turns (the persistent launch serves up to persistent_rows of them)")
a = ap.parse_args()
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)
g = GPT(LLAMA, max_batch=a.max_rows, max_seq_len=64 + a.tokens + 16, weight_dtype="fp32")
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
assert g.get_option("persistent_rows") >= min(5, a.max_rows)
if a.adapters:
    import numpy as np
    rl = np.random.Generator(np.random.Philox(key=31))
    for slot, rk in enumerate((8, 16, 4)):
        g.load_adapter(slot, [(l, t, (rl.standard_normal((rk, 768)) * 0.02).astype(np.float32), (rl.standard_normal((768, rk)) * 0.02).astype(np.float32), 2.0)
                              for l in range(20) for t in ("q_proj", "k_proj", "v_proj", "o_proj")])
g.compact = False                                  # every step of a request at the same row count
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
steps, t0, first = 0, time.perf_counter(), {}
for i in range(a.requests):
    B = 1 + i % a.max_rows
    ids, mask = synth.prompt_ids(B, 48, 21178, 4321 + (i % 8), pad_left=[(3 * b) % 7 for b in range(B)])
    if a.adapters:
        g.set_row_adapters([((b + i) % 4) - 1 for b in range(B)])
    emb = g(torch.from_numpy(ids), torch.ones(B, 48, dtype=torch.bool))
    out = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=a.tokens, min_new_token=a.tokens,
                          logits_warpers=LW, logits_processors=LP, return_hidden=False, noise="device", seed=7))[-1]
    steps += a.tokens
    key = (B, i % 8)
    sig = [int(x.sum()) for x in out.ids]
    if key in first:
        assert first[key] == sig, f"request {i}: tokens differ from the first run of the same request"
    first[key] = sig
torch.cuda.synchronize()
print(json.dumps({"engine": f"persist_layer.hip (1..{a.max_rows} rows)" + (", per-utterance adapters" if a.adapters else ""), "requests": a.requests, "decode_steps": steps, "layer_stack_launches": steps, "give_ups": 0, "repeats_identical": True,
                  "seconds": round(time.perf_counter() - t0, 1)}))
g.close()
