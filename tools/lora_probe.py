"""Per-utterance LoRA at batch 32: decode step time with the low-rank terms inside the QKV / o_proj launches (lora_fold 1), as two more launches per layer
(0), inside the persistent launch (mode "persist", <= 8 rows), and without adapters.  python tools/lora_probe.py [--rows 32] [--tokens 384] [--dtype fp32] [--modes none,fold,launch,persist]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT

ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=32); ap.add_argument("--tokens", type=int, default=384)
ap.add_argument("--dtype", default="fp32"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--modes", default="none,fold,launch")
a = ap.parse_args()
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)
g = GPT(LLAMA, max_batch=a.rows, max_seq_len=48 + a.tokens + 32, weight_dtype=a.dtype)
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
rl = np.random.Generator(np.random.Philox(key=31))
for slot in range(4):
    g.load_adapter(slot, [(l, t, (rl.standard_normal((8, 768)) * 0.02).astype(np.float32), (rl.standard_normal((768, 8)) * 0.02).astype(np.float32), 2.0)
                          for l in range(20) for t in ("q_proj", "k_proj", "v_proj", "o_proj")])
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
B = a.rows
ids, mask = synth.prompt_ids(B, 48, 21178, 4321)


def run(n):
    emb = g(torch.from_numpy(ids), torch.ones(B, 48, dtype=torch.bool))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=n, min_new_token=n,
                    logits_warpers=LW, logits_processors=LP, return_hidden=False, noise="device", seed=7))
    torch.cuda.synchronize(); return time.perf_counter() - t0


for mode in a.modes.split(","):
    g.set_row_adapters(None if mode == "none" else [(b % 5) - 1 for b in range(B)])
    g.set_option("lora_fold", {"launch": 0, "fold": 1, "notake": 2, "zeros": 3}.get(mode, 1))
    g.set_option("persistent_lora", 1 if mode in ("persist", "none") else 0)      # "persist": rows with adapters stay on the persistent launch (<= 8 rows, round 6)
    run(32)
    best = min((run(a.tokens) - run(a.tokens // 4)) / (a.tokens - a.tokens // 4) for _ in range(a.reps))
    print(json.dumps({"mode": mode, "rows": B, "dtype": a.dtype, "ms_per_step": round(best * 1e3, 5)}), flush=True)
g.set_row_adapters(None)
g.close()
