"""Host-inclusive wall clock of GPT.generate (what a pipeline user sees, as opposed to bench.py's device-side step
time): prompt 48, K new tokens, EOS disabled.  python tools/gen_wall.py [--batch B] [--steps K] [--noise torch|device]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models.gpt import GPT

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=512)
ap.add_argument("--noise", default="torch")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--text", action="store_true", help="refine-text pass: infer_text=True on the 21178-way head (top-p 0.7, top-k 20, T 0.7)")
a = ap.parse_args()
cfg = synth.GPT_REAL
g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=max(a.batch, 1),
        max_seq_len=48 + a.steps + 8, weight_dtype=a.dtype)
g.load_state_dict(synth.gpt_state_dict(cfg, 7))
ids = torch.from_numpy(synth.prompt_ids(a.batch, 48, cfg["num_text_tokens"], 3)[0]).cuda()
emb = g(ids, torch.ones(a.batch, 48, dtype=torch.bool, device="cuda"))
temp = torch.tensor([0.3] * 4)
lw = [type('P', (), dict(top_p=0.7, min_tokens_to_keep=3))(), type('K', (), dict(top_k=20))()]
lp = [type('R', (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
out = []
for r in range(a.reps + 1):
    torch.manual_seed(11)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if a.text:
        res = list(g.generate(emb, ids, torch.tensor([0.7]), eos_token=21177, max_new_token=a.steps, min_new_token=a.steps, noise=a.noise, seed=5,
                              logits_warpers=lw, logits_processors=[], infer_text=True, ensure_non_empty=False))
    else:
        res = list(g.generate(emb, ids, temp, eos_token=625, max_new_token=a.steps, min_new_token=a.steps, noise=a.noise, seed=5, logits_warpers=lw,
                              logits_processors=lp, return_hidden=True, ensure_non_empty=False))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if r:
        out.append(dt)
n = int(res[-1].ids[0].shape[0])
best = min(out)
print(json.dumps({k: round(v, 2) for k, v in g.host_timing.items()}))
print(json.dumps({"batch": a.batch, "dtype": a.dtype, "steps": n, "noise": a.noise, "wall_ms": round(best * 1e3, 2),
                  "us_per_step_wall": round(best / n * 1e6, 1), "tokens_per_s_wall": round(a.batch * n / best, 1)}))
