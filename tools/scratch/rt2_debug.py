import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)
g = GPT(LLAMA, max_batch=40, max_seq_len=400, weight_dtype="fp32")
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
def gen(B=24, P=40, N=16):
    ids, mask = synth.prompt_ids(B, P, 21178, 4321, pad_left=[(7 * i) % 36 for i in range(B)])
    emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
    res = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                          logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=7))[-1]
    return torch.stack(res.hiddens)
def d(a, b): return float((a - b).abs().max()), [float((a[:, t] - b[:, t]).abs().max()) for t in range(4)]
g.set_option("weight_prefetch_kb", 0); g.set_option("split_nbg2_rows", 99)
base = gen(); print("repeat", d(base, gen()))
g.set_option("weight_prefetch_kb", 96); print("pf96", d(base, gen())); g.set_option("weight_prefetch_kb", 0)
g.set_option("split_row_tiles", 2); r2 = gen(); print("rt2", d(base, r2)); print("rt2 repeat", d(r2, gen())); g.set_option("split_row_tiles", 1)
g.set_option("split_nbg2_rows", 17); n2 = gen(); print("nbg2", d(base, n2)); print("nbg2 repeat", d(n2, gen()))
g.set_option("split_row_tiles", 2); print("nbg2+rt2 vs nbg2", d(n2, gen()))
