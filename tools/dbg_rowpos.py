"""Round-3 check while porting the packed-residual hand-off to fp32: does a row's result depend on its position inside a 16-row group?
Developer script, not a test."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)
g = GPT(LLAMA, max_batch=32, max_seq_len=256, weight_dtype="fp16")
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
B, T, N = 32, 40, 40
rng = np.random.Generator(np.random.Philox(key=4))
pads = [int(p) for p in rng.integers(0, 30, size=B)]
ids, mask = synth.prompt_ids(B, T, 21178, 79, pad_left=pads)
q = torch.from_numpy(np.stack([synth.exp_noise(11, i, 4 * B, 626) for i in range(N)]))
def gen(ids, mask, q, compact=True, chunk=8):
    g.compact = compact; g.compact_chunk = chunk
    emb = g(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
    return list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                           min_new_token=N, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise=q))[-1]
a = gen(ids, mask, q)
b = gen(ids, mask, q)
print("same order twice: max diff", max(float((x - y).abs().max()) for x, y in zip(a.hiddens, b.hiddens)))
c = gen(ids, mask, q, compact=False)
print("compact off vs on: max diff", max(float((x - y).abs().max()) for x, y in zip(a.hiddens, c.hiddens)))
d = gen(ids, mask, q, compact=True, chunk=32)
print("chunk 32 vs 8: max diff", max(float((x - y).abs().max()) for x, y in zip(a.hiddens, d.hiddens)))
for s in range(0, N, 4):
    print("  step", s, "chunk8 vs compact-off diff", max(float((x[s] - y[s]).abs().max()) for x, y in zip(a.hiddens, c.hiddens)))
perm = list(range(B - 1, -1, -1))
qp = q.view(N, B, 4, 626)[:, perm].reshape(N, 4 * B, 626).contiguous()
r = gen(ids[perm], mask[perm], qp, compact=False)
print("reversed vs original (compact off): max diff", max(float((r.hiddens[i] - c.hiddens[perm[i]]).abs().max()) for i in range(B)))
