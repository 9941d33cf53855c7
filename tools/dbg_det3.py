"""dbg_det.py against library variants (CTTS_HIP_LIB): bisects which saturating store made the prompt pass irreproducible.  Developer script."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
NL = 20
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=NL)
MB, MS = 32, 256
g = GPT(LLAMA, max_batch=MB, max_seq_len=MS, weight_dtype="fp16")
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
B, T = 32, 10
ids, mask = synth.prompt_ids(B, T, 21178, 79)
q = torch.from_numpy(np.stack([synth.exp_noise(11, 0, 4 * B, 626)]))
kvs = []
for rep in range(3):
    g._kv.zero_()
    emb = g(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
    o = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=1,
                        min_new_token=1, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise=q))[-1]
    torch.cuda.synchronize()
    kv = g._kv.view(torch.float16).view(NL, 2, MB, 12, MS, 64)[:, :, :B, :, :T].float().clone()
    kvs.append((kv, torch.stack(o.hiddens).clone()))
for a, b in ((0, 1), (1, 2)):
    d = (kvs[a][0] - kvs[b][0]).abs()
    per = d.amax(dim=(2, 3, 4, 5))       # [layer][k/v]
    first = [(l, w) for l in range(NL) for w in range(2) if per[l, w] > 0][:4]
    print("runs", a, b, "hidden diff", float((kvs[a][1] - kvs[b][1]).abs().max()), "first differing (layer, k/v):", first)
    if first:
        l, w = first[0]
        dd = d[l, w]                       # [B][12][T][64]
        idx = (dd > 0).nonzero()
        print("   count", idx.shape[0], "examples (seq, head, slot, dim):", idx[:12].tolist())
        print("   values", [(float(kvs[a][0][l, w][tuple(i)]), float(kvs[b][0][l, w][tuple(i)])) for i in idx[:6].tolist()])
