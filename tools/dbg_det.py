"""Round-3 bisect of the irreproducible fp16 prompt pass (profiles/README.md, "finding"): runs one prompt pass three times on one engine
and reports the first KV-cache / hidden element that differs between runs.  Developer script, not a test."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)
for wd in ("fp16", "fp32"):
    g = GPT(LLAMA, max_batch=32, max_seq_len=256, weight_dtype=wd)
    g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
    for (B, T, N) in ((32, 40, 3), (32, 40, 12), (4, 10, 12), (12, 10, 12), (1, 40, 12)):
        rng = np.random.Generator(np.random.Philox(key=4))
        pads = [int(p) for p in rng.integers(0, T - 5, size=B)]
        ids, mask = synth.prompt_ids(B, T, 21178, 79, pad_left=pads)
        q = torch.from_numpy(np.stack([synth.exp_noise(11, i, 4 * B, 626) for i in range(N)]))
        outs = []
        for rep in range(3):
            emb = g(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
            o = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                                min_new_token=N, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise=q))[-1]
            outs.append(torch.stack(o.hiddens).clone())
        d01 = (outs[0] - outs[1]).abs().amax(dim=(0, 2)).tolist()
        d12 = (outs[1] - outs[2]).abs().amax(dim=(0, 2)).tolist()
        print(wd, (B, T, N), "per-step max diff run0-run1:", [round(x, 6) for x in d01][:12], "run1-run2:", [round(x, 6) for x in d12][:6], flush=True)
    g.close()
