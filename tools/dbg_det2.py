"""dbg_det.py over several (batch, prompt length) shapes: which shapes reproduce the run-to-run difference.  Developer script."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=int(os.environ.get("NL", "20")))
g = GPT(LLAMA, max_batch=32, max_seq_len=256, weight_dtype="fp16")
cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = LLAMA["num_hidden_layers"]
g.load_state_dict(synth.gpt_state_dict(cfg, 1234))
for (B, T, zero_pad) in ((32, 40, False), (32, 40, True), (32, 10, True), (12, 40, True), (17, 40, True), (16, 40, True), (32, 2, True), (32, 3, True), (20, 4, True)):
    rng = np.random.Generator(np.random.Philox(key=4))
    pads = [0] * B if zero_pad else [int(p) for p in rng.integers(0, T - 5, size=B)]
    ids, mask = synth.prompt_ids(B, T, 21178, 79, pad_left=pads)
    N = 1
    q = torch.from_numpy(np.stack([synth.exp_noise(11, i, 4 * B, 626) for i in range(N)]))
    outs = []
    for rep in range(3):
        emb = g(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
        o = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                            min_new_token=N, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise=q))[-1]
        outs.append(torch.stack(o.hiddens).clone())
    d = (outs[0] - outs[1]).abs().amax(dim=(1, 2))
    print((B, T, zero_pad), "rows*T", B * T, "rows that differ:", [i for i in range(B) if d[i] > 0], "max", float(d.max()), flush=True)
