"""Free-running token agreement of the fp16 mode with the fp32 goldens minted from the reference (same torch seed, noise="torch").
Writes one JSON object per golden: length of the common prefix and the fraction of equal tokens per sequence.
    python tools/fp16_agreement.py > profiles/r02_fp16_token_agreement.json     (GPU box)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth                      # noqa: E402
from chatttsplus_amd.hip_models import GPT             # noqa: E402
from tests.helpers import gen_case_inputs, load_golden  # noqa: E402

LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)

out = {}
for name in ("gpt_real_b1", "gpt_real_b2_pad", "gpt_real_greedy", "gpt_real_b4_ragged", "gpt_real_regen"):
    z, meta = load_golden(name)
    sd, ids, mask, spk = gen_case_inputs(meta, synth.GPT_REAL)
    res = {}
    for wd in ("fp32", "fp16"):
        g = GPT(LLAMA, max_batch=4, max_seq_len=128, weight_dtype=wd)
        g.load_state_dict(sd)
        ids_t = torch.from_numpy(ids)
        emb = g(ids_t, torch.ones(ids.shape[:2], dtype=torch.bool), spk_emb=torch.from_numpy(spk) if spk is not None else None,
                spk_emb_ids=int(meta["spk_id"]) if spk is not None else None)
        temp = float(meta["temperature"]) if "temperature" in meta else 0.3
        torch.manual_seed(int(meta["torch_seed"]))
        o = list(g.generate(emb, ids_t, torch.tensor([temp] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=int(meta["max_new"]),
                            min_new_token=int(meta["min_new"]), logits_warpers=LW, logits_processors=LP, noise="torch"))[-1]
        rows = []
        for b, n in enumerate(z["lens"]):
            got = o.ids[b].cpu().numpy()
            ref = z["ids"][b, :n].astype(np.int64)
            m = min(len(got), len(ref))
            eq = (got[:m] == ref[:m]).all(1)
            prefix = int(np.argmin(eq)) if not eq.all() else m
            rows.append(dict(ref_len=int(n), got_len=int(len(got)), common_prefix_steps=prefix, equal_token_fraction=round(float((got[:m] == ref[:m]).mean()), 4)))
        res[wd] = rows
        g.close()
    out[name] = res
print(json.dumps(out, indent=1))
