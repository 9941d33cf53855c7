"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

from chatttsplus_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = {k[5:]: z[k] for k in z.files if k.startswith("meta_")}
    return z, meta


def gen_case_inputs(meta, cfg):
    """Regenerates weights / prompt / speaker of a generate() golden from its stored seeds."""
    sd = synth.gpt_state_dict(cfg, int(meta["weight_seed"]))
    if "eos_boost" in meta:
        for i in range(4):
            sd[f"head_code.{i}.parametrizations.weight.original0"][625] *= float(meta["eos_boost"])
    B, T = int(meta["B"]), int(meta["T"])
    pad = [int(x) for x in np.atleast_1d(meta["pad_left"])]
    ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], int(meta["prompt_seed"]), pad_left=pad)
    spk = None
    if "spk_pos" in meta and int(meta["spk_pos"]) >= 0:
        ids[:, int(meta["spk_pos"]), :] = int(meta["spk_id"])
        spk = synth.speaker_vector(int(meta["spk_seed"]))
    return sd, ids, mask, spk
