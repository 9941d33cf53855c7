"""Batched DVAE-decoder + Vocos only (for profiling): 32 utterances x 272 tokens, 3 calls."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import Synth
B, n = 32, 272
s = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2 * n + 64, max_batch=B)
s.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234)); s.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
rng = np.random.Generator(np.random.Philox(key=1))
hs = [torch.from_numpy(rng.standard_normal((n, 768)).astype(np.float32)).cuda() for _ in range(B)]
for _ in range(3):
    w = s.decode_batch(hs)
torch.cuda.synchronize()
print("ok", w[0].shape)
