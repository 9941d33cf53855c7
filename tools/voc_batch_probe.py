"""Batched DVAE-decoder + Vocos only (32 x 272 tokens, 5 calls): run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import Synth
pool = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=1100, max_batch=32)
pool.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234)); pool.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
hs = [torch.randn(272, 768, device="cuda") for _ in range(32)]
pool.decode_batch(hs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): pool.decode_batch(hs)
torch.cuda.synchronize()
print(f"32 x 272 batched: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
