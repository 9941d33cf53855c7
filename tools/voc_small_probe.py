"""DVAE-decoder + Vocos on small batches (B utterances x T tokens, 10 calls): python tools/voc_small_probe.py B T  -- run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import Synth
B, T = int(sys.argv[1]), int(sys.argv[2])
pool = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2 * T + 64, max_batch=max(B, 1))
pool.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234)); pool.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
hs = [torch.randn(T, 768, device="cuda") for _ in range(B)]
pool.decode_batch(hs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): pool.decode_batch(hs)
torch.cuda.synchronize()
print(f"{B} x {T} tokens: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call")
