"""Times the DVAE-decoder + Vocos chain for several utterance lengths (single stream) and a 32-utterance batch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import Synth

def main():
    pool = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=4200, max_batch=32)
    pool.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234)); pool.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
    s0 = pool
    for n in (64, 272, 528, 1024, 2048):
        h = torch.randn(n, 768, device="cuda")
        for _ in range(2): s0.vocos_decode(s0.dvae_decode(h))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): mel = s0.dvae_decode(h)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(5): s0.vocos_decode(mel)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"n={n:5d}: dvae {(t1 - t0) / 5 * 1e3:7.3f} ms  vocos {(t2 - t1) / 5 * 1e3:7.3f} ms", flush=True)
    hs = [torch.randn(272, 768, device="cuda") for _ in range(32)]
    pool.decode_batch(hs); torch.cuda.synchronize(); t0 = time.perf_counter()
    pool.decode_batch(hs); torch.cuda.synchronize()
    print(f"32 x 272 batched (synth_batch): {(time.perf_counter() - t0) * 1e3:.2f} ms")
    t0 = time.perf_counter()
    for h in hs: s0.vocos_decode(s0.dvae_decode(h))
    torch.cuda.synchronize()
    print(f"32 x 272 on 1 stream : {(time.perf_counter() - t0) * 1e3:.2f} ms")

if __name__ == "__main__":
    main()
