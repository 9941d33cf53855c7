"""Per-kernel-class cost inside the decode graph without a profiler: run the step with only one class of layer
kernel enabled (CTTS_ABLATE mask) and subtract the heads+sampler-only step.  Results are numerically garbage, timing only."""
import os
import subprocess
import sys

B = sys.argv[1] if len(sys.argv) > 1 else "1"
names = ["qkv", "attn", "o_proj", "gate|up", "down"]
res = {}
for mask, label in [(31, "heads+sampler only")] + [(31 ^ (1 << i), names[i]) for i in range(5)] + [(0, "all")]:
    env = dict(os.environ, CTTS_ABLATE=str(mask), CTTS_PROBE_B=B)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gpu_probe.py"), "fp16"], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("fp16 B=")][0]
    us = float(line.split("step")[1].split("us")[0])
    res[label] = us
    print(f"{label:20s} {us:8.1f} us/step", flush=True)
base = res["heads+sampler only"]
for n in names:
    print(f"  {n:10s}: {(res[n] - base) / 20:6.2f} us per launch (incl. its 1.6 us boundary)")
print(f"  sum of parts {base + sum(res[n] - base for n in names):.1f} vs all {res['all']:.1f}")
