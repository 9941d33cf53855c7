"""Wall clock of the three ways of serving 128 ragged utterances on 32 decode rows (bench.queue_leg) for one dtype.
usage: python tools/queue_probe.py [fp32|fp16] [utterances] [rows]"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chatttsplus_amd import synth
from chatttsplus_amd.hip_models import GPT

wd = sys.argv[1] if len(sys.argv) > 1 else "fp32"
NU = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
g = GPT(bench.LLAMA, max_batch=max(32, rows), max_seq_len=640, weight_dtype=wd, device=str(dev))
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
import numpy as np
spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
if len(sys.argv) > 4 and sys.argv[4] == "sweep":
    for ch in (4, 8, 16):
        for am in (1, 2, 4, 8):
            g.compact_chunk = ch
            r = bench.queue_leg(g, dev, spk, 0, NU=NU, rows=rows, admit_min=am, modes=("continuous",))
            print(json.dumps({"dtype": wd, "chunk": ch, "admit_min": am, **r["continuous"]}), flush=True)
else:
    r = bench.queue_leg(g, dev, spk, 0, NU=NU, rows=rows)
    r["dtype"] = wd
    print(json.dumps(r))
