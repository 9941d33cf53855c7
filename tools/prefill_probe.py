"""Prompt-pass timing probe: python tools/prefill_probe.py B P [dtype] [option=value ...] -> ms per prompt pass (begin + prefill + first sample), 3 runs
per option set ("a=1,b=2" is one set; several sets are timed interleaved on ONE engine, two rounds).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import _lib, synth                                   # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT, sampler_cfg_from_objects    # noqa: E402

B, P = int(sys.argv[1]), int(sys.argv[2])
wd = sys.argv[3] if len(sys.argv) > 3 else "fp16"
init = dict((kv.split("=")[0], int(kv.split("=")[1])) for a in sys.argv[4:] if a.startswith("init:") for kv in a[5:].split(","))      # "init:name=value": options set before the weights load
optsets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[4:] if not a.startswith("init:")] or [{}]
dev = torch.device("cuda", 0)
g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=B, max_seq_len=P + 40, weight_dtype=wd, device="cuda:0", options=init or None)
g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
rngpad = [(7 * b) % max(1, P // 3) for b in range(B)]
ids, mask = synth.prompt_ids(B, P, 21178, 5, pad_left=rngpad)
emb = g(torch.from_numpy(ids).to(dev), torch.ones(B, P, dtype=torch.bool, device=dev))
lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, 16, 16, lw, lp, 4)
out_ids = torch.zeros(B, 16, 4, dtype=torch.int32, device=dev)
fin = torch.zeros(B, dtype=torch.int32, device=dev); end = torch.zeros(B, dtype=torch.int32, device=dev)
io = _lib.GenIO(ids=out_ids.data_ptr(), hiddens=None, finish=fin.data_ptr(), end_idx=end.data_ptr(), noise=None, n_draws=0, seed=1)
msk = torch.from_numpy(mask).to(dev).to(torch.int32)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for rnd in range(2 if len(optsets) > 1 else 1):
    for opts in optsets:
        for k, v in opts.items():
            g.set_option(k, v)
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(g._lib.ctts_gpt_begin(g._h, B, P, msk.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
            _lib.check(g._lib.ctts_gpt_prefill(g._h, emb.data_ptr(), st), "prefill")
            _lib.check(g._lib.ctts_gpt_sample(g._h, st), "sample")
            torch.cuda.synchronize()
            print(f"B={B} P={P} {wd} {dict(init, **opts) if (opts or init) else ''}: prompt pass {1e3 * (time.perf_counter() - t0):.3f} ms  first ids {out_ids[0, 0].tolist()} {out_ids[B - 1, 0].tolist()}", flush=True)
