"""Experiment: do independent decode chains (disjoint utterance subsets, one hipGraph chain per HIP stream) overlap on one MI355X?
A decode step is ~100 dependent launches that each leave most of the chip idle; utterances are independent, so a batch can be
split into C chains that the command processor runs concurrently from C hardware queues.

    python tools/mb/multichain.py            # prints us/step for (chains x rows): 1x32, 2x16, 4x8, 1x16, 1x8 ...
"""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chatttsplus_amd import _lib, synth                                   # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT, sampler_cfg_from_objects    # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = synth.GPT_REAL
sd = synth.gpt_state_dict(cfg, 1234)
P, W, K = 48, 16, 256


def make(B, seed):
    g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=B,
            max_seq_len=P + W + K + 8, weight_dtype="fp16", device=str(dev))
    g.load_state_dict(sd)
    ids, mask = synth.prompt_ids(B, P, cfg["num_text_tokens"], seed)
    emb = g(torch.from_numpy(ids).to(dev), torch.ones(B, P, dtype=torch.bool, device=dev))
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, W + K, W + K, lw, lp, 4)
    keep = dict(ids=torch.zeros(B, W + K, 4, dtype=torch.int32, device=dev), fin=torch.zeros(B, dtype=torch.int32, device=dev),
                end=torch.zeros(B, dtype=torch.int32, device=dev), msk=torch.from_numpy(mask).to(dev).to(torch.int32), emb=emb, sc=sc)
    io = _lib.GenIO(ids=keep["ids"].data_ptr(), hiddens=None, finish=keep["fin"].data_ptr(), end_idx=keep["end"].data_ptr(), noise=None,
                    n_draws=0, seed=seed)
    keep["io"] = io
    return g, keep


def run(chains, rows):
    gs = [make(rows, 1234 + i) for i in range(chains)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(chains)]
    lib = gs[0][0]._lib
    for (g, k), s in zip(gs, streams):
        st = C.c_void_p(s.cuda_stream)
        _lib.check(lib.ctts_gpt_begin(g._h, rows, P, k["msk"].data_ptr(), C.byref(k["sc"]), C.byref(k["io"]), st), "begin")
        _lib.check(lib.ctts_gpt_prefill(g._h, k["emb"].data_ptr(), st), "prefill")
        _lib.check(lib.ctts_gpt_sample(g._h, st), "sample")
        _lib.check(lib.ctts_gpt_decode(g._h, W - 1, 1, st), "warm")
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    # interleave the graph launches of the chains in small slices so no queue runs dry while the host feeds another
    for i in range(0, K, 32):
        for (g, k), s in zip(gs, streams):
            _lib.check(lib.ctts_gpt_decode(g._h, 32, 1, C.c_void_p(s.cuda_stream)), "decode")
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    for g, k in gs:
        assert int(k["end"].min().item()) == W + K, int(k["end"].min().item())
    us = dt / K * 1e6
    print(json.dumps(dict(chains=chains, rows_per_chain=rows, batch=chains * rows, us_per_step=round(us, 1),
                          tok_s=round(chains * rows * K / dt, 0))), flush=True)
    del gs


if __name__ == "__main__":
    for c, r in [(1, 32), (2, 16), (4, 8), (8, 4), (1, 16), (1, 8), (2, 32), (4, 16), (4, 32), (2, 1), (4, 1), (1, 1)]:
        run(c, r)
