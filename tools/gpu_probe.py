"""Perf probe: per-step time of the captured decode graph with real work vs. with every kernel exiting at its
first instruction (all_done set) -- the latter is the launch/boundary floor of the 102-kernel step."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import _lib, synth  # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT, sampler_cfg_from_objects  # noqa: E402


def main():
    wd = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=32, max_seq_len=1024, weight_dtype=wd)
    g.load_state_dict(sd)
    lib, h, dev = g._lib, g._h, g.device
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    Bs = [int(os.environ['CTTS_PROBE_B'])] if os.environ.get('CTTS_PROBE_B') else (1, 2, 4, 8, 16, 32)
    for B in Bs:
        P, N = 48, 300
        ids, mask = synth.prompt_ids(B, P, 21178, 1)
        emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
        sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, N, N, lw, lp, 4)
        out_ids = torch.zeros(B, N, 4, dtype=torch.int32, device=dev); hid = torch.zeros(B, N, 768, device=dev)
        fin = torch.zeros(B, dtype=torch.int32, device=dev); end = torch.zeros(B, dtype=torch.int32, device=dev)
        io = _lib.GenIO(ids=out_ids.data_ptr(), hiddens=hid.data_ptr(), finish=fin.data_ptr(), end_idx=end.data_ptr(), noise=None, n_draws=0, seed=1)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        msk = torch.from_numpy(mask).to(dev).to(torch.int32)
        _lib.check(lib.ctts_gpt_begin(h, B, P, msk.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
        _lib.check(lib.ctts_gpt_prefill(h, emb.data_ptr(), st), "prefill")
        _lib.check(lib.ctts_gpt_sample(h, st), "sample")
        _lib.check(lib.ctts_gpt_decode(h, 20, 1, st), "warm")
        ms = C.c_float(0)
        _lib.check(lib.ctts_gpt_time_decode(h, 200, C.byref(ms), st), "time")
        work = ms.value
        _lib.check(lib.ctts_gpt_decode(h, 100, 1, st), "finish")          # runs past max_new -> all_done
        steps, alld = C.c_int32(0), C.c_int32(0)
        _lib.check(lib.ctts_gpt_progress(h, C.byref(steps), C.byref(alld), st), "progress")
        _lib.check(lib.ctts_gpt_time_decode(h, 200, C.byref(ms), st), "time")
        print(f"{wd} B={B:2d}: step {work * 1e3:8.1f} us ({B / work * 1e3:9.0f} tok/s)  empty-graph floor {ms.value * 1e3:7.1f} us  (all_done={alld.value}, steps={steps.value})", flush=True)


if __name__ == "__main__":
    main()
