"""Feasibility probe (round 6): can two launch chains overlap on one GPU?  A 32-row decode step is ~102 dependent launches that each leave most of the chip idle
(0.29 of the HBM roofline).  Here the batch is split over TWO engines (own weights, own KV, own stream) whose decode graphs are enqueued alternately from one host
thread; the wall clock of both finishing K steps is compared with one engine decoding all the rows.
usage: python tools/two_stream_probe.py [rows_per_engine=16] [engines=2] [K=128]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chatttsplus_amd import _lib, synth
from chatttsplus_amd.hip_models.gpt import GPT, sampler_cfg_from_objects

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
P, S0 = 48, 200
sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)


def make(rows):
    g = GPT(bench.LLAMA, max_batch=rows, max_seq_len=P + S0 + 10 * K + 64, weight_dtype="fp32", device=str(dev))
    g.load_state_dict(sd)
    return g


def begin(g, rows, seed, stream):
    ids, mask = synth.prompt_ids(rows, P, synth.GPT_REAL["num_text_tokens"], seed)
    ids_t = torch.from_numpy(ids).to(dev)
    with torch.cuda.stream(stream):
        emb = g(ids_t, torch.ones(rows, P, dtype=torch.bool, device=dev))
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    n = S0 + 10 * K + 16
    sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, n, n, lw, lp, 4)
    keep = dict(out=torch.zeros(rows, n, 4, dtype=torch.int32, device=dev), fin=torch.zeros(rows, dtype=torch.int32, device=dev), end=torch.zeros(rows, dtype=torch.int32, device=dev),
                msk=torch.from_numpy(mask).to(dev).to(torch.int32), emb=emb)
    io = _lib.GenIO(ids=keep["out"].data_ptr(), hiddens=None, finish=keep["fin"].data_ptr(), end_idx=keep["end"].data_ptr(), noise=None, n_draws=0, seed=seed)
    st = C.c_void_p(stream.cuda_stream)
    torch.cuda.synchronize(dev)
    _lib.check(g._lib.ctts_gpt_begin(g._h, rows, P, keep["msk"].data_ptr(), C.byref(sc), C.byref(io), st), "begin")
    _lib.check(g._lib.ctts_gpt_prefill(g._h, emb.data_ptr(), st), "prefill")
    _lib.check(g._lib.ctts_gpt_sample(g._h, st), "sample")
    _lib.check(g._lib.ctts_gpt_decode(g._h, S0, 1, st), "decode")
    torch.cuda.synchronize(dev)
    return keep, st


def timed(engines, chunk=8):
    """engines: [(g, st)]: K steps each, enqueued alternately in chunks"""
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(K // chunk):
        for g, st in engines:
            _lib.check(g._lib.ctts_gpt_decode(g._h, chunk, 1, st), "decode")
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / K * 1e3


one = make(R * NE)
s_one = torch.cuda.Stream(dev)
k1, st1 = begin(one, R * NE, 7, s_one)
t_one = [timed([(one, st1)]) for _ in range(3)]
gs, keeps = [], []
for e in range(NE):
    g = make(R); s = torch.cuda.Stream(dev)
    k, st = begin(g, R, 7 + e, s)
    gs.append((g, st)); keeps.append((k, s))
t_alone = [timed([gs[0]]) for _ in range(3)]
t_two = [timed(gs) for _ in range(3)]
print(json.dumps({"rows_per_engine": R, "engines": NE, "ms_per_step_one_engine_all_rows": round(min(t_one[1:]), 5), "ms_per_step_one_engine_its_rows_alone": round(min(t_alone[1:]), 5),
                  "ms_per_step_all_engines_concurrently": round(min(t_two[1:]), 5), "tokens_per_s_one_engine": round(R * NE / min(t_one[1:]) * 1e3, 1),
                  "tokens_per_s_concurrent": round(R * NE / min(t_two[1:]) * 1e3, 1)}), flush=True)
