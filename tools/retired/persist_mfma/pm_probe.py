"""The persistent MFMA decode stack (persist_mfma.hip) against the launch chain on ONE engine: token ids / hidden states of a short generation per batch
size, then the step time of both paths in bench.py's window, then (optional) the last layer's phase marks of every workgroup.

usage: python tools/pm_probe.py [--batches 5,8,16,17,24,32] [--steps 24] [--no-time] [--marks B] [--opt name=value ...]
Every line printed is one JSON object (profiles/r05_pm_probe*.jsonl)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chatttsplus_amd import _lib, synth  # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT  # noqa: E402

LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]


def gen(g, B, P, N, pad=None, seed=7):
    ids, mask = synth.prompt_ids(B, P, 21178, 4321, pad_left=pad)
    emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
    res = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                          logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=seed))[-1]
    return res.ids, res.hiddens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="5,8,16,17,24,32")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--prompt", type=int, default=40)
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--time-steps", type=int, default=64)
    ap.add_argument("--marks", type=int, default=0, help="batch size whose last-layer phase marks are dumped (0 = none)")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    Bs = [int(x) for x in args.batches.split(",") if x]
    dev = torch.device("cuda", 0)
    g = GPT(bench.LLAMA, max_batch=max(Bs + [args.marks, 1]), max_seq_len=48 + 16 + 512 + 16, weight_dtype="fp32", device=str(dev))
    g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
    for kv in args.opt:
        k, v = kv.split("=")
        g.set_option(k, int(v))
    rows_max = g.get_option("mfma_rows") or 32          # (the mode is opt-in: off by default)
    g.set_option("mfma_rows", rows_max)
    print(json.dumps({"mfma_rows": rows_max, "mfma_rows_min": g.get_option("mfma_rows_min"), "persistent_rows": g.get_option("persistent_rows")}), flush=True)
    spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
    if not args.no_check:
        # the reference = the launch chain on the SAME arithmetic: packed-residual path with the in-launch split-K combine from 5 rows on (its defaults start at 9)
        g.set_option("split_rows", 4); g.set_option("down_splitk_rows", 5)
        for B in Bs:
            pad = [(7 * b) % 13 for b in range(B)]
            g.set_option("mfma_rows", 0)
            ref_ids, ref_h = gen(g, B, args.prompt, args.steps, pad)
            g.set_option("mfma_rows", rows_max)
            try:
                ids, hid = gen(g, B, args.prompt, args.steps, pad)
            except Exception as e:
                print(json.dumps({"B": B, "error": str(e)[:300]}), flush=True)
                continue
            same = all(torch.equal(a, b) for a, b in zip(ids, ref_ids))
            first = [int((a != b).any(-1).to(torch.int32).argmax()) if not torch.equal(a, b) else -1 for a, b in zip(ids, ref_ids)]
            err0 = max(float((a[0] - b[0]).abs().max()) for a, b in zip(hid, ref_h))
            errs = max(float((a - b).abs().max()) for a, b in zip(hid, ref_h))
            bit = all(torch.equal(a, b) for a, b in zip(hid, ref_h))
            print(json.dumps({"B": B, "steps": args.steps, "ids_equal_launch_chain": same, "first_differing_step_per_row": first if not same else None,
                              "hidden_maxabs_step0": err0, "hidden_maxabs_all": errs, "hidden_bitwise": bit,
                              "hidden_ref_rms": float(torch.stack(list(ref_h)).pow(2).mean().sqrt())}), flush=True)
    g.set_option("split_rows", 8); g.set_option("down_splitk_rows", 9)
    if not args.no_time:
        leg = bench.Leg(g, dev, 0, 1)
        for B in Bs:
            out = {"B": B}
            for name, val in (("launch_chain", 0), ("persistent_mfma", rows_max)):
                g.set_option("mfma_rows", val)
                try:
                    r = leg.run(B, 48, args.time_steps, 8, spk=spk)
                    s = bench.summarize(r, 1)
                    out[name] = {"ms_per_step": s["step_ms_hip_events"], "frac": s["frac_of_8TBps"]}
                except BaseException as e:
                    out[name] = {"error": str(e)[:300]}
            g.set_option("mfma_rows", rows_max)
            print(json.dumps(out), flush=True)
    if args.marks:
        B = args.marks
        g.set_option("mfma_rows", rows_max)
        g.set_option("mfma_timestamps", 1)
        g.use_graph = False
        try:
            gen(g, B, args.prompt, 8)
        finally:
            g.use_graph = True
        buf = np.zeros(256 * 16, dtype=np.uint64)
        n = C.c_size_t(0)
        _lib.check(g._lib.ctts_gpt_debug_read(g._h, b"pm_ts", buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(n), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "debug_read")
        t = buf.reshape(256, 16).astype(np.int64)
        names = ["p1_ready", "p1_pub", "p2_ready", "p2_pub", "p3_ready", "p3_pub", "p4_ready", "p4_pub", "p5_ready", "p5_ticket", "p5_end"]
        valid = t > 0
        t0 = t[valid].min() if valid.any() else 0
        us = (t - t0) / 100.0                       # wall_clock64 ticks at 100 MHz
        summary = {}
        for i, nm in enumerate(names):
            col = us[:, i][valid[:, i]]
            if col.size:
                summary[nm] = {"n": int(col.size), "min": round(float(col.min()), 2), "median": round(float(np.median(col)), 2), "max": round(float(col.max()), 2)}
        print(json.dumps({"marks_B": B, "last_layer_us_since_first_mark": summary}), flush=True)
        g.set_option("mfma_timestamps", 0)


if __name__ == "__main__":
    main()
