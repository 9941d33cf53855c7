"""Persistent decode layer (persist_layer.hip) on the GPU: correctness per phase, agreement with the launch path, step times, per-edge prices.

  1. ONE-layer engine, one decode step at batch 1 and 3: every in-launch hand-off (q|k|v after RoPE, attention output, x + attention, silu(gate) * up)
     is read back from the granule buffers (ctts_gpt_debug_read "pl_g") and compared with a float64 torch evaluation of the layer from the same
     inputs (plain torch on the synthetic weights -- not the oracle); the hidden row of the step checks the down projection.
  2. 20-layer engine: free-running generate() with device noise, persistent layers off / on: token ids must be identical, hiddens to rounding;
     batch 1, 2, 3, 4; a 600-token context; a context that crosses 1024 keys mid-generation (hands over to the launch path).
  3. Step time in bench.py's window, interleaved rounds (off / on), batch 1, 2, 4.
  4. Per-workgroup phase marks of the last launch (wall_clock64, 100 MHz) -> the measured price of every edge.
usage: python tools/persist_probe.py [--skip-times] > gpurun_out/persist_probe.jsonl"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chatttsplus_amd import _lib, synth  # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT, rope_table  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--skip-times", action="store_true")
ap.add_argument("--skip-layer", action="store_true")
ap.add_argument("--skip-checks", action="store_true")
ap.add_argument("--marks-prompt", type=int, default=48)
ap.add_argument("--marks-rows", default="1,2,3,4,5", help="row counts of the phase marks (6..8: the two-item attention workgroups)")
ap.add_argument("--adapters", action="store_true", help="phase marks with a per-utterance LoRA adapter on every row (round 6: the LORA kernels)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
MAXR = 5                       # rows this probe drives
GMAXR = 8                      # PL_MAXR: rows the library sizes its granule buffers for (persist.h)
G_QKV, G_ATT, G_X1, G_ACT = GMAXR * 12 * 192, GMAXR * 768, GMAXR * 768, GMAXR * 3072


def out(**kw):
    print(json.dumps(kw), flush=True)


def debug_read(g, name, nbytes):
    buf = np.zeros(nbytes, dtype=np.uint8)
    got = C.c_size_t(0)
    _lib.check(g._lib.ctts_gpt_debug_read(g._h, name.encode(), buf.ctypes.data_as(C.c_void_p), nbytes, C.byref(got), g._stream()), f"debug_read({name})")
    return buf[:got.value]


def granules(g):
    raw = debug_read(g, "pl_g", (G_QKV + G_ATT + G_X1 + G_ACT) * 8).view(np.uint32).reshape(-1, 2)
    val, tag = raw[:, 0].copy().view(np.float32), raw[:, 1].copy()
    o = [0, G_QKV, G_QKV + G_ATT, G_QKV + G_ATT + G_X1, G_QKV + G_ATT + G_X1 + G_ACT]
    gs = [(val[o[i]:o[i + 1]], tag[o[i]:o[i + 1]]) for i in range(4)]
    # the act rows travel as 16-byte granules {v0, v1, v2, tag + bits(v0) + bits(v1) + bits(v2)}, six per (row, producer workgroup) (persist_layer.hip PL_ACT16):
    # granule f holds columns 16 (f / 6) + 3 (f % 6) + {0, 1, 2} of the [rows][3072] array; unpack to one (value, tag) per column
    a16 = raw[o[3]:o[4]].reshape(-1)[:MAXR * 192 * 6 * 4].reshape(-1, 4)
    tg = (a16[:, 3] - a16[:, 0] - a16[:, 1] - a16[:, 2]).astype(np.uint32)
    av, at = np.zeros(MAXR * 3072, np.float32), np.zeros(MAXR * 3072, np.uint32)
    f = np.arange(a16.shape[0]); col = 16 * (f // 6) + 3 * (f % 6)
    for c in range(3):
        ok = (f % 6 < 5) | (c == 0)
        av[col[ok] + c] = a16[ok, c].copy().view(np.float32); at[col[ok] + c] = tg[ok]
    gs[3] = (av, at)
    return gs


def make(cfg_layers, max_batch, max_seq, options=None):
    llama = dict(bench.LLAMA, num_hidden_layers=cfg_layers)
    g = GPT(llama, max_batch=max_batch, max_seq_len=max_seq, weight_dtype="fp32", device=str(dev), options=options or {})
    sd = synth.gpt_state_dict(dict(synth.GPT_REAL, num_hidden_layers=cfg_layers), 1234)
    g.load_state_dict(sd)
    return g, sd


def gen(g, B, P, N, persist, seed=7, pad_left=None):
    g.set_option("persistent_rows", persist)
    cfg = synth.GPT_REAL
    ids, mask = synth.prompt_ids(B, P, cfg["num_text_tokens"], 4321, pad_left=pad_left)
    ids_t = torch.from_numpy(ids).to(dev)
    emb = g(ids_t, torch.ones(B, P, dtype=torch.bool, device=dev))
    res = list(g.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                          logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=seed))[-1]
    return [i.cpu() for i in res.ids], [h.cpu() for h in res.hiddens], emb.cpu()


def layer_reference(sd, emb, x_new, T):
    """float64 evaluation of layer 0 for ONE sequence: prompt rows emb[T, 768] (cache), decode row x_new[768] at position T."""
    f = lambda k: torch.from_numpy(sd[k]).double()
    eps = 1e-6
    rope = torch.from_numpy(rope_table(T + 2)).double()

    def norm(x, w):
        return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w

    def rot(v, pos):                       # v [..., 12, 64]
        c, s = rope[pos, :32], rope[pos, 32:]
        a, b = v[..., :32], v[..., 32:]
        return torch.cat((a * c - b * s, b * c + a * s), -1)

    p = "gpt.layers.0."
    X = torch.cat((emb.double(), x_new.double()[None]), 0)                 # [T + 1, 768]
    h = norm(X, f(p + "input_layernorm.weight"))
    q = (h @ f(p + "self_attn.q_proj.weight").T).view(T + 1, 12, 64)
    k = (h @ f(p + "self_attn.k_proj.weight").T).view(T + 1, 12, 64)
    v = (h @ f(p + "self_attn.v_proj.weight").T).view(T + 1, 12, 64)
    pos = torch.arange(T + 1)
    q = torch.stack([rot(q[t], t) for t in range(T + 1)]); k = torch.stack([rot(k[t], t) for t in range(T + 1)])
    qn, kn, vn = q[T], k[T], v[T]
    att = torch.einsum("hd,thd->ht", qn, k) / 8.0
    att = torch.softmax(att, -1)
    ao = torch.einsum("ht,thd->hd", att, v).reshape(768)
    x1 = X[T] + ao @ f(p + "self_attn.o_proj.weight").T
    h2 = norm(x1, f(p + "post_attention_layernorm.weight"))
    gate, up = h2 @ f(p + "mlp.gate_proj.weight").T, h2 @ f(p + "mlp.up_proj.weight").T
    act = torch.nn.functional.silu(gate) * up
    x2 = x1 + act @ f(p + "mlp.down_proj.weight").T
    hid = norm(x2, f("gpt.norm.weight"))
    return dict(qkv=torch.stack((qn, kn, vn), 1).reshape(12 * 192), att=ao, x1=x1, act=act, hid=hid)       # qkv -> [head][q 64 | k 64 | v 64]


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---- 1. one layer, phase by phase -----------------------------------------------------------------------------------------------------
if not args.skip_layer:
    g1, sd1 = make(1, 4, 128)
    for B in (1, 3):
        P = 24
        ids, hid, emb = gen(g1, B, P, 3, persist=4)           # step 0 = prompt pass; steps 1, 2 = decode (persistent); granules hold the LAST launch (step 2)
        st = debug_read(g1, "pl_state", 8).view(np.uint32)
        gs = granules(g1)
        epoch = (int(st[0]) - 1) * 32                          # tag of the last launch's layer 0 (launch counter * 32 + layer)
        worst = {}
        for r in range(B):
            # decode step 2 of row r: input = embedding of the tokens sampled at step 1, cache = prompt + token 0's row + token 1's row: rebuild the rows
            embc = torch.from_numpy(np.stack([sd1[f"emb_code.{i}.weight"] for i in range(4)]))
            x_tok = [sum(embc[i, int(ids[r][s, i])] for i in range(4)) for s in range(2)]           # ((e0 + e1) + e2) + e3
            seq = torch.cat((emb[r], x_tok[0][None]), 0)
            ref = layer_reference(sd1, seq, x_tok[1], P + 1)
            got = dict(qkv=gs[0][0][r * 12 * 192:(r + 1) * 12 * 192], att=gs[1][0][r * 768:(r + 1) * 768], x1=gs[2][0][r * 768:(r + 1) * 768],
                       act=gs[3][0][r * 3072:(r + 1) * 3072], hid=hid[r][2].numpy())
            tags_ok = all(bool((gs[i][1][r * n:(r + 1) * n] == epoch).all()) for i, n in enumerate((12 * 192, 768, 768, 3072)))
            for k in ref:
                worst[k] = max(worst.get(k, 0.0), relerr(got[k], ref[k].numpy()))
            worst["tags_ok"] = worst.get("tags_ok", True) and tags_ok
        out(check="one_layer_phases_vs_float64", B=B, epoch=epoch, error_word=int(st[1]), rel_err=worst, ok=bool(max(v for k, v in worst.items() if k != "tags_ok") < 2e-5 and worst["tags_ok"]))
    g1.close()

# ---- 2. twenty layers, launch path vs persistent layers -----------------------------------------------------------------------------------
MARK_ROWS = [int(x) for x in args.marks_rows.split(",")]
g, _ = make(20, max(MAXR, max(MARK_ROWS)), 1400)
for (B, P, N, pad) in (() if args.skip_checks else ((1, 48, 64, None), (2, 40, 48, [0, 9]), (3, 33, 40, [0, 5, 17]), (4, 48, 40, [3, 0, 11, 20]), (5, 36, 40, [0, 4, 9, 2, 13]), (5, 420, 24, None), (1, 600, 24, None), (1, 1000, 48, None))):
    a_ids, a_hid, _ = gen(g, B, P, N, persist=0, pad_left=pad)
    b_ids, b_hid, _ = gen(g, B, P, N, persist=MAXR, pad_left=pad)
    same = all(torch.equal(x, y) for x, y in zip(a_ids, b_ids))
    herr = max(relerr(y.numpy(), x.numpy()) for x, y in zip(a_hid, b_hid))
    st = debug_read(g, "pl_state", 8).view(np.uint32)
    out(check="generate_launches_vs_persistent", B=B, prompt=P, tokens=N, ids_identical=bool(same), hidden_rel_err=herr, error_word=int(st[1]), epoch=int(st[0]))

# ---- 3. step times ---------------------------------------------------------------------------------------------------------------------------
if not args.skip_times:
    spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
    leg = bench.Leg(g, dev, 0, 1)
    for B in (1, 2, 4, 5):
        arms = {"launches": (0, 0), "persistent_one_launch": (MAXR, 0), "persistent_launch_per_layer": (MAXR, 1)}
        times = {k: [] for k in arms}
        for rnd in range(4):
            for k, (pr, lpl) in arms.items():
                g.set_option("persistent_rows", pr)
                g.set_option("persistent_layers_per_launch", lpl)
                r = leg.run(B, 48, 64, 8, spk=spk)
                if rnd:
                    times[k].append(r["ev_ms"] / r["K"])
        g.set_option("persistent_layers_per_launch", 0)
        out(check="step_time_ms", B=B, **{k: {"median": round(statistics.median(t), 5), "min": round(min(t), 5)} for k, t in times.items()},
            ratio_one_launch=round(statistics.median(times["persistent_one_launch"]) / statistics.median(times["launches"]), 4))

# ---- 4. per-edge prices from the phase marks of the LAST layer of one launch (eager launches) -------------------------------------------
if True:
    spk = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
    leg = bench.Leg(g, dev, 0, 1)
    g.set_option("persistent_rows", max(MAXR, max(MARK_ROWS)))
    g.set_option("persistent_timestamps", 1)
    if args.adapters:
        rl = np.random.Generator(np.random.Philox(key=31))
        for slot in range(2):
            g.load_adapter(slot, [(l, t, (rl.standard_normal((8, 768)) * 0.02).astype(np.float32), (rl.standard_normal((768, 8)) * 0.02).astype(np.float32), 2.0)
                                  for l in range(20) for t in ("q_proj", "k_proj", "v_proj", "o_proj")])
    for B in MARK_ROWS:
        if args.adapters:
            g.set_row_adapters([b % 2 for b in range(B)])
        for rep in range(4):
            leg.run(B, args.marks_prompt, 4, 4, spk=spk, use_graph=0, gen_tokens=0)
            ts = debug_read(g, "pl_ts", 256 * 10 * 8).view(np.uint64).reshape(256, 10).astype(np.float64) * 0.01        # us
        natt = 12 * B if B <= 5 else 6 * B             # (6..8 rows: two items per attention workgroup, the first item's edge wave writes the marks)
        gem, att = ts[:192], ts[192:192 + natt, :3]
        t0 = min(gem[:, 0].min(), att[:, 0].min())
        names = ["start", "x_loaded", "x_gathered", "qkv_published", "attention_gathered", "x1_published", "x1_gathered", "act_published", "act_gathered", "end"]
        med = {n: round(float(np.median(gem[:, i] - t0)), 2) for i, n in enumerate(names)}
        mx = {n: round(float((gem[:, i] - t0).max()), 2) for i, n in enumerate(names)}
        amed = {n: round(float(np.median(att[:, i] - t0)), 2) for i, n in enumerate(["start", "qkv_gathered", "attention_published"])}
        fine = ts[192:192 + natt]
        out(check="attention_phase_fine_marks_us", B=B, since_qkv_gathered={n: round(float(np.median(fine[:, i] - fine[:, 1])), 2) for n, i in
            (("b1_passed_wave0", 3), ("scores_and_max", 4), ("exp_pv", 5), ("cross_lane_sums", 6), ("b2_passed_wave8", 7), ("published", 2))})
        out(check="phase_marks_us_last_layer", B=B, adapters=bool(args.adapters), layers_in_launch=20, gemv_median=med, gemv_max=mx, attention_median=amed,
            edges_us={"entry_to_first_barrier": med["x_loaded"], "mean_layer": round((med["end"] - med["x_loaded"]) / 20.0, 3),
                      "qkv_phase": round(med["qkv_published"] - med["x_gathered"], 2), "qkv_edge": round(amed["qkv_gathered"] - med["qkv_published"], 2),
                      "attention_phase": round(amed["attention_published"] - amed["qkv_gathered"], 2),
                      "attention_edge": round(med["attention_gathered"] - amed["attention_published"], 2),
                      "o_proj_phase": round(med["x1_published"] - med["attention_gathered"], 2), "x1_edge": round(med["x1_gathered"] - med["x1_published"], 2),
                      "gate_up_phase": round(med["act_published"] - med["x1_gathered"], 2), "act_edge": round(med["act_gathered"] - med["act_published"], 2),
                      "down_phase_to_end": round(med["end"] - med["act_gathered"], 2), "launch_total_last_workgroup": mx["end"]})
    g.set_option("persistent_timestamps", 0)
g.close()
