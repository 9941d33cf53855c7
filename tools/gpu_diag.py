"""First-contact diagnostic on the GPU box: isolates each fused kernel of the GPT path by zeroing
sub-blocks / changing the layer count, and prints max-abs differences against the oracle.
Not a test (asserts nothing); run as  python tools/gpu_diag.py  [fp32|fp16]."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth  # noqa: E402
from chatttsplus_amd.hip_models.gpt import GPT  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def make(cfg, sd, wd, max_batch=4, max_seq=128):
    g = GPT(dict(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                 num_attention_heads=cfg["num_attention_heads"], num_hidden_layers=cfg["num_hidden_layers"]),
            num_audio_tokens=cfg["num_audio_tokens"], num_text_tokens=cfg["num_text_tokens"], num_vq=4,
            max_batch=max_batch, max_seq_len=max_seq, weight_dtype=wd)
    g.load_state_dict(sd)
    return g


def run_case(name, L, wd, zero=(), B=1, T=8, N=4, pad=None):
    cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = L
    sd = synth.gpt_state_dict(cfg, 1234)
    for k in list(sd):
        if any(z in k for z in zero):
            sd[k] = np.zeros_like(sd[k])
    ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], 21, pad_left=pad)
    o = ref_cpu.OracleGPT(sd, cfg["num_attention_heads"])
    emb = o.embed(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    sp = ref_cpu.SamplerParams(min_new_token=N)
    torch.manual_seed(5)
    ref = o.generate(emb, torch.from_numpy(ids), sp, attention_mask=torch.from_numpy(mask), max_new_token=N, trace_logits=True)
    g = make(cfg, sd, wd)
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    torch.manual_seed(5)
    emb_d = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    print(f"  emb diff {float((emb_d.cpu() - emb).abs().max()):.3g}")
    out = list(g.generate(emb_d, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask),
                          max_new_token=N, min_new_token=N, logits_warpers=lw, logits_processors=lp, return_hidden=True))[-1]
    for b in range(B):
        hd = out.hiddens[b].cpu()
        hr = ref.hiddens[b]
        n = min(hd.shape[0], hr.shape[0])
        per_step = (hd[:n] - hr[:n]).abs().amax(dim=1).tolist()
        same = torch.equal(out.ids[b].cpu()[:n], ref.ids[b][:n])
        print(f"[{name} {wd} L={L} zero={zero}] row {b}: n={hd.shape[0]}/{hr.shape[0]} ids_equal={same} "
              f"hidden max|d| per step: {[f'{v:.2e}' for v in per_step]}  (|h| max {float(hr.abs().max()):.2f})")
        if not same:
            print("   hip ids:", out.ids[b].cpu()[:n].tolist())
            print("   ref ids:", ref.ids[b][:n].tolist())
    lg = g.last_logits(B).cpu().reshape(B * 4, -1)
    print(f"  last-step logits max|d| {float((lg - ref.logits_trace[-1]).abs().max()):.3e} (|logit| max {float(ref.logits_trace[-1].abs().max()):.2f})")
    del g
    torch.cuda.synchronize()


def main():
    wds = sys.argv[1:] or ["fp32", "fp16"]
    t0 = time.time()
    for wd in wds:
        for name, L, zero, kw in [
            ("heads-only", 0, (), {}),
            ("attn-only", 1, ("down_proj",), {}),
            ("mlp-only", 1, ("o_proj",), {}),
            ("one-layer", 1, (), {}),
            ("two-layer", 2, (), {}),
            ("full", 20, (), {}),
            ("full-B2-pad", 20, (), dict(B=2, T=10, pad=[0, 3], N=6)),
            ("full-B3-long", 20, (), dict(B=3, T=40, pad=[0, 7, 19], N=6)),
        ]:
            try:
                run_case(name, L, wd, zero, **kw)
            except Exception as e:  # keep going: this is a diagnostic
                import traceback
                traceback.print_exc()
                print(f"[{name} {wd}] FAILED: {e}")
    print("diag done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
