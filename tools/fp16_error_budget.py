"""Where does the fp16-mode error come from?  CPU analysis with the oracle (test infrastructure): the fp32 oracle runs free, its
token ids are forced through copies of the oracle whose (a) packed weights, (b) MFMA B operands (normalised rows, attention
output, SwiGLU output) and (c) cached K/V are rounded to fp16 like the HIP fp16 mode does; prints the teacher-forced hidden
rel-RMS error of every combination.   python tools/fp16_error_budget.py [N]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth          # noqa: E402
from oracle import ref_cpu                 # noqa: E402

HD = 64


def r16(x):
    return x.half().float()


class Emu(ref_cpu.OracleGPT):
    def __init__(self, sd, w16, a16, kv16, norm_folded=True):
        super().__init__(sd, 12)
        self.w16, self.a16, self.kv16 = w16, a16, kv16
        self.W = {}
        for l in range(self.L):
            p = f"gpt.layers.{l}."
            ln1, ln2 = self.sd[p + "input_layernorm.weight"], self.sd[p + "post_attention_layernorm.weight"]
            for n, ln in (("self_attn.q_proj", ln1), ("self_attn.k_proj", ln1), ("self_attn.v_proj", ln1), ("mlp.gate_proj", ln2), ("mlp.up_proj", ln2)):
                w = self.sd[p + n + ".weight"] * ln[None, :]            # RMSNorm weight folded into the columns (gpt_engine.hip pack)
                self.W[p + n] = r16(w) if w16 else w
            for n in ("self_attn.o_proj", "mlp.down_proj"):
                w = self.sd[p + n + ".weight"]
                self.W[p + n] = r16(w) if w16 else w

    def A(self, x):
        return r16(x) if self.a16 else x

    def forward(self, x, attn_mask, position_ids):
        B, q, _ = x.shape
        past = self.kv_len
        Ltot = past + q
        minv = torch.finfo(torch.float32).min
        cache_position = torch.arange(past, Ltot)
        causal = torch.full((q, Ltot), minv)
        if q != 1:
            causal = torch.triu(causal, diagonal=1)
        causal = causal * (torch.arange(Ltot) > cache_position.reshape(-1, 1))
        mask4 = causal[None, None].expand(B, 1, -1, -1).clone()
        pad = (mask4 + attn_mask[:, None, None, :Ltot].float()) == 0
        mask4 = mask4.masked_fill(pad, minv)
        cos, sin = self._rope(position_ids)
        cos = cos[:, None]; sin = sin[:, None]
        for l in range(self.L):
            p = f"gpt.layers.{l}."
            res = x
            h = self.A(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
            qs = F.linear(h, self.W[p + "self_attn.q_proj"]).view(B, q, self.nh, HD).transpose(1, 2)
            ks = F.linear(h, self.W[p + "self_attn.k_proj"]).view(B, q, self.nh, HD).transpose(1, 2)
            vs = F.linear(h, self.W[p + "self_attn.v_proj"]).view(B, q, self.nh, HD).transpose(1, 2)
            qs = qs * cos + self._rotate_half(qs) * sin
            ks = ks * cos + self._rotate_half(ks) * sin
            if self.kv16:
                ks, vs = r16(ks), r16(vs)
            self.kc[l, :, :, past:Ltot] = ks
            self.vc[l, :, :, past:Ltot] = vs
            K = self.kc[l, :, :, :Ltot]; V = self.vc[l, :, :, :Ltot]
            att = torch.softmax(torch.matmul(qs, K.transpose(-1, -2)) / math.sqrt(HD) + mask4, dim=-1)
            o = self.A(torch.matmul(att, V).transpose(1, 2).reshape(B, q, self.H))
            x = res + F.linear(o, self.W[p + "self_attn.o_proj"])
            res = x
            h = self.A(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
            g = F.linear(h, self.W[p + "mlp.gate_proj"]); u = F.linear(h, self.W[p + "mlp.up_proj"])
            x = res + F.linear(self.A(F.silu(g) * u), self.W[p + "mlp.down_proj"])
        self.kv_len = Ltot
        return self._rms(x, self.sd["gpt.norm.weight"])


class RefHalf(ref_cpu.OracleGPT):
    """Emulation of the reference's own GPU path (pipeline:37-41: the whole model .half()): every weight and every op output
    rounded to fp16 (matmul accumulation itself kept exact, as tensor-core fp16 GEMMs accumulate in fp32); RMSNorm variance and
    softmax in fp32 like llama.py:82-87 / SDPA.  Indicative only -- there is no fp16 GPU reference to run here."""

    def forward(self, x, attn_mask, position_ids):
        B, q, _ = x.shape
        past = self.kv_len
        Ltot = past + q
        minv = torch.finfo(torch.float32).min
        cache_position = torch.arange(past, Ltot)
        causal = torch.full((q, Ltot), minv)
        if q != 1:
            causal = torch.triu(causal, diagonal=1)
        causal = causal * (torch.arange(Ltot) > cache_position.reshape(-1, 1))
        mask4 = causal[None, None].expand(B, 1, -1, -1).clone()
        pad = (mask4 + attn_mask[:, None, None, :Ltot].float()) == 0
        mask4 = mask4.masked_fill(pad, minv)
        cos, sin = self._rope(position_ids)
        cos = r16(cos[:, None]); sin = r16(sin[:, None])
        W = lambda k: r16(self.sd[k])
        x = r16(x)

        def norm(x, w):
            return r16(r16(w) * r16(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)))
        for l in range(self.L):
            p = f"gpt.layers.{l}."
            res = x
            h = norm(x, self.sd[p + "input_layernorm.weight"])
            qs = r16(F.linear(h, W(p + "self_attn.q_proj.weight"))).view(B, q, self.nh, HD).transpose(1, 2)
            ks = r16(F.linear(h, W(p + "self_attn.k_proj.weight"))).view(B, q, self.nh, HD).transpose(1, 2)
            vs = r16(F.linear(h, W(p + "self_attn.v_proj.weight"))).view(B, q, self.nh, HD).transpose(1, 2)
            qs = r16(r16(qs * cos) + r16(self._rotate_half(qs) * sin))
            ks = r16(r16(ks * cos) + r16(self._rotate_half(ks) * sin))
            self.kc[l, :, :, past:Ltot] = ks
            self.vc[l, :, :, past:Ltot] = vs
            K = self.kc[l, :, :, :Ltot]; V = self.vc[l, :, :, :Ltot]
            att = r16(torch.softmax(r16(torch.matmul(qs, K.transpose(-1, -2)) / math.sqrt(HD)) + mask4, dim=-1))
            o = r16(torch.matmul(att, V)).transpose(1, 2).reshape(B, q, self.H)
            x = r16(res + r16(F.linear(o, W(p + "self_attn.o_proj.weight"))))
            res = x
            h = norm(x, self.sd[p + "post_attention_layernorm.weight"])
            g = r16(F.linear(h, W(p + "mlp.gate_proj.weight"))); u = r16(F.linear(h, W(p + "mlp.up_proj.weight")))
            x = r16(res + r16(F.linear(r16(r16(F.silu(g)) * u), W(p + "mlp.down_proj.weight"))))
        self.kv_len = Ltot
        return norm(x, self.sd["gpt.norm.weight"])


def forced_hiddens(m, emb, ids, forced, N):
    B, T = ids.shape[:2]
    m.alloc_cache(B, T + N + 1)
    mask = torch.ones(B, T + N + 1, dtype=torch.long)
    pos = torch.arange(T)[None].expand(B, -1)
    hid = [m.forward(emb, mask, pos)[:, -1]]
    for i in range(1, N):
        x = m.embed_code(forced[:, i - 1:i])
        hid.append(m.forward(x, mask, torch.full((B, 1), T + i - 1))[:, -1])
    return torch.stack(hid, 1)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    torch.set_num_threads(16)
    sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    ids, mask = synth.prompt_ids(1, 48, 21178, 701)
    o = ref_cpu.OracleGPT(sd, 12)
    emb = o.embed(torch.from_numpy(ids), torch.ones(1, 48, dtype=torch.bool))
    ref = o.generate(emb, torch.from_numpy(ids), ref_cpu.SamplerParams(min_new_token=N), attention_mask=torch.from_numpy(mask), max_new_token=N,
                     noise=ref_cpu.SeededNoise(9))
    forced = torch.stack(list(ref.ids), 0)
    href = torch.stack(list(ref.hiddens), 0)
    with torch.no_grad():
        for (w, a, kv) in ([(1, 1, 1)] if os.environ.get('QUICK') else [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (1, 1, 1)]):
            h = forced_hiddens(Emu(sd, w, a, kv), emb, torch.from_numpy(ids), forced, N)
            rel = float((h - href).pow(2).mean().sqrt() / href.pow(2).mean().sqrt())
            print(f"fp16 weights {w}  fp16 MFMA operands {a}  fp16 KV {kv}:  hidden rel-RMS {rel:.3e}", flush=True)
        h = forced_hiddens(RefHalf(sd, 12), emb, torch.from_numpy(ids), forced, N)
        rel = float((h - href).pow(2).mean().sqrt() / href.pow(2).mean().sqrt())
        print(f"emulated reference GPU path (model.half(), every op output in fp16):  hidden rel-RMS {rel:.3e}", flush=True)


if __name__ == "__main__":
    main()
