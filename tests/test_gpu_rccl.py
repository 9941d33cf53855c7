"""RCCL executed on the GPU box: `infer_sharded` through a real "nccl" process group.

north_star: "RCCL broadcast of speaker embeddings over xGMI".  Every other N > 1 test of the repo runs on gloo (RCCL refuses two ranks on one
device), so before this test the calls `bench.py` makes at --gpus > 1 -- `init_process_group("nccl", device_id=...)`, `dist.broadcast` /
`all_reduce` / `all_gather` / `barrier` on DEVICE tensors -- had never run anywhere.  A one-rank nccl group takes the same code path (communicator
creation, stream ordering against the engine's launches, device-side collectives); `chatttsplus_amd.dist` runs its collectives whenever a group
exists, also at world size 1.

A child process (so the group never leaks into the pytest process) serves the same request twice -- without a group, then inside a one-rank nccl
group created exactly as bench.py creates it -- and asserts: the speaker table that came out of the RCCL broadcast equals the one that went in; the
seed broadcast returns rank 0's draw; every utterance gets the same token ids and waveform; the all-reduced lengths agree.
Reference counterpart: the sequential, state-free slice loop pipelines/chattts_plus_pipeline.py:391-397.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_UTT = 10


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _child(port, ckpt_dir, out_path):
    import torch.distributed as dist
    from chatttsplus_amd import dist as cdist, synth
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams, load_config
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, "configs", "infer", "chattts_plus_hip.yaml"))
    cfg["MODELS"]["gpt"]["kwargs"].update(weight_dtype="fp32", max_batch=8, max_seq_len=128)
    cfg["MODELS"].pop("dvae_encode", None)
    tok = synth.toy_tokenizer(os.path.join(ckpt_dir, "tok"))
    pipe = ChatTTSPlusPipeline(cfg, device="cuda:0", tokenizer=tok, checkpoint_dir=ckpt_dir, max_frames=2 * 40 + 64, vocoder_batch=8)
    texts = synth.toy_texts(N_UTT, 2, 24, seed=5)
    rng = np.random.Generator(np.random.Philox(key=9))
    limits = [int(x) for x in rng.integers(6, 33, size=N_UTT)]
    spk_index = [int(x) for x in rng.integers(0, 3, size=N_UTT)]
    table = torch.from_numpy(np.stack([synth.speaker_vector(40 + i) for i in range(3)]))
    params = InferCodeParams(prompt="[speed_5]", max_new_token=40, min_new_token=2, show_tqdm=False)

    def serve(continuous):
        ids = []
        torch.manual_seed(4321)                 # no explicit noise_seed: rank 0 draws it and broadcast_seed carries it (through RCCL when a group exists)
        mine, wavs, lens = pipe.infer_sharded(list(texts), speaker_index=spk_index, speaker_table=table, params_infer_code=params,
                                              slice_size=6, continuous=continuous, max_new_tokens_per_utterance=limits, ids_out=ids)
        return mine, [w.cpu() for w in wavs], lens, [i.cpu() for i in ids]

    plain = {c: serve(c) for c in (False, True)}
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)           # exactly bench.py's call
    try:
        assert dist.get_backend() == "nccl"
        got_table = cdist.broadcast_speakers(table, 3, 768, dev)
        assert got_table.is_cuda and torch.equal(got_table.cpu(), table.float()), "the RCCL broadcast changed the speaker table"
        assert cdist.broadcast_seed(123456789012345, dev) == 123456789012345
        # the collectives of bench.py's timed legs, on device tensors
        dist.barrier()
        t = torch.tensor([1.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per = [torch.zeros(1, device=dev, dtype=torch.float64)]
        dist.all_gather(per, torch.tensor([2.5], device=dev, dtype=torch.float64))
        assert float(t.item()) == 1.5 and float(per[0].item()) == 2.5
        grouped = {c: serve(c) for c in (False, True)}
        dist.barrier()
    finally:
        dist.destroy_process_group()
    for c in (False, True):
        m0, w0, l0, i0 = plain[c]
        m1, w1, l1, i1 = grouped[c]
        assert m0 == m1 == list(range(N_UTT)) and l0 == l1, f"continuous={c}: partition / all-reduced lengths differ under the nccl group"
        assert len(set(l0)) > 3 and all(1 <= n <= lim for n, lim in zip(l0, limits))
        for u in range(N_UTT):
            assert torch.equal(i0[u], i1[u]), f"continuous={c}: utterance {u} got other token ids under the nccl group"
            assert torch.equal(w0[u], w1[u]), f"continuous={c}: utterance {u} got another waveform under the nccl group"
    torch.save(dict(ok=True, lens=plain[False][2]), out_path)


def test_infer_sharded_through_a_one_rank_nccl_group(tmp_path):
    from chatttsplus_amd import synth
    ckpt = synth.write_checkpoints(str(tmp_path / "ckpt"), 1234, full=False)
    out = str(tmp_path / "out.pt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_child, args=(_free_port(), ckpt, out))
    p.start()
    p.join(timeout=900)
    assert p.exitcode == 0, f"the nccl child exited with {p.exitcode}"
    got = torch.load(out, weights_only=True)
    assert got["ok"] and len(got["lens"]) == N_UTT


def test_bench_force_pg_runs_the_multi_rank_path_on_rccl():
    """`bench.py --gpus 1 --force-pg`: the barriers, the MAX all-reduce, the per-rank gather and the speaker broadcast of the N > 1 path on a one-rank RCCL group,
    ONE JSON line on stdout (RCCL's own banner is kept off it), the CPU baseline after the group is gone."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-extras", "--cpu-steps", "4", "--force-pg"],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["process_group"] == {"backend": "nccl", "world_size": 1, "forced_at_world_1": True}
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["value"] > 0 and d["cpu_baseline"]["value"] > 0 and len(d["per_rank_tokens_per_s"]) == 1
