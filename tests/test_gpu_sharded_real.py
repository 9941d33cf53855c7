"""BASELINE configs[3] rehearsed with the REAL engine on more than one rank.

Two processes share cuda:0 (one process per rank, gloo process group: RCCL refuses two ranks on one device), each builds the
real ChatTTSPlusPipeline (hip_models.GPT / Synth on synthetic real-size checkpoints) and calls `infer_sharded` on the same 14 ragged
utterances -- the snake partition, the speaker-table broadcast from rank 0, the seed broadcast, per-rank slices / continuous batching, the
vocoder, the length all-reduce.  The parent then runs the same request at world 1 and asserts what DESIGN section 6 promises: every utterance
gets the SAME token ids on whatever rank / slice / decode row it was served (bit-exact) and its waveform agrees to <= 1e-4 rel.
Reference counterpart: the sequential, state-free slice loop pipelines/chattts_plus_pipeline.py:391-397.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_UTT = 14
SEED = 77


def _request():
    from chatttsplus_amd import synth
    texts = synth.toy_texts(N_UTT, 2, 24, seed=5)
    rng = np.random.Generator(np.random.Philox(key=9))
    limits = [int(x) for x in rng.integers(6, 33, size=N_UTT)]
    spk_index = [int(x) for x in rng.integers(0, 3, size=N_UTT)]
    return texts, limits, spk_index


def _pipeline(ckpt_dir, device="cuda:0"):
    from chatttsplus_amd import synth
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, "configs", "infer", "chattts_plus_hip.yaml"))
    cfg["MODELS"]["gpt"]["kwargs"].update(weight_dtype="fp32", max_batch=4, max_seq_len=128)
    cfg["MODELS"].pop("dvae_encode", None)
    tok = synth.toy_tokenizer(os.path.join(ckpt_dir, "tok"))
    return ChatTTSPlusPipeline(cfg, device=device, tokenizer=tok, checkpoint_dir=ckpt_dir, max_frames=2 * 40 + 64, vocoder_batch=8)


def _run(pipe, rank, continuous):
    from chatttsplus_amd import synth
    from chatttsplus_amd.pipeline import InferCodeParams
    texts, limits, spk_index = _request()
    table = torch.from_numpy(np.stack([synth.speaker_vector(40 + i) for i in range(3)])) if rank == 0 else None
    params = InferCodeParams(prompt="[speed_5]", max_new_token=40, min_new_token=2, show_tqdm=False)
    ids = []
    torch.manual_seed(1000 + rank)            # must not matter: the request seed is explicit, the noise is keyed by utterance
    mine, wavs, lens = pipe.infer_sharded(list(texts), speaker_index=spk_index, speaker_table=table, params_infer_code=params, noise_seed=SEED,
                                          slice_size=3, continuous=continuous, max_new_tokens_per_utterance=limits, ids_out=ids)
    return mine, [w.cpu() for w in wavs], lens, [i.cpu() for i in ids]


def _worker(rank, world, port, ckpt_dir, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = _pipeline(ckpt_dir)
        for continuous in (False, True):
            mine, wavs, lens, ids = _run(pipe, rank, continuous)
            torch.save(dict(mine=mine, wavs=wavs, lens=lens, ids=ids), os.path.join(out_dir, f"r{rank}_c{int(continuous)}.pt"))
            dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_infer_sharded_world2_real_engine_equals_world1(tmp_path):
    from chatttsplus_amd import synth
    ckpt = synth.write_checkpoints(str(tmp_path / "ckpt"), 1234, full=False)
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, ckpt, out_dir)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    # world 1, same entry point, in this process
    texts, limits, _ = _request()
    pipe = _pipeline(ckpt)
    for continuous in (False, True):
        mine1, wavs1, lens1, ids1 = _run(pipe, 0, continuous)
        assert mine1 == list(range(N_UTT))
        assert all(1 <= n <= lim for n, lim in zip(lens1, limits)) and len(set(lens1)) > 3          # ragged, limits respected
        assert [int(i.shape[0]) for i in ids1] == lens1
        seen = []
        for r in range(world):
            got = torch.load(os.path.join(out_dir, f"r{r}_c{int(continuous)}.pt"), weights_only=True)
            assert got["lens"] == lens1, f"rank {r} (continuous={continuous}): all-reduced lengths differ from the world-1 run"
            assert 0 < len(got["mine"]) < N_UTT
            for j, u in enumerate(got["mine"]):
                assert torch.equal(got["ids"][j], ids1[u]), f"utterance {u} on rank {r} (continuous={continuous}): token ids differ from world 1"
                a, b = got["wavs"][j].numpy(), wavs1[u].numpy()
                assert a.shape == b.shape
                rel = float(np.sqrt(np.mean((a - b) ** 2))) / max(float(np.sqrt(np.mean(b ** 2))), 1e-20)
                assert rel <= 1e-4, f"utterance {u} on rank {r} (continuous={continuous}): waveform rms-rel {rel}"
            seen += got["mine"]
        assert sorted(seen) == list(range(N_UTT))
    # the two serving modes agree with each other too (an utterance's tokens do not depend on how it was batched)
