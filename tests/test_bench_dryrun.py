"""bench.py's multi-rank control flow (env contract of torch.distributed.run, speaker broadcast, barriers, MAX-over-ranks
timing, single JSON line from rank 0) rehearsed on CPU with gloo, world size 2."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_bench_dry_run_world2():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--dry-run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["ms_per_step"] * 8 / 1e3 >= 0.09            # the slower rank (0.1 s) defines the time
    # the sharded-request leg (BASELINE configs[3]) rehearsed through the real ChatTTSPlusPipeline.infer_sharded host code: both ranks serve a
    # share, every utterance keeps its tokens (digest == the world-1 digest), the lengths are the targets
    # what an N > 1 line must carry: the CPU baseline (rank 0, after the group is gone), north_star's 512-token-prompt leg
    # and the batch-32 RTF
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert len(d["extra"]["prompt512_batch1"]["per_rank_tokens_per_s"]) == 2 and d["extra"]["batch32"]["rtf_end_to_end"] > 0
    assert d["process_group"] == {"backend": "gloo", "world_size": 2, "forced_at_world_1": False}
    sr = d["extra"]["sharded_request"]
    assert sr["utterances"] == 48 and len(sr["per_rank_useful_tokens"]) == 2 and all(t > 0 for t in sr["per_rank_useful_tokens"])
    assert sum(sr["per_rank_useful_tokens"]) == sr["useful_tokens"] and sr["load_imbalance_max_over_mean"] < 1.2
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "4"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    sr1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])["extra"]["sharded_request"]
    assert sr1["ids_digest"] == sr["ids_digest"] and sr1["lengths_digest"] == sr["lengths_digest"]


def test_bench_dry_run_single():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "4"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["process_group"] is None


def test_bench_force_pg_takes_the_group_path_at_world_1():
    """`--force-pg`: a one-rank process group and every rendezvous of the N > 1 path at --gpus 1 (gloo here; "nccl" = RCCL on the GPU box, where the same
    switch makes the first contact with RCCL a one-GPU run instead of the driver's 8-GPU one)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "4", "--force-pg"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["process_group"] == {"backend": "gloo", "world_size": 1, "forced_at_world_1": True}
    assert d["extra"]["sharded_request"]["utterances"] == 48 and d["cpu_baseline"]["value"] > 0


def test_bench_self_spawns_ranks_without_launcher():
    """`python bench.py --gpus 2` with no launcher env re-executes itself through torch.distributed.run (a plain
    --gpus N invocation must work); world size and per-rank rates are visible in the JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and len(d["per_rank_tokens_per_s"]) == 2
    assert d["per_rank_tokens_per_s"][0] > d["per_rank_tokens_per_s"][1]          # rank 1 sleeps longer
