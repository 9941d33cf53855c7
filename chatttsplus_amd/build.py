"""Builds libctts_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libctts_hip.so")
SOURCES = ["gpt_engine.hip", "skinny_gemm.hip", "persist_layer.hip", "prefill_gemm.hip", "prefill_split.hip", "lora.hip", "attention.hip", "sampler.hip", "vocoder.hip", "encoder.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "ctts_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """`diag=True` (python -m chatttsplus_amd.build --force --diag) compiles the diagnostic switches in (-DCTTS_DIAG: CTTS_SPLITS,
    CTTS_XH, CTTS_NBG2_ROWS, ... read once at create); the product build ignores the environment apart from CTTS_PASS_ROWS."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        # kernarg preload: the leading scalar kernel arguments arrive in SGPRs at wave launch instead of through a first s_load
        # round trip (one serial scalar-cache miss per launch; a decode step is ~100 dependent launches)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=8"] + (["-DCTTS_DIAG"] if diag else []) + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out = (out or "") + "\n[build] hipcc timed out after 900 s"
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"[build] {src} failed", file=sys.stderr)
            if not verbose:
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, timeout=900)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, diag="--diag" in sys.argv)
