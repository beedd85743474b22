"""Builds bert4rec_amd/libb4r_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["b4r_gemm.hip", "b4r_gemm_rx.hip", "b4r_rowops.hip", "b4r_attn.hip", "b4r_attn_rx.hip", "b4r_head_rx.hip", "b4r_head32.hip", "b4r_ffn_rx.hip", "b4r_ffn32w.hip", "b4r_attn_block.hip", "b4r_attn32.hip", "b4r_rank.hip", "b4r_model.hip"]
OUT = os.path.join(HERE, "libb4r_hip.so")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]   # every translation unit is rebuilt
    deps = srcs + headers + [os.path.join(os.path.dirname(HERE), "include", "b4r.h")]
    if not force and _newer(OUT, deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
