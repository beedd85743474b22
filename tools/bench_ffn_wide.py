"""Times b4r_ffn_wide_fwd / _bwd (hidden 128 / 256) on the benchmark shapes: python tools/bench_ffn_wide.py [--n 51200]"""
import argparse
import ctypes as C
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from bert4rec_amd import _lib  # noqa: E402


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=51200)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    dev = "cuda"
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for H, I in ((128, 512), (256, 1024)):
        N = a.n
        g = torch.Generator(device=dev).manual_seed(1)
        r = lambda *sh, sc=1.0: torch.randn(*sh, device=dev, generator=g) * sc
        x1, W1, b1, W2, b2 = r(N, H), r(H, I, sc=0.05), r(I, sc=0.1), r(I, H, sc=0.05), r(H, sc=0.1)
        g2, be2, dz2 = 1 + 0.1 * r(H), 0.1 * r(H), r(N, H, sc=1e-3)
        out = {k: torch.empty(sh, device=dev) for k, sh in dict(z2=(N, H), x2=(N, H), mean2=(N,), rstd2=(N,), f=(N, I), fpre=(N, I),
                                                               df=(N, I), dx1=(N, H)).items()}
        st = torch.zeros(16, dtype=torch.int32, device=dev); st[0] = 7; st[1] = 3
        scratch = torch.empty(lib.b4r_ffn_wide_scratch_floats(H, I), device=dev)
        d = _lib.FfnDesc()
        d.N, d.H, d.I = N, H, I
        d.x1, d.W1, d.b1, d.W2, d.b2 = P(x1), P(W1), P(b1), P(W2), P(b2)
        d.ln_gamma, d.ln_beta, d.ln_eps = P(g2), P(be2), 1e-12
        d.rng, d.drop_stream, d.drop_rate = P(st), 5, 0.2
        d.z2, d.x2, d.mean2, d.rstd2 = P(out["z2"]), P(out["x2"]), P(out["mean2"]), P(out["rstd2"])
        d.scratch, d.dz2 = P(scratch), P(dz2)

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.reps * 1e3

        t_inf = timed(lambda: _lib.check(lib.b4r_ffn_wide_fwd(C.byref(d), None, None, s)))
        t_keep = timed(lambda: _lib.check(lib.b4r_ffn_wide_fwd(C.byref(d), P(out["f"]), P(out["fpre"]), s)))
        t_bwd = timed(lambda: _lib.check(lib.b4r_ffn_wide_bwd(C.byref(d), P(out["fpre"]), P(out["df"]), P(out["dx1"]), 1, s)))
        gf = 4.0 * N * H * I * 3 / 1e12   # executed TFLOP of the two three-term products
        print(f"H={H} I={I} N={N}: fwd (inference) {t_inf:7.1f} us  fwd (keeping f, fpre) {t_keep:7.1f} us  bwd (df, dx1) {t_bwd:7.1f} us"
              f"   executed {gf / (t_inf * 1e-6):6.0f} / {gf / (t_keep * 1e-6):6.0f} / {gf / (t_bwd * 1e-6):6.0f} TFLOP/s (incl. the pack launch)")


if __name__ == "__main__":
    main()
