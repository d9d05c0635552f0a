#!/usr/bin/env python3
"""Single-sequence latency of config 3 under different library settings, in ONE process and interleaved so that box-to-box and
run-to-run noise cancels:  python3 tools/latency_sweep.py QSVC_J2K_SPLIT=0,2,4 [QSVC_MQ_LANES=16,8] [--rounds 3] [--steps 8]
Prints, per setting, the median / min of the per-round mean ms per encode (whole step: analyse + texture, host included)."""
import argparse
import itertools
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("vars", nargs="+", help="NAME=v1,v2,...")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--clip", default="pan_noise")
    a = ap.parse_args()
    from qsvc_amd import hip
    from qsvc_amd.pipeline import CompressParams, Compressor
    from qsvc_amd.synth import make_clip
    W, H = 1920, 1080
    os.environ.setdefault("QSVC_T2_THREADS", "8")
    prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=8, block_size_min=8, search_range=4, SRLs=5, lossless=True)
    e = hip.Engine(0)
    c = Compressor(e, prm, None)
    c.configure()
    c.load(make_clip(a.clip, c.pictures, W, H, 43))

    def step():
        c.analyze()
        return c.texture(0, copy=False)

    names = [v.split("=")[0] for v in a.vars]
    grid = list(itertools.product(*[v.split("=")[1].split(",") for v in a.vars]))
    res = {g: [] for g in grid}
    for r in range(a.rounds):
        for g in (grid if r % 2 == 0 else grid[::-1]):
            for n, v in zip(names, g):
                os.environ[n] = v
            step(); step()
            e.sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            e.sync()
            res[g].append((time.perf_counter() - t0) / a.steps * 1e3)
    for g in grid:
        xs = res[g]
        print(" ".join(f"{n}={v}" for n, v in zip(names, g)), "median %.3f min %.3f ms" % (statistics.median(xs), min(xs)), ["%.2f" % x for x in xs], flush=True)


if __name__ == "__main__":
    main()
