#!/usr/bin/env python3
"""RCCL halo exchange self-test.

  python3 tools/rccl_selftest.py             1 rank: unique id, a 1-rank communicator, an exchange without neighbours
  python3 tools/rccl_selftest.py --ranks N   N rank processes, one GPU each (line topology): every rank analyses its GOP shard of one
                                             synthetic sequence with qsvc_halo_exchange (ncclSend / ncclRecv over xGMI) between decorrelate
                                             and update; the parent then analyses the WHOLE sequence on GPU 0 and compares every rank's
                                             subbands with the corresponding pictures byte for byte -- a wrong or missing halo changes the
                                             boundary pictures of low_t.  Skips (exit 0, says so) when fewer than N GPUs are visible.
The parent makes no GPU call before the children have exited (it only counts devices)."""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, BS, S0, G, TRLS = 256, 192, 16, 4, 8, 4  # one GOP of 8 pictures per rank, 3 temporal levels


def analyse(e, hip, clip, exchange=None):
    """-> [(high_t, low_t, types_t)] of the shard resident in engine e."""
    import numpy as np
    e.seq_config(W, H)
    e.put_frames(hip.LOW, 0, clip)
    P, S, res = clip.shape[0], S0, []
    for t in range(1, TRLS):
        e.motion_estimate(t, P, BS, S)
        e.decorrelate(t, P, BS, S)
        if exchange:
            e.halo_exchange(t)
        e.update(t, P, BS, 0.25)
        nB = P // 2
        res.append((e.get_frames(hip.HIGH, t, 0, nB), e.get_frames(hip.LOW, t, 0, nB + 1), e.get_frame_types(t, nB)))
        P, S = (P + 1) // 2, min(2 * S, 128)
    return res


def child(rank, world, uid_hex, outdir):
    import numpy as np
    from qsvc_amd import hip
    from qsvc_amd.synth import make_clip
    clip = make_clip("edge", G * world + 1, W, H, 49)[G * rank:G * rank + G + 1].copy()
    e = hip.Engine(rank)
    e.comm_init(rank, world, bytes.fromhex(uid_hex))
    res = analyse(e, hip, clip, exchange=True)
    for t, (hi, lo, ty) in enumerate(res, 1):
        np.save(os.path.join(outdir, f"r{rank}_high_{t}.npy"), hi)
        np.save(os.path.join(outdir, f"r{rank}_low_{t}.npy"), lo)
        open(os.path.join(outdir, f"r{rank}_types_{t}"), "wb").write(ty)
    e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--child", nargs=4, metavar=("RANK", "WORLD", "UID", "DIR"))
    a = ap.parse_args()
    if a.child:
        child(int(a.child[0]), int(a.child[1]), a.child[2], a.child[3])
        return 0
    if a.ranks <= 1:
        import numpy as np
        from qsvc_amd import hip
        from qsvc_amd.synth import make_clip
        uid = hip.Engine.comm_unique_id()
        print("unique id ok", len(uid))
        e = hip.Engine(0)
        e.comm_init(0, 1, uid)
        print("comm_init(1 rank) ok")
        e.seq_config(64, 64)
        e.put_frames(hip.LOW, 0, make_clip("pan", 5, 64, 64, 1))
        e.motion_estimate(1, 5, 16, 4); e.decorrelate(1, 5, 16, 4)
        e.halo_exchange(1)   # no neighbours: packs nothing, clears both halos
        e.update(1, 5, 16, 0.25)
        print("halo_exchange(no neighbours) ok")
        e.close()
        return 0
    import torch  # device_count() alone does not initialise the GPU runtime
    have = torch.cuda.device_count()
    if have < a.ranks:
        print(f"rccl_selftest: SKIPPED -- {a.ranks} ranks asked for, {have} GPU(s) visible")
        return 0
    with tempfile.TemporaryDirectory() as d:
        # the unique id comes from a short-lived helper process so that this one stays off the GPU
        uid = subprocess.check_output([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from qsvc_amd import hip; "
                                       "print(hip.Engine.comm_unique_id().hex())" % ROOT], text=True).strip()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(a.ranks), uid, d]) for r in range(a.ranks)]
        rcs = [p.wait(timeout=600) for p in procs]
        if any(rcs):
            print("rccl_selftest: FAILED, rank exit codes", rcs)
            return 1
        import numpy as np
        from qsvc_amd import hip
        from qsvc_amd.synth import make_clip
        e = hip.Engine(0)
        whole = analyse(e, hip, make_clip("edge", G * a.ranks + 1, W, H, 49))
        e.close()
        for r in range(a.ranks):
            per = G
            for t, (hi, lo, ty) in enumerate(whole, 1):
                per //= 2  # B pictures per rank at this level
                got_hi, got_lo = np.load(os.path.join(d, f"r{r}_high_{t}.npy")), np.load(os.path.join(d, f"r{r}_low_{t}.npy"))
                assert np.array_equal(got_hi, hi[r * per:(r + 1) * per]), f"rank {r} high_{t}"
                assert np.array_equal(got_lo, lo[r * per:(r + 1) * per + 1]), f"rank {r} low_{t}: boundary pictures differ (halo exchange)"
                assert open(os.path.join(d, f"r{r}_types_{t}"), "rb").read() == ty[r * per:(r + 1) * per], f"rank {r} frame_types_{t}"
        print(f"rccl_selftest: OK -- {a.ranks} ranks, {TRLS - 1} levels, halo records exchanged over RCCL, shards == whole sequence")
    return 0


if __name__ == "__main__":
    sys.exit(main())
