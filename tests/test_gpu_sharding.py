"""GPU: two rank processes (both on GPU 0, gloo host-staged halo exchange) == one process over the whole sequence."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gpu_shards_equal_single_process(tmp_path):
    from qsvc_amd import hip
    from qsvc_amd.pipeline import CompressParams, Compressor
    from qsvc_amd.synth import make_clip
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "shard_check.py"), str(tmp_path)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    W, H = 192, 128
    clip = make_clip("pan_noise", 17, W, H, 21)
    e = hip.Engine(0)
    try:
        prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=16, block_size_min=16, search_range=4, SRLs=4, lossless=True)
        comp = Compressor(e, prm)
        comp.configure(); comp.load(clip)
        comp.analyze()
        out = comp.texture()
        low3, high1, high3 = e.get_frames(hip.LOW, 3, 0, 3), e.get_frames(hip.HIGH, 1, 0, 8), e.get_frames(hip.HIGH, 3, 0, 2)
    finally:
        e.close()
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["low3"][-1], r1["low3"][0])
    np.testing.assert_array_equal(np.concatenate([r0["low3"], r1["low3"][1:]]), low3)
    np.testing.assert_array_equal(np.concatenate([r0["high1"], r1["high1"]]), high1)
    np.testing.assert_array_equal(np.concatenate([r0["high3"], r1["high3"]]), high3)
    assert r0["low_stream0"].tobytes() == out[(hip.LOW, 3)][0]
    assert r1["low_stream0"].tobytes() == out[(hip.LOW, 3)][2 * 3]   # rank 1 starts at its second low picture


def test_two_ranks_two_sequences_each_batched_halo_records(tmp_path):
    """Two rank processes with TWO sequences in flight each (what bench.py does with --sessions): the sequences' halo records of a
    level leave as one message per neighbour through qsvc_halo_batch_export_host / _import_host -- the record layout, slot assignment
    and event ordering between the contexts' streams that the RCCL batch (qsvc_halo_exchange_batch) uses -- and each sequence equals
    its single-process run."""
    from qsvc_amd import hip
    from qsvc_amd.pipeline import CompressParams, Compressor
    from qsvc_amd.synth import make_clip
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "shard_check.py"), str(tmp_path), "2"], env=dict(env, RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    W, H = 192, 128
    e = hip.Engine(0)
    try:
        prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=16, block_size_min=16, search_range=4, SRLs=4, lossless=True)
        for k in range(2):
            comp = Compressor(e, prm)
            comp.configure(); comp.load(make_clip("pan_noise", 17, W, H, 21 + k))
            comp.analyze()
            out = comp.texture()
            low3, high1, high3 = e.get_frames(hip.LOW, 3, 0, 3), e.get_frames(hip.HIGH, 1, 0, 8), e.get_frames(hip.HIGH, 3, 0, 2)
            sfx = "" if k == 0 else f"_s{k}"
            r0, r1 = np.load(tmp_path / f"rank0{sfx}.npz"), np.load(tmp_path / f"rank1{sfx}.npz")
            np.testing.assert_array_equal(np.concatenate([r0["low3"], r1["low3"][1:]]), low3, err_msg=f"sequence {k}")
            np.testing.assert_array_equal(np.concatenate([r0["high1"], r1["high1"]]), high1, err_msg=f"sequence {k}")
            np.testing.assert_array_equal(np.concatenate([r0["high3"], r1["high3"]]), high3, err_msg=f"sequence {k}")
            assert r1["low_stream0"].tobytes() == out[(hip.LOW, 3)][2 * 3]
    finally:
        e.close()


def test_one_rank_rccl_communicator_and_batched_exchange_in_a_process_that_imported_torch(tmp_path):
    """What bench.py does at WORLD_SIZE > 1, as far as one GPU allows: torch imported (gloo only), librccl loaded by the library, ONE
    communicator on the first of two contexts, qsvc_halo_exchange_batch on a line of one rank (no neighbour: nothing moves, every
    context's halo slots are cleared) -- the load / init / group-call path must work and change nothing."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
import torch                                   # CPU only, as in bench.py
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
W, H, bs = 128, 96, 16
engs = [hip.Engine(0) for _ in range(2)]
uid = hip.Engine.comm_unique_id()
engs[0].comm_init(0, 1, uid)
want = []
for k, e in enumerate(engs):
    clip = make_clip("pan_noise", 5, W, H, 30 + k)
    e.seq_config(W, H); e.put_frames(hip.LOW, 0, clip)
    e.analyze_step(1, 5, bs, 4)
    want.append(e.get_frames(hip.LOW, 1, 0, 3))
    e.motion_estimate(1, 5, bs, 4); e.decorrelate(1, 5, bs, 4)
hip.Engine.halo_exchange_batch(engs, 1)
for k, e in enumerate(engs):
    e.update(1, 5, bs, 0.25)
    assert np.array_equal(e.get_frames(hip.LOW, 1, 0, 3), want[k]), k
for e in engs:
    e.close()
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
