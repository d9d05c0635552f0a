"""CPU, world_size 2 (and 4: ranks with two neighbours) over gloo: the host-side GOP sharding (qsvc_amd/sharding.py + pipeline.Compressor.analyze)
with a CPU test double as the engine.  Two shards with exchanged boundary records must reproduce the
single-piece analysis of the whole sequence, and agree on the shared boundary picture."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from fake_engine import FakeEngine
from qsvc_amd import sharding
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, G = 96, 64, 4
clip = make_clip("pan_noise", G * world + 1, W, H, 9)
first, last = sharding.shard_range(rank, world, 1, G)
prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=3, GOPs=1, block_size=16, block_size_min=16, search_range=4)
comp = Compressor(FakeEngine(), prm, sharding.HostExchange(dist, rank, world))
comp.configure(); comp.load(clip[first:last + 1])
comp.analyze()
e = comp.e
np.savez(os.path.join(sys.argv[2], f"rank{rank}.npz"), low1=e.get_frames(0, 1, 0, 3), low2=e.get_frames(0, 2, 0, 2),
         high1=e.get_frames(1, 1, 0, 2), high2=e.get_frames(1, 2, 0, 1))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_shards_equal_whole_sequence(oracle, tmp_path, world):
    """world 4: ranks 1 and 2 have a neighbour on either side (two sends and two receives per level in one batch), which is what six
    of the eight ranks of a node do."""
    from qsvc_amd.synth import make_clip
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + 10 * world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(worker), ROOT, str(tmp_path)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    W, H = 96, 64
    clip = make_clip("pan_noise", 4 * world + 1, W, H, 9)
    whole = oracle.analyze(clip, W, H, 3, 16, 4)   # world 2: 9 pictures, levels of 9 and 5 pictures
    rk = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(world - 1):
        np.testing.assert_array_equal(rk[r]["low1"][-1], rk[r + 1]["low1"][0], err_msg=f"level-1 boundary picture {r} | {r + 1} differs")
        np.testing.assert_array_equal(rk[r]["low2"][-1], rk[r + 1]["low2"][0], err_msg=f"level-2 boundary picture {r} | {r + 1} differs")
    for lv, key in ((0, "1"), (1, "2")):
        np.testing.assert_array_equal(np.concatenate([rk[0]["low" + key]] + [z["low" + key][1:] for z in rk[1:]]), whole[lv]["low"])
        np.testing.assert_array_equal(np.concatenate([z["high" + key] for z in rk]), whole[lv]["high"])


WORKER_BATCH = r'''
import os, sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from fake_engine import FakeEngine
from qsvc_amd import sharding
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, G, S = 96, 64, 4, 2
first, last = sharding.shard_range(rank, world, 1, G)
prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=3, GOPs=1, block_size=16, block_size_min=16, search_range=4)
engs = [FakeEngine() for _ in range(S)]
group = sharding.BatchExchange(sharding.HostExchange(dist, rank, world), engs)
comps = [Compressor(engs[k], prm, group.member(k)) for k in range(S)]
for k, c in enumerate(comps):
    c.configure(); c.load(make_clip("pan_noise", G * world + 1, W, H, 9 + k)[first:last + 1])
th = [threading.Thread(target=c.analyze) for c in comps]
[t.start() for t in th]; [t.join() for t in th]
assert group.error is None, group.error
for k, e in enumerate(engs):
    np.savez(os.path.join(sys.argv[2], f"rank{rank}_s{k}.npz"), low1=e.get_frames(0, 1, 0, 3), low2=e.get_frames(0, 2, 0, 2),
             high1=e.get_frames(1, 1, 0, 2), high2=e.get_frames(1, 2, 0, 1))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_shards_two_sequences_each_in_one_batched_exchange(oracle, tmp_path):
    """Two ranks, two sequences in flight on each (one engine and host thread per sequence): the sequences' halo records travel as ONE
    message per neighbour and level (sharding.BatchExchange), and every sequence still equals its single-piece analysis."""
    from qsvc_amd.synth import make_clip
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER_BATCH)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29542", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(worker), ROOT, str(tmp_path)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    W, H = 96, 64
    for k in range(2):
        whole = oracle.analyze(make_clip("pan_noise", 9, W, H, 9 + k), W, H, 3, 16, 4)
        r0, r1 = np.load(tmp_path / f"rank0_s{k}.npz"), np.load(tmp_path / f"rank1_s{k}.npz")
        np.testing.assert_array_equal(np.concatenate([r0["low1"], r1["low1"][1:]]), whole[0]["low"])
        np.testing.assert_array_equal(np.concatenate([r0["high1"], r1["high1"]]), whole[0]["high"])
        np.testing.assert_array_equal(np.concatenate([r0["low2"], r1["low2"][1:]]), whole[1]["low"])
        np.testing.assert_array_equal(np.concatenate([r0["high2"], r1["high2"]]), whole[1]["high"])


def test_shard_ranges():
    from qsvc_amd import sharding
    assert sharding.shard_range(0, 8, 1, 8) == (0, 8)
    assert sharding.shard_range(7, 8, 1, 8) == (56, 64)
    assert sharding.shard_range(1, 2, 2, 8) == (16, 32)
