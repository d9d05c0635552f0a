"""2-process sharding check on ONE GPU (both ranks use device 0, halo records host-staged over gloo): S sequences in flight per rank
(argv[2], default 1), one engine and host thread each, their halo records batched into one message per neighbour and level
(sharding.BatchExchange -> qsvc_halo_batch_export_host / _import_host: the record layout and stream ordering of the RCCL batch).
Rank outputs are written to argv[1]; tests/test_gpu_sharding.py compares them with single-process runs."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from qsvc_amd import hip, sharding
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, G = 192, 128, 8
first, last = sharding.shard_range(rank, world, 1, G)
prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=1, block_size=16, block_size_min=16, search_range=4, SRLs=4, lossless=True)
engs = [hip.Engine(0) for _ in range(S)]
transport = sharding.HostExchange(dist, rank, world)
group = sharding.BatchExchange(transport, engs) if S > 1 else None
comps = [Compressor(engs[k], prm, group.member(k) if group else transport) for k in range(S)]
outs = [None] * S


def work(k):
    c = comps[k]
    c.configure(); c.load(make_clip("pan_noise", G * world + 1, W, H, 21 + k)[first:last + 1])
    c.analyze()
    outs[k] = c.texture(0 if rank == 0 else 1)


th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
[t.start() for t in th]; [t.join() for t in th]
assert all(o is not None for o in outs) and (group is None or group.error is None)
for k, e in enumerate(engs):
    out = outs[k]
    np.savez(os.path.join(sys.argv[1], f"rank{rank}.npz" if k == 0 else f"rank{rank}_s{k}.npz"), low3=e.get_frames(hip.LOW, 3, 0, 2),
             high1=e.get_frames(hip.HIGH, 1, 0, 4), high3=e.get_frames(hip.HIGH, 3, 0, 1), nstreams=np.array([len(v) for v in out.values()]),
             low_stream0=np.frombuffer(out[(hip.LOW, 3)][0], dtype=np.uint8))
    e.close()
dist.barrier(); dist.destroy_process_group()
