"""2-process sharding check on ONE GPU (both ranks use device 0, halo records host-staged over gloo):
rank outputs are written to argv[1]; tests/test_gpu_sharding.py compares them with a single-process run."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from qsvc_amd import hip, sharding
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, G = 192, 128, 8
clip = make_clip("pan_noise", G * world + 1, W, H, 21)
first, last = sharding.shard_range(rank, world, 1, G)
prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=1, block_size=16, block_size_min=16, search_range=4, SRLs=4, lossless=True)
e = hip.Engine(0)
comp = Compressor(e, prm, sharding.HostExchange(dist, rank, world))
comp.configure(); comp.load(clip[first:last + 1])
comp.analyze()
out = comp.texture(0 if rank == 0 else 1)
np.savez(os.path.join(sys.argv[1], f"rank{rank}.npz"), low3=e.get_frames(hip.LOW, 3, 0, 2), high1=e.get_frames(hip.HIGH, 1, 0, 4),
         high3=e.get_frames(hip.HIGH, 3, 0, 1), nstreams=np.array([len(v) for v in out.values()]),
         low_stream0=np.frombuffer(out[(hip.LOW, 3)][0], dtype=np.uint8))
e.close()
dist.barrier(); dist.destroy_process_group()
