#!/usr/bin/env python3
"""bench.py -- the `mctf compress` hot path on N MI355X (one process per GPU).

Metric (BASELINE.json): 1080p encode fps (+ EBCOT code-blocks/s), with the dominant kernel priced against the
HBM roofline and the CPU path timed beside it.

Workload (BASELINE.json configs[2]): 1920x1080 4:2:0, one 16-picture GOP pair =
17 pictures per GPU, TRLs=4 (3 MCTF levels: ME + predict + update, block 8, search range 4->8->16), then the
JPEG 2000 stage (reversible 5/3, Clevels=4, 64x64 code-blocks, EBCOT Tier-1, Tier-2 with SOP) for every
temporal subband picture.  `--sessions S` (default 12, with GPU_MAX_HW_QUEUES=12 so that every session's stream has its own hardware queue; round 2: 6 -> 3237, 8 -> 3700, 12 -> 4112, 16 -> 4035 fps; end of round 3, 20-step runs: 10 to 16 sessions within a few per cent of each other around 8 300) such sequences are encoded concurrently per GPU, one encoder context
and host thread each: EBCOT Tier-1 is a long serial chain per code-block that leaves most of the chip idle, and
independent sequences fill it.  One step = every session finishing one whole encode, from pictures
resident in HBM to code-streams in host memory.  Synthetic input (seeded pan + noise), no reference numbers exist
(vs_baseline null).

  python bench.py [--gpus N] [--steps K] [--warmup W]
Launched for N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; torch is
used only for rendezvous/barrier/max-reduce over gloo (it never touches the GPU in these processes).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# One hardware queue per session stream (the HIP runtime's default of 4 makes a fifth stream share a queue and serialise
# behind another session's kernels).  Must be set before the runtime initialises; a caller's own setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, ROOT)

# BASELINE.json configs[2] (the judged default) and configs[3] ("4K 32-frame GOP, 9/7 irreversible, 4 temporal + 5 spatial levels"):
# geometry, TRLs, GOPs, block size, first search range, SRLs, pictures per GPU window, default transform, sessions in flight, PMC file
CONFIGS = {
    3: dict(W=1920, H=1080, TRLS=4, GOPS=2, BS=8, S0=4, SRLS=5, G=16, reversible=1, sessions=12, pmc="r04_pmc_hbm.json", name="1080p"),
    4: dict(W=3840, H=2160, TRLS=5, GOPS=2, BS=16, S0=4, SRLS=6, G=32, reversible=0, sessions=4, pmc="r04_cfg4_pmc_hbm.json", name="4K"),
}
W, H, TRLS, GOPS, BS, S0, SRLS = 1920, 1080, 4, 2, 8, 4, 5   # overwritten from CONFIGS[--config] in main()
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2   # wave64 instructions/s: 1024 SIMD-32 units, 2 cycles each (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_FILE = "r04_pmc_hbm.json"  # tools/refresh_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes, one sequence in flight


def codeblocks_per_frame(w, h, levels, cb=64):
    """64x64 code-blocks of one 4:2:0 picture coded as three single-component streams with `levels` DWT levels (T.800 B.5/B.7)."""
    def plane(pw, ph):
        n, ws, hs = 0, [pw], [ph]
        for _ in range(levels):
            ws.append((ws[-1] + 1) // 2); hs.append((hs[-1] + 1) // 2)
        blocks = lambda bw, bh: ((bw + cb - 1) // cb) * ((bh + cb - 1) // cb) if bw > 0 and bh > 0 else 0
        n += blocks(ws[levels], hs[levels])
        for d in range(levels, 0, -1):
            lw, lh, hw, hh = ws[d], hs[d], ws[d - 1] - ws[d], hs[d - 1] - hs[d]
            n += blocks(hw, lh) + blocks(lw, hh) + blocks(hw, hh)
        return n
    return plane(w, h) + 2 * plane(w // 2, h // 2)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(reversible=True, frames=None):
    """The CPU path on the SAME workload as one GPU step's sequence (17 pictures, 3 MCTF levels, 17 subband pictures coded), one
    core.  MCTF: the reference's own tools (oracle/_ref, compiled from /root/reference at build time) when present, else the
    restatement; JPEG 2000: the T.800 restatement (the reference's Kakadu does not exist here), hence kind "port"."""
    from oracle import oracle as O
    from oracle import j2k as J
    from qsvc_amd.synth import make_clip
    full = GOPS * 2 ** (TRLS - 1) + 1
    # a bounded sample (about 10-30 s of one core): the whole sequence of one GPU step at 1080p (17 pictures, 3 levels); at 4K the
    # first 5 pictures through 2 temporal levels (the per-picture cost of a level does not depend on the level count)
    if frames is None:
        frames, trls = (full, TRLS) if W * H <= 1920 * 1088 else (5, 3)
    else:  # --cpu-baseline-pictures: a prefix of 2^j + 1 pictures through as many temporal levels as it carries (at most TRLS - 1)
        j = (frames - 1).bit_length() - 1
        if frames < 3 or frames > full or (1 << j) + 1 != frames:
            raise SystemExit(f"--cpu-baseline-pictures must be 2^j + 1 between 3 and {full}, got {frames}")
        trls = min(TRLS, j + 1)
    clip = make_clip("pan_noise", frames, W, H, 43)
    t0 = time.perf_counter()
    if O.have_ref():
        lv = O.ref_analyze(clip, W, H, trls, BS, S0)
        mctf_kind = "reference tools (oracle/_ref)"
    else:
        lv = O.analyze(clip, W, H, trls, BS, S0)
        mctf_kind = "restatement (oracle/mctf_oracle.c)"
    t1 = time.perf_counter()
    pics = [f for l in lv for f in l["high"]] + list(lv[-1]["low"])
    assert len(pics) == frames
    ncb = 0
    for f in pics:
        for pl in (f[:W * H].reshape(H, W), f[W * H:W * H + W * H // 4].reshape(H // 2, W // 2), f[W * H + W * H // 4:].reshape(H // 2, W // 2)):
            _, st = J.encode_plane(pl, reversible=bool(reversible), levels=SRLS - 1, want_stats=True)
            ncb += st["codeblocks"]
    t2 = time.perf_counter()
    # A second CPU figure for the JPEG 2000 half, from an optimised production encoder instead of the restatement: OpenJPEG (through
    # Pillow, where the image has it), same planes, same parameters (5/3 lossless or 9/7, Clevels, 64x64 blocks), one thread.  The
    # reference's Kakadu does not exist here; this is the nearest independent CPU codec.  Reported beside `value`, not in it.
    opj = None
    try:
        import io
        from PIL import Image, features
        if features.check_codec("jpg_2000"):
            t3 = time.perf_counter()
            nbytes = 0
            for f in pics:
                for pl in (f[:W * H].reshape(H, W), f[W * H:W * H + W * H // 4].reshape(H // 2, W // 2), f[W * H + W * H // 4:].reshape(H // 2, W // 2)):
                    bio = io.BytesIO()
                    kw = dict(irreversible=not reversible, num_resolutions=SRLS, codeblock_size=(64, 64), no_jp2=True)
                    if not reversible:
                        kw.update(quality_mode="rates", quality_layers=[8])   # (a lossy rate of the order the reference's slope gives; the time barely depends on it)
                    Image.fromarray(pl).save(bio, format="JPEG2000", **kw)
                    nbytes += bio.tell()
            t4 = time.perf_counter()
            opj = {"j2k_s": round(t4 - t3, 3), "frames_per_s_with_reference_mctf": round(frames / ((t1 - t0) + (t4 - t3)), 4), "codestream_bytes": nbytes,
                   "what": "OpenJPEG via Pillow %s, single thread, the same %d planes" % (getattr(__import__("PIL"), "__version__", "?"), 3 * frames)}
    except Exception as ex:  # noqa: BLE001  (no Pillow / no OpenJPEG in the image: the figure is simply absent)
        opj = {"error": repr(ex)}
    return {"value": round(frames / (t2 - t0), 4), "unit": "frames/s", "cores": 1, "kind": "port", "j2k_openjpeg": opj,
            "sample": f"{'the whole' if frames == full else 'the first'} {frames}{'' if frames == full else ' of ' + str(full)} pictures of one step's sequence: {trls - 1} MCTF levels via {mctf_kind} ({t1 - t0:.2f} s) + "
                      f"{'5/3' if reversible else '9/7'} DWT/EBCOT of the {frames} subband pictures via the T.800 restatement "
                      f"({t2 - t1:.2f} s, {ncb} code-blocks); single thread",
            "mctf_s": round(t1 - t0, 3), "j2k_s": round(t2 - t1, 3), "codeblocks_per_s": round(ncb / (t2 - t1), 1)}


def spawn_ranks(n):
    """`python bench.py --gpus N` from a plain shell: N fresh rank processes (one per GPU), started BEFORE this process makes
    any GPU call (it never does).  Rank 0's stdout (the JSON line) is ours; a failing rank fails the whole run."""
    import socket
    import subprocess
    try:
        import torch  # device_count() alone does not initialise the GPU runtime
        have = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        have = None
    if have is not None and have < n and os.environ.get("QSVC_BENCH_DEVICE") is None:  # (test hook: all ranks on one GPU)
        log(f"bench.py: --gpus {n} asked for, {have} visible: refusing (a silent fall-back to fewer GPUs would mislabel the result)")
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = time.time() + 1500
    alive = list(range(n))
    while alive:
        for r in list(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.remove(r)
            if c != 0:
                log(f"bench.py: rank {r} exited with {c}")
                rc = rc or (c if 0 < c < 256 else 1)
        if (rc or time.time() > deadline) and alive:  # one rank failed: the others would wait for it forever
            for r in alive:
                procs[r].terminate()
            for r in alive:
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            rc = rc or 1
            break
        time.sleep(0.05)
    return rc


def cgroup_cpu():
    """CPU time used / throttled by this container so far (cgroup v2 cpu.stat, microseconds) and its quota in CPUs (cpu.max); None
    where the files do not exist.  The GPU boxes of the pool give a one-GPU job 16 CPUs whatever the machine has."""
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        return {"usage_us": int(st["usage_usec"]), "throttled_us": int(st.get("throttled_usec", 0)), "quota_cpus": None if q[0] == "max" else int(q[0]) / int(q[1])}
    except (OSError, KeyError, ValueError, IndexError):
        return None


def task_cpu():
    """CPU seconds of this process's live threads by thread name (/proc/self/task/*/stat: utime + stime)."""
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open(f"/proc/self/task/{tid}/stat").read()
            except OSError:
                continue
            name = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[name] = out.get(name, 0.0) + (int(f[11]) + int(f[12])) / tick
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-pictures", type=int, default=None,
                    help="pictures of the step's sequence the single-core CPU baseline codes (2^j + 1; default: the whole sequence at "
                         "1080p, 5 at 4K).  The baseline is a reported figure, not part of the timed region; `sample` says what ran")
    ap.add_argument("--no-decode", action="store_true", help="skip the non-headline decode extra")
    ap.add_argument("--no-config4", action="store_true", help="skip the config4 block (one 4K 33-picture 9/7 sequence, BASELINE's \"HBM roofline run\") of the default line")
    ap.add_argument("--no-cli", action="store_true", help="skip cli_wall_s (bin/mcj2k compress from low_0 on disk to the .j2c files)")
    ap.add_argument("--no-iso", action="store_true", help="skip the single-sequence pass (profiling runs: exactly warmup + steps encodes per session)")
    ap.add_argument("--clip", default="pan_noise")
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration: 3 = 1080p 16-picture GOP pair, 5/3 lossless (the judged default); "
                         "4 = 4K 32-picture GOP pair, 9/7, TRLs 5, SRLs 6, one layer")
    ap.add_argument("--reversible", type=int, default=None, choices=(0, 1),
                    help="1 = 5/3 reversible lossless, 0 = 9/7 irreversible, one layer (default: 1 for config 3, 0 for config 4)")
    ap.add_argument("--host-cores", type=int, default=0,
                    help="pin this rank to K host cores (cores [local_rank*K, local_rank*K + K)) before any GPU call: what one rank of an "
                         "8-rank node gets of the host (Tier-2, launches) -- e.g. --gpus 1 --host-cores 32 on a 256-core host")
    ap.add_argument("--gops", type=int, default=None,
                    help="GOPs per sequence (the reference's --GOPs: pictures = GOPs * 2^(TRLs-1) + 1; default 2 = the 16-picture window of "
                         "BASELINE config 3); longer sequences make every temporal level's launch and the texture batch fatter")
    ap.add_argument("--sessions", type=int, default=None,
                    help="17-picture sequences encoded concurrently per GPU (one encoder context + host thread each); "
                         "a step = every session finishing one whole encode")
    args = ap.parse_args()
    global W, H, TRLS, GOPS, BS, S0, SRLS, PMC_FILE
    cfg = CONFIGS[args.config]
    W, H, TRLS, GOPS, BS, S0, SRLS, PMC_FILE = cfg["W"], cfg["H"], cfg["TRLS"], cfg["GOPS"], cfg["BS"], cfg["S0"], cfg["SRLS"], cfg["pmc"]
    if args.gops:
        GOPS = args.gops
    if args.reversible is None:
        args.reversible = cfg["reversible"]
    if args.sessions is None:
        args.sessions = cfg["sessions"]
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if args.host_cores > 0:  # before anything touches the GPU runtime (its helper threads inherit the mask)
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        avail = sorted(os.sched_getaffinity(0))
        mine = avail[(lr * args.host_cores) % len(avail):][:args.host_cores] or avail[:args.host_cores]
        os.sched_setaffinity(0, set(mine))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: refusing to measure a different job than asked for")
        sys.exit(2)

    # Native libraries (gloo, RCCL) print banners on fd 1; stdout must carry exactly ONE JSON line, so fd 1 points at
    # stderr until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch  # noqa: F401  (CPU only: rendezvous, barrier, max-reduce)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()

    import numpy as np
    from qsvc_amd import hip, sharding
    from qsvc_amd.pipeline import CompressParams, Compressor
    from qsvc_amd.synth import make_clip

    import threading
    S = max(1, args.sessions)
    # Tier-2 worker threads per encoder context: the host's cores shared by every rank of the node and every session of a rank
    # (8 sessions x 8 workers x 8 ranks would be 512 threads); a caller's own setting wins
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    user_t2 = os.environ.get("QSVC_T2_THREADS")
    ncpu = len(os.sched_getaffinity(0))  # the cores this process may use (a container's share, not the machine's count)
    t2_timed = int(user_t2) if user_t2 else max(1, min(16 if S == 1 else 8, ncpu // max(1, local_world * S)))
    dev = os.environ.get("QSVC_BENCH_DEVICE")  # test hook: several ranks on one GPU (forces the host-staged exchange)
    devno = int(dev) if dev is not None else (local if world > 1 else 0)
    engs = [hip.Engine(devno) for _ in range(S)]
    # The motion-field coding (compress.py:197-212) runs on the sequence's own context, in the textures' batch (qsvc_compress_code).
    # QSVC_BENCH_MOTION=beside is an experiment of round 4: a second context per sequence codes the motion beside the texture
    # coding (measured: 4 450 instead of 6 830 fps with 12 sequences in flight, 14.4 instead of 8.4 ms for one sequence alone -- the two
    # contexts' streams get in each other's way more than the 1.7 ms serial chain they could hide)
    motion_mode = os.environ.get("QSVC_BENCH_MOTION", "batch")
    mv_engs = [hip.Engine(devno) for _ in range(S)] if motion_mode == "beside" else [None] * S
    for e in engs:
        e.set_tier2_threads(t2_timed)  # an explicit per-context setting (include/qsvc_hip.h), not an environment variable read mid-run
    eng = engs[0]
    transport, exchange_kind = None, "none"
    if world > 1:
        import torch
        # RCCL over xGMI: ONE communicator per rank (held by session 0's context), whatever the number of sequences in flight -- their
        # halo records of a level travel as one message per neighbour (qsvc_halo_exchange_batch).  The id travels through gloo.  Every
        # rank takes the same branch at every step, so a failure anywhere degrades ALL ranks to the host-staged transport (a
        # transport fallback, never a compute fallback).
        uids = [None]
        if rank == 0 and dev is None and os.environ.get("QSVC_EXCHANGE", "rccl") == "rccl":
            try:
                uids = [hip.Engine.comm_unique_id()]
            except Exception as ex:
                log(f"[rank 0] RCCL unavailable ({ex})")
        dist.broadcast_object_list(uids, src=0)
        ok = 0
        if uids[0] is not None:
            try:
                transport = sharding.RcclExchange(engs[0], rank, world, uids[0])
                ok = 1
            except Exception as ex:
                log(f"[rank {rank}] qsvc_comm_init failed ({ex})")
        flag = torch.tensor([ok])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 1:
            exchange_kind = "rccl"
        else:
            transport = sharding.HostExchange(dist, rank, world)
            exchange_kind = "gloo-host"

    prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=TRLS, GOPs=GOPS, block_size=BS, block_size_min=BS,
                         search_range=S0, SRLs=SRLS, lossless=bool(args.reversible), reversible=bool(args.reversible), nLayers=1)

    # all sequences of this rank reach a level's exchange point together; the last one runs ONE batched exchange for all of them
    lock = sharding.BatchExchange(transport, engs) if world > 1 else None
    comps = [Compressor(engs[k], copy.deepcopy(prm), lock.member(k) if lock else None, mv_engs[k]) for k in range(S)]
    comp = comps[0]
    P = comp.pictures
    # weak scaling: rank r holds pictures [16r, 16r+16] of one long synthetic sequence (boundary picture shared)
    G = GOPS * 2 ** (TRLS - 1)   # pictures a rank's window advances by (cfg["G"] at the default --gops)
    assert P == G + 1
    seq_local = make_clip(args.clip, G * world + 1, W, H, 43)[G * rank:G * rank + P].copy()

    def variant(clip, k):
        """Session k's sequence: the synthetic clip mirrored (k & 1: left-right, k & 2: top-bottom) and, from the fifth session on,
        shifted in brightness -- 8 distinct sequences of identical statistics without generating 8 clips."""
        if k == 0:
            return clip
        out = np.empty_like(clip)
        for f_in, f_out in zip(clip, out):
            o = 0
            for (w, h) in ((W, H), (W // 2, H // 2), (W // 2, H // 2)):
                pl = f_in[o:o + w * h].reshape(h, w)
                if k & 1:
                    pl = pl[:, ::-1]
                if k & 2:
                    pl = pl[::-1, :]
                f_out[o:o + w * h] = pl.reshape(-1)
                o += w * h
        if k >= 4:
            out = np.clip(out.astype(np.int16) + 5 * (k // 4), 0, 255).astype(np.uint8)
        return out

    for k, c in enumerate(comps):
        c.configure()
        c.load(variant(seq_local, k))
    low_first = 0 if rank == 0 else 1  # the shared boundary picture of low_T is coded by the left shard
    nbytes_of = [0] * S

    phase = {"analyze_calls_return": 0.0, "motion_call": 0.0, "texture_call": 0.0, "n": 0}  # wall clock of session 0's calls (diagnostic)

    mv_bytes_of = [0] * S

    def step(k):
        t_a = time.perf_counter()
        comps[k].analyze()
        t_b = time.perf_counter()
        # compress.py:197-228: the motion fields are coded between the analysis and the textures (decorrelators on the host side of the
        # library, the 16-bit component planes of all levels lossless) -- here in the same batch as the textures
        if motion_mode == "off":    # diagnosis only: the round-3 step without the motion coding (its line is marked)
            t_c = t_b
            out, mcs = comps[k].texture(low_first, copy=False), {}
        elif motion_mode == "inline":   # two calls, the motion planes' Tier-1 chain in front of the textures'
            _, mcs = comps[k].motion(copy=True)
            t_c = time.perf_counter()
            out = comps[k].texture(low_first, copy=False)
        elif mv_engs[k] is None:   # the motion fields' component planes ride in the textures' batch (qsvc_compress_code): ONE call
            t_c = t_b
            out, mcs = comps[k].code(low_first, copy=False)  # code-streams stay in the engine's host buffer (no Python copies)
        else:
            fut = comps[k].motion_async()
            t_c = time.perf_counter()
            out = comps[k].texture(low_first, copy=False)
            _, mcs = fut.result()
        if k == 0:
            phase["analyze_calls_return"] += t_b - t_a; phase["motion_call"] += t_c - t_b; phase["texture_call"] += time.perf_counter() - t_c; phase["n"] += 1
        nbytes_of[k] = sum(len(c) for v in out.values() for c in v)
        mv_bytes_of[k] = sum(len(c) for v in mcs.values() for c in v)

    failures = []

    sess_cpu = [0.0]

    def run_steps(n):
        """n steps; a step = every session completing one whole encode.  Any session's exception fails the run."""
        def work(k):
            c0 = time.thread_time()
            try:
                for _ in range(n):
                    step(k)
                engs[k].sync()
                sess_cpu[0] += time.thread_time() - c0   # (the session thread's own CPU time: Python + the library calls it makes)
            except BaseException as ex:  # noqa: BLE001
                failures.append((k, repr(ex)))
                if lock is not None:
                    lock.abort()
        if S == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
            [t.start() for t in th]
            [t.join() for t in th]
        if failures:
            for k, ex in failures:
                log(f"[rank {rank}] session {k} failed: {ex}")
            os._exit(1)  # no JSON line; peers blocked in a collective are reaped by the launcher
        assert all(b > 0 for b in nbytes_of), f"a session produced no code-stream: {nbytes_of}"

    def barrier():
        for e in engs:
            e.sync()
        if dist is not None:
            dist.barrier()

    # the MQ coder kernel in work: the library's default is the two-wave pipeline (64 code-blocks per wave pair); QSVC_MQ_PIPE=0
    # selects the one-wave form (QSVC_MQ_LANES blocks per wave).  With several sequences in flight the pipeline runs in its small-LDS
    # form (two ring slots, a third of the staging area: its workgroups leave room on a CU for the other sequences' kernels) -- a
    # per-context switch (qsvc_set_tuning); the single-sequence figure below uses the library's default.  A caller's own QSVC_MQ_LDS wins.
    user_lds = os.environ.get("QSVC_MQ_LDS")
    small = (user_lds == "small") if user_lds is not None else S > 1
    for e in engs:
        e.set_tuning("mq_lds_small", int(small))
    mq_symbol = (("k_t1_mq_pipe<64, 2, 116, 48>" if small else "k_t1_mq_pipe<64, 4, 196, 144>") if os.environ.get("QSVC_MQ_PIPE", "1") != "0"
                 else "k_t1_mq<%s>" % os.environ.get("QSVC_MQ_LANES", "64"))
    shard_check = None
    if world > 1:
        # Self-validation of the multi-GPU path, once, before anything is timed: one step with the halo exchange (every rank, in
        # lock-step), then THIS rank analyses on its own the window of its shard plus one GOP on either side, without any exchange
        # (a boundary picture of level t depends on fewer than 2^t pictures on either side, less than one GOP), and the shard's
        # high_t / low_t / frame_types_t must equal the middle of that window byte for byte -- a wrong, missing or mis-ordered halo
        # record changes the boundary pictures of low_t.
        import hashlib
        run_steps(1)
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        got = [(sha(engs[0].get_frames(hip.HIGH, t, 0, Pt // 2)), sha(engs[0].get_frames(hip.LOW, t, 0, Pt // 2 + 1)),
                bytes(engs[0].get_frame_types(t, Pt // 2))) for (t, Pt, St, bst) in comp.levels]
        lo, hi = max(0, G * (rank - 1)), min(G * world, G * (rank + 2))
        wide = make_clip(args.clip, G * world + 1, W, H, 43)[lo:hi + 1].copy()
        chk = hip.Engine(devno)
        chk.seq_config(W, H)
        chk.put_frames(hip.LOW, 0, wide)
        Pw, off, bad = wide.shape[0], G * rank - lo, []
        for (t, Pt, St, bst), (h_sha, l_sha, types) in zip(comp.levels, got):
            chk.analyze_step(t, Pw, bst, St)
            off //= 2
            nBw, nB = Pw // 2, Pt // 2
            if sha(chk.get_frames(hip.HIGH, t, 0, nBw)[off:off + nB]) != h_sha: bad.append(f"high_{t}")
            if sha(chk.get_frames(hip.LOW, t, 0, nBw + 1)[off:off + nB + 1]) != l_sha: bad.append(f"low_{t}")
            if bytes(chk.get_frame_types(t, nBw))[off:off + nB] != types: bad.append(f"frame_types_{t}")
            Pw = (Pw + 1) // 2
        chk.close()
        flag = torch.tensor([0 if bad else 1])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if bad:
            log(f"[rank {rank}] shard check FAILED: {bad} differ from the locally analysed window (halo exchange: {exchange_kind})")
        if flag.item() != 1:
            os._exit(1)  # no JSON line for a run whose shards are wrong
        shard_check = "ok"
    run_steps(args.warmup)
    for e in engs:
        e.set_profiling(True)
    barrier()
    phase.update(analyze_calls_return=0.0, motion_call=0.0, texture_call=0.0, n=0)
    cpu0, tk0 = cgroup_cpu(), task_cpu()
    sess_cpu[0] = 0.0
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    cpu1, tk1, sess_s = cgroup_cpu(), task_cpu(), sess_cpu[0]
    nbytes = nbytes_of[0]
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    timings = eng.timings()
    stats = eng.j2k_stats()
    host_cpu = None
    if cpu0 and cpu1:   # what the timed region cost on the host: the whole container (every session, Tier-2 and HIP runtime thread)
        host_cpu = {"cpu_ms_per_step": round((cpu1["usage_us"] - cpu0["usage_us"]) / 1e3 / max(args.steps, 1), 2),
                    "cpus_busy": round((cpu1["usage_us"] - cpu0["usage_us"]) / 1e6 / dt, 2), "quota_cpus": cpu1["quota_cpus"],
                    "throttled_ms_in_timed_region": round((cpu1["throttled_us"] - cpu0["throttled_us"]) / 1e3, 1),
                    "source": "cgroup v2 cpu.stat / cpu.max of the container" + (" (all %d ranks of this node together)" % local_world if local_world > 1 else "")}
        # who used it: the session threads (their own CPU clocks), then the live threads by name -- Tier-2 workers ("qsvc-t2"), the
        # main thread and whatever the HIP / ROCr runtime keeps (its signal handlers and helpers carry the process name)
        by = {k: tk1.get(k, 0.0) - tk0.get(k, 0.0) for k in tk1}
        host_cpu["cpus_by_thread"] = dict({"session_threads": round(sess_s / dt, 2)}, **{k: round(v / dt, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1]) if v / dt >= 0.05})
    phase_timed = {k: round(v / max(phase["n"], 1) * 1e3, 3) for k, v in phase.items() if k != "n"}
    for e in engs:
        e.set_profiling(False)
    serial_ms, iso = None, {}
    if world == 1 and not args.no_iso:  # ONE sequence alone on the chip: its latency, and every kernel group's ISOLATED duration (HIP events)
        if user_lds is None:
            engs[0].set_tuning("mq_lds_small", 0)
        t2_single = t2_timed
        if user_t2 is None:  # one sequence alone: its Tier-2 may use the cores the other sessions' would
            t2_single = max(1, min(16, ncpu // max(1, local_world)))  # (16: 7.0 ms; 8: 7.6; 32: noisy)
            engs[0].set_tier2_threads(t2_single)
        step(0)
        engs[0].sync()
        ISO_N = 5  # the latency of one sequence: nothing else on the chip, no profiling events in its stream
        t1 = time.perf_counter()
        for _ in range(ISO_N):
            step(0)
        engs[0].sync()
        serial_ms = (time.perf_counter() - t1) / ISO_N * 1e3
        engs[0].set_profiling(True)  # then the kernel groups' isolated durations (HIP events around each group)
        ISO_N = 3
        for _ in range(ISO_N):
            step(0)
        engs[0].sync()
        iso = {k: (ms / ISO_N, n // ISO_N) for k, (ms, n) in engs[0].timings().items()}
        engs[0].set_profiling(False)

    decode_extra = None
    if world == 1 and not args.no_decode:
        # Non-headline extra: the same sequence back from its code-streams -- JPEG 2000 decode of every subband
        # picture (one batch), then the inverse temporal levels (un_update, correlate; synthesize.py) down to low_0, in a fresh
        # context that holds nothing but the streams, the filtered motion and the frame types.  Wall clock from streams in host
        # memory to low_0 in HBM; the decoded planes make the host round trip the C ABI's decode entry point implies.
        try:
            from qsvc_amd import motion as motion_chain
            e0 = engs[0]
            comp.analyze()
            _, mcs = comp.motion()
            mcs = {t: [bytes(c) for c in v] for t, v in mcs.items()}
            out = comp.texture(0, copy=True)
            types = {t: bytes(e0.get_frame_types(t, Pt // 2)) for (t, Pt, St, bst) in comp.levels}
            items = [(hip.HIGH, t, Pt // 2) for (t, Pt, St, bst) in comp.levels] + [(hip.LOW, comp.levels[-1][0], comp.levels[-1][1] // 2 + 1)]
            streams = [cs for (kind, t, n) in items for cs in out[(kind, t)]]
            dec = hip.Engine(devno)
            dec.seq_config(W, H)

            parts = {"motion_streams_and_correlators (beside the textures)": 0.0, "textures": 0.0, "inverse_mctf (incl. waiting for the motion side)": 0.0}

            dec_m = hip.Engine(devno)   # the motion streams on a context (and stream) of their own, beside the textures
            dec_m.seq_config(W, H)
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1)

            def motion_side():
                # motion_expand.py: the residues of every level from their code-streams (one batch), then the two correlators.  (The
                # component planes are the reference's raw 16-bit words: a negative residue is a 16-bit-plane block, 60 000 decisions
                # of ONE lane -- 11 ms whatever else the chip does, which is why this runs beside the textures, not before them.)
                ta = time.perf_counter()
                mplanes = dec_m.j2k_decode_planes([cs for (t, Pt, St, bst) in comp.levels for cs in mcs[t]])
                res, k = {}, 0
                for (t, Pt, St, bst) in comp.levels:
                    nf, BY, BX = Pt // 2, H // bst, W // bst
                    comps4 = np.stack(mplanes[k:k + 4 * nf]).reshape(nf, 4, BY, BX).transpose(1, 0, 2, 3)
                    res[t] = motion_chain.mux_components(np.ascontiguousarray(comps4))
                    k += 4 * nf
                fields = motion_chain.field_chain(res, TRLS)
                parts["motion_streams_and_correlators (beside the textures)"] += time.perf_counter() - ta
                return fields

            def decode_once():
                tb = time.perf_counter()
                fut = pool.submit(motion_side)
                # texture_expand.py: every subband's pictures from their code-streams, one batch, straight into the resident sequences
                dec.j2k_decode_multi([(kind, t, out[(kind, t)]) for (kind, t, n) in items])
                tc = time.perf_counter()
                fields = fut.result()
                for (t, Pt, St, bst) in comp.levels:
                    dec.put_motion(hip.MOTION_FILTERED, t, fields[t], bst)
                    dec.put_frame_types(t, types[t])
                for (t, Pt, St, bst) in reversed(comp.levels):
                    dec.synthesize_step(t, Pt, bst, St)
                dec.sync()
                td = time.perf_counter()
                parts["textures"] += tc - tb; parts["inverse_mctf (incl. waiting for the motion side)"] += td - tc
            decode_once()
            DN = 3
            for k_ in parts:
                parts[k_] = 0.0
            t1 = time.perf_counter()
            for _ in range(DN):
                decode_once()
            dms = (time.perf_counter() - t1) / DN * 1e3
            back = dec.get_frames(hip.LOW, 0, 0, P)
            d = back.astype(np.float64) - variant(seq_local, 0).astype(np.float64)
            mse = float((d * d).mean())
            pool.shutdown()
            dec_m.close()
            dec.close()
            # (lossless texture coding does not make the chain lossless: the reference's update / un_update pair truncates
            #  v + r * update_factor and is not an exact inverse, update.cpp:93-115 -- hence a PSNR, not an equality)
            decode_extra = {"ms_per_sequence": round(dms, 3), "frames_per_s": round(P / (dms * 1e-3), 1), "pictures": P,
                            "parts_ms": {k_: round(v_ / DN * 1e3, 2) for k_, v_ in parts.items()},
                            "psnr_vs_input_db": round(99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse), 2),
                            "what": "what `mcj2k expand` does: the motion residues from their code-streams + the two correlators (on a second context, beside "
                                    "the textures), JPEG 2000 decode of all subband pictures (one batch), inverse MCTF to low_0; one sequence, streams in host memory -> pictures in HBM"}
        except Exception as ex:  # an extra: never fail the headline on it
            decode_extra = {"error": repr(ex)}

    # ---- extras of the default line (rank 0, one GPU, config 3): the drop-in executable including disk I/O, and BASELINE's configs[3] ----
    cli_wall, config4, longer = None, None, None
    if rank == 0 and world == 1 and args.config == 3 and not args.no_cli:
        # SURVEY.md 8d(1) "including disk I/O": `bin/mcj2k compress` on this very configuration, from low_0 on disk to the code-stream
        # files, as a fresh process each time (interpreter start, library load and GPU initialisation included, as for the reference's
        # tools).  First run = cold; then the default form (every intermediate the reference's stages leave behind is written:
        # even_t, odd_t, high_t, low_t, motion_t, prediction_even_t) and the lean form (--keep_intermediates=0: only the code-stream).
        import shutil, subprocess, tempfile
        try:
            d = tempfile.mkdtemp(prefix="qsvc_cli_", dir="/tmp")
            variant(seq_local, 0).tofile(os.path.join(d, "low_0"))
            cmd = [os.path.join(ROOT, "bin", "mcj2k"), "compress", f"--pixels_in_x={W}", f"--pixels_in_y={H}", f"--TRLs={TRLS}", f"--GOPs={GOPS}",
                   f"--block_size={BS}", f"--block_size_min={BS}", f"--SRLs={SRLS}", f"--search_range={S0}",
                   "--lossless=1" if args.reversible else "--nLayers=1"]
            walls = []
            for extra in ([], [], ["--keep_intermediates=0"]):
                for f in os.listdir(d):
                    if f != "low_0":
                        os.remove(os.path.join(d, f))
                t1 = time.perf_counter()
                rr = subprocess.run(cmd + extra, cwd=d, capture_output=True, text=True, timeout=600)
                walls.append(time.perf_counter() - t1)
                if rr.returncode != 0:
                    raise RuntimeError(rr.stderr[-300:])
            nfiles = len([f for f in os.listdir(d) if f.endswith(".j2c")])
            shutil.rmtree(d, ignore_errors=True)
            cli_wall = {"cold_s": round(walls[0], 3), "warm_s": round(walls[1], 3), "warm_code_stream_only_s": round(walls[2], 3), "pictures": P,
                        "frames_per_s_warm": round(P / walls[1], 2), "j2c_files": nfiles, "command": " ".join(["mcj2k"] + cmd[1:]),
                        "what": "wall clock of one fresh `mcj2k compress` process: low_0 read from disk, MCTF + motion coding + texture coding on the "
                                "GPU, every output file written; cold = first run on this box; code_stream_only = --keep_intermediates=0"}
        except Exception as ex:  # an extra: never fail the headline on it
            cli_wall = {"error": repr(ex)}
    if rank == 0 and world == 1 and args.config == 3 and not args.no_config4:
        # BASELINE.json configs[3] ("4K 32-frame GOP, 9/7 irreversible, 4 temporal + 5 spatial levels, HBM roofline run"): ONE 33-picture
        # sequence alone on the chip, 2 warm-up + 4 timed encodes (with one warm-up the first timed encode still ran 15 % long), with its isolated streaming-kernel table -- in a child process (the
        # globals of this one describe config 3), after this one has released its contexts
        import subprocess
        for e in engs + [m for m in mv_engs if m is not None]:
            e.close()
        engs, mv_engs = [], []
        try:
            rr = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "4", "--sessions", "1", "--steps", "4", "--warmup", "2",
                                 "--no-cpu-baseline", "--no-decode", "--no-cli", "--no-config4"], capture_output=True, text=True, timeout=900)
            c4 = json.loads(rr.stdout.strip().splitlines()[-1])
            config4 = {k: c4[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "codeblocks_per_s", "roofline",
                                          "roofline_streaming", "kernels") if k in c4}
            config4["workload"] = c4["config"]["workload"]
            config4["sequences_in_flight"] = 1
        except Exception as ex:
            config4 = {"error": repr(ex)}
        # The same configuration on longer sequences (the reference's --GOPs: one video of several GOPs instead of many two-GOP
        # clips): every temporal level's launch and the texture batch are three times fatter, so the launches that are one round of
        # workgroups long (levels 2-3 of the MCTF, levels 2-4 of the DWT) weigh less.  Reported beside `value`, never in it.
        try:
            rr = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "3", "--sessions", "4", "--gops", "6", "--steps", "10", "--warmup", "3",
                                 "--no-cpu-baseline", "--no-decode", "--no-cli", "--no-config4", "--no-iso"], capture_output=True, text=True, timeout=600)
            lg = json.loads(rr.stdout.strip().splitlines()[-1])
            longer = {"value": lg["value"], "unit": lg["unit"], "ms_per_step": lg["ms_per_step"], "steps": lg["steps"], "warmup": lg["warmup"],
                      "sequences_in_flight_per_gpu": 4, "gops_per_sequence": 6, "pictures_per_step_total": lg["config"]["pictures_per_step_total"],
                      "workload": lg["config"]["workload"]}
        except Exception as ex:
            longer = {"error": repr(ex)}

    if rank == 0:
        pics_total = (G * world + 1) * S   # unique pictures per step: S independent sequences of G*N+1 pictures
        ms_step = dt / args.steps * 1e3
        fps = pics_total * args.steps / dt
        F = W * H * 3 // 2
        ncb_frame = codeblocks_per_frame(W, H, SRLS - 1)
        ncb_step = ncb_frame * P  # per rank
        # ---- algorithmic bytes per STEP of this rank's session 0 (F = 1.5 W H, N = W H) -----------------
        nB = [lv[1] // 2 for lv in comp.levels]      # B pictures per temporal level
        N = W * H
        dwt_key = "j2k_dwt53" if args.reversible else "j2k_dwt97"
        coef_bytes = 2 if args.reversible else 4      # Tier-1 input: int16 (5/3) / 32-bit quantiser indices (9/7), SURVEY.md 8d
        t1_bytes = ncb_step * 4096 * coef_bytes + nbytes   # EBCOT T1: 4096 coefficients per 64x64 block + emitted bytes
        alg = {
            "j2k_t1_mq": t1_bytes, "j2k_t1_cm": t1_bytes,
            "predict": sum(nB) * 4 * F,                # per B picture: read 2 even + 1 odd, write high
            "update": sum(n + 1 for n in nB) * 4 * F - len(nB) * 2 * F,   # per even picture <= 4 F (the two ends of a level 3 F)
            "me_search_l0": sum(nB) * (3 * N + 8 * (H // BS) * (W // BS)),
            dwt_key: int(P * 1.5 * N * ((3 + 4 / 3) if args.reversible else (5 + 8 / 3))),  # u8 in, int16 / float32 out, L levels
        }
        per, host = {}, {}
        for k, (ms, n) in timings.items():
            d = {"ms_per_step": round(ms / args.steps, 4), "launches_per_step": n // max(args.steps, 1)}
            (host if k.startswith("host_") else per)[k] = d
        pmc = {}
        try:  # HBM bytes per encode from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        except Exception:
            pass

        # kernels whose long streams are read 16 bytes per lane: on gfx950 FETCH_SIZE reports HALF of such a stream
        # (MI355X_MICROARCH.md, HBM counters); the MQ front reads its decisions as uint4, the context modelling its magnitude rows and
        # the gather its segments the same way.  Their own data confirm it: k_t1_cm2 writes 278 MB of decisions per sequence and the MQ
        # kernel's raw fetch is 146 MB.
        WIDE16 = ("k_t1_mq_pipe", "k_t1_mq<", "k_t1_cm2", "k_gather")

        def traffic_of(prefixes):
            """HBM bytes per step (= one encode of one sequence) of a kernel group: FETCH_SIZE + WRITE_SIZE of every kernel whose name
            starts with one of `prefixes`, summed over the launches of one encode (tools/pmc_summary.py), with the fetch DOUBLED for the
            kernels of WIDE16 and raw for the others (1-, 4- and 8-byte accesses: calibrated on a kernel with exactly known byte counts,
            DESIGN.md 6)."""
            tot, seen = 0, False
            for kn, e in pmc.items():
                if any(kn.startswith(px) for px in prefixes) and "hbm_bytes_per_step_raw" in e:
                    seen = True
                    tot += e["hbm_bytes_per_step_raw"]
                    if kn.startswith(WIDE16):
                        tot += e["fetch_bytes_per_launch_raw"] * e["launches_profiled"] / max(e.get("encodes_profiled", 5), 1)
            return int(tot) if seen else None

        sq = {}
        try:  # SQ counters of the same kernels (tools/refresh_profiles.sh: three --pmc passes, one sequence in flight)
            sq = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE.replace("pmc_hbm", "pmc_sq"))))
        except Exception:
            pass

        def valu_of(prefixes, iso_ms):
            """Vector-ALU accounting of a kernel group: wave-instructions per step (SQ_INSTS_VALU x launches / encodes profiled), the time
            they take at the chip's vector issue peak (1024 SIMD-32 units x 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12
            wave-instructions/s, MI355X_MICROARCH.md) and their share of the group's isolated duration.  The MCTF and DWT kernels work on
            bytes and 16-bit pairs: the useful rate is lane-operations per byte, not bytes per second."""
            tot, seen = 0.0, False
            for kn, e in sq.items():
                if isinstance(e, dict) and any(kn.startswith(px) for px in prefixes) and "SQ_INSTS_VALU" in e:
                    seen = True
                    tot += e["SQ_INSTS_VALU"] * e.get("launches", 0) / max(e.get("encodes_profiled", 5), 1)
            if not seen:
                return None
            floor_ms = tot / VALU_PEAK_WAVE_INSTS * 1e3
            return {"wave_insts_per_step": int(tot), "floor_ms_at_issue_peak": round(floor_ms, 4),
                    "frac_of_issue_peak": round(floor_ms / iso_ms, 4) if iso_ms and iso_ms > 0 else None}

        dom = "j2k_t1_mq"
        a = alg[dom] / (per[dom]["ms_per_step"] * 1e-3) / 1e9 if dom in per and per[dom]["ms_per_step"] > 0 else 0.0
        mq_traffic = traffic_of([mq_symbol]) or traffic_of(["k_t1_mq_pipe<64, 4, 196, 144>"])
        cm_traffic = traffic_of(["k_t1_cm2"])
        roof = {"kernel": dom, "kernel_symbol": mq_symbol, "bound": "hbm", "achieved": round(a, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 6),
                "traffic": mq_traffic,
                "traffic_source": ("profiles/" + PMC_FILE + " (2 x FETCH_SIZE + WRITE_SIZE per launch: the kernel reads its decisions 16 bytes per "
                                   "lane, the pattern FETCH_SIZE halves on gfx950; separate passes, one sequence in flight)") if mq_traffic is not None
                                  else "missing (kernel symbol not in profiles/" + PMC_FILE + ")",
                "tier1_traffic": (mq_traffic + cm_traffic) if (mq_traffic is not None and cm_traffic is not None) else None,
                "tier1_traffic_note": "context modelling (k_t1_cm2) + MQ coder together, fetch doubled for both: the one-byte-per-decision "
                                      "round trip through HBM on top of the algorithmic bytes",
                "launch_ms": per.get(dom, {}).get("ms_per_step"), "algorithmic_bytes_per_launch": int(alg[dom]),
                "algorithmic_bytes": f"SURVEY.md 8d: {ncb_step} code-blocks x 4096 coefficients x {coef_bytes} B + emitted code bytes, one launch per step",
                "note": "the launch with the largest share of the step; EBCOT Tier-1 is serial-dependency bound, not HBM bound (SURVEY.md H4/8d): "
                        "the fraction is quoted for completeness, the streaming kernels are in roofline_streaming"}
        # the kernels the HBM roofline is meaningful for, ISOLATED (one sequence in flight, measured live after the timed region)
        streaming = None
        if iso:
            groups = {
                dwt_key: ["k_dwt_tile", "k_dwt53_pk", "k_dwt53_rb", "k_dwt97_rb", "k_t1_nbps"],
                "predict": ["k_pred_", "k_mv_hist", "k_chroma_up"],
                "update": ["k_upd_"],
                "me_search_l0": ["k_me_piece<unsigned char", "k_me_search8<unsigned char", "k_me_search<unsigned char"],
            }
            streaming = {}
            for k, prefixes in groups.items():
                if k not in iso or iso[k][0] <= 0:
                    continue
                g = alg[k] / (iso[k][0] * 1e-3) / 1e9
                streaming[k] = {"ms_per_step_isolated": round(iso[k][0], 4), "launch_groups_per_step": iso[k][1],
                                "algorithmic_bytes_per_step": int(alg[k]), "achieved_GBs": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4),
                                "traffic": traffic_of(prefixes), "valu": valu_of(prefixes, iso[k][0])}
            for k, prefixes in (("j2k_t1_cm", ["k_t1_cm2"]), ("j2k_t1_mq", ["k_t1_mq"])):   # Tier-1: not streaming kernels, their vector work beside the others'
                if k in iso and iso[k][0] > 0:
                    streaming[k] = {"ms_per_step_isolated": round(iso[k][0], 4), "valu": valu_of(prefixes, iso[k][0])}
            streaming["note"] = ("isolated = one sequence in flight, HIP events on the context's stream around each kernel group; traffic = HBM bytes "
                                 "per step from profiles/" + PMC_FILE + " (FETCH_SIZE + WRITE_SIZE, raw: 1/4/8-byte accesses; null = not in that file); ME is ALU-bound "
                                 "by design (SURVEY.md 8d), its fraction is quoted for completeness; valu = SQ_INSTS_VALU of the same kernels from profiles/"
                                 + PMC_FILE.replace("pmc_hbm", "pmc_sq") + " against the vector issue peak (1.2288e12 wave-instructions/s): these kernels compute on "
                                 "bytes and 16-bit pairs, 9-30 lane-operations per algorithmic byte, so the vector units and the latency of their dependent "
                                 "phases bound them well before HBM does (DESIGN.md 4, Round 4)")

        t1ms = sum(per.get(k, {}).get("ms_per_step", 0) for k in ("j2k_t1_cm", "j2k_t1_mq"))
        line = {
            "metric": f"{cfg['name']} encode fps (MCTF + {'5/3' if args.reversible else '9/7'} DWT + EBCOT)", "value": round(fps, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int16 (int32 5/3 lifting in LDS, u16 MQ coder)" if args.reversible else "u8/int16 MCTF, f32 9/7 lifting, u16 MQ coder",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} 4:2:0, {P} pictures per GPU ({G}-picture GOP window + shared boundary), TRLs={TRLS} "
                                   f"(ME+predict+update, block {BS}, search {'/'.join(str(lv[2]) for lv in comp.levels)}), JPEG 2000 "
                                   + ("5/3 reversible lossless" if args.reversible else "9/7 irreversible, one layer at the reference's slope")
                                   + f", Clevels={SRLS - 1}, 64x64 code-blocks, SOP; clip=" + args.clip,
                       "baseline_config": args.config,
                       "pictures_per_step_total": pics_total, "sequences_in_flight_per_gpu": S,
                       "parallelism": f"gop-shard x{world}", "halo_exchange": exchange_kind,
                       "rccl_ranks": world if exchange_kind == "rccl" else 0, "shard_check": shard_check,
                       "rccl_communicators_per_rank": 1 if exchange_kind == "rccl" else 0,
                       "halo_messages_per_level_per_neighbour": 1 if world > 1 else 0,
                       "host_threads_per_rank": {"session": S, "motion_coding": S if mv_engs and mv_engs[0] is not None else 0, "tier2_workers": S * t2_timed},
                       "contexts_per_sequence": 2 if mv_engs and mv_engs[0] is not None else 1,
                       "motion_field_coding": {"batch": "in the textures' batch (qsvc_compress_code)", "inline": "its own batch before the textures",
                                               "beside": "on a second context", "off": "NOT CODED (diagnosis run, not a valid step)"}.get(motion_mode, motion_mode),
                       "host_cores_pinned": args.host_cores or None,
                       "tier2_threads_per_context": t2_timed, "host_cpus": len(os.sched_getaffinity(0))},
            "codeblocks_per_s": round(ncb_step * world * S * args.steps / dt, 1),
            "ebcot_codeblocks_per_s_kernel": round(ncb_step / (t1ms * 1e-3), 1) if t1ms else None,
            "codestream_bytes_per_step": int(nbytes), "codestream_bytes_per_step_all_sessions": int(sum(nbytes_of)),
            "motion_codestream_bytes_per_step": int(mv_bytes_of[0]), "mq_symbols_per_step": int(stats["symbols"]),
            "kernels": per, "host_stages": host, "host_cpu": host_cpu, "session0_calls_ms": phase_timed, "roofline": roof, "roofline_streaming": streaming,
        }
        if serial_ms is not None:
            line["single_sequence"] = {"ms_per_encode": round(serial_ms, 3), "frames_per_s": round(P / (serial_ms * 1e-3), 2),
                                       "tier2_threads": t2_single,
                                       "kernels_ms_isolated": {k: round(v[0], 4) for k, v in iso.items()}}
        if decode_extra is not None:
            line["decode"] = decode_extra
        if cli_wall is not None:
            line["cli_wall_s"] = cli_wall
        if config4 is not None:
            line["config4"] = config4
        if longer is not None:
            line["longer_sequences"] = longer
        line["kernels_note"] = ("kernels{} are HIP-event durations on session 0's stream inside the timed region; with several sessions in "
                                "flight kernels of different sessions overlap, so they do not sum to ms_per_step; host_stages{} are wall-clock "
                                "milliseconds of the CPU stages of session 0 (rate allocation, Tier-2)")
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args.reversible, args.cpu_baseline_pictures)
            except Exception as ex:  # the baseline is a reported extra; never fail the GPU measurement on it
                line["cpu_baseline"] = {"error": repr(ex)}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    for e in engs + [m for m in mv_engs if m is not None]:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
