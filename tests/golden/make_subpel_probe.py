#!/usr/bin/env python3
"""Build container only.  Evidence for the refusal of --subpixel_accuracy > 0: run the reference's motion_estimate / decorrelate / update with subpixel_accuracy=1 on seeded clips,
twice under default malloc, once under MALLOC_PERTURB_, and report whether the outputs are run-to-run identical."""
import hashlib, os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from qsvc_amd.synth import make_clip
REF = O.REF_DIR
def run(clip, W, H, bs, S, sub, env_extra=None):
    d = tempfile.mkdtemp(prefix="subpel_")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MALLOC_")}
    if env_extra: env.update(env_extra)
    P = clip.shape[0]
    try:
        clip.tofile(os.path.join(d, "low_0"))
        def call(tool, *a):
            r = subprocess.run([os.path.join(REF, tool)] + [str(x) for x in a], cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return r.returncode
        rc = []
        rc.append(call("split", "--even_fn=even_1", "--low_fn=low_0", "--odd_fn=odd_1", f"--pictures={P}", f"--pixels_in_x={W}", f"--pixels_in_y={H}"))
        rc.append(call("motion_estimate", f"--block_size={bs}", "--border_size=0", "--even_fn=even_1", "--imotion_fn=imotion_1", "--motion_fn=motion_1",
             "--odd_fn=odd_1", f"--pictures={P}", f"--pixels_in_x={W}", f"--pixels_in_y={H}", f"--search_range={S}", f"--subpixel_accuracy={sub}"))
        rc.append(call("decorrelate", "--block_overlaping=0", f"--block_size={bs}", "--even_fn=even_1", "--frame_types_fn=frame_types_1", "--high_fn=high_1",
             "--motion_in_fn=motion_1", "--motion_out_fn=motion_filtered_1", "--odd_fn=odd_1", f"--pictures={P}", f"--pixels_in_x={W}",
             f"--pixels_in_y={H}", f"--search_range={S}", f"--subpixel_accuracy={sub}", "--always_B=0"))
        rc.append(call("update", f"--block_size={bs}", "--even_fn=even_1", "--frame_types_fn=frame_types_1", "--high_fn=high_1", "--low_fn=low_1",
             "--motion_fn=motion_filtered_1", f"--pictures={P}", f"--pixels_in_x={W}", f"--pixels_in_y={H}", f"--subpixel_accuracy={sub}", "--update_factor=0.25"))
        out = {}
        for f in ("motion_1", "motion_filtered_1", "frame_types_1", "high_1", "low_1"):
            p = os.path.join(d, f)
            out[f] = hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) else None
        return rc, out
    finally:
        shutil.rmtree(d, ignore_errors=True)
HERE = os.path.dirname(os.path.abspath(__file__))
report = []
cases = [("pan", 42, 5, 64, 64, 16, 4), ("pan_noise", 43, 5, 128, 96, 16, 4), ("edge", 47, 5, 96, 64, 16, 4), ("saturating", 45, 5, 352, 288, 32, 4)]
for sub in (1, 2):
    for kind, seed, P, W, H, bs, S in cases:
        clip = make_clip(kind, P, W, H, seed)
        a = run(clip, W, H, bs, S, sub); b = run(clip, W, H, bs, S, sub)
        c = run(clip, W, H, bs, S, sub, {"MALLOC_PERTURB_": "165"})
        d = run(clip, W, H, bs, S, sub, {"MALLOC_PERTURB_": "77"})
        print(f"sub={sub} {kind} {W}x{H}: rc={a[0]} repeat_identical={a[1]==b[1]} perturb165_identical={a[1]==c[1]} perturb77_vs_165={c[1]==d[1]}")
        diff = [k for k in a[1] if a[1][k] != c[1][k]]
        print("   files differing under MALLOC_PERTURB_=165:", diff)
        report.append(dict(subpixel_accuracy=sub, kind=kind, seed=seed, pictures=P, W=W, H=H, block_size=bs, search_range=S, exit_codes=a[0],
                           default_malloc=a[1], repeat_identical=a[1] == b[1], malloc_perturb_165=c[1], malloc_perturb_77=d[1],
                           files_depending_on_heap_fill=diff))
import json
json.dump(report, open(os.path.join(HERE, "subpel_probe.json"), "w"), indent=1)
