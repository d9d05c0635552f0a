"""`mctf update` (trunk/src/update.cpp:158-684 with -D ANALYZE): even_t, high_t, motion_filtered_t,
frame_types_t -> low_t.  Flags as update.cpp:191-205."""
from . import common as C


def main(argv=None):
    a = C.parse("Update step of the temporal lifting.", {
        "block_size": (16, int), "even_fn": ("even", str), "frame_types_fn": ("frame_types", str), "high_fn": ("high", str),
        "low_fn": ("low", str), "motion_fn": ("motion", str), "pictures": (33, int), "pixels_in_x": (352, int),
        "pixels_in_y": (288, int), "subpixel_accuracy": (0, int), "update_factor": (1.0 / 4, float)}, argv)
    from .. import hip
    import numpy as np
    W, H, bs, nB = a["pixels_in_x"], a["pixels_in_y"], a["block_size"], a["pictures"] // 2
    even = C.read_frames(a["even_fn"], W, H, nB + 1)
    low = C.interleave(even, np.zeros((nB, even.shape[1]), dtype=np.uint8))  # odd pictures are not read by update
    high = C.read_frames(a["high_fn"], W, H, nB)
    mv = C.read_motion(a["motion_fn"], nB, H // bs, W // bs)
    try:
        types = open(a["frame_types_fn"], "rb").read(nB)
    except OSError:
        C.die(f"unable to read \"{a['frame_types_fn']}\" ... aborting!")
    e = C.engine()
    try:
        e.seq_config(W, H)
        e.put_frames(hip.LOW, 0, low)
        e.put_frames(hip.HIGH, 1, high)
        e.put_motion(hip.MOTION_FILTERED, 1, mv, bs)
        e.put_frame_types(1, types.ljust(nB, b"I"))
        e.update(1, low.shape[0], bs, a["update_factor"])
        C.write_frames(a["low_fn"], e.get_frames(hip.LOW, 1, 0, nB + 1))
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
