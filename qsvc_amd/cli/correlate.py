"""`mctf correlate` (trunk/src/decorrelate.cpp:199-1078 without ANALYZE): even_t, high_t, motion_t, frame_types_t -> odd_t.
Flags as decorrelate.cpp:232-251."""
from . import common as C


def main(argv=None):
    a = C.parse("Inverse of the prediction step of the temporal lifting.", {
        "block_overlaping": (0, int), "block_size": (16, int), "even_fn": ("even", str), "frame_types_fn": ("frame_types", str),
        "high_fn": ("high", str), "motion_in_fn": ("motion_in", str), "odd_fn": ("odd", str), "pictures": (33, int),
        "pixels_in_x": (352, int), "pixels_in_y": (288, int), "search_range": (4, int), "subpixel_accuracy": (0, int)}, argv)
    from .. import hip
    import numpy as np
    W, H, bs, nB = a["pixels_in_x"], a["pixels_in_y"], a["block_size"], a["pictures"] // 2
    even = C.read_frames(a["even_fn"], W, H, nB + 1)
    high = C.read_frames(a["high_fn"], W, H, nB)
    mv = C.read_motion(a["motion_in_fn"], nB, H // bs, W // bs)
    try:
        types = open(a["frame_types_fn"], "rb").read(nB)
    except OSError:
        C.die(f"unable to read \"{a['frame_types_fn']}\" ... aborting!")
    e = C.engine()
    try:
        e.seq_config(W, H)
        C.apply_mctf_options(e, a)
        e.put_frames(hip.LOW, 0, C.interleave(even, np.zeros((nB, even.shape[1]), dtype=np.uint8)))
        e.put_frames(hip.HIGH, 1, high)
        e.put_motion(hip.MOTION_FILTERED, 1, mv, bs)
        e.put_frame_types(1, types.ljust(nB, b"I"))
        e.correlate(1, 2 * nB + 1, bs, a["search_range"])
        C.write_frames(a["odd_fn"], e.get_frames(hip.LOW, 0, 0, 2 * nB + 1)[1::2])
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
