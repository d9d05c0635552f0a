"""`mctf un_update` (trunk/src/update.cpp:158-684 without ANALYZE): low_t, high_t, motion_t, frame_types_t -> even_t.
Flags as update.cpp:191-205."""
from . import common as C


def main(argv=None):
    a = C.parse("Undo of the update step of the temporal lifting.", {
        "block_size": (16, int), "even_fn": ("even", str), "frame_types_fn": ("frame_types", str), "high_fn": ("high", str),
        "low_fn": ("low", str), "motion_fn": ("motion", str), "pictures": (33, int), "pixels_in_x": (352, int),
        "pixels_in_y": (288, int), "subpixel_accuracy": (0, int), "update_factor": (1.0 / 4, float)}, argv)
    # --subpixel_accuracy is accepted and, as in update.cpp (which parses it at :266 and never uses it), has no effect: the vectors
    # are applied as whole pixels
    from .. import hip
    W, H, bs, nB = a["pixels_in_x"], a["pixels_in_y"], a["block_size"], a["pictures"] // 2
    low = C.read_frames(a["low_fn"], W, H, nB + 1)
    high = C.read_frames(a["high_fn"], W, H, nB)
    mv = C.read_motion(a["motion_fn"], nB, H // bs, W // bs)
    try:
        types = open(a["frame_types_fn"], "rb").read(nB)
    except OSError:
        C.die(f"unable to read \"{a['frame_types_fn']}\" ... aborting!")
    e = C.engine()
    try:
        e.seq_config(W, H)
        e.put_frames(hip.LOW, 1, low)
        e.put_frames(hip.HIGH, 1, high)
        e.put_motion(hip.MOTION_FILTERED, 1, mv, bs)
        e.put_frame_types(1, types.ljust(nB, b"I"))
        e.un_update(1, 2 * nB + 1, bs, a["update_factor"])
        C.write_frames(a["even_fn"], e.get_frames(hip.LOW, 0, 0, 2 * nB + 1)[0::2])
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
