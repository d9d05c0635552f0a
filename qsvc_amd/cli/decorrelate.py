"""`mctf decorrelate` (trunk/src/decorrelate.cpp:199-1078 with -D ANALYZE): even_t, odd_t, motion_t ->
high_t, frame_types_t, motion_filtered_t, prediction_<even_fn>.  Flags as decorrelate.cpp:232-251."""
from . import common as C


def main(argv=None):
    a = C.parse("Block-based time-domain motion decorrelation.", {
        "block_overlaping": (0, int), "block_size": (16, int), "even_fn": ("even", str),
        "frame_types_fn": ("frame_types", str), "high_fn": ("high", str), "motion_in_fn": ("motion_in", str),
        "motion_out_fn": ("motion_out", str), "odd_fn": ("odd", str), "pictures": (33, int), "pixels_in_x": (352, int),
        "pixels_in_y": (288, int), "search_range": (4, int), "subpixel_accuracy": (0, int), "always_B": (0, int)}, argv)
    from .. import hip
    W, H, bs, nB = a["pixels_in_x"], a["pixels_in_y"], a["block_size"], a["pictures"] // 2
    low = C.interleave(C.read_frames(a["even_fn"], W, H, nB + 1), C.read_frames(a["odd_fn"], W, H, nB))
    mv = C.read_motion(a["motion_in_fn"], nB, H // bs, W // bs)
    e = C.engine()
    try:
        e.seq_config(W, H, keep_prediction=True)
        C.apply_mctf_options(e, a)
        e.put_frames(hip.LOW, 0, low)
        e.put_motion(hip.MOTION_RAW, 1, mv, bs)
        e.decorrelate(1, low.shape[0], bs, a["search_range"], a["always_B"])
        C.write_frames(a["high_fn"], e.get_frames(hip.HIGH, 1, 0, nB))
        with open(a["frame_types_fn"], "wb") as f:
            f.write(e.get_frame_types(1, nB))
        C.write_motion(a["motion_out_fn"], e.get_motion(hip.MOTION_FILTERED, 1, nB, bs))
        C.write_frames("prediction_" + a["even_fn"], e.get_frames(hip.PRED, 1, 0, nB))
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
