"""`mctf motion_estimate` (trunk/src/motion_estimate.cpp:490-912): even_t, odd_t -> motion_t on the GPU.
Flags as motion_estimate.cpp:516-530 (incl. --border_size and --subpixel_accuracy).  One deliberate difference: an existing
motion file is overwritten (the reference exits 1, which aborts the pipeline)."""
from . import common as C


def main(argv=None):
    a = C.parse("Block-based time-domain motion estimation.", {
        "block_size": (32, int), "border_size": (0, int), "even_fn": ("even", str), "imotion_fn": ("imotion", str),
        "motion_fn": ("motion", str), "odd_fn": ("odd", str), "pictures": (9, int), "pixels_in_x": (352, int),
        "pixels_in_y": (288, int), "search_range": (4, int), "subpixel_accuracy": (0, int)}, argv)
    from .. import hip
    W, H, bs, nB = a["pixels_in_x"], a["pixels_in_y"], a["block_size"], a["pictures"] // 2
    low = C.interleave(C.read_frames(a["even_fn"], W, H, nB + 1), C.read_frames(a["odd_fn"], W, H, nB))
    e = C.engine()
    try:
        e.seq_config(W, H)
        C.apply_mctf_options(e, a)
        e.put_frames(hip.LOW, 0, low)
        e.motion_estimate(1, low.shape[0], bs, a["search_range"])
        C.write_motion(a["motion_fn"], e.get_motion(hip.MOTION_RAW, 1, nB, bs))
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
