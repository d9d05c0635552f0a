"""`mctf split` (trunk/src/split.cpp:40-342 with -D ANALYZE): low_{t-1} -> even_t, odd_t (lazy wavelet).
Pure file I/O in the reference and here; on the device the split is an index stride, not a kernel."""
from . import common as C


def main(argv=None):
    a = C.parse("Lazy wavelet split.", {
        "even_fn": ("even", str), "low_fn": ("low", str), "odd_fn": ("odd", str), "pictures": (33, int),
        "pixels_in_x": (352, int), "pixels_in_y": (288, int)}, argv)
    nB = a["pictures"] // 2
    low = C.read_frames(a["low_fn"], a["pixels_in_x"], a["pixels_in_y"], 2 * nB + 1)
    C.write_frames(a["even_fn"], low[0::2])
    C.write_frames(a["odd_fn"], low[1::2])


if __name__ == "__main__":
    main()
