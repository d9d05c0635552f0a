"""`mctf merge` (trunk/src/split.cpp:40-342 without ANALYZE): even_t, odd_t -> low_{t-1}.  Pure file I/O, as `split`.
Flags as the reference binary takes them from synthesize_step.py:138-146 (getopt prefixes of even_fn/low_fn/odd_fn)."""
from . import common as C


def main(argv=None):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    # getopt_long accepts the unambiguous prefixes --even= / --low= / --odd= the driver passes
    argv = [x.replace("--even=", "--even_fn=").replace("--low=", "--low_fn=").replace("--odd=", "--odd_fn=") for x in argv]
    a = C.parse("Inverse of the lazy wavelet split.", {
        "even_fn": ("even", str), "low_fn": ("low", str), "odd_fn": ("odd", str), "pictures": (33, int),
        "pixels_in_x": (352, int), "pixels_in_y": (288, int)}, argv)
    nB = a["pictures"] // 2
    even = C.read_frames(a["even_fn"], a["pixels_in_x"], a["pixels_in_y"], nB + 1)
    odd = C.read_frames(a["odd_fn"], a["pixels_in_x"], a["pixels_in_y"], nB)
    C.write_frames(a["low_fn"], C.interleave(even, odd))


if __name__ == "__main__":
    main()
