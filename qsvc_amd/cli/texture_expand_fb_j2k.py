"""`mctf texture_expand_fb_j2k` (trunk/src/texture_expand_fb_j2k.py:40-206): <file>_{Y,U,V}_NNNN.j2c -> the temporal subband
file `file` (planar I420 u8 pictures).  The per-picture `kdu_expand` processes (:153-162) become one batched GPU decode
(qsvc_j2k_decode_planes).  Kept as the reference has them: the pictures are APPENDED to `file` (`cat ... >> file`, :178-181),
and a component whose .j2c is missing is replaced by mid-grey 128 of the LUMA size pixels_in_x * pixels_in_y, also for U and V
(:170-176).  --rate (a Kakadu bit-rate cut, "optional and developing" in the reference) is not implemented: refused when > 0."""
import os

import numpy as np

from . import common as C


def expand_file(e, file, pictures, W, H, layers=0):
    names = [f"{file}_{comp}_{i:04d}.j2c" for i in range(pictures) for comp in "YUV"]
    present = [n for n in names if os.path.exists(n)]
    planes = dict(zip(present, e.j2k_decode_planes([open(n, "rb").read() for n in present], max_layers=layers))) if present else {}
    with open(file, "ab") as out:
        for n in names:
            if n in planes:
                out.write(np.ascontiguousarray(planes[n], dtype=np.uint8).tobytes())
            else:
                out.write(bytes([128]) * (W * H))  # the reference's neutral texture: luma-sized for every component


def main(argv=None):
    a = C.parse("Expands the the LFB y HFB texture data using JPEG 2000.", {
        "file": ("", str), "rate": (0.0, float), "pictures": (33, int), "pixels_in_x": (352, int), "pixels_in_y": (288, int),
        "subband": (0, int), "layers": (0, int)}, argv)
    if a["rate"] > 0:
        C.die("--rate (kdu_expand -rate) is not implemented on this path; use --layers to decode fewer quality layers")
    from .. import hip
    e = C.engine()
    try:
        expand_file(e, a["file"], a["pictures"], a["pixels_in_x"], a["pixels_in_y"], a["layers"])
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
