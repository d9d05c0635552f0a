"""`mctf texture_compress_fb_j2k` (trunk/src/texture_compress_fb_j2k.py:40-301): one temporal subband file ->
<file>_{Y,U,V}_NNNN.j2c (one raw JPEG 2000 code-stream per picture and component) + the size index <file>.j2c.
The per-picture `demux | split | kdu_compress` process chain (:155-195) becomes one batched GPU encode.
--reversible / --base_step are build extensions (the reference hard-codes Creversible=no, :186)."""
import os

from . import common as C


def header_size(data: bytes) -> int:
    """header_size.c:42-98: count the SOP markers FF 91 00 04; returns the value printed after 'OUT'.

    The tool's scanner consumes the bytes it compares (an FF that fails the 0x91 test is not looked at again as a
    first byte), and numByte = 1 + number of fgetc calls, one of which returns EOF.  Same state machine here, hopping
    from FF to FF with bytes.find instead of running Python per byte."""
    data = bytes(data)
    n = len(data)
    cont, start, first, total = 0, -1, 0, 0
    i = 0
    while True:
        p = data.find(b"\xff", i)
        if p < 0:
            break
        # c1 = data[p]; the next fgetc results are consumed whether they match or not
        if data[p + 1:p + 2] != b"\x91":
            i = p + 2
        elif data[p + 2:p + 3] != b"\x00":
            i = p + 3
        elif data[p + 3:p + 4] != b"\x04":
            i = p + 4
        else:
            num = p + 5  # numByte after the four reads
            if cont == 0:
                first = num - 4 - 1
            else:
                total += 7 if (num - start) == 7 else 6
            start = num
            cont += 1
            i = p + 4
    num = n + 2  # n bytes + the read that returned EOF, counted from 1
    total += 7 if (num - start + 1) == 7 else 6
    return first + total + 2


def encode_file(e, file, pictures, W, H, slopes, nLayers, SRLs, reversible=False, base_step=1.0 / 256, frames=None,
                device_seq=None):
    """Encode `pictures` pictures of `file`; frames already on the device when device_seq = (kind, t)."""
    from .. import hip
    prm = hip.j2k_params(reversible=reversible, levels=max(SRLs - 1, 0), layers=nLayers, slopes=slopes, base_step=base_step)
    if device_seq is None:
        e.seq_config(W, H)
        e.put_frames(hip.LOW, 0, frames if frames is not None else C.read_frames(file, W, H, pictures))
        device_seq = (hip.LOW, 0)
    streams = e.j2k_encode_frames(device_seq[0], device_seq[1], 0, pictures, prm)
    write_streams(file, streams)
    return streams


def write_streams(file, streams):
    """<file>_{Y,U,V}_NNNN.j2c + cumulative size index <file>.j2c (texture_compress_fb_j2k.py:280-301)."""
    total = 0
    with open(file + ".j2c", "w") as idx:
        for i in range(len(streams) // 3):
            for k, comp in enumerate("YUV"):
                cs = streams[3 * i + k]
                with open(f"{file}_{comp}_{i:04d}.j2c", "wb") as f:
                    f.write(cs)
                total += len(cs) - header_size(cs)
            idx.write(str(total) + "\n")


def main(argv=None):
    a = C.parse("Compress the texture data of one temporal subband using JPEG 2000.", {
        "file": ("", str), "nLayers": (5, int), "pictures": (33, int), "pixels_in_x": (352, int), "pixels_in_y": (288, int),
        "quantization": ("45000", str), "subband": (0, int), "SRLs": (5, int), "reversible": (0, int), "base_step": (1.0 / 256, float)}, argv)
    from .. import hip
    slopes = [int(s) for s in a["quantization"].replace('"', "").split(",") if s]
    e = C.engine()
    try:
        encode_file(e, a["file"], a["pictures"], a["pixels_in_x"], a["pixels_in_y"], slopes, a["nLayers"], a["SRLs"],
                    bool(a["reversible"]), a["base_step"])
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


if __name__ == "__main__":
    main()
