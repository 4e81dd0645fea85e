"""Minimal 16-bit greyscale PNG writer (the reference saves predictions with skimage.io.imsave as uint16,
test.py:96-100; scikit-image is not a dependency here)."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def imsave_uint16(path, img):
    a = np.ascontiguousarray(np.asarray(img).astype(np.uint16))
    if a.ndim != 2:
        raise ValueError("expected a 2-D image, got shape %r" % (a.shape,))
    h, w = a.shape
    rows = a.astype(">u2").tobytes()
    stride = 2 * w
    raw = b"".join(b"\x00" + rows[i * stride:(i + 1) * stride] for i in range(h))      # filter type 0 per scanline
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) \
        + _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b"")
    with open(path, "wb") as fh:
        fh.write(png)


def imread_uint16(path):
    """Inverse of imsave_uint16 (filter 0, greyscale 16-bit only) -- used by the tests."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert depth == 16 and ctype == 0
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    out = np.empty((h, w), np.uint16)
    for i in range(h):
        line = raw[i * (2 * w + 1):(i + 1) * (2 * w + 1)]
        assert line[0] == 0
        out[i] = np.frombuffer(line[1:], dtype=">u2")
    return out
