"""probav_amd.pngio: the 16-bit greyscale PNG writer test.py saves predictions with (the reference uses skimage.io.imsave on uint16
arrays, test.py:96-100; scikit-image is not a dependency here)."""
import struct
import zlib

import numpy as np
import pytest

from probav_amd.pngio import imread_uint16, imsave_uint16


def test_round_trip_and_file_structure(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 65536, size=(384, 384)).astype(np.uint16)
    img[0, 0], img[-1, -1], img[5, 7] = 0, 65535, 0x1234
    path = tmp_path / "imgset1306.png"
    imsave_uint16(str(path), img)
    np.testing.assert_array_equal(imread_uint16(str(path)), img)
    data = path.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    # IHDR: 384 x 384, bit depth 16, colour type 0 (greyscale), no interlace -- what skimage writes for a 2-D uint16 array
    n, tag = struct.unpack(">I4s", data[8:16])
    assert (n, tag) == (13, b"IHDR")
    assert struct.unpack(">IIBBBBB", data[16:29]) == (384, 384, 16, 0, 0, 0, 0)
    # every chunk's CRC is valid, samples are big-endian
    pos, idat = 8, b""
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF, tag
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert tag == b"IEND"
    raw = zlib.decompress(idat)
    assert raw[0] == 0 and raw[1:3] == bytes([0, 0]) and raw[1 + 2 * (384 * 0 + 0):][:2] == b"\x00\x00"
    row5 = raw[5 * (2 * 384 + 1):6 * (2 * 384 + 1)]
    assert row5[1 + 2 * 7:1 + 2 * 7 + 2] == b"\x12\x34"


def test_what_test_py_hands_over(tmp_path):
    """test.py casts the stitched float image (already clipped to [0, 2**16] and rounded) to uint16: 65536 wraps to 0 exactly as
    numpy's astype does in the reference."""
    f = np.array([[0.0, 1.0, 65535.0, 65536.0]], np.float64)
    imsave_uint16(str(tmp_path / "a.png"), f.astype(np.uint16))
    assert imread_uint16(str(tmp_path / "a.png")).tolist() == [[0, 1, 65535, 0]]
    with pytest.raises(ValueError):
        imsave_uint16(str(tmp_path / "b.png"), np.zeros((2, 2, 1), np.uint16))
