"""sdmi_write_png (save_images, src/bin/sample/main.rs:118-125) -- host code, no GPU."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest


def _decode(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr, types = 8, b"", None, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body), typ
        types.append(typ)
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    assert types[0] == b"IHDR" and types[-1] == b"IEND"
    w, h = hdr[:2]
    assert hdr[2:] == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("h,w", [(1, 1), (7, 5), (512, 512), (100, 300)])
def test_png_round_trip(tmp_path, h, w):
    from stable_diffusion_burn_amd._capi import check, load_library
    lib = load_library()
    img = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    path = tmp_path / "x.png"
    check(lib.sdmi_write_png(str(path).encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))
    np.testing.assert_array_equal(_decode(path.read_bytes()), img)


def test_png_errors(tmp_path):
    from stable_diffusion_burn_amd import SdmiError
    from stable_diffusion_burn_amd._capi import check, load_library
    lib = load_library()
    img = np.zeros((2, 2, 3), np.uint8)
    with pytest.raises(SdmiError):
        check(lib.sdmi_write_png(str(tmp_path / "no" / "dir.png").encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)), 2, 2))
    with pytest.raises(SdmiError):
        check(lib.sdmi_write_png(str(tmp_path / "z.png").encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 2))
