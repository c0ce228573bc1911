"""The BGZF member walk on the host (gramtools_amd.bgzf_members: what `gram` does before it hands a file to the device-side
decoder, gram_main.cpp: bgzf_member_at): offsets, sizes and trailers of the members without inflating anything (SAM spec 4.1).
No GPU needed."""
import gzip
import struct
import zlib

import pytest


def _bgzf(data: bytes, block=65280, eof=True, extra=b"") -> bytes:
    out = bytearray()
    for piece in [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof else []):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(piece) + c.flush()
        xlen = 6 + len(extra)
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra + b"BC" + struct.pack("<HH", 2, 12 + xlen + len(comp) + 8 - 1)
        out += comp + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


def test_members_are_found_without_inflating():
    from gramtools_amd import bgzf_members
    text = bytes(range(256)) * 700
    for extra in (b"", b"XY\x03\x00abc"):  # another extra subfield in front of BC
        data = _bgzf(text, block=30000, extra=extra)
        mem = bgzf_members(data)
        assert len(mem) == (len(text) + 29999) // 30000  # the empty EOF member is dropped
        back = b""
        for off, size, isize, crc in mem:
            piece = zlib.decompress(data[off:off + size], -15)
            assert len(piece) == isize and zlib.crc32(piece) & 0xFFFFFFFF == crc
            back += piece
        assert back == text
    assert bgzf_members(_bgzf(b"", eof=True)) == []


def test_anything_else_is_refused():
    from gramtools_amd import bgzf_members
    good = _bgzf(b"ACGT" * 5000, block=8000)
    for bad in (gzip.compress(b"ACGT" * 100), good[:-5], good + b"\0" * 20, b"\x1f\x8b\x08\x04" + b"\0" * 10, good[:40] + gzip.compress(b"x") + good[40:]):
        with pytest.raises(ValueError):
            bgzf_members(bad)
