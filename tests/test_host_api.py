"""Host-side mirror of the reference API: ParseMagicCookie cases follow tests/error_test.go:81-122."""
import struct

import pytest


def _cookie(fl=4096, ver=0, depth=16, pb=40, mb=10, kb=14, ch=2, maxrun=255, mfb=0, abr=0, rate=44100):
    return struct.pack(">IBBBBBBHIII", fl, ver, depth, pb, mb, kb, ch, maxrun, mfb, abr, rate)


def test_parse_magic_cookie_plain(pkg):
    c = pkg.ParseMagicCookie(_cookie(depth=24, ch=6, rate=96000))
    assert (c.FrameLength, c.BitDepth, c.NumChannels, c.PB, c.MB, c.KB, c.MaxRun, c.SampleRate) == (
        4096, 24, 6, 40, 10, 14, 255, 96000)


def test_parse_magic_cookie_wrapped(pkg):
    frma = struct.pack(">I4s4s", 12, b"frma", b"alac")
    alac = struct.pack(">I4sI", 36, b"alac", 0)
    for pre in (frma, alac, frma + alac):
        assert pkg.ParseMagicCookie(pre + _cookie()).FrameLength == 4096


@pytest.mark.parametrize("bad", [None, b"", b"\0" * 23])
def test_parse_magic_cookie_short(pkg, bad):
    with pytest.raises(pkg.ErrConfig) as e:
        pkg.ParseMagicCookie(bad)
    assert e.value.sentinel == pkg.ErrInvalidCookie


def test_parse_magic_cookie_bad_version(pkg):
    with pytest.raises(pkg.ErrConfig) as e:
        pkg.ParseMagicCookie(_cookie(ver=1))
    assert e.value.sentinel == pkg.ErrUnsupportedVersion


def test_status_word_rebuilds_reference_error_chain(pkg):
    e = pkg.status_error(1 | (2 << 8) | (3 << 12))
    assert str(e) == "decode failed: CPE: entropy decode V: alac: bitstream overrun"
    assert e.sentinel == pkg.ErrBitstreamOverrun and isinstance(e, pkg.ErrDecode)
    assert str(pkg.status_error(5)) == "decode failed: alac: unsupported element type (CCE/PCE)"
    assert str(pkg.status_error(3 | (1 << 8))) == "decode failed: SCE/LFE: alac: invalid frame header"
    assert str(pkg.status_error(1 | (4 << 8))) == "decode failed: FIL: alac: bitstream overrun"


def test_bytes_per_sample(pkg):
    assert [pkg.bytes_per_sample(d) for d in (16, 20, 24, 32)] == [2, 3, 3, 4]
    with pytest.raises(ValueError):
        pkg.bytes_per_sample(13)


def test_missing_extension_fails_loudly(pkg, monkeypatch, tmp_path):
    monkeypatch.setattr(pkg, "_LIB", None)
    monkeypatch.setattr(pkg, "_CSRC", str(tmp_path))
    with pytest.raises(ImportError):
        pkg.lib()
