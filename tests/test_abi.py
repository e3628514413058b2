"""The C-ABI library loads and exports every symbol include/alacgpu.h declares (no compute here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "alacgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alacgpu_[a-z_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("alacgpu_create", "alacgpu_destroy", "alacgpu_get_format", "alacgpu_decode_packet",
                 "alacgpu_decode_batch", "alacgpu_decode_batch_device"):
        assert must in names


def test_library_exports_every_declared_symbol(pkg):
    pkg.build()
    L = ctypes.CDLL(pkg.lib_path())
    for name in _declared():
        assert hasattr(L, name), name
    L.alacgpu_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.alacgpu_version()


def test_python_binding_covers_every_symbol(pkg):
    assert sorted(pkg._EXPORTS) == _declared()


def test_code_object_targets_gfx950(pkg):
    so = open(pkg.lib_path(), "rb").read()
    assert b"gfx950" in so and b"alac_decode" in so


def test_config_struct_layout_matches_header(pkg):
    # uint32 + 6 x uint8 + uint16 + 3 x uint32 = 24 bytes, natural alignment
    assert ctypes.sizeof(pkg.PacketConfig) == 24
    assert pkg.PacketConfig.MaxRun.offset == 10 and pkg.PacketConfig.SampleRate.offset == 20


def test_create_rejects_bad_configs_without_touching_the_gpu(pkg):
    """NewPacketDecoder's depth check (decoder.go:91-93) runs before any HIP call."""
    for kw in ({"BitDepth": 13}, {"BitDepth": 8}, {"NumChannels": 0}, {"NumChannels": 9}, {"FrameLength": 0}):
        with pytest.raises(pkg.ErrConfig):
            pkg.NewPacketDecoder(pkg.PacketConfig(**kw))
    try:
        pkg.NewPacketDecoder(pkg.PacketConfig(BitDepth=13))
    except pkg.ErrConfig as e:
        assert e.sentinel == pkg.ErrBitDepth and "13" in str(e)
