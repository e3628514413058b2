"""The boundary as a plain-C caller and as the cgo shim see it.

* tests/c_abi/smoke.c is compiled as C99 with -Wall -Wextra -pedantic -Werror against include/alacgpu.h only and linked
  with libalacgpu.so: the header is usable from C (what cgo compiles it as), every declared entry point links.
  On the GPU box it decodes the hand-derived KATs through alacgpu_decode_packet / alacgpu_decode_batch.
* go/alacgpu.go (the cgo shim a maintainer drops next to decoder.go, build tag `alacgpu`) is checked against the
  header: every C.alacgpu_* function and C.ALACGPU_* constant it uses is declared there, with the argument count it is
  called with. With a Go toolchain on PATH it is also run through gofmt (this image has none: the check is skipped)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "alacgpu.h")


@pytest.fixture(scope="module")
def smoke_bin(pkg, tmp_path_factory):
    pkg.build()
    out = str(tmp_path_factory.mktemp("c_abi") / "smoke")
    lib_dir = os.path.dirname(pkg.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-o", out, os.path.join(ROOT, "tests", "c_abi", "smoke.c"), "-L", lib_dir, "-lalacgpu",
                           "-Wl,-rpath," + lib_dir])
    return out


def test_header_is_c99_and_every_entry_point_links(smoke_bin):
    r = subprocess.run([smoke_bin, "--link-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "linked" in r.stdout


@pytest.mark.gpu
def test_c99_caller_decodes_the_known_answer_packets(smoke_bin):
    r = subprocess.run([smoke_bin], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 4


def _c_api():
    text = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    funcs = {}
    for m in re.finditer(r"\b(alacgpu_\w+)\s*\(([^;{]*?)\)\s*;", text):
        args = m.group(2).strip()
        funcs[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    consts = set(re.findall(r"\b(ALACGPU_\w+)\b", text))
    types = set(re.findall(r"\b(alacgpu_\w+)\b", text))
    return funcs, consts, types


def test_go_shim_matches_the_header():
    funcs, consts, types = _c_api()
    src = open(os.path.join(ROOT, "go", "alacgpu.go")).read()
    assert src.startswith("//go:build alacgpu")
    code = re.sub(r"//[^\n]*", "", src)
    used = set(re.findall(r"\bC\.(alacgpu_\w+)", code)) | set(re.findall(r"\bC\.(ALACGPU_\w+)", code))
    assert {"alacgpu_create", "alacgpu_destroy", "alacgpu_decode_packet", "alacgpu_decode_batch", "alacgpu_frame_bytes",
            "alacgpu_last_error", "ALACGPU_E_OK", "ALACGPU_E_CONFIG", "ALACGPU_E_DECODE", "ALACGPU_ERR_RANGE"} <= used
    for name in used:
        assert name in funcs or name in consts or name in types, name + " is not in include/alacgpu.h"
    for m in re.finditer(r"\bC\.(alacgpu_\w+)\(", code):  # calls: argument count as declared
        name = m.group(1)
        if name not in funcs:
            continue
        depth, i, n_args, seen = 1, m.end(), 0, False
        while depth:
            c = code[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                n_args += 1
            if depth and not c.isspace():
                seen = True
            i += 1
        assert (n_args + 1 if seen else 0) == funcs[name], name


def test_go_stream_decoder_stays_on_the_shim():
    """go/alacgpu_decoder.go (the read-ahead Decoder) is pure Go on top of GPUPacketDecoder: it must not reach into the C
    ABI itself, and what it uses of the shim must exist there."""
    src = open(os.path.join(ROOT, "go", "alacgpu_decoder.go")).read()
    shim = open(os.path.join(ROOT, "go", "alacgpu.go")).read()
    assert src.startswith("//go:build alacgpu")
    code = re.sub(r"//[^\n]*", "", src)
    assert not re.search(r"\bC\.", code) and 'import "C"' not in code
    for name in ("NewGPUPacketDecoder", "DecodeSamples", "statusErr", "Close", "Format"):
        assert name in code and re.search(r"func (\([^)]*\) )?%s\(" % name, shim), name
    # tabs only, no trailing blanks (gofmt's most visible rules; the real check needs a toolchain)
    for line in src.split("\n"):
        assert line == line.rstrip() and not line.startswith("    "), line


def test_go_files_are_gofmt_clean_when_a_toolchain_exists():
    gofmt = shutil.which("gofmt")
    if not gofmt:
        pytest.skip("no Go toolchain in this image (SURVEY.md §0)")
    r = subprocess.run([gofmt, "-l", os.path.join(ROOT, "go")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "", r.stdout + r.stderr
