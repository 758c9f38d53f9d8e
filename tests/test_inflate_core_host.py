"""The device inflater's decoder (metalign_amd/csrc/mg_inflate_core.h: DEFLATE / gzip as one wavefront decodes it) compiled for
the HOST and run lane by lane against zlib (tests/host_inflate_check.cpp): whole members at every level and strategy, jobs
entered at block starts the finder reports and resolved against the true window, count-only and overflowing jobs, members,
padding, header fields, stored and fixed blocks, truncated and bit-flipped streams, CRC combination.  (The GPU tests check the
same code as the device compiles it; this one runs where there is no GPU.)"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_the_decoder_as_the_host_compiles_it_equals_zlib(tmp_path):
    exe = str(tmp_path / "host_inflate_check")
    subprocess.check_call(["g++", "-O1", "-std=c++20", "-o", exe, os.path.join(HERE, "host_inflate_check.cpp"), "-lz"])
    out = subprocess.run([exe], capture_output=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith(b"ok "), out.stdout[-2000:].decode("utf-8", "replace")
