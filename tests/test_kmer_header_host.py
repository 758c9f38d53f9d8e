"""The kernels' k-mer header compiled for the host (tests/host_kmer_check.cpp) against the oracle: the 2-bit roller, the
first-multiply tables and the table-driven MurmurHash3 of metalign_amd/csrc/mg_kmer.h give, for EVERY k in 1..64 and for
the suffix hashes of the fused kernels, the oracle's canonical hash at every position — lower case, N and other
non-bases, sequences shorter than k.  (The GPU tests check the same code as the device compiles it; this one runs where
there is no GPU.)"""
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SUFFIXES = [(21, 51), (31, 51), (51, 51), (30, 60), (40, 60), (50, 60), (60, 60), (1, 64), (32, 64), (33, 64), (17, 33), (4, 5)]


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kmer") / "host_kmer_check")
    subprocess.check_call(["g++", "-O1", "-std=c++20", "-o", exe, os.path.join(HERE, "host_kmer_check.cpp")])
    return exe


def sequences():
    rng = np.random.default_rng(20261003)
    seqs = []
    for n in (0, 1, 3, 20, 21, 64, 65, 150, 400):
        seqs.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)))
    s = bytearray(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=300))
    for i in (7, 99, 100, 180):
        s[i] = ord("N")
    s[250] = ord("x")
    seqs.append(bytes(s))
    seqs.append(b"A" * 130)                      # its own reverse complement is all T: forward is canonical everywhere
    seqs.append(b"ACGT" * 40)                    # palindromic repeats: ties between the strands
    return seqs


def expected(seq, k):
    out = oracle.kmer_hashes(seq, k)  # (hashes, valid) per START position
    hashes, valid = out
    res = ["-"] * len(seq)
    for start in range(len(hashes)):
        if valid[start]:
            res[start + k - 1] = "%016x" % int(hashes[start])
    return res


def test_every_k_and_every_fused_suffix_equals_the_oracle(checker):
    oracle.build()
    seqs = sequences()
    out = subprocess.run([checker], input=b"\n".join(seqs) + b"\n", capture_output=True, check=True).stdout.decode().splitlines()
    it = iter(out)
    try:
        for seq in seqs:
            assert next(it) == "seq %d" % len(seq)
            for mode in (0, 1):  # hash(min(kmer, revcomp)) | min(hash(kmer), hash(revcomp)) % p  
                assert next(it) == "mode %d" % mode
                oracle.set_hash_mode(mode)
                for k in range(1, 65):
                    got = next(it).split()
                    assert got[0] == str(k)
                    assert got[1:] == expected(seq, k), "mode %d, k = %d, sequence of %d" % (mode, k, len(seq))
                for k, kmax in SUFFIXES:
                    got = next(it).split()
                    assert got[:3] == ["s", str(k), str(kmax)]
                    assert got[3:] == expected(seq, k), "mode %d, suffix k = %d of a %d-roller, sequence of %d" % (mode, k, kmax, len(seq))
    finally:
        oracle.set_hash_mode(0)
