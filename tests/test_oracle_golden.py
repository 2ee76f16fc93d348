"""Oracle vs the golden KATs derived from the reference's literal constant tables
(tests/golden/nthash_kat.json, generator tests/golden/gen_golden.py)."""
import numpy as np

from oracle import rbo


def test_seed_tables(golden):
    L = rbo.lib()
    nz = {int(k): int(v, 16) for k, v in golden["seedTab_nonzero"].items()}
    for c in range(256):
        assert L.rbo_seed(c) == nz.get(c, 0), c
    for c, row in golden["msTab_rows"].items():
        for j, v in enumerate(row):
            assert L.rbo_mstab(int(c), j) == int(v, 16)
    M = (1 << 64) - 1
    chk = 0
    for i in range(256):
        for j in range(64):
            chk = (chk + (i * 64 + j + 1) * L.rbo_mstab(i, j)) & M
    assert chk == int(golden["msTab_checksum"], 16)


def test_kat_from_scratch(golden):
    L = rbo.lib()
    for v in golden["kat"]:
        s, k = v["seq"].encode(), v["k"]
        f, r = L.rbo_ntp64(s, k), L.rbo_ntp64rc(s, k)
        assert f == int(v["fwd"], 16) and r == int(v["rev"], 16), v["seq"]
        fr = np.zeros(2, np.uint64)
        assert L.rbo_ntpc64(s, k, fr.ctypes.data) == int(v["canonical"], 16)
        assert [int(x, 16) for x in v["multi4_fwd"]] == [int(x) for x in rbo.ntm64(f, k, 4)]


def test_kat_rolling(golden):
    seq = golden["roll_seq"]
    for blk in golden["roll"]:
        k = blk["k"]
        h, fr = rbo.hash_region(seq, k, 1, rbo.CANON)
        assert [int(x, 16) for x in blk["fwd"]] == [int(x) for x in fr[:, 0]]
        assert [int(x, 16) for x in blk["rev"]] == [int(x) for x in fr[:, 1]]
        hf, _ = rbo.hash_region(seq, k, 1, rbo.FWD)
        hr, _ = rbo.hash_region(seq, k, 1, rbo.RC)
        assert [int(x, 16) for x in blk["fwd"]] == [int(x) for x in hf[:, 0]]
        assert [int(x, 16) for x in blk["rev"]] == [int(x) for x in hr[:, 0]]


def test_main_palindrome_and_u(golden):
    # NTHash.main (R/bloom/hash/NTHash.java:746-754): reverse-palindrome => f == r; U hashes as T
    a, b = golden["kat"][0], golden["kat"][1]
    assert a["fwd"] == a["rev"] == b["fwd"] == b["rev"]
