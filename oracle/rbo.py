"""ctypes binding of the CPU ORACLE (oracle/rb_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg — never from the product package (rna-bloom_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librb_oracle.so")

FWD, CANON, RC = 0, 1, 2
REVCOMP, COUNT_IF_PRESENT, STORE_READ_PAIRS = 1, 2, 4


def build(force=False):
    src = os.path.join(_HERE, "rb_oracle.c")
    hdr = os.path.join(_HERE, "rb_oracle.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


class AddStats(C.Structure):
    _fields_ = [("reads", C.c_int64), ("reads_skipped", C.c_int64), ("segments", C.c_int64),
                ("kmers", C.c_int64), ("pairs", C.c_int64)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u64, i64, i32, u32, f32 = C.c_uint64, C.c_int64, C.c_int, C.c_uint32, C.c_float
    vp, cp = C.c_void_p, C.c_char_p

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("rbo_seed", u64, C.c_uint)
    sig("rbo_mstab", u64, C.c_uint, i32)
    sig("rbo_ntp64", u64, cp, i32)
    sig("rbo_ntp64rc", u64, cp, i32)
    sig("rbo_ntpc64", u64, cp, i32, vp)
    sig("rbo_roll_f", u64, u64, C.c_uint, C.c_uint, i32)
    sig("rbo_roll_r", u64, u64, C.c_uint, C.c_uint, i32)
    sig("rbo_roll_f_back", u64, u64, C.c_uint, C.c_uint, i32)
    sig("rbo_ntm64", None, u64, vp, i32, i32)
    sig("rbo_combine", u64, u64, u64)
    sig("rbo_combine3", u64, u64, u64, u64)
    sig("rbo_hash_region", i64, cp, i64, i64, i32, i32, i32, vp, vp)
    sig("rbo_hash_pairs_region", i64, cp, i64, i64, i32, i32, i32, i32, vp, vp, vp)
    sig("rbo_neighbors", None, u64, u64, C.c_uint, i32, i32, i32, i32, vp, vp, vp)
    sig("rbo_variant", None, u64, u64, C.c_uint, C.c_uint, i32, i32, i32, i32, vp, vp, vp)
    sig("rbo_rng31", u32, u64, u64, u32)
    sig("rbo_minifloat_increment", C.c_uint8, C.c_uint8, u32)
    sig("rbo_minifloat_to_float", f32, C.c_uint8)
    sig("rbo_expected_size", i64, i64, f32, i32)
    sig("rbo_bloom_new", vp, i64, i32)
    sig("rbo_bloom_free", None, vp)
    sig("rbo_bloom_clear", None, vp)
    sig("rbo_bloom_add", None, vp, vp)
    sig("rbo_bloom_lookup", i32, vp, vp)
    sig("rbo_bloom_lookup_then_add", i32, vp, vp)
    sig("rbo_bloom_popcount", i64, vp)
    sig("rbo_bloom_fpr", f32, vp)
    sig("rbo_bloom_bytes", vp, vp, vp)
    sig("rbo_bloom_size", i64, vp)
    sig("rbo_cbf_new", vp, i64, i32)
    sig("rbo_cbf_free", None, vp)
    sig("rbo_cbf_clear", None, vp)
    sig("rbo_cbf_increment", None, vp, vp, u32)
    sig("rbo_cbf_increment_and_get", f32, vp, vp, u32)
    sig("rbo_cbf_get_count", f32, vp, vp)
    sig("rbo_cbf_popcount", i64, vp)
    sig("rbo_cbf_fpr", f32, vp)
    sig("rbo_cbf_bytes", vp, vp, vp)
    sig("rbo_graph_new", vp, i64, i64, i64, i32, i32, i32, i32, i32, i32, u64)
    sig("rbo_graph_free", None, vp)
    sig("rbo_graph_clear", None, vp)
    sig("rbo_graph_set_read_pair_distance", None, vp, i32)
    sig("rbo_graph_init_fragment_pairs", None, vp, i64, i32, i32)
    sig("rbo_graph_max_hash", i32, vp)
    sig("rbo_graph_ordinal", u64, vp)
    sig("rbo_graph_set_ordinal", None, vp, u64)
    for n in ("add", "add_if_absent", "add_count_if_present", "add_dbg_only", "add_count_only",
              "add_read_pair", "add_fragment_pair"):
        sig("rbo_graph_" + n, None, vp, vp)
    sig("rbo_graph_contains", i32, vp, vp)
    sig("rbo_graph_get_count", f32, vp, vp)
    sig("rbo_graph_lookup_read_pair", i32, vp, vp)
    sig("rbo_graph_lookup_fragment_pair", i32, vp, vp)
    for n in ("dbgbf", "cbf", "rpkbf", "fpkbf"):
        sig("rbo_graph_" + n, vp, vp)
    sig("rbo_graph_add_reads", None, vp, vp, vp, vp, i64, i32, C.c_uint, vp)
    sig("rbo_graph_add_reads_mt", None, vp, vp, vp, vp, i64, i32, C.c_uint, i32, vp)
    sig("rbo_graph_add_fastq_mt", None, vp, vp, i64, i64, i32, C.c_uint, i32, vp)
    sig("rbo_segments", i64, vp, vp, i64, i32, i32, vp, i64)
    sig("rbo_graph_get_kmers", i64, vp, vp, i64, vp, vp, vp)
    sig("rbo_graph_neighbors", None, vp, u64, u64, C.c_uint, i32, vp, vp, vp)
    sig("rbo_minimizers", i64, vp, i64, i32, i32, i32, vp, vp)
    sig("rbo_strobemers", i64, vp, i64, i32, i32, i32, i32, vp, vp, vp)
    sig("rbo_randstrobes", i64, vp, i64, i32, i32, i32, i32, i32, vp, vp)
    sig("rbo_strobe3", i64, vp, i64, i32, i32, i32, i32, vp, vp)
    sig("rbo_minimizers_next", i64, vp, i64, i32, i32, i32, vp, vp)
    sig("rbo_minimizer_set", i64, vp, i64, i32, i32, i32, u64, vp)
    sig("rbo_kmer_pair_hashes", i64, vp, i64, i32, i32, i32, vp)
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


def ntm64(base, k, m):
    out = np.zeros(m, np.uint64)
    lib().rbo_ntm64(int(base), _p(out), k, m)
    return out


def hash_region(seq, k, h, mode, start=0, end=None):
    seq = _b(seq)
    end = len(seq) if end is None else end
    n = max(0, end - start - k + 1)
    out = np.zeros((n, h), np.uint64)
    fr = np.zeros((n, 2), np.uint64)
    got = lib().rbo_hash_region(seq, start, end, k, h, mode, _p(out), _p(fr))
    assert got == n
    return out, fr


def hash_pairs_region(seq, k, h, d, mode, start=0, end=None):
    seq = _b(seq)
    end = len(seq) if end is None else end
    n = max(0, end - start - k - d + 1)
    p = np.zeros((n, h), np.uint64)
    l = np.zeros((n, h), np.uint64)
    r = np.zeros((n, h), np.uint64)
    got = lib().rbo_hash_pairs_region(seq, start, end, k, h, d, mode, _p(p), _p(l), _p(r))
    assert got == n
    return p, l, r


def neighbors(f, r, char_out, k, h, canonical, direction):
    of = np.zeros(4, np.uint64)
    orr = np.zeros(4, np.uint64)
    oh = np.zeros((4, h), np.uint64)
    lib().rbo_neighbors(int(f), int(r), char_out, k, h, int(canonical), direction, _p(of), _p(orr), _p(oh))
    return of, orr, oh


def variant(f, r, char_out, char_in, k, h, canonical, side):
    of = np.zeros(1, np.uint64)
    orr = np.zeros(1, np.uint64)
    oh = np.zeros(h, np.uint64)
    lib().rbo_variant(int(f), int(r), char_out, char_in, k, h, int(canonical), side, _p(of), _p(orr), _p(oh))
    return int(of[0]), int(orr[0]), oh


def segments(seq, qual, k, min_q):
    seq = _b(seq)
    cap = len(seq) // max(k, 1) + 2
    out = np.zeros((cap, 2), np.int64)
    sb = C.create_string_buffer(seq, len(seq))
    qb = C.create_string_buffer(_b(qual), len(seq)) if qual is not None else None
    n = lib().rbo_segments(C.cast(sb, C.c_void_p), C.cast(qb, C.c_void_p) if qb else None,
                           len(seq), k, min_q, _p(out), cap)
    return out[:n].copy()


def minimizers(seq, k, w, mode):
    seq = _b(seq)
    n = max(0, len(seq) - k + 1 - w + 1)
    oh = np.zeros(n, np.uint64)
    op = np.zeros(n, np.int64)
    sb = C.create_string_buffer(seq, len(seq))
    got = lib().rbo_minimizers(C.cast(sb, C.c_void_p), len(seq), k, w, mode, _p(oh), _p(op))
    return oh[:got], op[:got]


def strobemers(seq, k, n, wmin, wmax):
    seq = _b(seq)
    cap = max(0, len(seq) - k + 1)
    oh = np.zeros(cap, np.uint64)
    os_ = np.zeros(cap, np.int32)
    oe = np.zeros(cap, np.int32)
    sb = C.create_string_buffer(seq, len(seq))
    got = lib().rbo_strobemers(C.cast(sb, C.c_void_p), len(seq), k, n, wmin, wmax, _p(oh), _p(os_), _p(oe))
    return oh[:got], os_[:got], oe[:got]


def randstrobes(seq, k, n, wmin, wmax, canonical=False, slide=False):
    """StrobeHashIterator.next/get (slide=True: get) or CanonicalStrobeHashIterator.next/get: (hash, positions[n])"""
    seq = _b(seq)
    cap = max(0, len(seq) - k + 1)
    oh = np.zeros(cap, np.uint64); op = np.zeros((cap, n), np.int32)
    sb = C.create_string_buffer(seq, len(seq))
    got = lib().rbo_randstrobes(C.cast(sb, C.c_void_p), len(seq), k, n, wmin, wmax, (1 if canonical else 0) | (2 if slide else 0), _p(oh), _p(op))
    return oh[:got], op[:got]


def strobe3(seq, k, wmin, wmax, canonical=False):
    """Strobe3HashIterator / CanonicalStrobe3HashIterator: (hash, positions[3]) for p = getMin()..getMax()"""
    seq = _b(seq)
    cap = max(0, len(seq) - k + 1)
    oh = np.zeros(cap, np.uint64); op = np.zeros((cap, 3), np.int32)
    sb = C.create_string_buffer(seq, len(seq))
    got = lib().rbo_strobe3(C.cast(sb, C.c_void_p), len(seq), k, wmin, wmax, 1 if canonical else 0, _p(oh), _p(op))
    return oh[:got], op[:got]


def minimizers_next(seq, k, w, mode):
    seq = _b(seq)
    n = max(0, len(seq) - k + 1 - w + 1)
    oh = np.zeros(n, np.uint64); op = np.zeros(n, np.int64)
    sb = C.create_string_buffer(seq, len(seq))
    got = lib().rbo_minimizers_next(C.cast(sb, C.c_void_p), len(seq), k, w, mode, _p(oh), _p(op))
    return oh[:got], op[:got]


def minimizer_set(seq, k, w, mode, stale=0):
    seq = _b(seq)
    out = np.zeros(max(1, len(seq)), np.uint64)
    sb = C.create_string_buffer(seq, max(1, len(seq)))
    got = lib().rbo_minimizer_set(C.cast(sb, C.c_void_p), len(seq), k, w, mode, int(stale), _p(out))
    return out[:got]


def kmer_pair_hashes(seq, k, shift, canonical):
    seq = _b(seq)
    out = np.zeros(max(0, len(seq) - k + 1), np.uint64)
    sb = C.create_string_buffer(seq, len(seq))
    got = lib().rbo_kmer_pair_hashes(C.cast(sb, C.c_void_p), len(seq), k, shift, 1 if canonical else 0, _p(out))
    return out[:got]


def pack_reads(reads, quals=None):
    """list of bytes -> (concatenated uint8 array, qual array or None, int64 offsets[n+1])."""
    lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    seq = np.frombuffer(b"".join(reads), np.uint8).copy() if len(reads) else np.zeros(0, np.uint8)
    q = None
    if quals is not None:
        q = np.frombuffer(b"".join(quals), np.uint8).copy() if len(quals) else np.zeros(0, np.uint8)
        assert q.size == seq.size
    return seq, q, off


class Graph:
    """Sequential oracle graph — mirrors rnabloom.graph.BloomFilterDeBruijnGraph."""

    def __init__(self, dbgbf_bits, cbf_bytes, pkbf_bits, dbg_h=2, cbf_h=2, pk_h=2, k=25,
                 stranded=False, use_read_pairs=True, rng_seed=0):
        self.L = lib()
        self.k, self.stranded = k, stranded
        self.h = max(dbg_h, cbf_h)
        self.pk_h = pk_h
        self.g = self.L.rbo_graph_new(dbgbf_bits, cbf_bytes, pkbf_bits, dbg_h, cbf_h, pk_h, k,
                                      int(stranded), int(use_read_pairs), rng_seed)

    def __del__(self):
        if getattr(self, "g", None):
            self.L.rbo_graph_free(self.g)
            self.g = None

    def clear(self):
        self.L.rbo_graph_clear(self.g)

    def set_read_pair_distance(self, d):
        self.L.rbo_graph_set_read_pair_distance(self.g, d)

    def init_fragment_pairs(self, bits, pk_h, d):
        self.L.rbo_graph_init_fragment_pairs(self.g, bits, pk_h, d)

    def _h(self, h):
        a = np.ascontiguousarray(h, np.uint64)
        return a

    def add(self, h):
        a = self._h(h); self.L.rbo_graph_add(self.g, _p(a))

    def add_if_absent(self, h):
        a = self._h(h); self.L.rbo_graph_add_if_absent(self.g, _p(a))

    def add_count_if_present(self, h):
        a = self._h(h); self.L.rbo_graph_add_count_if_present(self.g, _p(a))

    def add_dbg_only(self, h):
        a = self._h(h); self.L.rbo_graph_add_dbg_only(self.g, _p(a))

    def add_count_only(self, h):
        a = self._h(h); self.L.rbo_graph_add_count_only(self.g, _p(a))

    def add_read_pair(self, h):
        a = self._h(h); self.L.rbo_graph_add_read_pair(self.g, _p(a))

    def add_fragment_pair(self, h):
        a = self._h(h); self.L.rbo_graph_add_fragment_pair(self.g, _p(a))

    def contains(self, h):
        a = self._h(h); return bool(self.L.rbo_graph_contains(self.g, _p(a)))

    def get_count(self, h):
        a = self._h(h); return float(self.L.rbo_graph_get_count(self.g, _p(a)))

    def lookup_read_pair(self, h):
        a = self._h(h); return bool(self.L.rbo_graph_lookup_read_pair(self.g, _p(a)))

    def lookup_fragment_pair(self, h):
        a = self._h(h); return bool(self.L.rbo_graph_lookup_fragment_pair(self.g, _p(a)))

    def add_reads(self, seq, qual, offsets, min_q=3, flags=0, threads=1):
        st = AddStats()
        n = len(offsets) - 1
        self.L.rbo_graph_add_reads_mt(self.g, _p(seq), _p(qual), _p(offsets), n, min_q, flags,
                                      threads, C.byref(st))
        return st

    def add_fastq(self, text, max_read_len, min_q=3, flags=0, threads=1):
        """FASTQ text (uint8 array) through the reader lock + workers; ordinals follow the order records were pulled"""
        st = AddStats()
        self.L.rbo_graph_add_fastq_mt(self.g, _p(text), text.size, max_read_len, min_q, flags, threads, C.byref(st))
        return st

    def _bloom_bytes(self, which):
        b = getattr(self.L, "rbo_graph_" + which)(self.g)
        if not b:
            return None
        n = C.c_int64()
        p = self.L.rbo_bloom_bytes(b, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()

    def dbgbf_bytes(self):
        return self._bloom_bytes("dbgbf")

    def rpkbf_bytes(self):
        return self._bloom_bytes("rpkbf")

    def fpkbf_bytes(self):
        return self._bloom_bytes("fpkbf")

    def cbf_bytes(self):
        c = self.L.rbo_graph_cbf(self.g)
        n = C.c_int64()
        p = self.L.rbo_cbf_bytes(c, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()

    def folds(self):
        """(dbgbf, cbf, rpkbf) digests computed in place (rbo_fold = the host twin of rb_filter_fold): no copy of the filters"""
        L = self.L
        L.rbo_fold.restype = C.c_uint64
        L.rbo_fold.argtypes = [C.c_void_p, C.c_int64]
        out = []
        n = C.c_int64()
        for get in (L.rbo_graph_dbgbf, None, L.rbo_graph_rpkbf):
            if get is None:
                p = L.rbo_cbf_bytes(L.rbo_graph_cbf(self.g), C.byref(n))
            else:
                b = get(self.g)
                if not b:
                    out.append(0); continue
                p = L.rbo_bloom_bytes(b, C.byref(n))
            out.append(int(L.rbo_fold(C.cast(p, C.c_void_p), n.value)))
        return tuple(out)

    def popcounts(self):
        L = self.L
        r = L.rbo_graph_rpkbf(self.g)
        return (L.rbo_bloom_popcount(L.rbo_graph_dbgbf(self.g)), L.rbo_cbf_popcount(L.rbo_graph_cbf(self.g)),
                L.rbo_bloom_popcount(r) if r else 0)

    def fprs(self):
        L = self.L
        r = L.rbo_graph_rpkbf(self.g)
        return (L.rbo_bloom_fpr(L.rbo_graph_dbgbf(self.g)), L.rbo_cbf_fpr(L.rbo_graph_cbf(self.g)),
                L.rbo_bloom_fpr(r) if r else 0.0)

    def get_kmers(self, seq):
        seq = _b(seq)
        n = max(0, len(seq) - self.k + 1)
        f = np.zeros(n, np.uint64); r = np.zeros(n, np.uint64); c = np.zeros(n, np.float32)
        sb = C.create_string_buffer(seq, len(seq))
        got = self.L.rbo_graph_get_kmers(self.g, C.cast(sb, C.c_void_p), len(seq), _p(f), _p(r), _p(c))
        assert got == n
        return f, r, c

    def neighbors(self, f, r, char_out, direction):
        of = np.zeros(4, np.uint64); orr = np.zeros(4, np.uint64); c = np.zeros(4, np.float32)
        self.L.rbo_graph_neighbors(self.g, int(f), int(r), char_out, direction, _p(of), _p(orr), _p(c))
        return of, orr, c


def walk_max_cov(og, seed, direction, bound, min_cov=1.0, target=None, k=25, stranded=False):
    """Restatement of the loop GraphUtils.getMaxCoveragePath runs in each direction (R/util/GraphUtils.java:1603-1620
    to the right with Kmer.getMaxCovSuccessor, :1629-1672 to the left with getMaxCovPredecessor; R/graph/Kmer.java:301-355:
    the FIRST strict maximum of graph.getCount among A,C,G,T with count >= minKmerCov).  Returns (appended bases as
    bytes, counts, reason): 0 no neighbour, 1 the best neighbour equals the target (not appended), 2 it equals a k-mer
    appended before (not appended), 3 bound reached, 4 the seed has a base outside ACGTU."""
    seed = _b(seed).upper().replace(b"U", b"T")
    if len(seed) != k or any(c not in b"ACGT" for c in seed):
        return b"", [], 4
    target = _b(target).upper().replace(b"U", b"T") if target is not None else None
    _, fr = hash_region(seed, k, 1, 1)
    f, r = int(fr[0, 0]), int(fr[0, 1])
    cur = seed
    seen = set()
    out, counts = bytearray(), []
    while len(out) < bound:
        char_out = cur[0] if direction == 0 else cur[-1]
        f4, r4, c4 = og.neighbors(f, r, char_out, direction)
        best = -1
        best_c = -1.0
        for i in range(4):
            if c4[i] >= min_cov and c4[i] > best_c:
                best, best_c = i, float(c4[i])
        if best < 0:
            return bytes(out), counts, 0
        nb = b"ACGT"[best:best + 1]
        nxt = (cur[1:] + nb) if direction == 0 else (nb + cur[:-1])
        if target is not None and nxt == target:
            return bytes(out), counts, 1
        if nxt in seen:
            return bytes(out), counts, 2
        seen.add(nxt)
        out += nb; counts.append(best_c)
        cur, f, r = nxt, int(f4[best]), int(r4[best])
    return bytes(out), counts, 3


def get_max_coverage_path(og, left, right, bound, min_cov=1.0, k=25, low_complexity=None, trace=None):
    """Restatement of GraphUtils.getMaxCoveragePath(graph, left, right, bound, lookahead, minKmerCov)
    (R/util/GraphUtils.java:1591-1675), statement by statement, over the oracle graph.  `low_complexity` is
    SeqUtils.isLowComplexityShort (R/util/SeqUtils.java:499-543), passed in so that this file stays free of host code."""
    def step(cur, direction):
        """Kmer.getMaxCovSuccessor / getMaxCovPredecessor, R/graph/Kmer.java:301-355"""
        _, fr = hash_region(cur, k, 1, 1)
        f4, r4, c4 = og.neighbors(int(fr[0, 0]), int(fr[0, 1]), cur[0] if direction == 0 else cur[-1], direction)
        best, best_c = -1, -1.0
        for i in range(4):
            if c4[i] >= min_cov and c4[i] > best_c:
                best, best_c = i, float(c4[i])
        if best < 0:
            return None
        nb = b"ACGT"[best:best + 1]
        return (cur[1:] + nb) if direction == 0 else (nb + cur[:-1])

    left, right = _b(left), _b(right)
    left_set, left_path = set(), []
    best = left
    for _ in range(bound):
        best = step(best, 0)
        if best is None:
            break
        if best == right:
            if trace is not None: trace.append("from the left")
            return left_path
        if best in left_set:
            break
        left_set.add(best); left_path.append(best)
    right_set, right_path = set(), []
    best = right
    for _ in range(bound):
        best = step(best, 1)
        if best is None:
            break
        if best == left:
            if trace is not None: trace.append("from the right")
            return right_path
        if best in right_set:
            return None
        if best in left_set:
            if low_complexity(best):
                return None
            right_path.insert(0, best)
            for idx in range(len(left_path) - 1, -1, -1):
                if left_path[idx] == best:
                    if trace is not None: trace.append("walks meet")
                    return left_path[:idx] + right_path
        else:
            right_set.add(best); right_path.insert(0, best)
    return None


def greedy_extend(og, source, direction, lookahead, bound, k=25, gate=None, stranded=False):
    """Restatement of GraphUtils.greedyExtendRight / greedyExtendLeft(graph, source, lookahead, bound)
    (R/util/GraphUtils.java:1961-1976 / :1906-1921), greedyExtendRightOnce / LeftOnce (:501-529 / :564-592) and
    getMaxMedianCoverageRight / Left (:248-310 / :375-438), statement by statement over the oracle graph.
    A k-mer here is (bytes, count, f, r).  Returns (appended bases, their counts)."""
    def neighbours(km):
        """Kmer.getSuccessors / getPredecessors(k, numHash, graph): count >= 1, order A,C,G,T (R/graph/Kmer.java:199-255)"""
        b, _, f, r = km
        f4, r4, c4 = og.neighbors(f, r, b[0] if direction == 0 else b[-1], direction)
        out = []
        for i in range(4):
            if gate is not None:     # Kmer.getSuccessors(k, numHash, graph, bf) :257-299: bf.lookup(hVals) first; hVals[0] is canonical
                fi, ri = int(f4[i]), int(r4[i])
                sf, sr = fi - (1 << 64) if fi >> 63 else fi, ri - (1 << 64) if ri >> 63 else ri
                if not gate(fi if (stranded or sf <= sr) else ri):
                    continue
            if c4[i] >= 1:
                nb = b"ACGT"[i:i + 1]
                out.append(((b[1:] + nb) if direction == 0 else (nb + b[:-1]), float(c4[i]), int(f4[i]), int(r4[i])))
        return out

    def max_median_coverage(src):
        nbrs = neighbours(src)
        if not nbrs:
            return 0.0 if lookahead > 0 else src[1]
        path = [src]
        cursor = nbrs.pop(0)
        path.append(cursor)
        frontier = [nbrs]
        best = 0.0
        while frontier:
            if len(path) < lookahead:
                nbrs = neighbours(cursor)
                if nbrs:
                    cursor = nbrs.pop(0)
                    path.append(cursor)
                    frontier.append(nbrs)
                    continue
            if len(path) == lookahead:
                cov = min(km[1] for km in path)
                if best < cov:
                    best = cov
            while frontier:
                nbrs = frontier[-1]
                path.pop()
                if not nbrs:
                    frontier.pop()
                else:
                    cursor = nbrs.pop(0)
                    path.append(cursor)
                    break
        return best

    def extend_once(src):
        cands = neighbours(src)
        if not cands:
            return None
        if len(cands) == 1:
            return cands[0]
        best_cov, best = -1.0, None
        for km in cands:
            c = max_median_coverage(km)
            if c > best_cov:
                best, best_cov = km, c
            elif c == best_cov and km[1] > best[1]:
                best = km
        return best

    source = _b(source)
    _, fr = hash_region(source, k, 1, 1)
    nxt = (source, 0.0, int(fr[0, 0]), int(fr[0, 1]))
    out, counts = bytearray(), []
    for _ in range(bound):
        nxt = extend_once(nxt)
        if nxt is None:
            break
        out += nxt[0][-1:] if direction == 0 else nxt[0][:1]
        counts.append(nxt[1])
    return bytes(out), counts


def naive_extend(og, seed, direction, mode, bound=0, min_cov=1.0, terminators=b"", k=25, cap=4096):
    """GraphUtils.naiveExtendRight / naiveExtendLeft restated statement by statement over the oracle graph, with k-mers as
    byte strings and every count taken from get_kmers of the k-mer STRING (no rolling): R/util/GraphUtils.java:6780-6833
    (terminators, mode 0), :6835-6886 (bounded, mode 1), :6888-6933 (NoBackChecks, mode 2) and their Left twins.
    Kmer.hasDepthLeft / hasDepthRight always return true (R/graph/Kmer.java:407-486).  Returns (appended bases, reason)."""
    seed = _b(seed).upper().replace(b"U", b"T")
    if len(seed) != k or any(c not in b"ACGT" for c in seed):
        return b"", 4

    def count(km):
        return float(og.get_kmers(km)[2][0])

    def neighbours(km):          # getSuccessors / getPredecessors(k, numHash, graph, result, minKmerCov): A,C,G,T order
        out = []
        for c in b"ACGT":
            nb = km[1:] + bytes([c]) if direction == 0 else bytes([c]) + km[:-1]
            if count(nb) >= min_cov:
                out.append(nb)
        return out

    def back_variants(km):       # getLeftVariants (right walk) / getRightVariants (left walk), minKmerCov = 1
        pos = 0 if direction == 0 else k - 1
        return [km[:pos] + bytes([c]) + km[pos + 1:] for c in b"ACGT" if c != km[pos] and count(km[:pos] + bytes([c]) + km[pos + 1:]) >= 1]

    terms = set()
    t = _b(terminators).upper().replace(b"U", b"T")
    for p in range(len(t) - k + 1):
        terms.add(t[p:p + k])
    used, result, length = set(), [], 0
    nbrs = neighbours(seed)
    best = seed
    while nbrs:
        if mode != 2 and back_variants(best):
            return _join(result, direction), 1
        if len(nbrs) == 1:
            cand = nbrs.pop()
        else:
            return _join(result, direction), 2        # hasDepth* is true for every neighbour: the second one ends the walk
        if mode == 0:
            if cand in terms or cand in used:
                return _join(result, direction), 5
            if len(result) >= cap:
                return _join(result, direction), 6
        if mode == 2 and (cand == seed or (result and cand == result[-1])):
            return _join(result, direction), 7
        best = cand
        result.append(best); used.add(best)
        if mode != 0:
            length += 1
            if length > bound:
                return _join(result, direction), 3
        nbrs = neighbours(best)
    return _join(result, direction), 0


def _join(kmers, direction):
    return bytes(km[-1] if direction == 0 else km[0] for km in kmers)


def get_kmers_min_coverage(og, seq, min_cov):
    """Restatement of HashFunction.getKmers(seq, numHash, graph, minCoverage), R/bloom/hash/HashFunction.java:86-134,
    with Python lists standing in for the ArrayLists (object identity matters: `longestSegment != currentSegment`).
    Returns the list of (k-mer index, count) of the returned segment."""
    f, r, c = og.get_kmers(seq)
    current = []
    longest = current
    current_min = longest_min = float("inf")
    longest_len = 0
    for i, x in enumerate(c):
        if x >= min_cov:
            current.append((i, float(x)))
            current_min = min(current_min, float(x))
        elif current:
            if longest is not current:
                n = len(current)
                if n > longest_len or (n == longest_len and current_min > longest_min):
                    longest, longest_min, longest_len = current, current_min, n
            current = []
            current_min = float("inf")
    return longest
