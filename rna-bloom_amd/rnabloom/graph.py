"""rnabloom.graph — host-side mirror of R/graph/BloomFilterDeBruijnGraph.java over the C ABI.

Method names and argument meaning follow the reference class; array arguments are numpy arrays of
BASE hash values (hashVals[0]); the library expands them with NTM64 using the graph's k.
All filter state lives in HBM behind the C handle; nothing is computed on the host.
"""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import check, lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class ReadBatch:
    """A batch of reads resident on the device in the packed 2-bit + validity format."""

    def __init__(self, handle, device):
        self.h = handle
        self.device = device

    @classmethod
    def from_ascii(cls, seq, qual, offsets, min_base_qual=3, device=0):
        seq = np.ascontiguousarray(seq, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.int64)
        q = None if qual is None else np.ascontiguousarray(qual, np.uint8)
        h = C.c_void_p()
        check(lib.rb_batch_create_ascii(device, _ptr(seq), _ptr(q), _ptr(offsets), len(offsets) - 1,
                                        min_base_qual, C.byref(h)))
        return cls(h, device)

    @classmethod
    def from_reads(cls, reads, quals=None, min_base_qual=3, device=0):
        lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
        off = np.zeros(len(reads) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        seq = np.frombuffer(b"".join(reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
        q = None
        if quals is not None:
            q = np.frombuffer(b"".join(quals), np.uint8) if len(quals) else np.zeros(0, np.uint8)
        return cls.from_ascii(seq, q, off, min_base_qual, device)

    @classmethod
    def synthetic(cls, n_pairs, genome_bases, read_len=150, frag_mean=300, frag_sd=30, sub_rate=0.001,
                  n_rate=1e-4, expr_sigma=2.0, seed=0x5EED, tx_min=500, tx_max=4000, device=0, pair_offset=0, total_pairs=0):
        """pair_offset/total_pairs: this batch is the slice [pair_offset, +n_pairs) of a set of total_pairs pairs"""
        p = N.SynthParams(n_pairs, genome_bases, read_len, frag_mean, frag_sd, sub_rate, n_rate, expr_sigma,
                          seed, tx_min, tx_max, pair_offset, total_pairs)
        h = C.c_void_p()
        check(lib.rb_batch_create_synthetic(device, C.byref(p), C.byref(h)))
        return cls(h, device)

    def info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.rb_batch_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"n_reads": a.value, "n_bases": b.value, "device_bytes": c.value}

    @property
    def n_reads(self):
        return self.info()["n_reads"]

    def download(self, first=0, n=None):
        n = self.n_reads - first if n is None else n
        off = np.zeros(n + 1, np.int64)
        check(lib.rb_batch_download_ascii(self.h, first, n, None, _ptr(off)))
        seq = np.zeros(int(off[-1]), np.uint8)
        check(lib.rb_batch_download_ascii(self.h, first, n, _ptr(seq), _ptr(off)))
        return seq, off

    def nthash(self, k, mode, first=0, n=None, with_positions=False):
        """{,Canonical,ReverseComplement}NTHashIterator over every usable segment (mode 0/1/2)."""
        n = self.n_reads - first if n is None else n
        cnt = C.c_int64()
        check(lib.rb_nthash_batch(self.h, k, mode, first, n, C.byref(cnt), None, None, None))
        h0 = np.zeros(cnt.value, np.uint64)
        rd = np.zeros(cnt.value, np.uint32) if with_positions else None
        ps = np.zeros(cnt.value, np.uint32) if with_positions else None
        if cnt.value:
            check(lib.rb_nthash_batch(self.h, k, mode, first, n, C.byref(cnt), _ptr(h0), _ptr(rd), _ptr(ps)))
        return (h0, rd, ps) if with_positions else h0

    def close(self):
        if self.h:
            lib.rb_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BloomFilterDeBruijnGraph:
    """R/graph/BloomFilterDeBruijnGraph.java:75-104 constructor signature (+ device, rngSeed)."""

    def __init__(self, dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k,
                 stranded, useReadPairedKmers, device=0, rngSeed=0, maxBatchKmers=0, groupBits=0):
        self.p = N.GraphParams(dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash,
                               k, int(stranded), int(useReadPairedKmers), device, groupBits, rngSeed, maxBatchKmers)
        self.h = C.c_void_p()
        check(lib.rb_graph_create(C.byref(self.p), C.byref(self.h)))
        self.k = k
        self.stranded = bool(stranded)
        self.device = device

    # ---- lifetime ----
    def destroy(self):
        if self.h:
            lib.rb_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def clearDbgbf(self): check(lib.rb_graph_clear(self.h, 1))
    def clearCbf(self): check(lib.rb_graph_clear(self.h, 2))
    def clearRpkbf(self): check(lib.rb_graph_clear(self.h, 4))
    def clearFpkbf(self): check(lib.rb_graph_clear(self.h, 8))
    def clearAllBf(self): check(lib.rb_graph_clear(self.h, 15))

    # ---- parameters ----
    def getK(self): return self.k
    def isStranded(self): return self.stranded
    def getMaxNumHash(self): return max(self.p.dbgbf_num_hash, self.p.cbf_num_hash)
    def setReadPairedKmerDistance(self, d): check(lib.rb_graph_set_read_paired_kmer_distance(self.h, d))
    def setFragPairedKmerDistance(self, d): check(lib.rb_graph_set_frag_paired_kmer_distance(self.h, d))

    def initializePairKmersBloomFilter(self, pkbfNumBits, pkbfNumHash):
        check(lib.rb_graph_init_fragment_pairs(self.h, pkbfNumBits, pkbfNumHash))

    def getOpOrdinal(self):
        v = C.c_uint64()
        check(lib.rb_graph_get_op_ordinal(self.h, C.byref(v)))
        return v.value

    # ---- stage-1 insert ----
    def addBatch(self, batch, reverseComplement=False, incrementIfPresent=False, storeReadPairedKmers=False,
                 first=0, n=None):
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        st = N.AddStats()
        if n is None and first == 0:
            check(lib.rb_graph_add_batch(self.h, batch.h, flags, C.byref(st)))
        else:
            n = batch.n_reads - first if n is None else n
            check(lib.rb_graph_add_batch_range(self.h, batch.h, first, n, flags, C.byref(st)))
        return st

    def addReads(self, seq, qual, offsets, minBaseQual=3, reverseComplement=False, incrementIfPresent=False,
                 storeReadPairedKmers=False):
        """host ASCII reads (the FastqToGraphWorker boundary): rb_graph_add_reads pins the buffers, uploads and
        2-bit encodes them chunk by chunk, overlapped with the insert pipeline"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        seq = np.ascontiguousarray(np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else seq, dtype=np.uint8)
        if qual is not None:
            qual = np.ascontiguousarray(np.frombuffer(qual, np.uint8) if isinstance(qual, (bytes, bytearray)) else qual, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        st = N.AddStats()
        check(lib.rb_graph_add_reads(self.h, _ptr(seq), _ptr(qual) if qual is not None else None, _ptr(off), off.size - 1, minBaseQual,
                                     flags, C.byref(st)))
        return st

    def addPairs(self, batch, which=N.RPKBF, reverseComplement=False, existingKmersOnly=False, first=0, n=None):
        """PairedKmersToGraphWorker (R/RNABloom.java:436-524): only the paired k-mers of the reads, into rpkbf (or fpkbf)"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_PAIRS_IF_PRESENT if existingKmersOnly else 0)
        n = batch.n_reads - first if n is None else n
        st = N.AddStats()
        check(lib.rb_graph_add_pairs(self.h, batch.h, first, n, which, flags, C.byref(st)))
        return st

    def addFragments(self, batch, loadPairedKmers=True, first=0, n=None):
        """FragmentsToGraphWorker (R/RNABloom.java:1463-1539): k-mers into dbgbf only, plus read- and fragment-paired k-mers"""
        n = batch.n_reads - first if n is None else n
        st = N.AddStats()
        check(lib.rb_graph_add_fragments(self.h, batch.h, first, n, int(loadPairedKmers), C.byref(st)))
        return st

    # ---- per-hash mutators (arrays are applied in order) ----
    def _apply(self, op, h0):
        a = _u64(np.atleast_1d(h0))
        check(lib.rb_graph_apply(self.h, op, _ptr(a), a.size))

    def add(self, h0): self._apply(N.OP_ADD, h0)
    def addIfAbsent(self, h0): self._apply(N.OP_ADD_IF_ABSENT, h0)
    def addCountIfPresent(self, h0): self._apply(N.OP_ADD_COUNT_IF_PRESENT, h0)
    def addDbgOnly(self, h0): self._apply(N.OP_ADD_DBG_ONLY, h0)
    def addCountOnly(self, h0): self._apply(N.OP_ADD_COUNT_ONLY, h0)
    def addReadSingleKmerPair(self, h0): self._apply(N.OP_ADD_READ_PAIR, h0)
    def addFragmentSingleKmerPair(self, h0): self._apply(N.OP_ADD_FRAG_PAIR, h0)

    # ---- queries ----
    def contains(self, h0):
        a = _u64(np.atleast_1d(h0)); out = np.zeros(a.size, np.uint8)
        check(lib.rb_graph_contains(self.h, _ptr(a), a.size, _ptr(out)))
        return out.astype(bool)

    def getCount(self, h0):
        a = _u64(np.atleast_1d(h0)); out = np.zeros(a.size, np.float32)
        check(lib.rb_graph_count(self.h, _ptr(a), a.size, _ptr(out)))
        return out

    def _lookup(self, which, h0):
        a = _u64(np.atleast_1d(h0)); out = np.zeros(a.size, np.uint8)
        check(lib.rb_filter_lookup(self.h, which, _ptr(a), a.size, _ptr(out)))
        return out.astype(bool)

    def lookupThenAdd(self, which, h0):
        """BloomFilter.lookupThenAdd for an array, in array order (R/bloom/BloomFilter.java:147-155)"""
        a = _u64(np.atleast_1d(h0)); out = np.zeros(a.size, np.uint8)
        check(lib.rb_filter_lookup_then_add(self.h, which, _ptr(a), a.size, _ptr(out)))
        return out.astype(bool)

    def lookupReadKmerPair(self, pairHash): return self._lookup(N.RPKBF, pairHash)
    def lookupFragmentKmerPair(self, pairHash): return self._lookup(N.FPKBF, pairHash)

    def getCbfCount(self, h0):
        a = _u64(np.atleast_1d(h0)); out = np.zeros(a.size, np.float32)
        check(lib.rb_filter_get_count(self.h, _ptr(a), a.size, _ptr(out)))
        return out

    def getKmers(self, reads):
        """getKmers(String) for a list of sequences -> (koffsets, f, r, count)."""
        lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
        off = np.zeros(len(reads) + 1, np.int64); np.cumsum(lens, out=off[1:])
        seq = np.frombuffer(b"".join(reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
        ko = np.zeros(len(reads) + 1, np.int64)
        check(lib.rb_graph_kmers(self.h, _ptr(seq), _ptr(off), len(reads), _ptr(ko), None, None, None))
        t = int(ko[-1])
        f = np.zeros(t, np.uint64); r = np.zeros(t, np.uint64); c = np.zeros(t, np.float32)
        if t:
            check(lib.rb_graph_kmers(self.h, _ptr(seq), _ptr(off), len(reads), _ptr(ko), _ptr(f), _ptr(r), _ptr(c)))
        return ko, f, r, c

    def getNeighbors(self, f, r, charOut, direction):
        """4 successors (direction 0) / predecessors (1) of each k-mer: (f4, r4, count4) shaped [n,4]."""
        f = _u64(np.atleast_1d(f)); r = _u64(np.atleast_1d(r))
        ch = np.ascontiguousarray(np.atleast_1d(charOut), np.uint8)
        n = f.size
        f4 = np.zeros((n, 4), np.uint64); r4 = np.zeros((n, 4), np.uint64); c4 = np.zeros((n, 4), np.float32)
        check(lib.rb_graph_neighbors(self.h, _ptr(f), _ptr(r), _ptr(ch), n, direction, _ptr(f4), _ptr(r4), _ptr(c4)))
        return f4, r4, c4

    def walkMaxCov(self, seeds, direction, bound, minKmerCov=1.0, targets=None, hashes=True, counts=True):
        """Batched greedy maximum-coverage walks (the loop around Kmer.getMaxCovSuccessor / getMaxCovPredecessor,
        R/graph/Kmer.java:301-355, as GraphUtils.getMaxCoveragePath runs it, R/util/GraphUtils.java:1591-1675).
        seeds / targets: k-mers as bytes.  Returns (bases[n, bound], f[n, bound], r[n, bound], count[n, bound], len[n],
        reason[n]); reason: 0 dead end, 1 reached the target, 2 met a k-mer of the walk again, 3 bound, 4 invalid seed."""
        n = len(seeds)
        k = self.k
        sb = np.frombuffer(b"".join(seeds), np.uint8) if n else np.zeros(0, np.uint8)
        assert sb.size == n * k, "every seed must have k bases"
        tb = None
        if targets is not None:
            tb = np.frombuffer(b"".join(targets), np.uint8)
            assert tb.size == n * k
        bases = np.zeros((n, bound), np.uint8); ln = np.zeros(n, np.int32); reason = np.zeros(n, np.uint8)
        f = np.zeros((n, bound), np.uint64) if hashes else None
        r = np.zeros((n, bound), np.uint64) if hashes else None
        c = np.zeros((n, bound), np.float32) if counts else None
        check(lib.rb_graph_walk(self.h, _ptr(sb), _ptr(tb) if tb is not None else None, n, direction, bound, float(minKmerCov),
                                _ptr(bases), _ptr(f) if hashes else None, _ptr(r) if hashes else None, _ptr(c) if counts else None,
                                _ptr(ln), _ptr(reason)))
        return bases, f, r, c, ln, reason

    def greedyExtend(self, seeds, direction, lookahead, bound, counts=True, bf=None):
        """GraphUtils.greedyExtendRight (direction 0) / greedyExtendLeft (1) for many source k-mers at once
        (R/util/GraphUtils.java:1961-1976, :1906-1921; with bf — a rnabloom.bloom.BloomFilter — the gated variants
        :1978-1993, :1940-1955).  Returns (bases[n, bound], count[n, bound] or None, len[n], reason[n])."""
        n = len(seeds)
        sb = np.frombuffer(b"".join(seeds), np.uint8) if n else np.zeros(0, np.uint8)
        assert sb.size == n * self.k, "every seed must have k bases"
        bases = np.zeros((n, bound), np.uint8); ln = np.zeros(n, np.int32); reason = np.zeros(n, np.uint8)
        c = np.zeros((n, bound), np.float32) if counts else None
        check(lib.rb_graph_greedy_extend(self.h, bf._g.h if bf is not None else None, _ptr(sb), n, direction, lookahead, bound, _ptr(bases),
                                         _ptr(c) if counts else None, _ptr(ln), _ptr(reason)))
        return bases, c, ln, reason

    # ---- filter state ----
    def filterSize(self, which):
        s, nb, h = C.c_int64(), C.c_int64(), C.c_int()
        check(lib.rb_filter_size(self.h, which, C.byref(s), C.byref(nb), C.byref(h)))
        return s.value, nb.value, h.value

    def popcount(self, which):
        v = C.c_int64()
        check(lib.rb_filter_popcount(self.h, which, C.byref(v)))
        return v.value

    def _fpr(self, which):
        v = C.c_float()
        check(lib.rb_filter_fpr(self.h, which, C.byref(v)))
        return v.value

    def getDbgbfFPR(self): return self._fpr(N.DBGBF)
    def getCbfFPR(self): return self._fpr(N.CBF)
    def getRpkbfFPR(self): return self._fpr(N.RPKBF)
    def getPkbfFPR(self): return self._fpr(N.FPKBF)
    def getFPR(self): return np.float32(self.getDbgbfFPR()) * np.float32(self.getCbfFPR())

    def exportFilter(self, which):
        _, nb, _ = self.filterSize(which)
        out = np.zeros(nb, np.uint8)
        check(lib.rb_filter_export(self.h, which, _ptr(out), nb))
        return out

    def importFilter(self, which, data):
        a = np.ascontiguousarray(data, np.uint8)
        check(lib.rb_filter_import(self.h, which, _ptr(a), a.size))

    # ---- instrumentation ----
    def profileEnable(self, on=True): check(lib.rb_graph_profile_enable(self.h, int(on)))

    def profileGet(self, reset=True):
        p = N.Profile()
        check(lib.rb_graph_profile_get(self.h, C.byref(p), int(reset)))
        return {p.name[i].decode(): (p.ms[i], p.launches[i]) for i in range(p.n)}


# ---- sketching (BASELINE config 5): hash-only, no graph needed ----
def _pack(reads):
    lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    seq = np.frombuffer(b"".join(reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
    return seq, off


def minimizers(reads, k, w, mode=1, device=0):
    """MinimizerHashIterator over each read: (offsets[n+1], hash, pos)."""
    seq, off = _pack(reads)
    mo = np.zeros(len(reads) + 1, np.int64)
    check(lib.rb_minimizers(device, _ptr(seq), _ptr(off), len(reads), k, w, mode, _ptr(mo), None, None))
    t = int(mo[-1])
    h = np.zeros(t, np.uint64); p = np.zeros(t, np.int64)
    if t:
        check(lib.rb_minimizers(device, _ptr(seq), _ptr(off), len(reads), k, w, mode, _ptr(mo), _ptr(h), _ptr(p)))
    return mo, h, p


def strobemers(reads, k, n, wmin, wmax, device=0):
    """StrobeHashIterator.getInterval over each read: (offsets[n+1], hash, start, end)."""
    seq, off = _pack(reads)
    so = np.zeros(len(reads) + 1, np.int64)
    check(lib.rb_strobemers(device, _ptr(seq), _ptr(off), len(reads), k, n, wmin, wmax, _ptr(so), None, None, None))
    t = int(so[-1])
    h = np.zeros(t, np.uint64); s = np.zeros(t, np.int32); e = np.zeros(t, np.int32)
    if t:
        check(lib.rb_strobemers(device, _ptr(seq), _ptr(off), len(reads), k, n, wmin, wmax, _ptr(so), _ptr(h), _ptr(s), _ptr(e)))
    return so, h, s, e
