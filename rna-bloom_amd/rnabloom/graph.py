"""rnabloom.graph — host-side mirror of R/graph/BloomFilterDeBruijnGraph.java over the C ABI.

Method names and argument meaning follow the reference class; array arguments are numpy arrays of
BASE hash values (hashVals[0]); the library expands them with NTM64 using the graph's k.
All filter state lives in HBM behind the C handle; nothing is computed on the host.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N
from ._native import check, lib


def _java_float(x):
    """Float.toString: shortest decimal that round-trips, plain notation for 1e-3 <= |x| < 1e7, else d.dddE[-]n"""
    x = float(np.float32(x))
    if x != x: return "NaN"
    if x in (float("inf"), float("-inf")): return "Infinity" if x > 0 else "-Infinity"
    if x == 0: return "0.0"
    r = np.format_float_scientific(np.float32(x), unique=True, trim="0")      # e.g. 1.e-05 / 1.5e-05
    mant, exp = r.split("e")
    if mant.endswith("."): mant += "0"
    e = int(exp)
    if -3 <= e < 7:
        p = np.format_float_positional(np.float32(x), unique=True, trim="0")
        return p + "0" if p.endswith(".") else p
    return "%sE%d" % (mant, e)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class PackedHost:
    """Packed reads in HOST memory (the build's batch format, include/rb_capi.h "read batches packed in HOST memory"): numpy views
    codes (u64 per 32 bases), valid (u32 per 32 bases), len (u32 per read) — over pinned memory when pinned=True."""

    def __init__(self, n_reads, n_words, pinned=True):
        self.n_reads, self.n_words = int(n_reads), int(n_words)
        self._raw = []

        def arr(count, dtype):
            nbytes = max(1, count) * np.dtype(dtype).itemsize
            if not pinned:
                return np.zeros(count, dtype)
            p = C.c_void_p()
            check(lib.rb_host_alloc(nbytes, C.byref(p)))
            self._raw.append(p)
            return np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype, count)
        self.codes = arr(self.n_words, np.uint64)
        self.valid = arr(self.n_words, np.uint32)
        self.len = arr(self.n_reads, np.uint32)

    def nbytes(self):
        return self.n_words * 12 + self.n_reads * 4

    def words_before(self, r):
        """word offset of read r (uniform-length reads are the common case; the general case is a prefix sum)"""
        if not hasattr(self, "_woff"):
            self._woff = np.zeros(self.n_reads + 1, np.int64)
            np.cumsum((self.len.astype(np.int64) + 31) >> 5, out=self._woff[1:])
        return int(self._woff[r])

    def close(self):
        self.codes = self.valid = self.len = None
        for p in self._raw:
            lib.rb_host_free(p)
        self._raw = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackedStream:
    """rb_packed_stream: two device batches taking turns; begin() starts the upload of a chunk of a PackedHost and returns,
    finish() waits for it and returns a (borrowed) ReadBatch for addBatch — the caller inserts chunk c between begin(c + 1) and
    finish(c + 1), so the upload runs beside the insert."""

    def __init__(self, max_reads, max_words, device=0):
        self.h = C.c_void_p()
        check(lib.rb_packed_stream_create(device, max_reads, max_words, C.byref(self.h)))
        self.device = device

    def begin(self, ph, first=0, n=None):
        n = ph.n_reads - first if n is None else n
        w0, w1 = ph.words_before(first), ph.words_before(first + n)
        check(lib.rb_packed_stream_begin(self.h, ph.codes.ctypes.data + 8 * w0, ph.valid.ctypes.data + 4 * w0, ph.len.ctypes.data + 4 * first, n, w1 - w0))

    def finish(self):
        b = C.c_void_p()
        check(lib.rb_packed_stream_finish(self.h, C.byref(b)))
        rb = ReadBatch(b, self.device)
        rb.borrowed = True
        return rb

    def close(self):
        if self.h:
            lib.rb_packed_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadBatch:
    """A batch of reads resident on the device in the packed 2-bit + validity format."""
    borrowed = False                     # a batch owned by a PackedStream: never destroyed from here

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

    def downloadPacked(self, first=0, n=None, pinned=True):
        """reads [first, first + n) in the build's host batch format (include/rb_capi.h: codes u64 / valid u32 per 32 bases, len u32 per
        read) -> PackedHost; pinned: the arrays live in hipHostMalloc'ed memory (uploads at link speed, no registration pass)"""
        n = self.n_reads - first if n is None else n
        nw = C.c_int64()
        check(lib.rb_batch_download_packed(self.h, first, n, None, None, None, C.byref(nw)))
        ph = PackedHost(n, nw.value, pinned)
        check(lib.rb_batch_download_packed(self.h, first, n, _ptr(ph.codes), _ptr(ph.valid), _ptr(ph.len), C.byref(nw)))
        return ph

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
        if self.h and not self.borrowed:
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
    def getMaxNumHash(self): return getattr(self, "dbgbfCbfMaxNumHash", max(self.p.dbgbf_num_hash, self.p.cbf_num_hash))
    def setReadPairedKmerDistance(self, d):
        check(lib.rb_graph_set_read_paired_kmer_distance(self.h, d)); self.readPairedKmersDistance = int(d)
    def setFragPairedKmerDistance(self, d):
        check(lib.rb_graph_set_frag_paired_kmer_distance(self.h, d)); self.fragmentPairedKmersDistance = int(d)
    def getReadPairedKmerDistance(self): return getattr(self, "readPairedKmersDistance", -1)     # :52-56: -1 until set
    def getFragPairedKmerDistance(self): return getattr(self, "fragmentPairedKmersDistance", -1)
    def getDbgbfNumHash(self): return self.p.dbgbf_num_hash
    def getCbfNumHash(self): return self.p.cbf_num_hash
    def getPkbfNumHash(self): return self.p.pkbf_num_hash

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

    def addPacked(self, ph, reverseComplement=False, incrementIfPresent=False, storeReadPairedKmers=False, first=0, n=None, pieceReads=0):
        """reads [first, first + n) of a PackedHost through rb_graph_add_packed: ONE insert whose input arrives piece by piece on a copy stream
        while the pipeline already works on what is there (FastqToGraphWorker's loop over reads that are already packed in host memory)"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        n = ph.n_reads - first if n is None else n
        w0, w1 = ph.words_before(first), ph.words_before(first + n)
        st = N.AddStats()
        check(lib.rb_graph_add_packed(self.h, ph.codes.ctypes.data + 8 * w0, ph.valid.ctypes.data + 4 * w0, ph.len.ctypes.data + 4 * first,
                                      n, w1 - w0, pieceReads, flags, C.byref(st)))
        return st

    def prefetchPacked(self, ph, first=0, n=None, pieceReads=0):
        """start the upload of reads [first, first + n) of a PackedHost and return: a later addPacked of the same range picks it up where it is
        (rb_graph_prefetch_packed: the next file travels while this one is inserted)"""
        n = ph.n_reads - first if n is None else n
        w0, w1 = ph.words_before(first), ph.words_before(first + n)
        check(lib.rb_graph_prefetch_packed(self.h, ph.codes.ctypes.data + 8 * w0, ph.valid.ctypes.data + 4 * w0, ph.len.ctypes.data + 4 * first, n, w1 - w0, pieceReads))

    def addFastq(self, text, minBaseQual=3, reverseComplement=False, incrementIfPresent=False, storeReadPairedKmers=False):
        """the text of a FASTQ file (bytes / uint8 array / np.memmap) through FastqToGraphWorker's loop, R/RNABloom.java:526-643:
        uploaded as it is, records found and 2-bit encoded on the GPU; returns (stats, records)"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        t = np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        st = N.AddStats(); n = C.c_int64()
        check(lib.rb_graph_add_fastq(self.h, _ptr(t), t.size, minBaseQual, flags, C.byref(st), C.byref(n)))
        return st, n.value

    def addFastqFile(self, path, minBaseQual=3, reverseComplement=False, incrementIfPresent=False, storeReadPairedKmers=False):
        """a FASTQ file (plain or gzip, detected by its magic bytes) streamed through FastqToGraphWorker's loop: the next piece is
        read / inflated, uploaded and parsed while the current one is inserted; returns (stats, records)"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        st = N.AddStats(); n = C.c_int64()
        check(lib.rb_graph_add_fastq_file(self.h, os.fsencode(path), minBaseQual, flags, C.byref(st), C.byref(n)))
        return st, n.value

    def addFastaFile(self, path, reverseComplement=False, incrementIfPresent=False, storeReadPairedKmers=False):
        """the same for a FASTA file (FastaToGraphWorker's loop)"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        st = N.AddStats(); n = C.c_int64()
        check(lib.rb_graph_add_fasta_file(self.h, os.fsencode(path), flags, C.byref(st), C.byref(n)))
        return st, n.value

    def addFasta(self, text, reverseComplement=False, incrementIfPresent=False, storeReadPairedKmers=False):
        """the text of a FASTA file through FastaToGraphWorker's loop (R/RNABloom.java:645-732), records found on the GPU;
        returns (stats, records)"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_COUNT_IF_PRESENT if incrementIfPresent else 0) \
            | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        t = np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        st = N.AddStats(); n = C.c_int64()
        check(lib.rb_graph_add_fasta(self.h, _ptr(t), t.size, flags, C.byref(st), C.byref(n)))
        return st, n.value

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

    def getKmers(self, reads, out=None):
        """getKmers(String) for a list of sequences -> (koffsets, f, r, count); out = (f, r, count) arrays to fill (large enough)."""
        lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
        off = np.zeros(len(reads) + 1, np.int64); np.cumsum(lens, out=off[1:])
        seq = np.frombuffer(b"".join(reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
        ko = np.zeros(len(reads) + 1, np.int64)
        check(lib.rb_graph_kmers(self.h, _ptr(seq), _ptr(off), len(reads), _ptr(ko), None, None, None))
        t = int(ko[-1])
        if out is None:
            f = np.zeros(t, np.uint64); r = np.zeros(t, np.uint64); c = np.zeros(t, np.float32)
        else:
            f, r, c = (a[:t] for a in out)
        if t:
            check(lib.rb_graph_kmers(self.h, _ptr(seq), _ptr(off), len(reads), _ptr(ko), _ptr(f), _ptr(r), _ptr(c)))
        return ko, f, r, c

    def batchCounts(self, batch, first=0, n=None, koffsets=None, to_host=True, out=None):
        """The count profile of getKmers for reads [first, first + n) of a resident ReadBatch (rb_graph_batch_counts): one float per
        window, no hashes.  koffsets None: an [n, stride] array (stride = longest read - k + 1, shorter reads zero-padded) flattened;
        else packed rows at koffsets[i].  to_host False: the counts stay on the device (a torch tensor is returned)."""
        n = batch.n_reads - first if n is None else n
        stride = C.c_int64(0)
        ko = None if koffsets is None else np.ascontiguousarray(koffsets, np.int64)
        check(lib.rb_graph_batch_counts(self.h, batch.h, first, 0, None, None, 0, C.byref(stride)))
        total = int(ko[-1]) if ko is not None else n * stride.value
        if to_host:
            out = np.empty(total, np.float32) if out is None else out
            assert out.dtype == np.float32 and out.size >= total and out.flags.c_contiguous
            check(lib.rb_graph_batch_counts(self.h, batch.h, first, n, _ptr(ko) if ko is not None else None, _ptr(out), 0, None))
            return out[:total]
        import torch
        out = torch.empty(total, dtype=torch.float32, device="cuda:%d" % self.device) if out is None else out
        check(lib.rb_graph_batch_counts(self.h, batch.h, first, n, _ptr(ko) if ko is not None else None, C.c_void_p(out.data_ptr()), 1, None))
        return out

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

    # ---- the reference's convenience methods, batched (R/graph/BloomFilterDeBruijnGraph.java; all device work goes through
    # the calls above) ----
    def greedyExtendOnce(self, seeds, direction, lookahead, bf=None):
        """GraphUtils.greedyExtendRightOnce / greedyExtendLeftOnce(graph, source, lookahead[, bf]) (R/util/GraphUtils.java:501-625)
        for many source k-mers: the chosen neighbour's base (b"" where the source has none) and its count"""
        bases, cnt, ln, _ = self.greedyExtend(seeds, direction, lookahead, 1, bf=bf)
        return [bytes(bases[i, :ln[i]]) for i in range(len(seeds))], cnt[:, 0]

    def naiveExtend(self, seeds, direction, mode=1, bound=0, minKmerCov=1.0, terminators=None, cap=4096, maxTipLength=None):
        """GraphUtils.naiveExtendRight / naiveExtendLeft (R/util/GraphUtils.java:6780-7112) for many seed k-mers.  mode 0: the
        forms with a terminator set — terminators[i] = a sequence whose k-mers stop walk i (plus the k-mers it added); mode 1:
        bounded; mode 2: NoBackChecks.  maxTipLength is accepted and unused, as in the reference's effect (Kmer.hasDepth* always
        answers true).  Returns (list of appended bases, reason[n])."""
        n = len(seeds)
        sd = np.frombuffer(b"".join(s if isinstance(s, bytes) else s.encode() for s in seeds), np.uint8)
        if sd.size != n * self.k: raise ValueError("naiveExtend: every seed must be one k-mer")
        width = cap if mode == 0 else bound + 1
        ob = np.zeros((n, max(1, width)), np.uint8); ol = np.zeros(n, np.int32); orr = np.zeros(n, np.uint8)
        tseq = toff = None
        if mode == 0:
            tseq, toff = _pack([t if isinstance(t, bytes) else t.encode() for t in (terminators if terminators is not None else [b""] * n)])
            if tseq.size == 0: tseq = np.zeros(1, np.uint8)
        check(lib.rb_graph_naive_extend(self.h, _ptr(sd), n, direction, mode, bound, cap, C.c_float(minKmerCov), _ptr(tseq) if tseq is not None else None,
                                        _ptr(toff) if toff is not None else None, _ptr(ob), _ptr(ol), _ptr(orr)))
        return [bytes(ob[i, :ol[i]]) for i in range(n)], orr

    def _base_hash(self, f, r):
        """hashVals[0] of a k-mer: forward hash when stranded, signed minimum of both strands otherwise (NTHash.java:449-475)"""
        return f if self.stranded else np.where(r.view(np.int64) < f.view(np.int64), r, f)

    def getCounts(self, h0):                                            # getCounts(String[]) :578-590, from hashes
        return self.getCount(h0)

    def getKmer(self, kmer):
        """getKmer(String) :592-594 -> (fHashVal, rHashVal, count) of one k-mer string"""
        ko, f, r, c = self.getKmers([kmer if isinstance(kmer, bytes) else kmer.encode()])
        if ko[-1] != 1: raise ValueError("getKmer: the string must hold exactly one k-mer")
        return int(f[0]), int(r[0]), float(c[0])

    def containsSeq(self, kmers):                                       # contains(String) :534-536 for many k-mer strings
        ko, f, r, _ = self.getKmers([x if isinstance(x, bytes) else x.encode() for x in kmers])
        if int(ko[-1]) != len(kmers): raise ValueError("containsSeq: every string must hold exactly one k-mer")
        return self.contains(self._base_hash(f, r))

    def isValidSeq(self, seqs):
        """isValidSeq(String) :1181-1194 for many sequences: every k-mer of the sequence is in dbgbf (True for a sequence
        shorter than k, as in the reference, whose loop then never runs)"""
        seqs = [x if isinstance(x, bytes) else x.encode() for x in seqs]
        ko, f, r, _ = self.getKmers(seqs)
        hit = self.contains(self._base_hash(f, r)) if f.size else np.zeros(0, bool)
        return [bool(hit[int(ko[i]):int(ko[i + 1])].all()) for i in range(len(seqs))]

    _ALT = {ord("A"): b"CGT", ord("C"): b"AGT", ord("G"): b"ACT", ord("T"): b"ACG", ord("U"): b"ACG"}   # SeqUtils.java:51-55

    def _variants(self, kmer, pos):
        """getLeftVariants / getRightVariants(String): the k-mer is hashed once, its three alternatives come from the
        variant iterators on the device (rb_graph_neighbors directions 2 / 3) and are kept when the graph contains them"""
        kmer = kmer if isinstance(kmer, bytes) else kmer.encode()
        f, r, _ = self.getKmer(kmer)
        _, _, c4 = self.getNeighbors([f], [r], [kmer[pos]], 2 if pos == 0 else 3)
        out = []
        for c in self._ALT.get(kmer[pos], b"ACGT"):            # getAltNucleotides order, SeqUtils.java:51-55
            if c4[0][b"ACGT".index(c)] > 0:
                out.append((kmer[:pos] + bytes([c]) + kmer[pos + 1:]).decode())
        return out

    def getLeftVariants(self, kmer): return self._variants(kmer, 0)                 # :1056-1068
    def getRightVariants(self, kmer): return self._variants(kmer, self.k - 1)       # :1109-1121

    def getSuccessors(self, f, r, charOut, minKmerCov=1.0):
        """Kmer.getSuccessors(k, numHash, graph, minKmerCov) (R/graph/Kmer.java:199-226) for many k-mers: the four
        candidates in A,C,G,T order with a mask of those whose count reaches minKmerCov -> (f4, r4, count4, keep4)"""
        f4, r4, c4 = self.getNeighbors(f, r, charOut, 0)
        return f4, r4, c4, c4 >= np.float32(minKmerCov)

    def getPredecessors(self, f, r, charOut, minKmerCov=1.0):           # Kmer.java:228-255
        f4, r4, c4 = self.getNeighbors(f, r, charOut, 1)
        return f4, r4, c4, c4 >= np.float32(minKmerCov)

    # Kmer.hasPredecessors / hasSuccessors / hasAtLeastX* / getNum* (R/graph/Kmer.java:97-197) for many k-mers: the
    # neighbours graph.contains() accepts, i.e. those with getCount > 0 (getCount is cbf + 1 inside dbgbf, 0 outside)
    def getNumPredecessors(self, f, r, lastBase): return (self.getNeighbors(f, r, lastBase, 1)[2] > 0).sum(axis=1)
    def getNumSuccessors(self, f, r, firstBase): return (self.getNeighbors(f, r, firstBase, 0)[2] > 0).sum(axis=1)
    def hasPredecessors(self, f, r, lastBase): return self.getNumPredecessors(f, r, lastBase) > 0
    def hasSuccessors(self, f, r, firstBase): return self.getNumSuccessors(f, r, firstBase) > 0
    def hasAtLeastXPredecessors(self, f, r, lastBase, x): return self.getNumPredecessors(f, r, lastBase) >= x
    def hasAtLeastXSuccessors(self, f, r, firstBase, x): return self.getNumSuccessors(f, r, firstBase) >= x

    def _pair_hashes(self, f, r, d):
        from .graphutils import kmerPairHashValues
        f, r = _u64(np.atleast_1d(f)), _u64(np.atleast_1d(r))
        return kmerPairHashValues(f, r, d, self.stranded) if f.size > d > 0 else np.zeros(0, np.uint64)

    def addReadPairedKmers(self, f, r):
        """addReadPairedKmers(ArrayList<Kmer>) :485-494: k-mers i and i + readPairedKmersDistance of one sequence, given
        by their hash values, into rpkbf"""
        p = self._pair_hashes(f, r, self.getReadPairedKmerDistance())
        if p.size: self.addReadSingleKmerPair(p)

    def addFragmentPairKmers(self, f, r):                               # :474-483, into fpkbf
        p = self._pair_hashes(f, r, self.getFragPairedKmerDistance())
        if p.size: self.addFragmentSingleKmerPair(p)

    def _iterate(self, seqs, mode):
        b = ReadBatch.from_reads([x if isinstance(x, bytes) else x.encode() for x in seqs], device=self.device)
        try:
            return b.nthash(self.k, mode, with_positions=True)
        finally:
            b.close()

    def getHashIterator(self, seqs):
        """getHashIterator().start(seq) ... next() (:1196-1206) over many sequences at once: hashVals[0] of every k-mer of
        every usable (ACGT) segment, with its sequence number and position — the strand-specific iterator for a stranded
        graph, the canonical one otherwise"""
        return self._iterate(seqs, 0 if self.stranded else 1)

    def getReverseComplementHashIterator(self, seqs):                    # :1208-1214
        return self._iterate(seqs, 2 if self.stranded else 1)

    def getDbgbf(self): return _FilterOfGraph(self, N.DBGBF)            # :277-291: the filter objects of a graph, as views
    def getCbf(self): return _FilterOfGraph(self, N.CBF)
    def getRpkbf(self): return _FilterOfGraph(self, N.RPKBF) if self._has(N.RPKBF) else None
    def getFpkbf(self): return _FilterOfGraph(self, N.FPKBF) if self._has(N.FPKBF) else None
    def getFpkbfFPR(self): return self.getPkbfFPR()

    def destroyDbgbf(self): check(lib.rb_graph_destroy_filter(self.h, N.DBGBF))      # :249-254: the device memory is freed
    def destroyCbf(self): check(lib.rb_graph_destroy_filter(self.h, N.CBF))          # :256-261
    def destroyFpkbf(self): check(lib.rb_graph_destroy_filter(self.h, N.FPKBF))      # :263-268
    def destroyRpkbf(self): check(lib.rb_graph_destroy_filter(self.h, N.RPKBF))      # :270-275

    # ---- persistence: the reference's own files (a graph saved here loads in RNA-Bloom and the other way round) ----
    _EXT = {N.DBGBF: ".dbgbf", N.CBF: ".cbf", N.RPKBF: ".rpkbf", N.FPKBF: ".fpkbf"}     # :58-62

    def saveDesc(self, graphFile):                                      # :297-305
        with open(graphFile, "w") as w:
            w.write("dbgbfCbfMaxNumHash:%d\nstranded:%s\nk:%d\nreadPairedKmersDistance:%d\nfragmentPairedKmersDistance:%d\n" % (
                self.getMaxNumHash(), "true" if self.stranded else "false", self.k, self.getReadPairedKmerDistance(), self.getFragPairedKmerDistance()))

    def _save_filter(self, graphFile, which):
        """BloomFilter.save / CountingBloomFilter.save (R/bloom/BloomFilter.java:113-124, CountingBloomFilter.java:106-118):
        <graph><ext>.desc with size / numhash / fpr, <graph><ext> with the raw bytes"""
        size, _, h = self.filterSize(which)
        path = str(graphFile) + self._EXT[which]
        with open(path + ".desc", "w") as w:
            w.write("size:%d\nnumhash:%d\nfpr:%s\n" % (size, h, _java_float(self._fpr(which))))
        self.exportFilter(which).tofile(path)

    def _has(self, which):
        try: return self.filterSize(which)[0] > 0
        except Exception: return False

    def save(self, graphFile):                                          # :307-329 (fpkbf is written by savePkbf)
        self.saveDesc(graphFile)
        self._save_filter(graphFile, N.DBGBF); self._save_filter(graphFile, N.CBF)
        if self._has(N.RPKBF): self._save_filter(graphFile, N.RPKBF)

    def savePkbf(self, graphFile):                                      # :331-339
        self.saveDesc(graphFile); self._save_filter(graphFile, N.FPKBF)

    @staticmethod
    def _read_desc(path):
        d = {}
        with open(path) as r:
            for line in r:
                if ":" in line:
                    key, val = line.rstrip("\n").split(":", 1)
                    d[key] = val
        return d

    def restorePkbf(self, graphFile):                                   # :341-350
        import os
        path = str(graphFile) + self._EXT[N.FPKBF]
        d = self._read_desc(path + ".desc")
        if self._has(N.FPKBF): self.destroyFpkbf()                     # :345-347: a new filter with the file's size and numhash
        self.initializePairKmersBloomFilter(int(d["size"]), int(d["numhash"]))
        self.p.pkbf_num_hash = int(d["numhash"])
        self.importFilter(N.FPKBF, np.fromfile(path, np.uint8))

    def updateFragmentKmerDistance(self, graphFile):                    # :106-119
        d = self._read_desc(graphFile)
        if "fragmentPairedKmersDistance" in d: self.setFragPairedKmerDistance(int(d["fragmentPairedKmersDistance"]))

    @classmethod
    def fromFile(cls, graphFile, loadDbgBits=True, device=0, rngSeed=0):
        """BloomFilterDeBruijnGraph(File graphFile, boolean loadDbgBits) :121-189"""
        import os
        g = str(graphFile)
        d = cls._read_desc(g)
        db, cb = cls._read_desc(g + ".dbgbf.desc"), cls._read_desc(g + ".cbf.desc")
        rp = g + ".rpkbf"
        has_rp = os.path.isfile(rp) and os.path.isfile(rp + ".desc")
        rpd = cls._read_desc(rp + ".desc") if has_rp else {"size": "64", "numhash": "1"}
        self = cls(int(db["size"]), int(cb["size"]), int(rpd["size"]), int(db["numhash"]), int(cb["numhash"]), int(rpd["numhash"]),
                   int(d["k"]), d.get("stranded", "false").strip() == "true", has_rp, device=device, rngSeed=rngSeed)
        if loadDbgBits: self.importFilter(N.DBGBF, np.fromfile(g + ".dbgbf", np.uint8))
        self.importFilter(N.CBF, np.fromfile(g + ".cbf", np.uint8))
        if has_rp: self.importFilter(N.RPKBF, np.fromfile(rp, np.uint8))
        if "readPairedKmersDistance" in d and int(d["readPairedKmersDistance"]) > 0: self.setReadPairedKmerDistance(int(d["readPairedKmersDistance"]))
        if "fragmentPairedKmersDistance" in d and int(d["fragmentPairedKmersDistance"]) > 0: self.setFragPairedKmerDistance(int(d["fragmentPairedKmersDistance"]))
        fp = g + ".fpkbf"
        if os.path.isfile(fp) and os.path.isfile(fp + ".desc"): self.restorePkbf(g)     # pkbfNumHash from the loaded filter (:170-176)
        if "dbgbfCbfMaxNumHash" in d: self.dbgbfCbfMaxNumHash = int(d["dbgbfCbfMaxNumHash"])
        return self

    # ---- filter state ----
    def filterSize(self, which):
        s, nb, h = C.c_int64(), C.c_int64(), C.c_int()
        check(lib.rb_filter_size(self.h, which, C.byref(s), C.byref(nb), C.byref(h)))
        return s.value, nb.value, h.value

    def popcount(self, which):
        v = C.c_int64()
        check(lib.rb_filter_popcount(self.h, which, C.byref(v)))
        return v.value

    def fold(self, which):
        """64-bit digest of the filter's bytes, computed on the device (rb_filter_fold; fold_bytes is the host form)"""
        v = C.c_uint64()
        check(lib.rb_filter_fold(self.h, which, C.byref(v)))
        return v.value

    def _fpr(self, which):
        v = C.c_float()
        check(lib.rb_filter_fpr(self.h, which, C.byref(v)))
        self.__dict__.setdefault("_popcache", {})[which] = self.popcount(which)   # BloomFilter.java:185-194: getFPR() remembers its count
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


class _FilterOfGraph:
    """One filter of a graph seen through the reference's BloomFilter / CountingBloomFilter methods that take a hash value
    (R/bloom/BloomFilter.java:139-199, R/bloom/CountingBloomFilter.java:235-263); arrays in, arrays out."""

    def __init__(self, graph, which): self.g, self.which = graph, which
    def lookup(self, h0): return self.g.getCbfCount(h0) > 0 if self.which == N.CBF else self.g._lookup(self.which, h0)
    def getCount(self, h0):
        if self.which != N.CBF: raise TypeError("getCount is a counting-filter method")
        return self.g.getCbfCount(h0)
    def getFPR(self): return self.g._fpr(self.which)
    def getPopCount(self): return self.g.__dict__.get("_popcache", {}).get(self.which, -1)   # :201-203: cached by the last getFPR(), -1 before
    def getNumHash(self): return self.g.filterSize(self.which)[2]
    def getSize(self): return self.g.filterSize(self.which)[0]
    def toBytes(self): return self.g.exportFilter(self.which)


# ---- sketching (BASELINE config 5): hash-only, no graph needed ----
def _pack(reads):
    lens = np.fromiter((len(r) for r in reads), np.int64, len(reads))
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    seq = np.frombuffer(b"".join(reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
    return seq, off


def minimizers(reads, k, w, mode=1, device=0, out=None):
    """MinimizerHashIterator over each read: (offsets[n+1], hash, pos).  reads: list of bytes, or (seq, offsets) arrays;
    out = (hash, pos) arrays to fill."""
    seq, off = reads if isinstance(reads, tuple) else _pack(reads)
    reads = range(off.size - 1)
    mo = np.zeros(len(reads) + 1, np.int64)
    check(lib.rb_minimizers(device, _ptr(seq), _ptr(off), len(reads), k, w, mode, _ptr(mo), None, None))
    t = int(mo[-1])
    h, p = (np.zeros(t, np.uint64), np.zeros(t, np.int64)) if out is None else (out[0][:t], out[1][:t])
    if t:
        check(lib.rb_minimizers(device, _ptr(seq), _ptr(off), len(reads), k, w, mode, _ptr(mo), _ptr(h), _ptr(p)))
    return mo, h, p


def strobemers(reads, k, n, wmin, wmax, device=0, out=None):
    """StrobeHashIterator.getInterval over each read: (offsets[n+1], hash, start, end).  reads: list of bytes, or (seq, offsets)
    arrays; out = (hash, start, end) arrays to fill."""
    seq, off = reads if isinstance(reads, tuple) else _pack(reads)
    reads = range(off.size - 1)
    so = np.zeros(len(reads) + 1, np.int64)
    check(lib.rb_strobemers(device, _ptr(seq), _ptr(off), len(reads), k, n, wmin, wmax, _ptr(so), None, None, None))
    t = int(so[-1])
    h, s, e = (np.zeros(t, np.uint64), np.zeros(t, np.int32), np.zeros(t, np.int32)) if out is None else (out[0][:t], out[1][:t], out[2][:t])
    if t:
        check(lib.rb_strobemers(device, _ptr(seq), _ptr(off), len(reads), k, n, wmin, wmax, _ptr(so), _ptr(h), _ptr(s), _ptr(e)))
    return so, h, s, e


def _count_handle(counts_from):
    """the rb_graph handle behind a CountingBloomFilter / graph whose cbf is asked getCount(hash) on the device"""
    if counts_from is None:
        return None
    g = getattr(counts_from, "_g", counts_from)
    return g.h


def randstrobes(reads, k, n, wmin, wmax, canonical=False, slide=False, counts_from=None, device=0):
    """StrobeHashIterator.next()/get() (slide=True: get) or CanonicalStrobeHashIterator over each read:
    (offsets[n_reads+1], hash, positions[., n], counts or None).  counts_from: a CountingBloomFilter (or graph) whose
    getCount(hash) is evaluated in the same call (SeqSubsampler.strobemerBased's lookup half)."""
    seq, off = _pack(reads)
    so = np.zeros(len(reads) + 1, np.int64)
    flags = (N.STROBE_CANONICAL if canonical else 0) | (N.STROBE_SLIDE if slide else 0)
    check(lib.rb_randstrobes(device, _ptr(seq), _ptr(off), len(reads), k, n, wmin, wmax, flags, None, _ptr(so), None, None, None))
    t = int(so[-1])
    h = np.zeros(t, np.uint64); pos = np.zeros((t, n), np.int32)
    cnt = np.zeros(t, np.float32) if counts_from is not None else None
    if t:
        check(lib.rb_randstrobes(device, _ptr(seq), _ptr(off), len(reads), k, n, wmin, wmax, flags, _count_handle(counts_from), _ptr(so),
                                 _ptr(h), _ptr(pos), _ptr(cnt) if cnt is not None else None))
    return so, h, pos, cnt


def strobe3(reads, k, wmin, wmax, canonical=False, counts_from=None, device=0):
    """Strobe3HashIterator / CanonicalStrobe3HashIterator over each read: (offsets, hash, positions[., 3], counts or None)"""
    seq, off = _pack(reads)
    so = np.zeros(len(reads) + 1, np.int64)
    check(lib.rb_strobe3(device, _ptr(seq), _ptr(off), len(reads), k, wmin, wmax, int(canonical), None, _ptr(so), None, None, None))
    t = int(so[-1])
    h = np.zeros(t, np.uint64); pos = np.zeros((t, 3), np.int32)
    cnt = np.zeros(t, np.float32) if counts_from is not None else None
    if t:
        check(lib.rb_strobe3(device, _ptr(seq), _ptr(off), len(reads), k, wmin, wmax, int(canonical), _count_handle(counts_from), _ptr(so),
                             _ptr(h), _ptr(pos), _ptr(cnt) if cnt is not None else None))
    return so, h, pos, cnt


def kmerPairHashes(reads, k, shift, canonical=False, counts_from=None, device=0):
    """SeqSubsampler.kmerBased's pair hashes (gap = shift - k) of each read: (offsets, hash, counts or None)"""
    seq, off = _pack(reads)
    po = np.zeros(len(reads) + 1, np.int64)
    check(lib.rb_kmer_pair_hashes(device, _ptr(seq), _ptr(off), len(reads), k, shift, int(canonical), None, _ptr(po), None, None))
    t = int(po[-1])
    h = np.zeros(t, np.uint64)
    cnt = np.zeros(t, np.float32) if counts_from is not None else None
    if t:
        check(lib.rb_kmer_pair_hashes(device, _ptr(seq), _ptr(off), len(reads), k, shift, int(canonical), _count_handle(counts_from), _ptr(po),
                                      _ptr(h), _ptr(cnt) if cnt is not None else None))
    return po, h, cnt


def _windows(reads, k, w):
    return sum(max(0, len(r) - k + 1 - w + 1) for r in reads)


def nextMinimizers(reads, k, w, mode=1, device=0):
    """the minimizers MinimizerHashIterator.nextMinimizer() walks through: (offsets, hash, pos)"""
    seq, off = _pack(reads)
    cap = _windows(reads, k, w)
    mo = np.zeros(len(reads) + 1, np.int64)
    h = np.zeros(max(1, cap), np.uint64); p = np.zeros(max(1, cap), np.int64)
    check(lib.rb_minimizers_next(device, _ptr(seq), _ptr(off), len(reads), k, w, mode, _ptr(mo), _ptr(h), _ptr(p)))
    t = int(mo[-1])
    return mo, h[:t].copy(), p[:t].copy()


def getMinimizers(reads, k, w, mode=1, stale=None, device=0):
    """GraphUtils.getMinimizers for each read (sorted distinct window minimizers, signed order): (offsets, values).
    stale[i]: what the iterator's hVals[0] held before read i (only reads with numKmers <= w look at it; default 0)."""
    seq, off = _pack(reads)
    cap = sum(max(1, len(r) - k + 1 - w + 1) for r in reads)
    mo = np.zeros(len(reads) + 1, np.int64)
    out = np.zeros(max(1, cap), np.uint64)
    st = None if stale is None else np.ascontiguousarray(stale, np.uint64)
    check(lib.rb_minimizer_set(device, _ptr(seq), _ptr(off), len(reads), k, w, mode, _ptr(st) if st is not None else None, _ptr(mo), _ptr(out)))
    return mo, out[:int(mo[-1])].copy()


def fold_bytes(data, first_word=0):
    """Host restatement of rb_filter_fold for a filter's exported bytes (numpy): wrapping sum over the non-zero 32-bit
    little-endian words of splitmix64(global word number * 0x9E3779B97F4A7C15 + word)."""
    a = np.ascontiguousarray(data, np.uint8)
    pad = (-a.size) % 4
    if pad:
        a = np.concatenate([a, np.zeros(pad, np.uint8)])
    w = a.view("<u4").astype(np.uint64)
    nz = np.nonzero(w)[0]
    with np.errstate(over="ignore"):
        z = (nz.astype(np.uint64) + np.uint64(first_word)) * np.uint64(0x9E3779B97F4A7C15) + w[nz]
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(z.sum(dtype=np.uint64))
