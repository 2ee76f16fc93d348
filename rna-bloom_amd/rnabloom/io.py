"""Input formats of the reference at the C-ABI boundary: FASTQ text -> sequence / quality buffers
(rnabloom.io.FastqReader: R/io/FastqReader.java:140-186, gzip through R/util/FileUtils.java:50-57) and .nbits 2-bit
sequence files (R/io/NucleotideBitsReader.java / NucleotideBitsWriter.java).  Record splitting is the library's threaded
splitter; 2-bit encoding, quality segmentation and the .nbits bit permutation run on the GPU."""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import check, lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def splitFastq(text, threads=0, with_qual=True):
    """FASTQ text (bytes / uint8 array) -> (seq uint8[], qual uint8[] or None, offsets int64[n+1])"""
    t = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
    n = C.c_int64()
    check(lib.rb_fastq_split(_ptr(t), t.size, threads, None, None, None, 0, C.byref(n)))
    off = np.zeros(n.value + 1, np.int64)
    seq = np.zeros(max(1, t.size), np.uint8)
    qual = np.zeros(max(1, t.size), np.uint8) if with_qual else None
    check(lib.rb_fastq_split(_ptr(t), t.size, threads, _ptr(seq), _ptr(qual), _ptr(off), n.value, C.byref(n)))
    tot = int(off[-1])
    return seq[:tot], (qual[:tot] if with_qual else None), off


def gunzip(data, threads=0):
    """every member of a gzip byte string (GZIPInputStream semantics) -> uint8 array; BGZF members are inflated in parallel"""
    a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    n = C.c_size_t()
    check(lib.rb_gunzip(_ptr(a), a.size, threads, None, 0, C.byref(n)))
    out = np.empty(max(1, n.value), np.uint8)
    check(lib.rb_gunzip(_ptr(a), a.size, threads, _ptr(out), out.size, C.byref(n)))
    return out[:n.value]


def readText(path, threads=0):
    """FileUtils.getTextFileReader: the file's text as a uint8 array, inflated when the name ends in .gz (R/util/FileUtils.java:50-57)"""
    path = str(path)
    raw = np.fromfile(path, np.uint8)
    return gunzip(raw, threads) if path.lower().endswith(".gz") else raw


def readFastq(path, threads=0, with_qual=True):
    """getTextFileReader + FastqReader over a whole file (.gz by extension, as FileUtils.getTextFileReader decides)"""
    return splitFastq(readText(path, threads), threads, with_qual)


def batchFromFastq(text, device=0, minBaseQual=3, final=True, with_qual=True):
    """FASTQ text -> ReadBatch on the device, records found and encoded there; returns (batch, bytes consumed)"""
    from .graph import ReadBatch
    t = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
    h = C.c_void_p(); used = C.c_size_t()
    check(lib.rb_batch_create_fastq(device, _ptr(t), t.size, int(final), minBaseQual, int(with_qual), C.byref(h), C.byref(used)))
    return ReadBatch(h, device), used.value


def batchFromFasta(text, device=0, final=True):
    """FASTA text -> ReadBatch on the device (FastaReader.next semantics, records found there); returns (batch, bytes consumed, ended)"""
    from .graph import ReadBatch
    t = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
    h = C.c_void_p(); used = C.c_size_t(); ended = C.c_int32()
    check(lib.rb_batch_create_fasta(device, _ptr(t), t.size, int(final), C.byref(h), C.byref(used), C.byref(ended)))
    return ReadBatch(h, device), used.value, bool(ended.value)


def writeNbits(path, seq, offsets, append=False):
    """NucleotideBitsWriter: 4-byte big-endian length + 2-bit bases (first base in the top bits, value - 128) per sequence"""
    seq = np.ascontiguousarray(seq, np.uint8); off = np.ascontiguousarray(offsets, np.int64)
    need = C.c_size_t()
    check(lib.rb_nbits_encode(_ptr(seq), _ptr(off), off.size - 1, None, 0, C.byref(need)))
    out = np.zeros(max(1, need.value), np.uint8)
    check(lib.rb_nbits_encode(_ptr(seq), _ptr(off), off.size - 1, _ptr(out), out.size, C.byref(need)))
    with open(str(path), "ab" if append else "wb") as f:
        f.write(out[:need.value].tobytes())
    return need.value


def batchFromNbits(data, device=0, max_reads=-1):
    """NucleotideBitsReader over a byte string -> ReadBatch on the device (decoded there); returns (batch, bytes consumed)"""
    from .graph import ReadBatch
    a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    h = C.c_void_p(); used = C.c_size_t()
    check(lib.rb_batch_create_nbits(device, _ptr(a), a.size, max_reads, C.byref(h), C.byref(used)))
    return ReadBatch(h, device), used.value
