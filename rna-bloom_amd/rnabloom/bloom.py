"""rnabloom.bloom.BloomFilter / CountingBloomFilter / PairedKeysBloomFilter as stand-alone objects (the reference also
uses them outside the graph: screening filters, SeqSubsampler's counting filter), with the reference's method names.
Each object owns an rb_graph handle in which only its own filter has a real size — the kernels, the byte layout and the
arithmetic are the ones the graph uses, so everything the parity tests establish for dbgbf / cbf holds here.
Methods take arrays of base hashes (hashVals[0]); the library derives hashVals[1..] with NTM64 for the filter's k, as
HashFunction.getHashValues(long, int) does (R/bloom/hash/HashFunction.java:56-60)."""
import numpy as np

from . import _native as N
from .graph import BloomFilterDeBruijnGraph

_TINY = 64          # size of the filters an object does not use


class BloomFilter:
    """R/bloom/BloomFilter.java:40-258"""

    def __init__(self, size, numHash, k, device=0):
        self.size, self.numHash, self.k, self.device = int(size), int(numHash), int(k), int(device)
        self.popcount = -1                                                      # :48, set by getFPR
        self._g = BloomFilterDeBruijnGraph(self.size, _TINY, 0, self.numHash, 1, 1, self.k, True, False, device=device)

    def add(self, h0): self._g.addDbgOnly(h0)                                   # :133-141
    def lookup(self, h0): return self._g.contains(h0)                           # :170-182
    def lookupThenAdd(self, h0): return self._g.lookupThenAdd(N.DBGBF, h0)      # :143-155, in array order
    def getFPR(self):                                                           # :185-194 (remembers the popcount it used)
        self.popcount = self._g.popcount(N.DBGBF)
        return self._g.getDbgbfFPR()
    def getPopCount(self): return self.popcount                                # :201-203: the value cached by the last getFPR() (-1 before)
    def getOptimalSize(self, fpr):                                              # :205-213: from the popcount of the last getFPR()
        return self.getExpectedSize(self.popcount, fpr, self.numHash) if self.popcount > 0 else self.size
    def getNumHash(self): return self.numHash
    def getSize(self): return self.size
    def empty(self): self._g.clearDbgbf()                                       # :240-242
    def destroy(self): self._g.destroy()                                        # :244-246
    def toBytes(self): return self._g.exportFilter(N.DBGBF)                     # the bytes save() writes, :113-124
    def fromBytes(self, data): self._g.importFilter(N.DBGBF, data)
    def equivalent(self, other):                                                # :248-257
        return self.size == other.size and self.numHash == other.numHash and bool((self.toBytes() == other.toBytes()).all())

    @staticmethod
    def getExpectedSize(expNumElements, fpr, numHash):                          # :196-199
        return int(N.lib.rb_expected_size(int(expNumElements), float(fpr), int(numHash)))


class CountingBloomFilter:
    """R/bloom/CountingBloomFilter.java:41-339 (8-bit MiniFloat counters, conservative update).  The random draws of
    MiniFloat.increment come from the library's counter-based generator (seed, op ordinal), so a run is reproducible."""

    def __init__(self, size, numHash, k, device=0, rngSeed=0):
        self.size, self.numHash, self.k, self.device, self.rngSeed = int(size), int(numHash), int(k), int(device), int(rngSeed)
        self.popcount = -1
        self._g = BloomFilterDeBruijnGraph(_TINY, self.size, 0, 1, self.numHash, 1, self.k, True, False, device=device, rngSeed=rngSeed)

    def increment(self, h0): self._g.addCountOnly(h0)                           # :170-194, in array order
    def incrementAndGet(self, h0):                                              # :196-222, one call after the other
        h = np.ascontiguousarray(np.atleast_1d(h0), np.uint64)
        out = np.zeros(h.size, np.float32)
        N.check(N.lib.rb_filter_increment_and_get(self._g.h, h.ctypes.data, h.size, out.ctypes.data))
        return out
    def getCount(self, h0): return self._g.getCbfCount(h0)                      # :235-251
    def getFPR(self):                                                           # :254-263 (remembers the popcount it used: non-zero counters)
        self.popcount = self._g.popcount(N.CBF)
        return self._g.getCbfFPR()
    def getPopCount(self): return self.popcount                                # :280-282: cached by the last getFPR() (-1 before)
    def getNumHash(self): return self.numHash
    def getSize(self): return self.size
    def empty(self): self._g.clearCbf()
    def destroy(self): self._g.destroy()
    def toBytes(self): return self._g.exportFilter(N.CBF)
    def fromBytes(self, data): self._g.importFilter(N.CBF, data)

    def getBloomFilter(self, minCount):
        """:328-338 — the plain Bloom filter of the counters with MiniFloat.toFloat(count) >= minCount, built on the device"""
        bf = BloomFilter(self.size, self.numHash, self.k, device=self.device)
        N.check(N.lib.rb_cbf_to_bloom(self._g.h, float(minCount), bf._g.h, N.DBGBF))
        return bf


class PairedKeysBloomFilter(BloomFilter):
    """R/bloom/PairedKeysBloomFilter.java:40-231: one bit array addressed by the hash values of a k-mer PAIR — add / lookup /
    lookupThenAdd / getFPR / getOptimalSize / empty / destroy are BloomFilter's statements on `bitArrayPair` (:133-170,
    :205-230), so the class is BloomFilter under its own name.  (In the reference only the static getExpectedSize is
    reached, R/RNABloom.java:7010; the live pair filters are plain BloomFilters, R/graph/BloomFilterDeBruijnGraph.java:102, 354.)"""

    def getNumhash(self): return self.numHash                                   # :101-103 (sic)

    @staticmethod
    def getExpectedSize(expNumElements, fpr, numHash):                          # :213-216
        return BloomFilter.getExpectedSize(expNumElements, fpr, numHash)
