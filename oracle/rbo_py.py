"""oracle/rbo_py.py — a SECOND, independent restatement of the reference's sketch iterators (TEST INFRASTRUCTURE, like
everything under oracle/: imported by tests/ and tools/ only).

oracle/rb_oracle_sketch.c restates the Java loops statement by statement, and the HIP kernels were written against it —
a misreading of the Java would pass on both sides.  This file is written from the Java again in a different shape, so
that C oracle == Python oracle == HIP is a three-way check:

  * k-mer hashes from scratch for every k-mer (no rolling): NTP64 / NTP64 of the reverse complement with the literal
    seeds, the reverse strand through the `ch & 7` lookup (R/bloom/hash/NTHash.java:30, 39-43, 133-166, 318-373);
  * strobemers by brute force: every candidate of a window gets its combined hash, the winner is the unsigned minimum
    with the tie rule the comparison operator implies (`>=` keeps replacing: the LAST minimum; `>`: the FIRST) —
    StrobeHashIterator.java:73-164, CanonicalStrobeHashIterator.java:79-140, Strobe3HashIterator.java:78-151,
    CanonicalStrobe3HashIterator.java:85-225.  (get / getInterval slide the strobe across equal k-mer hashes; an equal
    k-mer hash gives an equal combined hash, which `>=` would take anyway: same result as next(), by construction here);
  * minimizers through a literal LongRollingWindow object with its circular buffer (R/util/LongRollingWindow.java:23-83)
    driven as MinimizerHashIterator drives it (R/bloom/hash/MinimizerHashIterator.java:42-112);
  * GraphUtils.getMinimizers as what it computes: the sorted set of the window minima (R/util/GraphUtils.java:2462-2549).

Pure Python integers: slow, for reads of a few hundred bases."""

M64 = (1 << 64) - 1
SEED = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
# forward strand: msTab[ch] (NTHash.java:45-131): the letters A C G T U in both cases, everything else 0
FWD = {ord(c): SEED[c.upper().replace("U", "T")] for c in "ACGTUacgtu"}
# reverse strand: msTab[ch & cpOff], cpOff = 7 (:30): rows 0..7 of the table are N T N G A A N C
REV = [0, SEED["T"], 0, SEED["G"], SEED["A"], SEED["A"], 0, SEED["C"]]


def rotl(v, s):
    s %= 64
    return ((v << s) | (v >> (64 - s))) & M64 if s else v


def s64(v):
    return v - (1 << 64) if v >> 63 else v


def kmer_hashes(seq, k):
    """(f, r) of every k-mer, each from scratch: f = xor_i rotl(seed(s_i), k-1-i) (:332-337), r = xor_i rotl(seed'(s_i), i)
    with seed' the complement's seed looked up through ch & 7 (:367-373)"""
    seq = bytes(seq)
    f, r = [], []
    for p in range(len(seq) - k + 1):
        fv = rv = 0
        for i in range(k):
            ch = seq[p + i]
            fv ^= rotl(FWD.get(ch, 0), k - 1 - i)
            rv ^= rotl(REV[ch & 7], i)
        f.append(fv); r.append(rv)
    return f, r


def mode_hashes(seq, k, mode):
    """hVals[0] of the NTHashIterator (0), CanonicalNTHashIterator (1: signed minimum, :488-494) or
    ReverseComplementNTHashIterator in forward position order (2)"""
    f, r = kmer_hashes(seq, k)
    if mode == 0:
        return f
    if mode == 2:
        return r
    return [rv if s64(rv) < s64(fv) else fv for fv, rv in zip(f, r)]


def combine(a, b):
    """HashFunction.combineHashValues (R/bloom/hash/HashFunction.java:260-263); 0x9e3779b9 is an int literal: sign-extended"""
    return a ^ ((b + 0xFFFFFFFF9E3779B9 + ((a << 6) & M64) + (b >> 2)) & M64)


def _argmin(values, first, last_wins):
    """index (offset by `first`) of the unsigned minimum; ties: the last one if the Java compares with >=, else the first"""
    m = min(values)
    idx = [i for i, v in enumerate(values) if v == m]
    return first + (idx[-1] if last_wins else idx[0]), m


def randstrobes(seq, k, n, wmin, wmax, canonical=False):
    """[(hash, [p, strobe positions...])] of StrobeHashIterator / CanonicalStrobeHashIterator for p = 0 .. max"""
    f, r = kmer_hashes(seq, k)
    nk = len(f)
    if len(seq) < k or not nk > wmax * (n - 1):
        return []
    out = []
    for p in range(nk - wmax * (n - 2) - wmin):                 # p <= max = numKmers - wMax*(n-2) - wMin - 1
        h, pos = f[p], [p]
        for s in range(n - 1):
            lo, hi = p + s * wmax + wmin, min(p + s * wmax + wmax, nk)
            q, h = _argmin([combine(h, f[i]) for i in range(lo, hi)], lo, True)
            pos.append(q)
        if canonical:                                           # reverse hashes of the same positions, back to front
            rh = r[pos[-1]]
            for q in reversed(pos[:-1]):
                rh = combine(r[q], rh)
            h = rh if s64(rh) < s64(h) else h                   # Math.min(long, long)
        out.append((h, pos))
    return out


def strobemer_intervals(seq, k, n, wmin, wmax):
    """StrobeHashIterator.getInterval(p): (hash, p, last strobe + k - 1)"""
    return [(h, pos[0], pos[-1] + k - 1) for h, pos in randstrobes(seq, k, n, wmin, wmax)]


def strobe3(seq, k, wmin, wmax, canonical=False):
    """[(hash, [pos1, p, pos3])] for p = getMin() .. getMax()"""
    f, r = kmer_hashes(seq, k)
    nk = len(f)
    if len(seq) < k or not nk > 2 * wmin:
        return []
    lo_p, hi_p = (wmax, nk - 1 - wmax) if canonical else (wmin, nk - 1 - wmin)
    out = []
    for p in range(lo_p, hi_p + 1):
        up = range(max(0, p - wmax + 1), p - wmin + 1)
        down = range(p + wmin, min(p + wmax, nk))
        p1, h1 = _argmin([combine(f[i], f[p]) for i in up], up[0], False)          # `>`: first minimum
        p3, h3 = _argmin([combine(h1, f[i]) for i in down], down[0], canonical)     # canonical: `>=`, plain: `>`
        if canonical:
            q3, g3 = _argmin([combine(r[i], r[p]) for i in down], down[0], True)    # reverse strand: downstream first, `>=`
            q1, g1 = _argmin([combine(g3, r[i]) for i in up], up[0], False)         # then upstream, `>`
            if h3 > g1:                                                              # Long.compareUnsigned(fh3, rh1) > 0
                out.append((g1, [q1, p, q3]))
                continue
        out.append((h3, [p1, p, p3]))
    return out


class LongRollingWindow:
    """R/util/LongRollingWindow.java:23-83, field by field"""

    def __init__(self, window):
        self.window = list(window)
        self.size = len(window)
        self.index = self.size - 1
        self.pos = self.index
        self._update_min_index()

    def set_index(self, i, pos):
        self.index, self.pos = i, pos

    def roll(self, new_val):
        self.pos += 1
        self.index += 1
        if self.index >= self.size:
            self.index = 0
        self.window[self.index] = new_val
        if self.min_index == self.index:
            self._update_min_index()
        elif new_val < self.window[self.min_index]:
            self.min_index = self.index

    def _update_min_index(self):
        self.min_index = 0
        m = self.window[0]
        for i in range(1, self.size):
            if self.window[i] < m:
                m, self.min_index = self.window[i], i

    def get_min(self):
        return self.window[self.min_index]

    def get_min_pos(self):
        if self.min_index > self.index:
            return self.pos - self.index - self.size + self.min_index
        return self.pos - self.index + self.min_index


def minimizers(seq, k, w, mode):
    """MinimizerHashIterator.next() for every window: [(window.getMin(), window.getMinPos())], hashes as unsigned"""
    h = [s64(v) for v in mode_hashes(seq, k, mode)]              # Java longs
    n_windows = len(h) - w + 1
    if len(seq) < k or n_windows <= 0:
        return []
    first = h[:w - 1] + [0]                                      # start(): w-1 hashes, a 0 in the last slot (:54-62)
    win = LongRollingWindow(first)
    win.set_index(w - 2, w - 2)
    out = []
    for p in range(n_windows):
        win.roll(h[p + w - 1])
        out.append((win.get_min() & M64, win.get_min_pos()))
    return out


def minimizers_next(seq, k, w, mode):
    """what SeqUtils.getMinimizerChainString reads (R/util/SeqUtils.java:1731-1757): nextMinimizer() while hasNext(), until the
    position stops moving: the first window's minimizer, then one entry per move of the minimum's position"""
    allw = minimizers(seq, k, w, mode)
    out = []
    for hv, pos in allw:
        if not out or pos > out[-1][1]:
            out.append((hv, pos))
    return out


def minimizer_set(seq, k, w, mode, stale=0):
    """GraphUtils.getMinimizers: sorted (signed) set of the window minima; numKmers <= windowSize: the one value
    min(stale, every k-mer hash) (:2480-2494, `stale` = what hvals[0] held before the first next())"""
    h = [s64(v) for v in mode_hashes(seq, k, mode)] if len(seq) >= k else []
    if len(seq) - k + 1 <= w:
        return [min([s64(stale)] + h) & M64]
    return [v & M64 for v in sorted({min(h[t:t + w]) for t in range(len(h) - w + 1)})]


def kmer_pair_hashes(seq, k, shift, canonical):
    """SeqSubsampler.kmerBased (R/util/SeqSubsampler.java:176-179, 266-268)"""
    f, r = kmer_hashes(seq, k)
    out = []
    for i in range(len(f) - shift):
        pf = combine(f[i], f[i + shift])
        if canonical:
            pr = combine(r[i + shift], r[i])
            pf = pr if s64(pr) < s64(pf) else pf
        out.append(pf)
    return out


# ---- stage 1 (round 4): a second restatement of the INSERT, in another shape than oracle/rb_oracle.c ------------------------
# rb_oracle.c follows FastqToGraphWorker and the filters byte array by byte array, with rolling hashes and a hand-written
# segmentation loop; the HIP pipeline was written against it.  This one is written from the Java again: every k-mer hashed from
# scratch (kmer_hashes above), the reads cut by the reference's OWN regular expressions (tests/golden/seq_patterns.json: the
# literal parts of SeqUtils.getPhred33Pattern / getNucleotideCharsPattern, run by Python's re), the filters as a set of bit
# indices and a dict of counters (no byte arrays, no words, no masks).  C oracle == this == HIP on the committed fixtures is
# a three-way check of everything between the hash layer and the saved files.  Counters below 16 only (no random draw:
# MiniFloat.increment is deterministic there, R/util/MiniFloat.java:31-37) — the fixtures are built that way.
MULTI_SEED = 0x90b45d39fb6da1fa      # NTHash.java:33
MULTI_SHIFT = 27                     # :32


def multi_hashes(base, k, m):
    """NTM64(bVal, hVal[], k, m) (NTHash.java:518-527): hVal[0] = b; hVal[i] = t ^ (t >>> 27), t = b * (i ^ (k * multiSeed))"""
    out = [base & M64]
    for i in range(1, m):
        t = (base * (i ^ ((k * MULTI_SEED) & M64))) & M64
        out.append(t ^ (t >> MULTI_SHIFT))
    return out


def index_of(h, size):
    """BloomFilter.getIndex (R/bloom/BloomFilter.java:108-111): (hashVal >>> 1) % size"""
    return (h >> 1) % size


def worker_segments(seq, qual, k, min_qual, patterns):
    """while (mQual.find()) { mSeq.region(...); while (mSeq.find()) ... } (R/RNABloom.java:572-577) with compiled patterns
    (qual pattern, seq pattern); qual None: the FASTA worker's single pattern (:677-697)"""
    mq, ms = patterns
    if qual is None:
        return [(m.start(), m.end()) for m in ms.finditer(seq)]
    return [(m.start(), m.end()) for q in mq.finditer(qual) for m in ms.finditer(seq, q.start(), q.end())]


class Stage1:
    """BloomFilterDeBruijnGraph with dbgbf / rpkbf as sets of bit indices and cbf as {index: byte}."""

    def __init__(self, dbg_bits, cbf_bytes, pk_bits, h_dbg, h_cbf, h_pk, k, stranded):
        self.k, self.stranded = k, stranded
        self.size = (dbg_bits, cbf_bytes, pk_bits)
        self.h = (h_dbg, h_cbf, h_pk)
        self.dbg, self.cbf, self.rpk = set(), {}, set()
        self.kmers = self.pairs = 0

    # -- BloomFilterDeBruijnGraph.add (R/graph/BloomFilterDeBruijnGraph.java:405-412)
    def add(self, base):
        hv = multi_hashes(base, self.k, max(self.h[0], self.h[1]))      # hashVals has max(dbgbfNumHash, cbfNumHash) entries (:90)
        # BloomFilter.lookupThenAdd (R/bloom/BloomFilter.java:147-155): getAndSet on every index, no early exit
        found = True
        for j in range(self.h[0]):
            i = index_of(hv[j], self.size[0])
            found = (i in self.dbg) and found
            self.dbg.add(i)
        if found:
            self.increment(hv)

    # -- CountingBloomFilter.increment (R/bloom/CountingBloomFilter.java:170-194)
    def increment(self, hv):
        idx = [index_of(hv[j], self.size[1]) for j in range(self.h[1])]
        mn = min(self.cbf.get(i, 0) for i in idx)                       # (the early exits at 0 change nothing: 0 is the minimum)
        assert mn < 16, "Stage1 restates the deterministic range of MiniFloat.increment only"
        updated = mn + 1                                                # b <= 7: +1; 8..15: 2^((b>>3)-1) = 1, the draw % 1 == 0 always
        for i in idx:
            if self.cbf.get(i, 0) == mn:                                # compareAndSwap(index, min, updated): "update min count only"
                self.cbf[i] = updated

    # -- addReadSingleKmerPair -> rpkbf.add (:455-457, BloomFilter.add :133-137)
    def add_pair(self, base):
        for hvj in multi_hashes(base, self.k, self.h[2]):
            self.rpk.add(index_of(hvj, self.size[2]))

    def add_reads(self, reads, quals, min_qual, patterns, reverse_complement=False, pair_distance=0):
        """FastqToGraphWorker.run (R/RNABloom.java:551-634) for one file; reads / quals: lists of bytes (quals None: FASTA)"""
        k, d = self.k, pair_distance
        for n, seq in enumerate(reads):
            if len(seq) < k:
                continue                                                # :566-569
            f, r = kmer_hashes(seq, k)
            if self.stranded:
                h0 = r if reverse_complement else f                     # ReverseComplementNTHashIterator / NTHashIterator (:540-545)
            else:
                h0 = [rv if s64(rv) < s64(fv) else fv for fv, rv in zip(f, r)]     # NTPC64: signed min (NTHash.java:465-475)
            for s, e in worker_segments(seq, None if quals is None else quals[n], k, min_qual, patterns):
                for p in range(s, e - k + 1):                           # itr.start(seq, start, end): positions start .. end - k
                    self.add(h0[p])
                    self.kmers += 1
                if d > 0:
                    for p in range(s, e - k - d + 1):                   # pitr: positions start .. end - k - d (PairedNTHashIterator.java:53-61)
                        if self.stranded and not reverse_complement:
                            pb = combine(f[p], f[p + d])                                   # PairedNTHashIterator :67-69
                        elif self.stranded:
                            pb = combine(r[p + d], r[p])                                   # ReverseComplementPaired... :40-42: combine(R, L)
                        else:
                            a, b = combine(f[p], f[p + d]), combine(r[p + d], r[p])        # CanonicalPaired... :43-45
                            pb = b if s64(b) < s64(a) else a                               # Math.min(long, long)
                        self.add_pair(pb)
                        self.pairs += 1

    # -- the saved bytes (UnsafeBitBuffer: bit i = byte i / 8, mask 1 << (i % 8), R/bloom/buffer/UnsafeBitBuffer.java:45-62; one byte per counter)
    @staticmethod
    def _bits_to_bytes(bits, size):
        out = bytearray((size + 7) // 8)
        for i in bits:
            out[i >> 3] |= 1 << (i & 7)
        return bytes(out)

    def dbgbf_bytes(self):
        return self._bits_to_bytes(self.dbg, self.size[0])

    def rpkbf_bytes(self):
        return self._bits_to_bytes(self.rpk, self.size[2])

    def cbf_bytes(self):
        out = bytearray(self.size[1])
        for i, v in self.cbf.items():
            out[i] = v
        return bytes(out)
