"""Host mirror of the part of rnabloom.util.GraphUtils that sits directly on the neighbour-extension loop, batched over
many k-mer pairs: the graph work (every getMaxCovSuccessor / getMaxCovPredecessor step) runs in two rb_graph_walk calls
per batch; what is left on the host is the bookkeeping of R/util/GraphUtils.java:1591-1675."""
import math

LOW_COMPLEXITY_THRESHOLD_SHORT_SEQ = 0.95          # R/util/SeqUtils.java:61
_IDX = {65: 0, 67: 1, 71: 2, 84: 3, 85: 3}         # nucleotideArrayIndex, R/util/SeqUtils.java:315-330 (upper case only)


def _java_round(x):
    return int(math.floor(x + 0.5))                # Math.round(float)


def isLowComplexityShort(seq):
    """R/util/SeqUtils.java:499-543 (String version): homopolymer / di- / tri-nucleotide repeats counted over a sliding
    window of three bases, then the two-letter content test.  length/2 and length/3 are integer divisions."""
    seq = bytes(seq)
    length = len(seq)
    t1 = min(32767, _java_round(length * LOW_COMPLEXITY_THRESHOLD_SHORT_SEQ))
    t2 = min(32767, _java_round((length // 2) * LOW_COMPLEXITY_THRESHOLD_SHORT_SEQ))
    t3 = min(32767, _java_round((length // 3) * LOW_COMPLEXITY_THRESHOLD_SHORT_SEQ))
    nf1 = [0] * 4
    nf2 = [[0] * 4 for _ in range(4)]
    nf3 = [[[0] * 4 for _ in range(4)] for _ in range(4)]
    c3, c2, c1 = _IDX[seq[0]], _IDX[seq[1]], _IDX[seq[2]]
    nf1[c3] += 1; nf1[c2] += 1; nf1[c1] += 1
    nf2[c3][c2] += 1; nf2[c2][c1] += 1
    nf3[c3][c2][c1] += 1
    for ch in seq[3:]:
        c3, c2, c1 = c2, c1, _IDX[ch]
        nf1[c1] += 1
        if nf1[c1] >= t1:
            return True
        nf2[c2][c1] += 1
        if nf2[c2][c1] >= t2:
            return True
        nf3[c3][c2][c1] += 1
        if nf3[c3][c2][c1] >= t3:
            return True
    return any(nf1[a] + nf1[b] >= t1 for a in range(4) for b in range(a + 1, 4))


def _kmers_right(seed, appended, k):
    s = seed + appended
    return [s[j + 1:j + 1 + k] for j in range(len(appended))]


def _kmers_left(seed, prepended, k):
    """k-mers of a left walk in the order they were found (nearest to the seed first)"""
    s = prepended[::-1] + seed
    n = len(prepended)
    return [s[n - 1 - j:n - 1 - j + k] for j in range(n)]


def getMaxCoveragePaths(graph, lefts, rights, bound, minKmerCov=1.0):
    """GraphUtils.getMaxCoveragePath(graph, left, right, bound, lookahead, minKmerCov) (R/util/GraphUtils.java:1591-1675) for
    many (left, right) pairs at once.  lefts / rights: k-mers as upper-case bytes.  Returns, per pair, the list of k-mers
    strictly between left and right along the path found, or None — exactly what the reference returns."""
    n = len(lefts)
    k = graph.k
    out = [None] * n
    bases, _, _, _, ln, reason = graph.walkMaxCov(lefts, 0, bound, minKmerCov, rights, hashes=False, counts=False)      # :1603-1620
    left_paths = []
    todo = []
    for i in range(n):
        lp = _kmers_right(lefts[i], bytes(bases[i, :ln[i]]), k)
        left_paths.append(lp)
        if reason[i] == 1:
            out[i] = lp                                                                     # :1609-1611
        elif reason[i] != 4:
            todo.append(i)
    if todo:
        bases, _, _, _, ln, reason = graph.walkMaxCov([rights[i] for i in todo], 1, bound, minKmerCov, [lefts[i] for i in todo], hashes=False, counts=False)   # :1629-1672
        for j, i in enumerate(todo):
            if reason[j] == 4:
                continue
            rp = _kmers_left(rights[i], bytes(bases[j, :ln[j]]), k)
            lp = left_paths[i]
            lset = {km: idx for idx, km in reversed(list(enumerate(lp)))}                    # first index of every k-mer
            last = {}
            for idx, km in enumerate(lp):
                last[km] = idx                                                              # descendingIterator finds the LAST equal one
            hit = next((d for d, km in enumerate(rp) if km in lset), None)
            if hit is not None:                                                             # :1644-1664: the right path meets the left path
                best = rp[hit]
                if isLowComplexityShort(best):
                    continue
                out[i] = lp[:last[best]] + [best] + rp[:hit][::-1]
            elif reason[j] == 1:                                                            # :1637-1639
                out[i] = rp[::-1]
    return out


# ---- the k-mer-list pair helpers of BloomFilterDeBruijnGraph (R/graph/BloomFilterDeBruijnGraph.java:474-526), batched over sequences ----
import numpy as np

from . import _native as N


def _combine(a, b):
    """HashFunction.combineHashValues, R/bloom/hash/HashFunction.java:260-263 (uint64 arithmetic wraps like Java's long)"""
    with np.errstate(over="ignore"):
        return a ^ (b + np.uint64(0xFFFFFFFF9E3779B9) + (a << np.uint64(6)) + (b >> np.uint64(2)))


def kmerPairHashValues(f, r, d, stranded):
    """Kmer.getKmerPairHashValue / CanonicalKmer.getKmerPairHashValue (R/graph/Kmer.java:65-67, CanonicalKmer.java:61-72)
    of the k-mers i and i + d of one sequence, for every i: combine(fL, fR), canonical: the signed minimum of that and
    combine(rR, rL)."""
    fl, fr = f[:-d], f[d:]
    p = _combine(fl, fr)
    if not stranded:
        q = _combine(r[d:], r[:-d])
        p = np.where(q.view(np.int64) < p.view(np.int64), q, p)
    return p


def _pairs_per_sequence(graph, seqs, d):
    ko, f, r, _ = graph.getKmers(seqs)
    out, spans = [], []
    for i in range(len(seqs)):
        a, b = int(ko[i]), int(ko[i + 1])
        n = b - a - d
        spans.append(max(n, 0))
        if n > 0:
            out.append(kmerPairHashValues(f[a:b], r[a:b], d, graph.stranded))
    return (np.concatenate(out) if out else np.zeros(0, np.uint64)), spans


def containsAllPairedKmers(graph, seqs, d):
    """graph.containsAllPairedKmers(kmers) for many sequences (:496-511): False for a sequence without a pair"""
    p, spans = _pairs_per_sequence(graph, seqs, d)
    hit = graph.lookupFragmentKmerPair(p) if p.size else np.zeros(0, bool)
    res, at = [], 0
    for n in spans:
        res.append(bool(n > 0 and hit[at:at + n].all()))
        at += n
    return res


def lookupAndAddAllPairedKmers(graph, seqs, d):
    """graph.lookupAndAddAllPairedKmers(kmers) for many sequences, in order (:513-526): per sequence the AND of
    fpkbf.lookupThenAdd over its pairs (True for a sequence without a pair) — a later sequence sees the pairs of the earlier ones."""
    p, spans = _pairs_per_sequence(graph, seqs, d)
    hit = graph.lookupThenAdd(N.FPKBF, p) if p.size else np.zeros(0, bool)
    res, at = [], 0
    for n in spans:
        res.append(bool(hit[at:at + n].all()))
        at += n
    return res


def getKmersMinCoverage(graph, seqs, minCoverage):
    """HashFunction.getKmers(seq, numHash, graph, minCoverage) (R/bloom/hash/HashFunction.java:86-134) for many sequences:
    the k-mers of the "longest" run with count >= minCoverage — with the reference's bookkeeping kept as it is: the
    list that is open when the scan starts is the result unless a LATER closed run replaces it (its length is never
    recorded, so any later closed run does), and a run still open at the end of the sequence is never compared.
    Returns per sequence (first k-mer index, number of k-mers, their counts)."""
    ko, _, _, c = graph.getKmers(seqs)
    out = []
    for i in range(len(seqs)):
        cnt = c[int(ko[i]):int(ko[i + 1])]
        cur_start, cur_len, cur_min = 0, 0, float("inf")
        best = None                      # None: the result is still the list that was open at the start
        first_open = True                # the current list IS that first list
        first = (0, 0)
        best_len, best_min = 0, float("inf")
        for j, x in enumerate(cnt):
            if x >= minCoverage:
                if cur_len == 0:
                    cur_start = j
                cur_len += 1
                cur_min = min(cur_min, float(x))
            elif cur_len:
                if first_open:
                    first = (cur_start, cur_len)
                elif cur_len > best_len or (cur_len == best_len and cur_min > best_min):
                    best, best_len, best_min = (cur_start, cur_len), cur_len, cur_min
                first_open = False
                cur_len, cur_min = 0, float("inf")
        if first_open:
            first = (cur_start, cur_len) if cur_len else (0, 0)
        s, n = best if best is not None else first
        out.append((s, n, cnt[s:s + n].copy()))
    return out
