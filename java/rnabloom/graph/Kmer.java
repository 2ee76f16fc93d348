package rnabloom.graph;

import java.util.ArrayDeque;
import java.util.Arrays;

import rnabloom.bloom.BloomFilter;

import static rnabloom.bloom.hash.HashFunction.combineHashValues;
import static rnabloom.bloom.hash.NTHash.NTP64RC;
import static rnabloom.util.SeqUtils.bytesToString;
import static rnabloom.util.SeqUtils.stringToBytes;

/**
 * Drop-in for src/rnabloom/graph/Kmer.java (:41-487) over the native graph: same fields, constructors and public methods.
 *
 * The reference asks the graph once per candidate base — a (Predecessors|Successors|LeftVariants|RightVariants)NTHashIterator rolls
 * one hash, graph.getCount(hVals) looks it up, four times per k-mer (:199-255, :301-405).  Here a k-mer's neighbourhood is ONE call:
 * NativeGraph.neighbors hashes the four candidates on the device (rb_graph_neighbors, csrc/rb_query.hip k_neighbors; order A, C, G, T
 * as NUCLEOTIDES_BYTES has it) and returns their forward hashes, reverse hashes and graph counts; the methods below only filter and
 * build objects.  The entry of a variants call whose base equals the replaced base is the k-mer itself and is skipped, which is what
 * iterating getAltNucleotides(charOut) does; for a charOut outside ACGTU all four are alternatives (SeqUtils.getAltNucleotides).
 *
 * A plain Kmer hashes the forward strand only (stranded graphs: HashFunction.getKmers builds Kmer there, CanonicalKmer otherwise).  Should
 * one be asked about a graph that is NOT stranded, the counts of the native call (canonical hashes) do not apply: the four forward hashes
 * it returned are looked up with one batched getCount instead.
 */
public class Kmer {

    public byte[] bytes;
    public float count;
    protected long fHashVal;

    protected static final byte[] ACGT = {'A', 'C', 'G', 'T'};

    public Kmer(String seq, int k, float count, long fHashVal) {
        this(stringToBytes(seq, k), count, fHashVal);
    }

    public Kmer(byte[] bytes, float count, long fHashVal) {
        this.bytes = bytes;
        this.count = count;
        this.fHashVal = fHashVal;
    }

    public long getHash() { return fHashVal; }

    public long getReverseComplementHash() { return NTP64RC(bytes, bytes.length); }

    public long getKmerPairHashValue(Kmer rightPartner) { return combineHashValues(this.fHashVal, rightPartner.fHashVal); }

    public boolean equals(Kmer other) { return Arrays.equals(bytes, other.bytes); }

    @Override
    public String toString() { return bytesToString(bytes, bytes.length); }

    @Override
    public int hashCode() { return (int) getHash(); }

    @Override
    public boolean equals(Object obj) {
        if (this == obj) return true;
        if (obj == null || getClass() != obj.getClass()) return false;
        return Arrays.equals(bytes, ((Kmer) obj).bytes);
    }

    // ---- one native call per neighbourhood ----

    /** the four candidates of one direction: forward hash, reverse hash, graph count, in base order A, C, G, T */
    protected static final class Hood {
        final long[] f = new long[4], r = new long[4];
        final float[] count = new float[4];
    }

    /** reverse-strand hash handed to the native call (a plain Kmer has none: stranded graphs never read it) */
    protected long reverseHashForNative() { return 0L; }

    /** does the native call's count (the GRAPH's hash of a candidate) answer for this class's hash of it? */
    protected boolean nativeCountsApply(BloomFilterDeBruijnGraph graph) { return graph.isStranded(); }

    /** this class's base hash of candidate i of a neighbourhood */
    protected long candidateHash(Hood h, int i) { return h.f[i]; }

    protected Kmer candidate(byte[] myBytes, float myCount, Hood h, int i) { return new Kmer(myBytes, myCount, h.f[i]); }

    protected final Hood hood(BloomFilterDeBruijnGraph graph, int direction, byte charOut) {
        Hood h = new Hood();
        NativeGraph.neighbors(graph.getHandle(), new long[]{fHashVal}, new long[]{reverseHashForNative()}, new byte[]{charOut}, 1, direction, h.f, h.r, h.count);
        if (!nativeCountsApply(graph)) {
            long[] base = new long[4];
            for (int i = 0; i < 4; ++i) base[i] = candidateHash(h, i);
            graph.getCountAll(base, 4, h.count);
        }
        return h;
    }

    protected static boolean isSelf(byte charOut, int i) {
        byte c = charOut == 'U' ? (byte) 'T' : charOut;
        return c == ACGT[i];
    }

    private byte[] shifted(int k, boolean left, byte in) {
        byte[] b = new byte[k];
        if (left) { System.arraycopy(bytes, 1, b, 0, k - 1); b[k - 1] = in; }
        else { System.arraycopy(bytes, 0, b, 1, k - 1); b[0] = in; }
        return b;
    }

    private int countPresent(BloomFilterDeBruijnGraph graph, int direction, byte charOut) {
        // graph.contains(hVals) = dbgbf.lookup; a k-mer in dbgbf has graph.getCount > 0 and vice versa (getCount is 0 unless dbgbf holds it, :562-570)
        Hood h = hood(graph, direction, charOut);
        int n = 0;
        for (int i = 0; i < 4; ++i) if (h.count[i] > 0) ++n;
        return n;
    }

    public boolean hasPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return countPresent(graph, NativeGraph.PREDECESSORS, bytes[k - 1]) > 0; }

    public boolean hasSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return countPresent(graph, NativeGraph.SUCCESSORS, bytes[0]) > 0; }

    public boolean hasAtLeastXPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph, int x) {
        return countPresent(graph, NativeGraph.PREDECESSORS, bytes[k - 1]) >= x;
    }

    public boolean hasAtLeastXSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph, int x) {
        return countPresent(graph, NativeGraph.SUCCESSORS, bytes[0]) >= x;
    }

    public int getNumPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return countPresent(graph, NativeGraph.PREDECESSORS, bytes[k - 1]); }

    public int getNumSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return countPresent(graph, NativeGraph.SUCCESSORS, bytes[0]); }

    public ArrayDeque<Kmer> getPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return getPredecessors(k, numHash, graph, 1); }

    public ArrayDeque<Kmer> getPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph, float minKmerCov) {
        ArrayDeque<Kmer> result = new ArrayDeque<>(4);
        getPredecessors(k, numHash, graph, result, minKmerCov);
        return result;
    }

    public void getPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph, ArrayDeque<Kmer> result, float minKmerCov) {
        Hood h = hood(graph, NativeGraph.PREDECESSORS, bytes[k - 1]);
        for (int i = 0; i < 4; ++i)
            if (h.count[i] >= minKmerCov) result.add(candidate(shifted(k, false, ACGT[i]), h.count[i], h, i));
    }

    public ArrayDeque<Kmer> getSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return getSuccessors(k, numHash, graph, 1); }

    public ArrayDeque<Kmer> getSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph, float minKmerCov) {
        ArrayDeque<Kmer> result = new ArrayDeque<>(4);
        getSuccessors(k, numHash, graph, result, minKmerCov);
        return result;
    }

    public void getSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph, ArrayDeque<Kmer> result, float minKmerCov) {
        Hood h = hood(graph, NativeGraph.SUCCESSORS, bytes[0]);
        for (int i = 0; i < 4; ++i)
            if (h.count[i] >= minKmerCov) result.add(candidate(shifted(k, true, ACGT[i]), h.count[i], h, i));
    }

    /** the neighbours that `bf` holds too (:257-299): one more batched lookup, of the four candidates in `bf` */
    private ArrayDeque<Kmer> gated(int k, BloomFilterDeBruijnGraph graph, BloomFilter bf, boolean successors) {
        ArrayDeque<Kmer> result = new ArrayDeque<>(4);
        Hood h = hood(graph, successors ? NativeGraph.SUCCESSORS : NativeGraph.PREDECESSORS, successors ? bytes[0] : bytes[k - 1]);
        long[] base = new long[4];
        for (int i = 0; i < 4; ++i) base[i] = candidateHash(h, i);
        byte[] in = new byte[4];
        bf.lookupAll(base, 4, in);
        for (int i = 0; i < 4; ++i)
            if (in[i] != 0 && h.count[i] > 0) result.add(candidate(shifted(k, successors, ACGT[i]), h.count[i], h, i));
        return result;
    }

    public ArrayDeque<Kmer> getPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph, BloomFilter bf) { return gated(k, graph, bf, false); }

    public ArrayDeque<Kmer> getSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph, BloomFilter bf) { return gated(k, graph, bf, true); }

    private Kmer maxCov(int k, BloomFilterDeBruijnGraph graph, float minKmerCov, boolean successor) {
        Hood h = hood(graph, successor ? NativeGraph.SUCCESSORS : NativeGraph.PREDECESSORS, successor ? bytes[0] : bytes[k - 1]);
        int best = -1;
        float bestCount = -1;
        for (int i = 0; i < 4; ++i)                         // the first of equal counts wins (strict >, A before C before G before T: :313, :341)
            if (h.count[i] >= minKmerCov && h.count[i] > bestCount) { bestCount = h.count[i]; best = i; }
        return best < 0 ? null : candidate(shifted(k, successor, ACGT[best]), bestCount, h, best);
    }

    public Kmer getMaxCovSuccessor(int k, int numHash, BloomFilterDeBruijnGraph graph, float minKmerCov) { return maxCov(k, graph, minKmerCov, true); }

    public Kmer getMaxCovPredecessor(int k, int numHash, BloomFilterDeBruijnGraph graph, float minKmerCov) { return maxCov(k, graph, minKmerCov, false); }

    private ArrayDeque<Kmer> variants(int k, BloomFilterDeBruijnGraph graph, float minKmerCov, boolean left) {
        ArrayDeque<Kmer> result = new ArrayDeque<>(4);
        final int at = left ? 0 : k - 1;
        final byte charOut = bytes[at];
        Hood h = hood(graph, left ? NativeGraph.LEFT_VARIANTS : NativeGraph.RIGHT_VARIANTS, charOut);
        for (int i = 0; i < 4; ++i) {
            if (isSelf(charOut, i) || !(h.count[i] >= minKmerCov)) continue;
            byte[] myBytes = Arrays.copyOf(bytes, k);
            myBytes[at] = ACGT[i];
            result.add(candidate(myBytes, h.count[i], h, i));
        }
        return result;
    }

    public ArrayDeque<Kmer> getLeftVariants(int k, int numHash, BloomFilterDeBruijnGraph graph) { return getLeftVariants(k, numHash, graph, 1); }

    public ArrayDeque<Kmer> getLeftVariants(int k, int numHash, BloomFilterDeBruijnGraph graph, float minKmerCov) { return variants(k, graph, minKmerCov, true); }

    public ArrayDeque<Kmer> getRightVariants(int k, int numHash, BloomFilterDeBruijnGraph graph) { return getRightVariants(k, numHash, graph, 1); }

    public ArrayDeque<Kmer> getRightVariants(int k, int numHash, BloomFilterDeBruijnGraph graph, float minKmerCov) { return variants(k, graph, minKmerCov, false); }

    /**
     * The reference walks the tree of ALL extensions without ever asking the graph (:407-445, :447-485: no contains / getCount inside the
     * loop), so its answer does not depend on the graph: the first branch reaches `depth` and the method returns true, for every depth.
     * No caller exists in the reference; the observable behaviour is kept.
     */
    public boolean hasDepthRight(int k, int numHash, BloomFilterDeBruijnGraph graph, int depth) { return true; }

    public boolean hasDepthLeft(int k, int numHash, BloomFilterDeBruijnGraph graph, int depth) { return true; }
}
