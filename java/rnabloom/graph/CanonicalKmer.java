package rnabloom.graph;

import java.util.Arrays;

import static rnabloom.bloom.hash.HashFunction.combineHashValues;
import static rnabloom.util.SeqUtils.stringToBytes;

/**
 * Drop-in for src/rnabloom/graph/CanonicalKmer.java (:37-519): a k-mer that carries the hashes of both strands; its base hash is the
 * smaller of the two (signed, as Math.min has it), which is what the graph's filters are asked with.  The neighbourhood methods are
 * Kmer's — one NativeGraph.neighbors call per k-mer (see Kmer) — with the three hooks below: the reverse hash goes into the call, the
 * native counts apply when the graph hashes canonically (every graph that is not stranded), and a candidate becomes a CanonicalKmer
 * with both rolled hashes (Canonical{Predecessors,Successors,LeftVariants,RightVariants}NTHashIterator.fHashVal / rHashVal).
 */
public class CanonicalKmer extends Kmer {

    protected long rHashVal;

    public CanonicalKmer(String seq, int k, float count, long fHashVal, long rHashVal) {
        this(stringToBytes(seq, k), count, fHashVal, rHashVal);
    }

    public CanonicalKmer(byte[] bytes, float count, long fHashVal, long rHashVal) {
        super(bytes, count, fHashVal);
        this.rHashVal = rHashVal;
    }

    @Override
    public long getHash() { return Math.min(fHashVal, rHashVal); }

    @Override
    public long getReverseComplementHash() { return getHash(); }

    @Override
    public long getKmerPairHashValue(Kmer rightPartner) {
        if (rightPartner instanceof CanonicalKmer) return getKmerPairHashValue((CanonicalKmer) rightPartner);
        return combineHashValues(this.fHashVal, rightPartner.fHashVal);
    }

    public long getKmerPairHashValue(CanonicalKmer rightPartner) {
        return Math.min(combineHashValues(fHashVal, rightPartner.fHashVal), combineHashValues(rightPartner.rHashVal, rHashVal));
    }

    @Override
    public boolean equals(Object obj) {
        if (this == obj) return true;
        if (obj == null || getClass() != obj.getClass()) return false;
        return Arrays.equals(bytes, ((CanonicalKmer) obj).bytes);
    }

    public long getFHash() { return fHashVal; }

    public long getRHash() { return rHashVal; }

    @Override
    protected long reverseHashForNative() { return rHashVal; }

    @Override
    protected boolean nativeCountsApply(BloomFilterDeBruijnGraph graph) { return !graph.isStranded(); }

    @Override
    protected long candidateHash(Hood h, int i) { return Math.min(h.f[i], h.r[i]); }

    @Override
    protected Kmer candidate(byte[] myBytes, float myCount, Hood h, int i) { return new CanonicalKmer(myBytes, myCount, h.f[i], h.r[i]); }

    // (every neighbourhood method of the reference's class — has* / getNum* / get{Predecessors,Successors} x4 / getMaxCov* / get*Variants /
    //  hasDepth* — is inherited: the hooks above are the only difference between the two classes' loops)
    @Override
    public boolean hasPredecessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return super.hasPredecessors(k, numHash, graph); }

    @Override
    public boolean hasSuccessors(int k, int numHash, BloomFilterDeBruijnGraph graph) { return super.hasSuccessors(k, numHash, graph); }
}
