package rnabloom.bloom;

import java.io.File;
import java.io.FileNotFoundException;
import java.io.IOException;
import rnabloom.bloom.hash.HashFunction;
import rnabloom.graph.NativeGraph;

/**
 * Drop-in for rnabloom.bloom.PairedKeysBloomFilter (src/rnabloom/bloom/PairedKeysBloomFilter.java:40-231): ONE bit array
 * addressed by the hash values of a k-mer pair.  Its add / lookup / lookupThenAdd / getFPR / getOptimalSize / empty / destroy are
 * BloomFilter's statements on `bitArrayPair` (:133-170, :205-230), so the class is a BloomFilter under its own method names.
 * (In the reference only the static getExpectedSize is reached, src/rnabloom/RNABloom.java:7010; the graph's live pair filters
 * are plain BloomFilters, src/rnabloom/graph/BloomFilterDeBruijnGraph.java:102, 354.)
 */
public class PairedKeysBloomFilter {
    private final BloomFilter pairs;

    public PairedKeysBloomFilter(long size, int numHash, HashFunction hashFunction) {
        pairs = new BloomFilter(size, numHash, hashFunction);
    }

    public PairedKeysBloomFilter(File desc, File pairBits, HashFunction hashFunction) throws FileNotFoundException, IOException {
        pairs = new BloomFilter(desc, pairBits, hashFunction);
    }

    public int getNumhash() { return pairs.getNumHash(); }            // (sic, :102)

    public void save(File desc, File bits) throws IOException { pairs.save(desc, bits); }

    public void add(final long hashValPair) { pairs.add(hashValPair); }

    public void add(final long[] hashValsPair) { pairs.add(hashValsPair); }

    public boolean lookup(final long hashValsPair) { return pairs.lookup(hashValsPair); }

    public boolean lookup(final long[] hashValsPair) { return pairs.lookup(hashValsPair); }

    public boolean lookupThenAdd(final long hashValsPair) { return pairs.lookupThenAdd(hashValsPair); }

    public boolean lookupThenAdd(final long[] hashVals) { return pairs.lookupThenAdd(hashVals); }

    public void destroy() { pairs.destroy(); }

    public void empty() { pairs.empty(); }

    public boolean equivalent(PairedKeysPartitionedBloomFilter bf) {
        return false;                 // the partitioned class is never constructed in the reference (SURVEY.md s7)
    }

    public float getFPR() { return pairs.getFPR(); }

    public static long getExpectedSize(long expNumElements, float fpr, int numHash) {
        return NativeGraph.expectedSize(expNumElements, fpr, numHash);
    }

    public long getOptimalSize(float fpr) { return pairs.getOptimalSize(fpr); }

    public long getPopCount() { return pairs.getPopCount(); }
}
