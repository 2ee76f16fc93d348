package rnabloom.bloom;

import java.io.File;
import java.io.FileNotFoundException;
import java.io.FileWriter;
import java.io.IOException;
import rnabloom.bloom.hash.HashFunction;
import rnabloom.graph.NativeGraph;

/**
 * Drop-in for rnabloom.bloom.CountingBloomFilter (src/rnabloom/bloom/CountingBloomFilter.java:41-339): 8-bit MiniFloat
 * counters with conservative update, resident in HBM.  Either a handle of its own whose counting filter is the only real
 * filter (SeqSubsampler's use) or a view of a graph's cbf (BloomFilterDeBruijnGraph.getCbf()).
 *
 * MiniFloat.increment draws from an UNSEEDED Math.random() in the reference (src/rnabloom/util/MiniFloat.java:34), so counts
 * of 16 and more are not reproducible there even between two runs; the library draws from a counter-based generator
 * (seed, op ordinal) with the same success probability 2^-s — see DESIGN.md s2.
 */
public class CountingBloomFilter implements CountingBloomFilterInterface {
    protected long handle;
    protected final boolean owner;
    protected int numHash;
    protected long size;
    protected HashFunction hashFunction;
    protected long popcount = -1;

    private static final long TINY = 64;

    public CountingBloomFilter(long size, int numHash, HashFunction hashFunction) {
        this.size = size;
        this.numHash = numHash;
        this.hashFunction = hashFunction;
        this.owner = true;
        this.handle = NativeGraph.create(TINY, size, 0, 1, numHash, 1, hashFunction.getK(), true, false, NativeGraph.defaultDevice(), 0L);
    }

    public CountingBloomFilter(File desc, File bytes, HashFunction hashFunction) throws FileNotFoundException, IOException {
        long[] sn = BloomFilter.readDesc(desc);
        this.size = sn[0];
        this.numHash = (int) sn[1];
        this.hashFunction = hashFunction;
        this.owner = true;
        this.handle = NativeGraph.create(TINY, size, 0, 1, numHash, 1, hashFunction.getK(), true, false, NativeGraph.defaultDevice(), 0L);
        BloomFilter.loadFile(handle, NativeGraph.CBF, bytes);
    }

    /** view of the counting filter of a graph's handle */
    public CountingBloomFilter(long graphHandle, long size, int numHash, HashFunction hashFunction) {
        this.handle = graphHandle;
        this.owner = false;
        this.size = size;
        this.numHash = numHash;
        this.hashFunction = hashFunction;
    }

    public void save(File desc, File bytes) throws IOException {
        try (FileWriter w = new FileWriter(desc, false)) {
            w.write("size:" + size + "\n" + "numhash:" + numHash + "\n" + "fpr:" + getFPR() + "\n");
        }
        BloomFilter.saveFile(handle, NativeGraph.CBF, bytes);
    }

    private static long[] one(long v) { return new long[]{v}; }

    @Override
    public void increment(String key) {
        final long[] hashVals = new long[numHash];
        hashFunction.getHashValues(key, numHash, hashVals);
        increment(hashVals);
    }

    public void increment(long hashVal) {
        NativeGraph.apply(handle, NativeGraph.OP_ADD_COUNT_ONLY, one(hashVal), 1);
    }

    public void increment(final long[] hashVals) { increment(hashVals[0]); }

    /** batched form: the conservative updates of n keys, applied in array order (order-exact, DESIGN.md s3) */
    public void incrementAll(final long[] baseHashes, int n) {
        NativeGraph.apply(handle, NativeGraph.OP_ADD_COUNT_ONLY, baseHashes, n);
    }

    public float incrementAndGet(final long[] hashVals) {
        float[] o = new float[1];
        NativeGraph.filterIncrementAndGet(handle, one(hashVals[0]), 1, o);
        return o[0];
    }

    @Override
    public float getCount(String key) {
        final long[] hashVals = new long[numHash];
        hashFunction.getHashValues(key, numHash, hashVals);
        return getCount(hashVals);
    }

    public float getCount(long hashVal) {
        float[] o = new float[1];
        NativeGraph.filterGetCount(handle, one(hashVal), 1, o);
        return o[0];
    }

    public float getCount(final long[] hashVals) { return getCount(hashVals[0]); }

    public void getCountAll(final long[] baseHashes, int n, float[] out) {
        NativeGraph.filterGetCount(handle, baseHashes, n, out);
    }

    @Override
    public float getFPR() {
        popcount = NativeGraph.popcount(handle, NativeGraph.CBF);     // non-zero counters (:254-263)
        return (float) Math.pow((double) popcount / (double) size, numHash);
    }

    public static long getExpectedSize(long expNumElements, float fpr, int numHash) {
        return NativeGraph.expectedSize(expNumElements, fpr, numHash);
    }

    public long getOptimalSize(float fpr) {
        return popcount > 0 ? getExpectedSize(popcount, fpr, numHash) : size;
    }

    public long getPopCount() { return popcount; }

    public int getNumHash() { return numHash; }

    public long getSize() { return size; }

    public void empty() { NativeGraph.clear(handle, 1 << NativeGraph.CBF); }

    public void destroy() {
        if (handle == 0) return;
        if (owner) NativeGraph.destroy(handle); else NativeGraph.destroyFilter(handle, NativeGraph.CBF);
        handle = 0;
    }

    public boolean equivalent(CountingBloomFilter bf) {
        return size == bf.size && numHash == bf.numHash
            && NativeGraph.popcount(handle, NativeGraph.CBF) == NativeGraph.popcount(bf.handle, NativeGraph.CBF)
            && NativeGraph.fold(handle, NativeGraph.CBF) == NativeGraph.fold(bf.handle, NativeGraph.CBF);
    }

    /** the plain Bloom filter of the counters with MiniFloat.toFloat(count) >= minCov, built on the device (:328-338) */
    public BloomFilter getBloomFilter(int minCov) {
        BloomFilter bf = new BloomFilter(size, numHash, hashFunction);
        NativeGraph.cbfToBloom(handle, (float) minCov, bf.handle, NativeGraph.DBGBF);
        return bf;
    }
}
