package rnabloom.bloom;

import java.io.BufferedReader;
import java.io.File;
import java.io.FileNotFoundException;
import java.io.FileReader;
import java.io.FileWriter;
import java.io.IOException;
import java.io.RandomAccessFile;
import java.nio.ByteBuffer;
import java.nio.channels.FileChannel;
import rnabloom.bloom.hash.HashFunction;
import rnabloom.graph.NativeGraph;

/**
 * Drop-in for the reference's rnabloom.bloom.BloomFilter (src/rnabloom/bloom/BloomFilter.java:40-258): same constructors and
 * public methods, the bit array lives in HBM behind librb_hip.so instead of an UnsafeBitBuffer.
 *
 * A filter is one bit filter of an rb_graph handle: either a handle of its own in which only that filter has a real size
 * (stand-alone use: screening filters), or a view of dbgbf / rpkbf / fpkbf of a BloomFilterDeBruijnGraph's handle
 * (getDbgbf() etc.).  Per-element methods call the batched natives with n = 1; bulk callers use the array forms.
 * Only hashVals[0] crosses the boundary: the library re-derives hashVals[1..] with NTM64 from the filter's k
 * (src/rnabloom/bloom/hash/NTHash.java:518-527), which is what every caller in the reference passes.
 */
public class BloomFilter implements BloomFilterInterface {
    protected long handle;            // rb_graph*
    protected final int which;        // NativeGraph.DBGBF / RPKBF / FPKBF
    protected final boolean owner;    // false: a view of a graph's filter
    protected int numHash;
    protected long size;
    protected HashFunction hashFunction;
    protected long popcount = -1;

    private static final long TINY = 64;   // size of the filters a stand-alone object does not use

    public BloomFilter(long size, int numHash, HashFunction hashFunction) {
        this.size = size;
        this.numHash = numHash;
        this.hashFunction = hashFunction;
        this.which = NativeGraph.DBGBF;
        this.owner = true;
        this.handle = NativeGraph.create(size, TINY, 0, numHash, 1, 1, hashFunction.getK(), true, false, NativeGraph.defaultDevice(), 0L);
    }

    public BloomFilter(File desc, File bits, HashFunction hashFunction) throws FileNotFoundException, IOException {
        this(desc, bits, hashFunction, true);
    }

    public BloomFilter(File desc, File bits, HashFunction hashFunction, boolean loadBits) throws FileNotFoundException, IOException {
        long[] sn = readDesc(desc);
        this.size = sn[0];
        this.numHash = (int) sn[1];
        this.hashFunction = hashFunction;
        this.which = NativeGraph.DBGBF;
        this.owner = true;
        this.handle = NativeGraph.create(size, TINY, 0, numHash, 1, 1, hashFunction.getK(), true, false, NativeGraph.defaultDevice(), 0L);
        if (loadBits) {
            loadFile(handle, which, bits);
        }
    }

    /** view of filter `which` of a graph's handle (BloomFilterDeBruijnGraph.getDbgbf / getRpkbf / getFpkbf) */
    public BloomFilter(long graphHandle, int which, long size, int numHash, HashFunction hashFunction) {
        this.handle = graphHandle;
        this.which = which;
        this.owner = false;
        this.size = size;
        this.numHash = numHash;
        this.hashFunction = hashFunction;
    }

    // ---- the .desc text and the raw bytes are the reference's own files (:64-124) ----
    /** {size, numhash} of a filter description file */
    public static long[] readDesc(File desc) throws IOException {
        long size = 0, numHash = 0;
        try (BufferedReader br = new BufferedReader(new FileReader(desc))) {
            for (String line = br.readLine(); line != null; line = br.readLine()) {
                String[] kv = line.split(":");
                if (kv[0].equals("size")) size = Long.parseLong(kv[1]);
                else if (kv[0].equals("numhash")) numHash = Integer.parseInt(kv[1]);
            }
        }
        return new long[]{size, numHash};
    }

    static void loadFile(long handle, int which, File bytes) throws IOException {
        long n = NativeGraph.filterSize(handle, which)[1];
        try (RandomAccessFile f = new RandomAccessFile(bytes, "r"); FileChannel ch = f.getChannel()) {
            if (n <= Integer.MAX_VALUE) {
                NativeGraph.importFilter(handle, which, ch.map(FileChannel.MapMode.READ_ONLY, 0, n), n);
            } else {                                    // direct buffers hold < 2 GiB: the JNI shim maps larger files itself
                NativeGraph.importFilterFromFile(handle, which, bytes.getPath(), n);
            }
        }
    }

    static void saveFile(long handle, int which, File bytes) throws IOException {
        long n = NativeGraph.filterSize(handle, which)[1];
        if (n <= Integer.MAX_VALUE) {
            try (RandomAccessFile f = new RandomAccessFile(bytes, "rw"); FileChannel ch = f.getChannel()) {
                f.setLength(n);
                NativeGraph.exportFilter(handle, which, ch.map(FileChannel.MapMode.READ_WRITE, 0, n), n);
            }
        } else {
            NativeGraph.exportFilterToFile(handle, which, bytes.getPath(), n);
        }
    }

    public void save(File desc, File bits) throws IOException {
        try (FileWriter w = new FileWriter(desc, false)) {
            w.write("size:" + size + "\n" + "numhash:" + numHash + "\n" + "fpr:" + getFPR() + "\n");
        }
        saveFile(handle, which, bits);
    }

    private static long[] one(long v) { return new long[]{v}; }

    @Override
    public void add(String key) {
        final long[] hashVals = new long[numHash];
        hashFunction.getHashValues(key, numHash, hashVals);
        add(hashVals);
    }

    public void add(final long[] hashVals) { add(hashVals[0]); }

    public void add(final long hashVal) {
        NativeGraph.apply(handle, opAdd(), one(hashVal), 1);
    }

    /** batched form: every base hash of the array (hashVals[0] of each key) */
    public void addAll(final long[] baseHashes, int n) {
        NativeGraph.apply(handle, opAdd(), baseHashes, n);
    }

    private int opAdd() {
        return which == NativeGraph.DBGBF ? NativeGraph.OP_ADD_DBG_ONLY
             : which == NativeGraph.RPKBF ? NativeGraph.OP_ADD_READ_PAIR : NativeGraph.OP_ADD_FRAG_PAIR;
    }

    public boolean lookupThenAdd(final long hashVal) {
        byte[] o = new byte[1];
        NativeGraph.filterLookupThenAdd(handle, which, one(hashVal), 1, o);
        return o[0] != 0;
    }

    public boolean lookupThenAdd(final long[] hashVals) { return lookupThenAdd(hashVals[0]); }

    /** batched form, in array order: out[i] != 0 iff every bit of element i was set before element i was added */
    public void lookupThenAddAll(final long[] baseHashes, int n, byte[] out) {
        NativeGraph.filterLookupThenAdd(handle, which, baseHashes, n, out);
    }

    /** the reference's lock-free variant (:157-162); on the device every add is an atomicOr */
    public void addCAS(final long[] hashVals) { add(hashVals); }

    @Override
    public boolean lookup(String key) {
        final long[] hashVals = new long[numHash];
        hashFunction.getHashValues(key, numHash, hashVals);
        return lookup(hashVals);
    }

    public boolean lookup(final long[] hashVals) { return lookup(hashVals[0]); }

    public boolean lookup(final long hashVal) {
        byte[] o = new byte[1];
        NativeGraph.filterLookup(handle, which, one(hashVal), 1, o);
        return o[0] != 0;
    }

    public void lookupAll(final long[] baseHashes, int n, byte[] out) {
        NativeGraph.filterLookup(handle, which, baseHashes, n, out);
    }

    @Override
    public float getFPR() {
        popcount = NativeGraph.popcount(handle, which);              // :185-194 remembers the popcount it used
        return (float) Math.pow((double) popcount / (double) size, numHash);
    }

    public static long getExpectedSize(long expNumElements, float fpr, int numHash) {
        return NativeGraph.expectedSize(expNumElements, fpr, numHash);
    }

    public long getPopCount() { return popcount; }                  // the value cached by the last getFPR(), as in the reference (:201-203)

    public long getOptimalSize(float fpr) {                          // :205-213
        return popcount > 0 ? getExpectedSize(popcount, fpr, numHash) : size;
    }

    public int getNumHash() { return numHash; }

    public long getSize() { return size; }

    public void empty() { NativeGraph.clear(handle, 1 << which); }

    public void destroy() {
        if (handle == 0) return;
        if (owner) NativeGraph.destroy(handle); else NativeGraph.destroyFilter(handle, which);
        handle = 0;
    }

    public boolean equivalent(BloomFilter bf) {                      // :248-257: same size, same numHash, same bits
        return size == bf.size && numHash == bf.numHash
            && NativeGraph.popcount(handle, which) == NativeGraph.popcount(bf.handle, bf.which)
            && NativeGraph.fold(handle, which) == NativeGraph.fold(bf.handle, bf.which);
    }
}
