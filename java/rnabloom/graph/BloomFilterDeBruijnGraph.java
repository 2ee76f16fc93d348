package rnabloom.graph;

import java.io.BufferedReader;
import java.io.File;
import java.io.FileNotFoundException;
import java.io.FileReader;
import java.io.FileWriter;
import java.io.IOException;
import java.util.ArrayDeque;
import java.util.ArrayList;
import java.util.Collection;
import java.util.HashMap;
import java.util.Iterator;
import rnabloom.bloom.BloomFilter;
import rnabloom.bloom.CountingBloomFilter;
import rnabloom.bloom.hash.*;
import static rnabloom.util.SeqUtils.*;

/**
 * Drop-in for the reference's rnabloom.graph.BloomFilterDeBruijnGraph (src/rnabloom/graph/BloomFilterDeBruijnGraph.java):
 * the same constructors, the same public methods, the same files on disk — with the four filters resident in HBM behind ONE
 * rb_graph handle (librb_hip.so through NativeGraph) instead of four Unsafe*Buffers.
 *
 * What changes for callers: nothing in the signatures.  Per-element methods (add(long[]), contains(long[]), getCount(long[]),
 * ...) call the batched natives with n = 1, so code that is not ported yet keeps working; the hot loops should move to the
 * batched forms at the end of this class (applyAll / containsAll / getCountAll) and stage 1 to
 * NativeGraph.addReads / addFastq (see NativeFastqToGraphWorker in this directory).  Only hashVals[0] crosses the boundary: the
 * library re-derives hashVals[1..] with NTM64 from the graph's k (src/rnabloom/bloom/hash/NTHash.java:518-527) — which is
 * what every hashVals array in the reference was made with.
 *
 * Kmer, CanonicalKmer, HashFunction and the iterators stay the reference's own classes: they call back into
 * getCount(long) / contains(long[]) of this class exactly as before.  (HashFunction needs one accessor the reference does not
 * have: `public int getK() { return k; }`.)
 */
public class BloomFilterDeBruijnGraph {
    private long handle;                       // rb_graph*: dbgbf, cbf, rpkbf, fpkbf on the device
    private BloomFilter dbgbf;                 // views of the handle's filters (getDbgbf() etc.)
    private CountingBloomFilter cbf;
    private BloomFilter fpkbf = null;
    private BloomFilter rpkbf = null;

    private int dbgbfNumHash;
    private int cbfNumHash;
    private int dbgbfCbfMaxNumHash;
    private final HashFunction hashFunction;
    private int k;
    private int kMinus1;
    private boolean stranded;
    private int fragmentPairedKmersDistance = -1;
    private int pkbfNumHash;
    private int readPairedKmersDistance = -1;

    private static final String EXT_DESC = ".desc", EXT_DBGBF = ".dbgbf", EXT_CBF = ".cbf", EXT_FPKBF = ".fpkbf", EXT_RPKBF = ".rpkbf";

    public BloomFilterDeBruijnGraph(long dbgbfNumBits,
                                    long cbfNumBytes,
                                    long pkbfNumBits,
                                    int dbgbfNumHash,
                                    int cbfNumHash,
                                    int pkbfNumHash,
                                    int k,
                                    boolean stranded,
                                    boolean useReadPairedKmers) {
        this.k = k;
        this.kMinus1 = k - 1;
        this.stranded = stranded;
        this.hashFunction = stranded ? new HashFunction(k) : new CanonicalHashFunction(k);
        this.dbgbfNumHash = dbgbfNumHash;
        this.cbfNumHash = cbfNumHash;
        this.pkbfNumHash = pkbfNumHash;
        this.dbgbfCbfMaxNumHash = Math.max(dbgbfNumHash, cbfNumHash);
        this.handle = NativeGraph.create(dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k, stranded,
                                         useReadPairedKmers, NativeGraph.defaultDevice(), 0L);
        this.dbgbf = new BloomFilter(handle, NativeGraph.DBGBF, dbgbfNumBits, dbgbfNumHash, hashFunction);
        this.cbf = new CountingBloomFilter(handle, cbfNumBytes, cbfNumHash, hashFunction);
        if (useReadPairedKmers) {
            this.rpkbf = new BloomFilter(handle, NativeGraph.RPKBF, pkbfNumBits, pkbfNumHash, hashFunction);
        }
    }

    private static HashMap<String, String> readLabels(File f) throws IOException {
        HashMap<String, String> m = new HashMap<>();
        try (BufferedReader br = new BufferedReader(new FileReader(f))) {
            for (String line = br.readLine(); line != null; line = br.readLine()) {
                String[] kv = line.split(":");
                if (kv.length >= 2) m.put(kv[0], kv[1]);
            }
        }
        return m;
    }

    public void updateFragmentKmerDistance(File graphFile) throws FileNotFoundException, IOException {
        String v = readLabels(graphFile).get("fragmentPairedKmersDistance");
        if (v != null) {
            fragmentPairedKmersDistance = Integer.parseInt(v);
            NativeGraph.setFragPairedKmerDistance(handle, fragmentPairedKmersDistance);
        }
    }

    public BloomFilterDeBruijnGraph(File graphFile, boolean loadDbgBits) throws FileNotFoundException, IOException {
        HashMap<String, String> d = readLabels(graphFile);
        if (d.containsKey("dbgbfCbfMaxNumHash")) dbgbfCbfMaxNumHash = Integer.parseInt(d.get("dbgbfCbfMaxNumHash"));
        if (d.containsKey("k")) { k = Integer.parseInt(d.get("k")); kMinus1 = k - 1; }
        if (d.containsKey("stranded")) stranded = Boolean.parseBoolean(d.get("stranded"));
        if (d.containsKey("fragmentPairedKmersDistance")) fragmentPairedKmersDistance = Integer.parseInt(d.get("fragmentPairedKmersDistance"));
        if (d.containsKey("readPairedKmersDistance")) readPairedKmersDistance = Integer.parseInt(d.get("readPairedKmersDistance"));
        this.hashFunction = stranded ? new HashFunction(k) : new CanonicalHashFunction(k);

        String base = graphFile.getPath();
        long[] dDesc = BloomFilter.readDesc(new File(base + EXT_DBGBF + EXT_DESC));
        long[] cDesc = BloomFilter.readDesc(new File(base + EXT_CBF + EXT_DESC));
        File rBits = new File(base + EXT_RPKBF), rDesc = new File(base + EXT_RPKBF + EXT_DESC);
        File fBits = new File(base + EXT_FPKBF), fDesc = new File(base + EXT_FPKBF + EXT_DESC);
        boolean hasR = rBits.isFile() && rDesc.isFile(), hasF = fBits.isFile() && fDesc.isFile();
        long[] rD = hasR ? BloomFilter.readDesc(rDesc) : new long[]{0, 1};
        dbgbfNumHash = (int) dDesc[1];
        cbfNumHash = (int) cDesc[1];
        handle = NativeGraph.create(dDesc[0], cDesc[0], rD[0], dbgbfNumHash, cbfNumHash, (int) rD[1], k, stranded, hasR, NativeGraph.defaultDevice(), 0L);
        dbgbf = new BloomFilter(handle, NativeGraph.DBGBF, dDesc[0], dbgbfNumHash, hashFunction);
        cbf = new CountingBloomFilter(handle, cDesc[0], cbfNumHash, hashFunction);
        if (loadDbgBits) importFile(NativeGraph.DBGBF, new File(base + EXT_DBGBF));
        importFile(NativeGraph.CBF, new File(base + EXT_CBF));
        if (hasF) {
            long[] fD = BloomFilter.readDesc(fDesc);
            pkbfNumHash = (int) fD[1];
            NativeGraph.initFragmentPairs(handle, fD[0], pkbfNumHash);
            fpkbf = new BloomFilter(handle, NativeGraph.FPKBF, fD[0], pkbfNumHash, hashFunction);
            importFile(NativeGraph.FPKBF, fBits);
        }
        if (hasR) {
            pkbfNumHash = (int) rD[1];
            rpkbf = new BloomFilter(handle, NativeGraph.RPKBF, rD[0], pkbfNumHash, hashFunction);
            importFile(NativeGraph.RPKBF, rBits);
        }
        if (readPairedKmersDistance >= 0) NativeGraph.setReadPairedKmerDistance(handle, readPairedKmersDistance);
        if (fragmentPairedKmersDistance >= 0) NativeGraph.setFragPairedKmerDistance(handle, fragmentPairedKmersDistance);
    }

    private void importFile(int which, File bytes) throws IOException {
        long n = NativeGraph.filterSize(handle, which)[1];
        NativeGraph.importFilterFromFile(handle, which, bytes.getPath(), n);
    }

    /** the rb_graph handle, for the batched natives (NativeGraph.addReads, addFastq, batchCounts, walk, ...) */
    public long getHandle() { return handle; }

    public HashFunction getHashFunction() { return this.hashFunction; }

    public int getDbgbfNumHash() { return dbgbfNumHash; }

    public int getCbfNumHash() { return cbfNumHash; }

    public int getPkbfNumHash() { return pkbfNumHash; }

    public int getMaxNumHash() { return dbgbfCbfMaxNumHash; }

    public void destroy() {
        if (handle != 0) { NativeGraph.destroy(handle); handle = 0; }
        dbgbf = null; cbf = null; rpkbf = null; fpkbf = null;
    }

    public void clearAllBf() { NativeGraph.clear(handle, 15); }

    public void clearDbgbf() { NativeGraph.clear(handle, 1 << NativeGraph.DBGBF); }

    public void clearCbf() { NativeGraph.clear(handle, 1 << NativeGraph.CBF); }

    public void clearFpkbf() { if (fpkbf != null) NativeGraph.clear(handle, 1 << NativeGraph.FPKBF); }

    public void clearRpkbf() { if (rpkbf != null) NativeGraph.clear(handle, 1 << NativeGraph.RPKBF); }

    public void destroyDbgbf() { if (dbgbf != null) { NativeGraph.destroyFilter(handle, NativeGraph.DBGBF); dbgbf = null; } }

    public void destroyCbf() { if (cbf != null) { NativeGraph.destroyFilter(handle, NativeGraph.CBF); cbf = null; } }

    public void destroyFpkbf() { if (fpkbf != null) { NativeGraph.destroyFilter(handle, NativeGraph.FPKBF); fpkbf = null; } }

    public void destroyRpkbf() { if (rpkbf != null) { NativeGraph.destroyFilter(handle, NativeGraph.RPKBF); rpkbf = null; } }

    public BloomFilter getDbgbf() { return dbgbf; }

    public CountingBloomFilter getCbf() { return cbf; }

    public BloomFilter getFpkbf() { return fpkbf; }

    public BloomFilter getRpkbf() { return rpkbf; }

    public boolean isStranded() { return stranded; }

    public void saveDesc(File graphFile) throws IOException {
        try (FileWriter w = new FileWriter(graphFile)) {
            w.write("dbgbfCbfMaxNumHash:" + dbgbfCbfMaxNumHash + "\n" + "stranded:" + stranded + "\n" + "k:" + k + "\n"
                    + "readPairedKmersDistance:" + readPairedKmersDistance + "\n"
                    + "fragmentPairedKmersDistance:" + fragmentPairedKmersDistance + "\n");
        }
    }

    public void save(File graphFile) throws IOException {
        saveDesc(graphFile);
        String base = graphFile.getPath();
        dbgbf.save(new File(base + EXT_DBGBF + EXT_DESC), new File(base + EXT_DBGBF));
        cbf.save(new File(base + EXT_CBF + EXT_DESC), new File(base + EXT_CBF));
        if (rpkbf != null) rpkbf.save(new File(base + EXT_RPKBF + EXT_DESC), new File(base + EXT_RPKBF));
    }

    public void savePkbf(File graphFile) throws IOException {
        saveDesc(graphFile);                       // the k-mer pair distance may have changed
        String base = graphFile.getPath();
        fpkbf.save(new File(base + EXT_FPKBF + EXT_DESC), new File(base + EXT_FPKBF));
    }

    public void restorePkbf(File graphFile) throws IOException {
        String base = graphFile.getPath();
        long[] fD = BloomFilter.readDesc(new File(base + EXT_FPKBF + EXT_DESC));
        if (fpkbf != null) NativeGraph.destroyFilter(handle, NativeGraph.FPKBF);
        pkbfNumHash = (int) fD[1];
        NativeGraph.initFragmentPairs(handle, fD[0], pkbfNumHash);
        fpkbf = new BloomFilter(handle, NativeGraph.FPKBF, fD[0], pkbfNumHash, hashFunction);
        importFile(NativeGraph.FPKBF, new File(base + EXT_FPKBF));
    }

    public void initializePairKmersBloomFilter(long pkbfNumBits, int pkbfNumHash) {
        if (fpkbf == null) {
            this.pkbfNumHash = pkbfNumHash;
            NativeGraph.initFragmentPairs(handle, pkbfNumBits, pkbfNumHash);
            fpkbf = new BloomFilter(handle, NativeGraph.FPKBF, pkbfNumBits, pkbfNumHash, hashFunction);
        } else {
            fpkbf.empty();
        }
    }

    public void setFragPairedKmerDistance(int d) { fragmentPairedKmersDistance = d; NativeGraph.setFragPairedKmerDistance(handle, d); }

    public int getFragPairedKmerDistance() { return fragmentPairedKmersDistance; }

    public void setReadPairedKmerDistance(int d) { readPairedKmersDistance = d; NativeGraph.setReadPairedKmerDistance(handle, d); }

    public int getReadPairedKmerDistance() { return readPairedKmersDistance; }

    public int getK() { return k; }

    /** the handle keeps the k it was created with (it salts NTM64): as in the reference, iterators made before a setK keep theirs */
    public void setK(int k) {
        this.k = k;
        this.kMinus1 = k - 1;
        this.hashFunction.setK(k);
    }

    public int getKMinus1() { return kMinus1; }

    public boolean isLowComplexity(Kmer kmer) { return isLowComplexity2(kmer.bytes); }

    public boolean isRepeatKmer(Kmer kmer) { return isRepeat(kmer.bytes); }

    // ---- per-element mutators: one native call with n = 1 (src/.../BloomFilterDeBruijnGraph.java:399-461) ----
    private static long[] one(long v) { return new long[]{v}; }

    private long[] hashesOf(String kmer) {
        final long[] hashVals = new long[dbgbfCbfMaxNumHash];
        hashFunction.getHashValues(kmer, dbgbfCbfMaxNumHash, hashVals);
        return hashVals;
    }

    public void add(String kmer) { add(hashesOf(kmer)); }

    public void add(final long[] hashVals) { NativeGraph.apply(handle, NativeGraph.OP_ADD, one(hashVals[0]), 1); }

    public void addIfAbsent(final long[] hashVals) { NativeGraph.apply(handle, NativeGraph.OP_ADD_IF_ABSENT, one(hashVals[0]), 1); }

    public void addCountIfPresent(final long[] hashVals) { NativeGraph.apply(handle, NativeGraph.OP_ADD_COUNT_IF_PRESENT, one(hashVals[0]), 1); }

    public void addDbgOnly(final long hashVal) { NativeGraph.apply(handle, NativeGraph.OP_ADD_DBG_ONLY, one(hashVal), 1); }

    public void addDbgOnly(final long[] hashVals) { addDbgOnly(hashVals[0]); }

    public void addCountOnly(final long[] hashVals) { NativeGraph.apply(handle, NativeGraph.OP_ADD_COUNT_ONLY, one(hashVals[0]), 1); }

    public void addReadSingleKmerPair(long[] pairingHashVals) { NativeGraph.apply(handle, NativeGraph.OP_ADD_READ_PAIR, one(pairingHashVals[0]), 1); }

    public void addFragmentSingleKmerPair(long[] pairingHashVals) { NativeGraph.apply(handle, NativeGraph.OP_ADD_FRAG_PAIR, one(pairingHashVals[0]), 1); }

    /** hashVals[0] of the pair (kmers[i], kmers[i + d]) for i < kmers.size() - d: what the four methods below hand to a pair filter */
    private long[] pairHashes(ArrayList<Kmer> kmers, int d) {
        final int n = kmers.size() - d;
        if (n <= 0) return new long[0];
        long[] h = new long[n];
        for (int i = 0; i < n; ++i) h[i] = kmers.get(i).getKmerPairHashValue(kmers.get(i + d));
        return h;
    }

    public void addFragmentPairKmers(ArrayList<Kmer> kmers) {
        long[] h = pairHashes(kmers, fragmentPairedKmersDistance);
        if (h.length > 0) NativeGraph.apply(handle, NativeGraph.OP_ADD_FRAG_PAIR, h, h.length);
    }

    public void addReadPairedKmers(ArrayList<Kmer> kmers) {
        long[] h = pairHashes(kmers, readPairedKmersDistance);
        if (h.length > 0) NativeGraph.apply(handle, NativeGraph.OP_ADD_READ_PAIR, h, h.length);
    }

    public boolean containsAllPairedKmers(ArrayList<Kmer> kmers) {
        long[] h = pairHashes(kmers, fragmentPairedKmersDistance);
        if (h.length == 0) return false;
        byte[] o = new byte[h.length];
        NativeGraph.filterLookup(handle, NativeGraph.FPKBF, h, h.length, o);
        for (byte b : o) if (b == 0) return false;
        return true;
    }

    /** one lookupThenAdd over the sequence's pairs IN ORDER (a later pair sees the bits of an earlier one), ANDed — :513-524 */
    public boolean lookupAndAddAllPairedKmers(ArrayList<Kmer> kmers) {
        long[] h = pairHashes(kmers, fragmentPairedKmersDistance);
        if (h.length == 0) return true;
        byte[] o = new byte[h.length];
        NativeGraph.filterLookupThenAdd(handle, NativeGraph.FPKBF, h, h.length, o);
        boolean all = true;
        for (byte b : o) all &= b != 0;
        return all;
    }

    public boolean lookupFragmentKmerPair(Kmer left, Kmer right) { return fpkbf.lookup(left.getKmerPairHashValue(right)); }

    public boolean lookupReadKmerPair(Kmer left, Kmer right) { return rpkbf.lookup(left.getKmerPairHashValue(right)); }

    // ---- queries ----
    public boolean contains(String kmer) { return contains(hashesOf(kmer)); }

    public boolean contains(final long[] hashVals) {
        byte[] o = new byte[1];
        NativeGraph.contains(handle, one(hashVals[0]), 1, o);
        return o[0] != 0;
    }

    public void increment(String kmer) { cbf.increment(kmer); }

    public float getCount(String kmer) { return getCount(hashesOf(kmer)); }

    /** dbgbf.lookup ? cbf.getCount + 1 : 0 (:552-570), evaluated on the device */
    public float getCount(final long hashVal) {
        float[] o = new float[1];
        NativeGraph.getCount(handle, one(hashVal), 1, o);
        return o[0];
    }

    public float getCount(final long[] hashVals) { return getCount(hashVals[0]); }

    public float getDbgbfFPR() { return NativeGraph.fpr(handle, NativeGraph.DBGBF); }

    public float getCbfFPR() { return NativeGraph.fpr(handle, NativeGraph.CBF); }

    public float getRpkbfFPR() { return NativeGraph.fpr(handle, NativeGraph.RPKBF); }

    public float getPkbfFPR() { return NativeGraph.fpr(handle, NativeGraph.FPKBF); }

    public float getFPR() { return getDbgbfFPR() * getCbfFPR(); }

    public Kmer getKmer(String kmer) { return hashFunction.getKmer(kmer, dbgbfCbfMaxNumHash, this); }

    public String getPrefix(String kmer) { return kmer.substring(0, kMinus1); }

    public String getSuffix(String kmer) { return kmer.substring(1, k); }

    public CharSequence getPrefixCharSeq(String kmer) { return kmer.subSequence(0, kMinus1); }

    public CharSequence getSuffixCharSeq(String kmer) { return kmer.subSequence(1, k); }

    /** the k-mers that differ from `kmer` in its first base and are in the graph */
    public ArrayDeque<String> getLeftVariants(String kmer) {
        ArrayDeque<String> result = new ArrayDeque<>(4);
        final String suffix = getSuffix(kmer);
        for (char c : getAltNucleotides(kmer.charAt(0))) {
            String v = c + suffix;
            if (contains(v)) result.add(v);
        }
        return result;
    }

    /** the same for the last base */
    public ArrayDeque<String> getRightVariants(String kmer) {
        ArrayDeque<String> result = new ArrayDeque<>(4);
        final String prefix = getPrefix(kmer);
        for (char c : getAltNucleotides(kmer.charAt(kMinus1))) {
            String v = prefix + c;
            if (contains(v)) result.add(v);
        }
        return result;
    }

    public float[] getCounts(String[] kmers) {
        long[] h = new long[kmers.length];
        for (int i = 0; i < kmers.length; ++i) h[i] = hashesOf(kmers[i])[0];
        float[] counts = new float[kmers.length];
        if (kmers.length > 0) NativeGraph.getCount(handle, h, h.length, counts);
        return counts;
    }

    /** every k-mer of seq is in dbgbf (:1181-1194): one batched lookup of the sequence's hashes */
    public boolean isValidSeq(String seq) {
        NTHashIterator itr = getHashIterator();
        itr.start(seq);
        long[] hVals = itr.hVals;
        ArrayList<Long> hs = new ArrayList<>();
        while (itr.hasNext()) {
            itr.next();
            hs.add(hVals[0]);
        }
        if (hs.isEmpty()) return true;
        long[] h = new long[hs.size()];
        for (int i = 0; i < h.length; ++i) h[i] = hs.get(i);
        byte[] o = new byte[h.length];
        NativeGraph.contains(handle, h, h.length, o);
        for (byte b : o) if (b == 0) return false;
        return true;
    }

    public NTHashIterator getHashIterator() { return hashFunction.getHashIterator(this.dbgbfCbfMaxNumHash); }

    public NTHashIterator getHashIterator(int numHash) { return hashFunction.getHashIterator(numHash); }

    public NTHashIterator getHashIterator(int numHash, int k) { return hashFunction.getHashIterator(numHash, k); }

    public NTHashIterator getReverseComplementHashIterator(int numHash) { return hashFunction.getReverseComplementHashIterator(numHash); }

    public NTHashIterator getReverseComplementHashIterator(int numHash, int k) { return hashFunction.getReverseComplementHashIterator(numHash, k); }

    public PairedNTHashIterator getPairedHashIterator(int d) { return hashFunction.getPairedHashIterator(this.pkbfNumHash, d); }

    public PairedNTHashIterator getReverseComplementPairedHashIterator(int d) { return hashFunction.getReverseComplementPairedHashIterator(this.pkbfNumHash, d); }

    public ArrayList<Kmer> getKmers(String seq) { return hashFunction.getKmers(seq, this.dbgbfCbfMaxNumHash, this); }

    public ArrayList<Kmer> getKmers(String seq, float minCoverage) { return hashFunction.getKmers(seq, this.dbgbfCbfMaxNumHash, this, minCoverage); }

    public ArrayList<Kmer> getKmers(String seq, int start, int end) { return hashFunction.getKmers(seq, start, end, this.dbgbfCbfMaxNumHash, this); }

    // ---- k-mer lists back to sequences: the first k-mer whole, then one base of every following k-mer ----
    private void appendWhole(StringBuilder sb, Kmer first) { for (byte b : first.bytes) sb.append((char) b); }

    public String assemble(ArrayDeque<Kmer> kmers) { return assemble((Collection<Kmer>) kmers); }

    public String assemble(ArrayList<Kmer> kmers, int start, int end) {
        StringBuilder sb = new StringBuilder(end - start + kMinus1);
        appendWhole(sb, kmers.get(start));
        for (int i = start + 1; i < end; ++i) sb.append((char) kmers.get(i).bytes[kMinus1]);
        return sb.toString();
    }

    public byte[] assembleBytes(ArrayList<Kmer> kmers, int start, int end) {
        byte[] out = new byte[end - start + kMinus1];
        System.arraycopy(kmers.get(start).bytes, 0, out, 0, k);
        for (int i = start + 1, p = k; i < end; ++i) out[p++] = kmers.get(i).bytes[kMinus1];
        return out;
    }

    public byte[] assembleReverseComplementBytes(ArrayList<Kmer> kmers, int start, int end) {
        byte[] fwd = assembleBytes(kmers, start, end);
        byte[] out = new byte[fwd.length];
        for (int i = 0; i < fwd.length; ++i) out[fwd.length - 1 - i] = complement(fwd[i]);
        return out;
    }

    public String assembleReverseOrder(ArrayDeque<Kmer> kmers) {
        StringBuilder sb = new StringBuilder(kmers.size() + kMinus1);
        Iterator<Kmer> itr = kmers.descendingIterator();
        if (itr.hasNext()) {
            appendWhole(sb, itr.next());
            while (itr.hasNext()) sb.append((char) itr.next().bytes[kMinus1]);
        }
        return sb.toString();
    }

    public String assembleReverseOrder(ArrayList<Kmer> kmers) {
        final int n = kmers.size();
        StringBuilder sb = new StringBuilder(n + kMinus1);
        appendWhole(sb, kmers.get(n - 1));
        for (int i = n - 2; i >= 0; --i) sb.append((char) kmers.get(i).bytes[kMinus1]);
        return sb.toString();
    }

    public String assemble(Collection<Kmer> kmers) {
        StringBuilder sb = new StringBuilder(kmers.size() + kMinus1);
        Iterator<Kmer> itr = kmers.iterator();
        if (itr.hasNext()) {
            appendWhole(sb, itr.next());
            while (itr.hasNext()) sb.append((char) itr.next().bytes[kMinus1]);
        }
        return sb.toString();
    }

    public String assembleFirstBase(Collection<Kmer> kmers) {
        StringBuilder sb = new StringBuilder(kmers.size());
        for (Kmer kmer : kmers) sb.append((char) kmer.bytes[0]);
        return sb.toString();
    }

    public String assembleLastBase(Collection<Kmer> kmers) {
        StringBuilder sb = new StringBuilder(kmers.size());
        for (Kmer kmer : kmers) sb.append((char) kmer.bytes[kMinus1]);
        return sb.toString();
    }

    // ---- batched forms (not in the reference): what ported hot loops call instead of n per-element calls ----
    /** op = NativeGraph.OP_*: the n base hashes are applied in array order, exactly as n per-element calls would be */
    public void applyAll(int op, long[] baseHashes, int n) { NativeGraph.apply(handle, op, baseHashes, n); }

    public void containsAll(long[] baseHashes, int n, byte[] out) { NativeGraph.contains(handle, baseHashes, n, out); }

    public void getCountAll(long[] baseHashes, int n, float[] out) { NativeGraph.getCount(handle, baseHashes, n, out); }
}
