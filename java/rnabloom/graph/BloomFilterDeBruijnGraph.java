package rnabloom.graph;

import java.io.*;
import java.nio.ByteBuffer;
import java.util.*;
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
 * Kmer and CanonicalKmer have drop-ins beside this file (one NativeGraph.neighbors call per k-mer neighbourhood); HashFunction and the
 * iterators stay the reference's own classes.  (HashFunction needs one accessor the reference does not have:
 * `public int getK() { return k; }`.)
 */
public class BloomFilterDeBruijnGraph {
    private long handle;                                   // rb_graph*: dbgbf, cbf, rpkbf, fpkbf live on the device behind it
    private BloomFilter dbgbf, rpkbf, fpkbf;               // thin views of the handle's filters (getDbgbf() etc.); null = absent
    private CountingBloomFilter cbf;
    private final HashFunction hashFunction;
    private int k, kMinus1;
    private boolean stranded;
    private int dbgbfNumHash, cbfNumHash, pkbfNumHash, dbgbfCbfMaxNumHash;
    private int readPairedKmersDistance = -1, fragmentPairedKmersDistance = -1;

    private static final String EXT_DESC = ".desc", EXT_DBGBF = ".dbgbf", EXT_CBF = ".cbf", EXT_FPKBF = ".fpkbf", EXT_RPKBF = ".rpkbf";

    public BloomFilterDeBruijnGraph(long dbgbfNumBits, long cbfNumBytes, long pkbfNumBits, int dbgbfNumHash, int cbfNumHash, int pkbfNumHash,
                                    int k, boolean stranded, boolean useReadPairedKmers) {
        this.k = k; this.kMinus1 = k - 1; this.stranded = stranded;
        this.dbgbfNumHash = dbgbfNumHash; this.cbfNumHash = cbfNumHash; this.pkbfNumHash = pkbfNumHash;
        this.dbgbfCbfMaxNumHash = dbgbfNumHash > cbfNumHash ? dbgbfNumHash : cbfNumHash;
        this.hashFunction = stranded ? new HashFunction(k) : new CanonicalHashFunction(k);
        this.handle = NativeGraph.create(dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k, stranded,
                                         useReadPairedKmers, NativeGraph.defaultDevice(), 0L);
        this.dbgbf = new BloomFilter(handle, NativeGraph.DBGBF, dbgbfNumBits, dbgbfNumHash, hashFunction);
        this.cbf = new CountingBloomFilter(handle, cbfNumBytes, cbfNumHash, hashFunction);
        this.rpkbf = useReadPairedKmers ? new BloomFilter(handle, NativeGraph.RPKBF, pkbfNumBits, pkbfNumHash, hashFunction) : null;
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
        final long[] h = new long[dbgbfCbfMaxNumHash];
        hashFunction.getHashValues(kmer, h.length, h);
        return h;
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

    /**
     * The k-mers that differ from `kmer` in one END base and are in the graph (:1056-1068, :1113-1124): the alternatives are hashed on
     * the host and looked up with ONE batched native call instead of one `contains` per alternative.
     */
    private ArrayDeque<String> endVariants(String kmer, boolean firstBase) {
        final char[] alts = getAltNucleotides(kmer.charAt(firstBase ? 0 : kMinus1));
        final String[] cand = new String[alts.length];
        final long[] base = new long[alts.length];
        final StringBuilder sb = new StringBuilder(kmer);
        for (int i = 0; i < alts.length; ++i) {
            sb.setCharAt(firstBase ? 0 : kMinus1, alts[i]);
            cand[i] = sb.toString();
            base[i] = hashesOf(cand[i])[0];
        }
        final byte[] present = new byte[alts.length];
        if (alts.length > 0) NativeGraph.contains(handle, base, alts.length, present);
        final ArrayDeque<String> found = new ArrayDeque<>(4);
        for (int i = 0; i < alts.length; ++i) if (present[i] != 0) found.add(cand[i]);
        return found;
    }

    public ArrayDeque<String> getLeftVariants(String kmer) { return endVariants(kmer, true); }

    public ArrayDeque<String> getRightVariants(String kmer) { return endVariants(kmer, false); }

    public float[] getCounts(String[] kmers) {
        long[] h = new long[kmers.length];
        for (int i = 0; i < kmers.length; ++i) h[i] = hashesOf(kmers[i])[0];
        float[] counts = new float[kmers.length];
        if (kmers.length > 0) NativeGraph.getCount(handle, h, h.length, counts);
        return counts;
    }

    /** every k-mer of seq is in dbgbf (:1181-1194): the iterator's base hashes are collected and looked up in one batched call */
    public boolean isValidSeq(String seq) {
        final int windows = seq.length() - k + 1;
        if (windows <= 0) return true;
        final long[] base = new long[windows];
        final NTHashIterator it = getHashIterator();
        it.start(seq);
        int n = 0;
        for (; it.hasNext() && n < windows; ++n) {
            it.next();
            base[n] = it.hVals[0];
        }
        if (n == 0) return true;
        final byte[] present = new byte[n];
        NativeGraph.contains(handle, base, n, present);
        for (int i = 0; i < n; ++i) if (present[i] == 0) return false;
        return true;
    }

    public NTHashIterator getHashIterator() { return hashFunction.getHashIterator(this.dbgbfCbfMaxNumHash); }

    public NTHashIterator getHashIterator(int numHash) { return hashFunction.getHashIterator(numHash); }

    public NTHashIterator getHashIterator(int numHash, int k) { return hashFunction.getHashIterator(numHash, k); }

    public NTHashIterator getReverseComplementHashIterator(int numHash) { return hashFunction.getReverseComplementHashIterator(numHash); }

    public NTHashIterator getReverseComplementHashIterator(int numHash, int k) { return hashFunction.getReverseComplementHashIterator(numHash, k); }

    public PairedNTHashIterator getPairedHashIterator(int d) { return hashFunction.getPairedHashIterator(this.pkbfNumHash, d); }

    public PairedNTHashIterator getReverseComplementPairedHashIterator(int d) { return hashFunction.getReverseComplementPairedHashIterator(this.pkbfNumHash, d); }

    /**
     * getKmers (:1224-1234 -> {Canonical,}HashFunction.getKmers, src/rnabloom/bloom/hash/CanonicalHashFunction.java:46-170): the
     * reference hashes window by window and asks graph.getCount per k-mer — here ONE NativeGraph.getKmers call returns forward hash,
     * reverse hash and count of every window of seq[start, end) (count 0 for a window with a character outside ACGTU, :73-78), and the
     * Kmer objects are built from the arrays.  Kmer / CanonicalKmer are the reference's own classes.
     */
    private static final class WindowProfile {
        final byte[] bytes; final long[] f, r; final float[] count; final int n;
        WindowProfile(byte[] bytes, long[] f, long[] r, float[] count, int n) { this.bytes = bytes; this.f = f; this.r = r; this.count = count; this.n = n; }
    }

    private WindowProfile profile(String seq, int start, int end) {
        final int len = end - start, n = len - k + 1;
        final byte[] ascii = new byte[Math.max(len, 0)];
        for (int i = 0; i < len; ++i) ascii[i] = (byte) seq.charAt(start + i);
        if (n <= 0) return new WindowProfile(ascii, new long[0], new long[0], new float[0], 0);
        final ByteBuffer text = ByteBuffer.allocateDirect(len);
        text.put(ascii);
        final long[] f = new long[n], r = new long[n], koff = new long[2];
        final float[] count = new float[n];
        NativeGraph.getKmers(handle, text, new long[]{0L, (long) len}, 1, koff, f, r, count);
        return new WindowProfile(ascii, f, r, count, n);
    }

    private Kmer kmerAt(WindowProfile w, int i) {
        final byte[] b = Arrays.copyOfRange(w.bytes, i, i + k);
        return stranded ? new Kmer(b, w.count[i], w.f[i]) : new CanonicalKmer(b, w.count[i], w.f[i], w.r[i]);
    }

    public ArrayList<Kmer> getKmers(String seq) { return getKmers(seq, 0, seq.length()); }

    public ArrayList<Kmer> getKmers(String seq, int start, int end) {
        final ArrayList<Kmer> out = new ArrayList<>();
        if (seq.length() < k) return out;
        final WindowProfile w = profile(seq, start, end);
        out.ensureCapacity(w.n);
        for (int i = 0; i < w.n; ++i) out.add(kmerAt(w, i));
        return out;
    }

    /**
     * "The longest stretch of consecutive k-mers with count >= minCoverage" as the reference computes it (:81-134), quirks included: a
     * stretch is ranked (longer wins, then the larger minimum count, then the earlier) only when a k-mer below the threshold ENDS it, and the
     * FIRST stretch is never ranked — it is what `longestSegment` points to from the start, so it is returned unless a later, ended stretch
     * takes over, and the first one that ends always does (it is compared with length 0).  A last stretch that runs to the end of the
     * sequence is returned only if it is the first.  (oracle/rbo.py::get_kmers_min_coverage is the same statement-by-statement.)
     */
    public ArrayList<Kmer> getKmers(String seq, float minCoverage) {
        final ArrayList<Kmer> out = new ArrayList<>();
        if (seq.length() < k) return out;
        final WindowProfile w = profile(seq, 0, seq.length());
        int keepAt = 0, keepLen = 0, stretches = 0, rankedLen = 0;
        float rankedMin = Float.MAX_VALUE;
        for (int i = 0; i < w.n; ) {
            if (!(w.count[i] >= minCoverage)) { ++i; continue; }
            int j = i;
            float low = Float.MAX_VALUE;
            for (; j < w.n && w.count[j] >= minCoverage; ++j) low = Math.min(low, w.count[j]);
            final int len = j - i;
            if (stretches++ == 0) { keepAt = i; keepLen = len; }
            else if (j < w.n && (len > rankedLen || (len == rankedLen && low > rankedMin))) { keepAt = i; keepLen = len; rankedLen = len; rankedMin = low; }
            i = j;
        }
        out.ensureCapacity(keepLen);
        for (int i = keepAt; i < keepAt + keepLen; ++i) out.add(kmerAt(w, i));
        return out;
    }

    // ---- k-mer lists back to sequences: all k bases of the first k-mer, then the last base of each one after it ----
    private String spell(Iterator<Kmer> kmers, int count) {
        final char[] text = new char[Math.max(count + kMinus1, 0)];
        int at = 0;
        while (kmers.hasNext()) {
            final byte[] b = kmers.next().bytes;
            if (at == 0) for (int i = 0; i < kMinus1; ++i) text[at++] = (char) b[i];
            text[at++] = (char) b[kMinus1];
        }
        return new String(text, 0, at);
    }

    private static Iterator<Kmer> backwards(final ArrayList<Kmer> list) {
        return new Iterator<Kmer>() {
            private int next = list.size() - 1;
            public boolean hasNext() { return next >= 0; }
            public Kmer next() { return list.get(next--); }
        };
    }

    public String assemble(Collection<Kmer> kmers) { return spell(kmers.iterator(), kmers.size()); }

    public String assemble(ArrayDeque<Kmer> kmers) { return spell(kmers.iterator(), kmers.size()); }

    public String assemble(ArrayList<Kmer> kmers, int start, int end) { return spell(kmers.subList(start, end).iterator(), end - start); }

    public String assembleReverseOrder(ArrayDeque<Kmer> kmers) { return spell(kmers.descendingIterator(), kmers.size()); }

    public String assembleReverseOrder(ArrayList<Kmer> kmers) { return spell(backwards(kmers), kmers.size()); }

    public byte[] assembleBytes(ArrayList<Kmer> kmers, int start, int end) {
        final byte[] text = new byte[end - start + kMinus1];
        final byte[] head = kmers.get(start).bytes;
        for (int i = 0; i < kMinus1; ++i) text[i] = head[i];
        for (int i = start; i < end; ++i) text[kMinus1 + i - start] = kmers.get(i).bytes[kMinus1];
        return text;
    }

    public byte[] assembleReverseComplementBytes(ArrayList<Kmer> kmers, int start, int end) {
        final byte[] text = assembleBytes(kmers, start, end);
        for (int lo = 0, hi = text.length - 1; lo <= hi; ++lo, --hi) {
            final byte a = complement(text[lo]), z = complement(text[hi]);
            text[lo] = z;
            text[hi] = a;
        }
        return text;
    }

    private String column(Collection<Kmer> kmers, int which) {
        final char[] text = new char[kmers.size()];
        int at = 0;
        for (Kmer kmer : kmers) text[at++] = (char) kmer.bytes[which];
        return new String(text);
    }

    public String assembleFirstBase(Collection<Kmer> kmers) { return column(kmers, 0); }

    public String assembleLastBase(Collection<Kmer> kmers) { return column(kmers, kMinus1); }

    // ---- batched forms (not in the reference): what ported hot loops call instead of n per-element calls ----
    /** op = NativeGraph.OP_*: the n base hashes are applied in array order, exactly as n per-element calls would be */
    public void applyAll(int op, long[] baseHashes, int n) { NativeGraph.apply(handle, op, baseHashes, n); }

    public void containsAll(long[] baseHashes, int n, byte[] out) { NativeGraph.contains(handle, baseHashes, n, out); }

    public void getCountAll(long[] baseHashes, int n, float[] out) { NativeGraph.getCount(handle, baseHashes, n, out); }
}
