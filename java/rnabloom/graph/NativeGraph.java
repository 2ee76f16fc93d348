package rnabloom.graph;

import java.nio.ByteBuffer;

/**
 * JNI surface of librb_hip.so (include/rb_capi.h): one static native method per C entry point a single-process host
 * needs.  Handles are the rb_graph / rb_batch pointers as long.  Large inputs (reads, filter bytes) travel in DIRECT
 * ByteBuffers; hash / result arrays are primitive arrays.  Every method throws (IllegalArgumentException,
 * IllegalStateException, OutOfMemoryError or RuntimeException carrying rb_last_error()) instead of returning a status.
 *
 * BloomFilterDeBruijnGraph keeps its public signature (src/rnabloom/graph/BloomFilterDeBruijnGraph.java:75-104,
 * 405-461, 534-590) and forwards to these methods; only hashVals[0] crosses the boundary — the library re-derives
 * hashVals[1..] with NTM64 (src/rnabloom/bloom/hash/NTHash.java:518-527) from the graph's k.
 * The multi-GPU phases (rb_shard_*) are driven by one process per GPU and are not part of this class.
 */
public final class NativeGraph {
    static { System.loadLibrary("rb_jni"); }          // librb_jni.so links librb_hip.so

    private NativeGraph() { }

    /** HIP device the drop-in classes create their handles on: -Drb.device=N (default 0) */
    public static int defaultDevice() { return Integer.getInteger("rb.device", 0); }

    // filters (RB_DBGBF .. RB_FPKBF), add flags (RB_ADD_*), per-hash ops (RB_OP_*), neighbour directions
    public static final int DBGBF = 0, CBF = 1, RPKBF = 2, FPKBF = 3;
    public static final int ADD_REVCOMP = 1, ADD_COUNT_IF_PRESENT = 2, ADD_STORE_READ_PAIRS = 4, ADD_PAIRS_IF_PRESENT = 8;
    public static final int OP_ADD = 0, OP_ADD_IF_ABSENT = 1, OP_ADD_COUNT_IF_PRESENT = 2, OP_ADD_DBG_ONLY = 3,
            OP_ADD_COUNT_ONLY = 4, OP_ADD_READ_PAIR = 5, OP_ADD_FRAG_PAIR = 6;
    public static final int SUCCESSORS = 0, PREDECESSORS = 1, LEFT_VARIANTS = 2, RIGHT_VARIANTS = 3;
    public static final int STROBE_CANONICAL = 1, STROBE_SLIDE = 2;

    // ---- graph lifetime ----
    public static native int version();
    public static native long create(long dbgbfNumBits, long cbfNumBytes, long pkbfNumBits, int dbgbfNumHash, int cbfNumHash,
                                     int pkbfNumHash, int k, boolean stranded, boolean useReadPairedKmers, int device, long rngSeed);
    public static native void destroy(long h);
    public static native void clear(long h, int whichMask);
    public static native void destroyFilter(long h, int which);
    public static native void setReadPairedKmerDistance(long h, int d);
    public static native void setFragPairedKmerDistance(long h, int d);
    public static native void initFragmentPairs(long h, long pkbfNumBits, int pkbfNumHash);
    public static native long getOpOrdinal(long h);
    public static native void setOpOrdinal(long h, long v);

    // ---- read batches resident on the device ----
    /** seq / qual (qual may be null): direct buffers of concatenated reads; offsets[nReads + 1]. */
    public static native long batchCreateAscii(int device, ByteBuffer seq, ByteBuffer qual, long[] offsets, int nReads, int minBaseQual);
    /** .nbits bytes (NucleotideBitsWriter format) -> batch; consumed[0] receives the number of bytes used. */
    public static native long batchCreateNbits(int device, ByteBuffer bytes, long nBytes, long maxReads, long[] consumed);
    public static native void batchDestroy(long b);
    /** {reads, bases, device bytes} */
    public static native long[] batchInfo(long b);

    // ---- reads packed in HOST memory (the build's batch format: codes 8 B + valid 4 B per 32 bases, len 4 B per read; little endian) ----
    /** pinned host memory as a direct buffer: uploads from it run at link speed.  Free with hostFree — the collector does not own it. */
    public static native ByteBuffer hostAlloc(long bytes);
    public static native void hostFree(ByteBuffer buf);
    /** reads [first, first + n) of a device batch into direct buffers; returns their word count (all buffers null: size query) */
    public static native long batchDownloadPacked(long b, long first, long n, ByteBuffer codes, ByteBuffer valid, ByteBuffer len);
    /** two device batches taking turns: begin() starts the upload of a chunk and returns, finish() waits and returns a BORROWED batch handle
     *  (addBatch it, never batchDestroy it) that stays valid until the begin() after next */
    public static native long packedStreamCreate(int device, long maxReads, long maxWords);
    public static native void packedStreamBegin(long s, ByteBuffer codes, ByteBuffer valid, ByteBuffer len, long wordOffset, long readOffset, long nReads, long nWords);
    public static native long packedStreamFinish(long s);
    public static native void packedStreamDestroy(long s);
    /** FastqToGraphWorker's loop over reads already packed in host memory: ONE insert; the input goes up in pieces of pieceReads reads (0: 2^20 doubling to 2^23) on a copy stream while the pipeline works on what has arrived */
    public static native long[] addPacked(long h, ByteBuffer codes, ByteBuffer valid, ByteBuffer len, long nReads, long nWords, long pieceReads, int flags);

    /** start the upload of a packed batch the next addPacked with the same buffers and sizes inserts; returns at once (the buffers must stay untouched until then) */
    public static native void prefetchPacked(long h, ByteBuffer codes, ByteBuffer valid, ByteBuffer len, long nReads, long nWords, long pieceReads);

    // ---- stage-1 inserts; every add returns {reads, kmers, pairs, distinct, conflictOps, sortedKmers} ----
    public static native long[] addBatch(long h, long batch, long first, long n, int flags);
    public static native long[] addPairs(long h, long batch, long first, long n, int which, int flags);
    public static native long[] addFragments(long h, long batch, long first, long n, boolean loadPairedKmers);
    public static native long[] addReads(long h, ByteBuffer seq, ByteBuffer qual, long[] offsets, int nReads, int minBaseQual, int flags);
    public static native void apply(long h, int op, long[] baseHashes, int n);

    // ---- queries (may be called from many threads on one handle) ----
    public static native void contains(long h, long[] baseHashes, int n, byte[] out);
    public static native void getCount(long h, long[] baseHashes, int n, float[] out);
    public static native void filterLookup(long h, int which, long[] baseHashes, int n, byte[] out);
    public static native void filterLookupThenAdd(long h, int which, long[] baseHashes, int n, byte[] out);
    public static native void filterGetCount(long h, long[] baseHashes, int n, float[] out);
    public static native void filterIncrementAndGet(long h, long[] baseHashes, int n, float[] out);
    /** Counts of getKmers for reads [first, first + n) of a resident batch (one float per window, no hashes): packed rows at koffsets[i]
     *  (koffsets[n + 1], koffsets[0] = 0), or — koffsets == null — rows of the returned stride (longest read - k + 1), zero-padded.
     *  Call with n = 0 to learn the stride before allocating `out`. */
    public static native long batchCounts(long h, long batch, long first, long n, long[] koffsets, float[] out);
    /** getKmers of nReads sequences: koffsets[nReads + 1] is filled; pass f == null to size the outputs first. */
    public static native void getKmers(long h, ByteBuffer seq, long[] offsets, int nReads, long[] koffsets, long[] f, long[] r, float[] count);
    public static native void neighbors(long h, long[] f, long[] r, byte[] charOut, int n, int direction, long[] f4, long[] r4, float[] count4);
    public static native void walk(long h, byte[] seeds, byte[] targets, int n, int direction, int bound, float minKmerCov,
                                   byte[] outBases, long[] outF, long[] outR, float[] outCount, int[] outLen, byte[] outReason);
    public static native void greedyExtend(long h, long gateHandleOr0, byte[] seeds, int n, int direction, int lookahead, int bound,
                                           byte[] outBases, float[] outCount, int[] outLen, byte[] outReason);

    /** GraphUtils.naiveExtendRight / Left for n seeds: mode 0 terminator forms (termSeq / termOff, capacity cap), 1 bounded, 2 NoBackChecks. */
    public static native void naiveExtend(long h, byte[] seeds, int n, int direction, int mode, int bound, int cap, float minKmerCov,
                                          byte[] termSeq, long[] termOff, byte[] outBases, int[] outLen, byte[] outReason);

    // ---- filter state ----
    /** {size, bytes, numHash} */
    public static native long[] filterSize(long h, int which);
    public static native long popcount(long h, int which);
    /** rb_filter_fold: 64-bit digest of the filter's bytes, computed on the device (compare filters without exporting them) */
    public static native long fold(long h, int which);
    public static native float fpr(long h, int which);
    public static native void exportFilter(long h, int which, ByteBuffer dst, long nBytes);
    public static native void importFilter(long h, int which, ByteBuffer src, long nBytes);
    /** the same for filters of 2 GiB and more (beyond a direct ByteBuffer): the shim maps the file */
    public static native void importFilterFromFile(long h, int which, String path, long nBytes);
    public static native void exportFilterToFile(long h, int which, String path, long nBytes);
    public static native long expectedSize(long expNumElements, float fpr, int numHash);
    public static native void cbfToBloom(long src, float minCov, long dst, int which);

    // ---- hash-only work over sequences (no graph needed; countIn: 0 or a handle whose cbf is asked getCount) ----
    /** returns the number of windows / strobemers written; offsetsOut[nReads + 1]. Call with outHash == null to size. */
    public static native long minimizers(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int w, int mode,
                                         long[] offsetsOut, long[] outHash, long[] outPos);
    public static native long minimizersNext(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int w, int mode,
                                             long[] offsetsOut, long[] outHash, long[] outPos);
    public static native long minimizerSet(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int w, int mode, long[] stale,
                                           long[] offsetsOut, long[] out);
    public static native long strobemers(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int n, int wMin, int wMax,
                                         long[] offsetsOut, long[] outHash, int[] outStart, int[] outEnd);
    public static native long randstrobes(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int n, int wMin, int wMax, int flags,
                                          long countIn, long[] offsetsOut, long[] outHash, int[] outPos, float[] outCount);
    public static native long strobe3(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int wMin, int wMax, boolean canonical,
                                      long countIn, long[] offsetsOut, long[] outHash, int[] outPos, float[] outCount);
    public static native long kmerPairHashes(int device, ByteBuffer seq, long[] offsets, int nReads, int k, int shift, boolean canonical,
                                             long countIn, long[] offsetsOut, long[] outHash, float[] outCount);

    // ---- input formats ----
    /** FASTQ text -> seq / qual (direct buffers of at least textLen bytes; qual may be null) + offsets; returns the record count.
     *  Call with offsets == null for the count only. */
    public static native long fastqSplit(ByteBuffer text, long textLen, int nThreads, ByteBuffer seq, ByteBuffer qual, long[] offsets);
    /** every member of a gzip byte string (direct buffers; dst == null: returns the uncompressed size); BGZF in parallel. */
    public static native long gunzip(ByteBuffer src, long n, int nThreads, ByteBuffer dst, long cap);
    /** FASTQ text (direct buffer) parsed and encoded on the GPU; isFinal = false: a piece of a longer input, consumed[0] = where the
     *  next piece starts. */
    public static native long batchCreateFastq(int device, ByteBuffer text, long textLen, boolean isFinal, int minBaseQual, boolean useQual, long[] consumed);
    /** FASTA text parsed on the GPU (FastaReader.next semantics); consumedAndEnded = {bytes consumed, 1 if an empty header line ended the iteration}. */
    public static native long batchCreateFasta(int device, ByteBuffer text, long textLen, boolean isFinal, long[] consumedAndEnded);
    /** FastaToGraphWorker's loop over the text of a whole FASTA file; nRecords[0] = records inserted. */
    public static native long[] addFasta(long h, ByteBuffer text, long textLen, int flags, long[] nRecords);
    /** FastqToGraphWorker's loop over the text of a whole file (pieces of 1 GiB, parsed on the GPU); nRecords[0] = records inserted. */
    public static native long[] addFastq(long h, ByteBuffer text, long textLen, int minBaseQual, int flags, long[] nRecords);
    /** a FASTQ / FASTA FILE (plain or gzip) streamed piece by piece: the next piece is read / inflated and parsed while the current one is inserted */
    public static native long[] addFastqFile(long h, String path, int minBaseQual, int flags, long[] nRecords);
    public static native long[] addFastaFile(long h, String path, int flags, long[] nRecords);
    /** NucleotideBitsWriter.write for nReads sequences; out == null: returns the size needed. */
    public static native long nbitsEncode(ByteBuffer seq, long[] offsets, int nReads, ByteBuffer out, long cap);
}
