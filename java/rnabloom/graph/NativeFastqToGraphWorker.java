package rnabloom.graph;

import java.nio.ByteBuffer;
import rnabloom.io.FastqReader;
import rnabloom.io.FastqRecord;
import rnabloom.io.FileFormatException;

/**
 * Batched replacement of RNABloom.FastqToGraphWorker (src/rnabloom/RNABloom.java:526-643).  The reference's worker pulls one
 * record at a time under the reader's lock, cuts it into [ACGTU]{k,} runs inside PHRED >= minQual runs with two regexes, rolls
 * ntHash over every run and calls graph.add / addCountIfPresent per k-mer and addReadSingleKmerPair per paired k-mer.  This
 * worker pulls up to BATCH_READS records, copies their sequence and quality lines into two direct buffers and hands them to
 * NativeGraph.addReads once: segmentation, hashing and the order-exact inserts happen on the GPU (same skip-if-shorter-than-k
 * rule, same quality threshold, same per-read order of k-mers; DESIGN.md s3).
 *
 * ONE worker per file keeps the reference's -t 1 order, which is the order the library's results are defined for; several
 * workers on one reader interleave their batches as the reference's threads interleave their records.
 * For plain-text files NativeGraph.addFastq(handle, mappedText, ...) needs no Java-side record handling at all.
 */
public class NativeFastqToGraphWorker implements Runnable {
    // A call of more than one piece (256 M bases) is streamed inside the library — lengths and offsets computed on the GPU, bases and qualities uploaded and
    // encoded piece by piece beside ONE insert (csrc/rb_packed.hip add_reads_streamed: 17.7 G k-mers/s for a whole file in one call); calls do not overlap
    // each other, so a batch is as large as two direct buffers may be: four pieces (2 x 1 GiB of direct memory, -XX:MaxDirectMemorySize permitting).
    public static final int BATCH_READS = 1 << 23;
    private static final int BATCH_BASES = 1 << 30;

    private final BloomFilterDeBruijnGraph graph;
    private final FastqReader fr;
    private final int minBaseQual, flags, k;
    private long numReads = 0;
    private boolean successful = false;
    private Exception exception = null;

    public NativeFastqToGraphWorker(BloomFilterDeBruijnGraph graph, FastqReader fr, int minBaseQual, boolean reverseComplement,
                                    boolean incrementIfPresent, boolean storeReadPairedKmers) {
        this.graph = graph;
        this.fr = fr;
        this.k = graph.getK();
        this.minBaseQual = minBaseQual;
        this.flags = (reverseComplement ? NativeGraph.ADD_REVCOMP : 0) | (incrementIfPresent ? NativeGraph.ADD_COUNT_IF_PRESENT : 0)
                   | (storeReadPairedKmers ? NativeGraph.ADD_STORE_READ_PAIRS : 0);
    }

    @Override
    public void run() {
        ByteBuffer seq = ByteBuffer.allocateDirect(BATCH_BASES), qual = ByteBuffer.allocateDirect(BATCH_BASES);
        long[] offsets = new long[BATCH_READS + 1];
        FastqRecord record = new FastqRecord();
        try {
            boolean more = true;
            while (more) {
                seq.clear(); qual.clear();
                int n = 0;
                while (n < BATCH_READS) {
                    fr.nextWithoutName(record);
                    if (record.seq == null) { more = false; break; }
                    final int len = record.seq.length();
                    if (len < k) continue;                               // :572-575: skip to the next read (not counted, :626)
                    ++numReads;
                    if (seq.remaining() < len) {                         // flush, then take this record into the next batch
                        flush(seq, qual, offsets, n);
                        seq.clear(); qual.clear(); n = 0;
                    }
                    for (int i = 0; i < len; ++i) { seq.put((byte) record.seq.charAt(i)); qual.put((byte) record.qual.charAt(i)); }
                    offsets[++n] = seq.position();
                }
                flush(seq, qual, offsets, n);
            }
            successful = true;
        } catch (FileFormatException | RuntimeException e) {
            exception = e;
        }
    }

    private void flush(ByteBuffer seq, ByteBuffer qual, long[] offsets, int n) {
        if (n > 0) NativeGraph.addReads(graph.getHandle(), seq, qual, offsets, n, minBaseQual, flags);
    }

    public boolean isSuccessful() { return successful; }

    public Exception getExceptionCaught() { return exception; }

    public long getReadCount() { return numReads; }
}
