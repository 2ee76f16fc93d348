/*
 * rb_capi.h — C ABI of librb_hip.so, the MI355X (gfx950) implementation of RNA-Bloom's k-mer
 * hashing / Bloom-filter de Bruijn graph hot path.
 *
 * This is the drop-in boundary: plain C, opaque handles, host pointers + sizes, int status codes
 * (0 = RB_OK; message via rb_last_error()).  The reference (bcgsc/RNA-Bloom v2.0.1, Java) has no
 * FFI seam of its own; the entry points below are what a JNI shim for
 * rnabloom.bloom.{BloomFilter,CountingBloomFilter}, rnabloom.bloom.hash.* and
 * rnabloom.graph.BloomFilterDeBruijnGraph would bind (INTEGRATION.md shows that shim).  Each
 * declaration cites the reference method(s) it replaces; R/ = src/rnabloom/ in the reference tree.
 *
 * Semantics are those of the reference run with ONE worker thread (-t 1): filters end up exactly as
 * if the reads had been processed one after another, k-mers left to right.
 * All entry points are synchronous: when they return, results are visible to the next call.
 */
#ifndef RB_CAPI_H
#define RB_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB_OK 0
#define RB_ERR_INVALID 1   /* bad argument                    (Java: IllegalArgumentException) */
#define RB_ERR_HIP 2       /* HIP runtime / device failure    (Java: RuntimeException)         */
#define RB_ERR_NOMEM 3     /* device or host allocation failed (Java: OutOfMemoryError)        */
#define RB_ERR_STATE 4     /* filter not initialised etc.     (Java: NullPointerException path) */

typedef struct rb_graph rb_graph; /* BloomFilterDeBruijnGraph + its 4 filters, device resident */
typedef struct rb_batch rb_batch; /* a batch of reads, device resident, 2-bit packed          */

/* which-filter selectors: R/graph/BloomFilterDeBruijnGraph.java:39-43 */
enum { RB_DBGBF = 0, RB_CBF = 1, RB_RPKBF = 2, RB_FPKBF = 3 };

/* BloomFilterDeBruijnGraph(long dbgbfNumBits, long cbfNumBytes, long pkbfNumBits, int dbgbfNumHash,
 *   int cbfNumHash, int pkbfNumHash, int k, boolean stranded, boolean useReadPairedKmers)
 * R/graph/BloomFilterDeBruijnGraph.java:75-104 */
typedef struct rb_graph_params {
    int64_t dbgbf_bits;           /* dbgbfNumBits  (any positive long, not a power of two) */
    int64_t cbf_bytes;            /* cbfNumBytes                                             */
    int64_t pkbf_bits;            /* pkbfNumBits   (rpkbf; fpkbf via rb_graph_init_fragment_pairs) */
    int32_t dbgbf_num_hash;       /* 1..RB_MAX_HASH */
    int32_t cbf_num_hash;
    int32_t pkbf_num_hash;
    int32_t k;                    /* 1..RB_MAX_K */
    int32_t stranded;             /* 0 => CanonicalHashFunction, 1 => HashFunction            */
    int32_t use_read_paired_kmers;
    int32_t device;               /* HIP device ordinal                                       */
    int32_t group_bits;           /* tuning: top hash bits the occurrence sort groups on (0 = default 32,
                                     max 64); any value gives identical results (DESIGN.md "split runs") */
    uint64_t rng_seed;            /* seed of the counter-based generator that replaces the
                                     unseeded Math.random() of R/util/MiniFloat.java:34        */
    int64_t max_batch_kmers;      /* 0 = default; upper bound on k-mers per internal sub-batch */
} rb_graph_params;

#define RB_MAX_HASH 8
#define RB_MAX_K 256

/* flags of rb_graph_add_* : constructor arguments of the stage-1 workers,
 * R/RNABloom.java:535-548 (FastqToGraphWorker) / :655-668 (FastaToGraphWorker) */
#define RB_ADD_REVCOMP 1u          /* reverseComplement: RC iterators (no-op when !stranded,
                                      R/bloom/hash/CanonicalHashFunction.java:188-206)          */
#define RB_ADD_COUNT_IF_PRESENT 2u /* incrementIfPresent: graph::addCountIfPresent             */
#define RB_ADD_STORE_READ_PAIRS 4u /* storeReadPairedKmers: graph.addReadSingleKmerPair        */
#define RB_ADD_PAIRS_IF_PRESENT 8u /* rb_graph_add_pairs: existingKmersOnly (both k-mers in dbgbf) */

typedef struct rb_add_stats {
    int64_t reads;    /* reads consumed (each takes one op ordinal)                 */
    int64_t kmers;    /* graph.add / addCountIfPresent calls performed              */
    int64_t pairs;    /* graph.addReadSingleKmerPair calls performed                */
    int64_t distinct; /* distinct k-mer hashes seen per sub-batch, summed           */
    int64_t conflict_ops; /* ops replayed in order because they shared a counter with another k-mer */
    int64_t sorted_kmers; /* occurrences that survived the no-op prefilter and were grouped (<= kmers) */
} rb_add_stats;

const char *rb_last_error(void); /* thread-local message of the last failing call */
int rb_version(void);
/* 16 hex digits identifying the kernel sources this library was built from (tools/csrc_id.py); no reference counterpart:
 * bench.py uses it to tie committed rocprofv3 counter summaries to the code they profiled */
const char *rb_build_id(void);

/* ---- graph lifetime: ctor :75-104, destroyXxx / clearXxx :332-350 (explicit off-heap lifetime,
 *      R/bloom/buffer/UnsafeByteBuffer.java:44,152-155) ---- */
int rb_graph_create(const rb_graph_params *p, rb_graph **out);
int rb_graph_destroy(rb_graph *g);
int rb_graph_clear(rb_graph *g, unsigned which_mask /* bit i = filter i; also resets the op ordinal when all */);
/* setReadPairedKmerDistance :375-377, setFragPairedKmerDistance :367-369 */
int rb_graph_set_read_paired_kmer_distance(rb_graph *g, int d);
int rb_graph_set_frag_paired_kmer_distance(rb_graph *g, int d);
/* initializePairKmersBloomFilter(long pkbfNumBits, int pkbfNumHash) :352-359 */
int rb_graph_init_fragment_pairs(rb_graph *g, int64_t pkbf_bits, int pkbf_num_hash);
int rb_graph_get_op_ordinal(rb_graph *g, uint64_t *out);
int rb_graph_set_op_ordinal(rb_graph *g, uint64_t v);

/* ---- read batches (the build's counterpart of FastqReader/FastaReader + the per-read regex
 *      segmentation of R/RNABloom.java:572-577, R/util/SeqUtils.java:1432-1438) ----
 * seq: concatenated ASCII bases; qual: concatenated PHRED+33 or NULL (FASTA: no quality pass);
 * offsets[n_reads+1]: read i = seq[offsets[i], offsets[i+1]).  A base is usable iff it is one of
 * ACGTUacgtu and (qual == NULL or '!'+min_base_qual <= qual <= '~').  Encoding to the packed
 * device format (2-bit codes + validity bit per base) runs on the GPU. */
int rb_batch_create_ascii(int device, const char *seq, const char *qual, const int64_t *offsets,
                          int64_t n_reads, int min_base_qual, rb_batch **out);
int rb_batch_destroy(rb_batch *b);
int rb_batch_info(const rb_batch *b, int64_t *n_reads, int64_t *n_bases, int64_t *device_bytes);
/* copy reads [first, first+n) back as ASCII with 'N' at unusable positions (for checkers) */
int rb_batch_download_ascii(const rb_batch *b, int64_t first, int64_t n, char *seq /* len = sum of lens */,
                            int64_t *offsets /* n+1 */);
/* synthetic paired-end reads generated on the device (bench data; SURVEY.md §8(d) model).
 * Produces 2*n_pairs reads: left reads 0..n_pairs-1 then right reads (as sequenced). */
typedef struct rb_synth_params {
    int64_t n_pairs;
    int64_t genome_bases;   /* G */
    int32_t read_len;       /* 150 */
    int32_t frag_mean, frag_sd;
    float sub_rate;         /* substitution errors (error bases are masked unusable, like PHRED 2) */
    float n_rate;
    float expr_sigma;       /* log-normal sigma; 0 => uniform expression */
    uint64_t seed;
    int32_t tx_min, tx_max; /* transcript length range */
    /* a slice of a larger read set (multi-GPU benches): this batch holds pairs
     * [pair_offset, pair_offset + n_pairs) of a set of total_pairs pairs drawn with the same seed,
     * so the ranks' batches together are exactly the single-GPU set.  total_pairs 0 = n_pairs. */
    int64_t pair_offset, total_pairs;
} rb_synth_params;
int rb_batch_create_synthetic(int device, const rb_synth_params *p, rb_batch **out);

/* ---- stage-1 insert: FastqToGraphWorker.run R/RNABloom.java:551-634 /
 *      FastaToGraphWorker.run :672-724, i.e. per k-mer graph.add (BloomFilterDeBruijnGraph.java:405-412)
 *      or addCountIfPresent (:424-428), per paired k-mer addReadSingleKmerPair (:455-457) ---- */
int rb_graph_add_batch(rb_graph *g, const rb_batch *b, unsigned flags, rb_add_stats *stats);
/* PairedKmersToGraphWorker.run (R/RNABloom.java:436-524): the paired k-mers (distance = the read- or fragment-paired
 * k-mer distance of `which` = RB_RPKBF / RB_FPKBF) of reads [first, first+n) into that pair filter — nothing else.
 * flags: RB_ADD_REVCOMP, RB_ADD_PAIRS_IF_PRESENT (existingKmersOnly, :466-482: only pairs whose two k-mers are both
 * in dbgbf).  stats: reads, pairs. */
int rb_graph_add_pairs(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int which, unsigned flags, rb_add_stats *stats);
/* FragmentsToGraphWorker.run (R/RNABloom.java:1463-1539): for every fragment graph.addDbgOnly of each k-mer; with
 * load_paired_kmers also addReadSingleKmerPair of its read-paired k-mers and, for fragments long enough to have one,
 * addFragmentSingleKmerPair of its fragment-paired k-mers.  Needs rb_graph_init_fragment_pairs + both distances.
 * stats: reads, kmers, pairs (read + fragment pairs). */
int rb_graph_add_fragments(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int load_paired_kmers, rb_add_stats *stats);
/* reads [first, first+n) of the batch only */
int rb_graph_add_batch_range(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, unsigned flags,
                             rb_add_stats *stats);
/* FastqToGraphWorker.run (R/RNABloom.java:526-643) over host ASCII reads: the result of rb_batch_create_ascii + rb_graph_add_batch + rb_batch_destroy,
 * as a pipeline.  The caller's arrays are registered for the call (best effort, slab by slab ahead of the copies) and handed back when it returns.
 * Input of more than one piece (256 M bases): ONE insert over a device batch sized for the whole call — lengths and word offsets are computed on the
 * GPU from `offsets`, bases and qualities follow piece by piece through two staging buffers and the 2-bit encode kernel on the handle's copy stream,
 * and the insert works on the pieces that have arrived (csrc/rb_packed.hip add_reads_streamed; 17.7 G k-mers/s for 50 M reads of 150 bases = 15 GB
 * over a 57 GB/s link, HISTORY "Round 6").  Smaller input: one chunk.  stats is added to, not cleared. */
int rb_graph_add_reads(rb_graph *g, const char *seq, const char *qual, const int64_t *offsets,
                       int64_t n_reads, int min_base_qual, unsigned flags, rb_add_stats *stats);

/* ---- per-hash mutators (h0 = hashVals[0]; the library expands it with NTM64 using the graph's k,
 *      R/bloom/hash/NTHash.java:518-527).  n ops are applied in array order; each takes one op
 *      ordinal.  op = one of RB_OP_*. ---- */
enum {
    RB_OP_ADD = 0,              /* add(long[])              :405-412 */
    RB_OP_ADD_IF_ABSENT = 1,    /* addIfAbsent              :414-422 */
    RB_OP_ADD_COUNT_IF_PRESENT = 2, /* addCountIfPresent    :424-428 */
    RB_OP_ADD_DBG_ONLY = 3,     /* addDbgOnly               :430-436 */
    RB_OP_ADD_COUNT_ONLY = 4,   /* addCountOnly             :438-440 */
    RB_OP_ADD_READ_PAIR = 5,    /* addReadSingleKmerPair    :455-457 (h0 = pair hash) */
    RB_OP_ADD_FRAG_PAIR = 6     /* addFragmentSingleKmerPair:459-461 */
};
int rb_graph_apply(rb_graph *g, int op, const uint64_t *h0, size_t n);

/* ---- queries ---- */
/* contains(long[]) :538-540 -> dbgbf.lookup R/bloom/BloomFilter.java:170-178 */
int rb_graph_contains(rb_graph *g, const uint64_t *h0, size_t n, uint8_t *out);
/* getCount(long[]) :562-570 = dbgbf.lookup ? cbf.getCount + 1 : 0 */
int rb_graph_count(rb_graph *g, const uint64_t *h0, size_t n, float *out);
/* BloomFilter.lookup(long) on any bit filter / CountingBloomFilter.getCount(long) :231-233 */
int rb_filter_lookup(rb_graph *g, int which, const uint64_t *h0, size_t n, uint8_t *out);
/* BloomFilter.lookupThenAdd (R/bloom/BloomFilter.java:147-155) for an array of base hashes, IN ARRAY ORDER: out[i] = 1 iff
 * every bit of element i was set before element i is added — by the state before the call or by an element earlier in
 * the array — and all bits are set on return.  This is what graph.lookupAndAddAllPairedKmers
 * (R/graph/BloomFilterDeBruijnGraph.java:513-526) and GraphUtils.java:640-650 AND together per sequence. */
int rb_filter_lookup_then_add(rb_graph *g, int which, const uint64_t *h0, size_t n, uint8_t *out);
int rb_filter_get_count(rb_graph *g, const uint64_t *h0, size_t n, float *out);
/* getKmers(String) :1224-1226 -> {Canonical,}HashFunction.getKmers: for every window of every
 * read of the batch: forward hash, reverse hash (0 when stranded), count (0 for windows that
 * contain a non-ACGTU base).  koffsets[n_reads+1] receives the per-read output offsets
 * (read i has max(0,len_i-k+1) windows); pass f=r=count=NULL to query sizes only.  The call works in pieces of 16 M k-mers
 * (320 MB of device scratch whatever the number of reads).  * On a shard handle (rb_graph_create_shard) only the hashes are local: count = 1 for a usable window, 0 otherwise; the counts of a
 * distributed graph come from one rb_shard_query_* exchange (rnabloom/sharded.py::ShardRank.getKmers). */
int rb_graph_kmers(rb_graph *g, const char *seq, const int64_t *offsets, int64_t n_reads,
                   int64_t *koffsets, uint64_t *f, uint64_t *r, float *count);
/* The counts of getKmers for reads that are already resident in HBM (an rb_batch): out[row(i) + p] = graph.getCount of window p of
 * read first + i (0 for a window with a non-ACGTU base) — the per-read count profile stage 2 reads first
 * (R/RNABloom.java:1984, 2097-2114: graph.getKmers(seq) before correctErrors); hashes are not returned (4 bytes per k-mer instead
 * of 20: the call is bound by the copy back).  koffsets (host, n + 1 entries, koffsets[0] = 0): row(i) = koffsets[i], read i must
 * have koffsets[i+1] - koffsets[i] = max(0, len_i - k + 1) windows; koffsets = NULL: rows of *stride_out = max_len - k + 1 counts
 * (the batch's longest read), shorter reads padded with zeros — for uniform reads that is the packed layout.  out_on_device != 0:
 * `out` is device memory of the graph's device and nothing is copied.  Host buffers are pinned for the call (see
 * rb_graph_add_reads).  A window is usable where the batch marks all its bases valid: for getKmers(String) semantics the batch
 * is one created without a quality threshold (min_base_qual 0 / no qualities).  Results equal rb_graph_kmers' counts and the
 * oracle's (tests/test_gpu_queries.py). */
int rb_graph_batch_counts(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, const int64_t *koffsets, float *out,
                          int out_on_device, int64_t *stride_out);
/* Kmer.getSuccessors/getPredecessors R/graph/Kmer.java:210-255, CanonicalKmer.java:226-270:
 * for each (f, r, char_out) the 4 neighbours in order A,C,G,T: forward hash, reverse hash and
 * graph.getCount.  direction 0 = successors (char_out = first base), 1 = predecessors
 * (char_out = last base); 2 = left variants (first base replaced, char_out = first base), 3 = right
 * variants (last base replaced, char_out = last base) — Kmer.getLeftVariants/getRightVariants
 * R/graph/Kmer.java:357-405 over R/bloom/hash/{,Canonical}{Left,Right}VariantsNTHashIterator.java; the
 * entry whose base equals char_out is the k-mer itself.  Callers apply minKmerCov to count4.  * On a shard handle: candidate hashes only (count4 = 0), counts by rb_shard_query_* (ShardRank.neighbors). */
int rb_graph_neighbors(rb_graph *g, const uint64_t *f, const uint64_t *r, const uint8_t *char_out,
                       size_t n, int direction, uint64_t *f4, uint64_t *r4, float *count4);

/* Greedy maximum-coverage walks, batched: the loop the reference runs around Kmer.getMaxCovSuccessor /
 * getMaxCovPredecessor (R/graph/Kmer.java:301-355) in GraphUtils.getMaxCoveragePath (R/util/GraphUtils.java:1591-1675)
 * — one JNI call per batch of walks instead of 4 graph.getCount calls per step.  seeds / targets: n x k bases
 * (ASCII, left to right; targets may be NULL).  direction 0 extends to the right (successors), 1 to the left
 * (predecessors).  Per step the best neighbour is the first strict maximum of graph.getCount among A,C,G,T
 * with count >= min_cov.  Walk i ends with out_reason[i] = 0 no such neighbour, 1 the best neighbour is the
 * target (not appended), 2 it is a k-mer the walk appended before (not appended; Kmer.equals: same bases),
 * 3 `bound` k-mers appended, 4 the seed holds a base outside ACGTU.  out_len[i] appended k-mers; for walk i and
 * step j < out_len[i]: out_bases[i*bound + j] the base added (last base of the k-mer for direction 0, first base
 * for direction 1), out_f / out_r / out_count (each may be NULL: 20 of the 21 bytes a step returns) its forward hash,
 * reverse hash, graph.getCount. */
int rb_graph_walk(rb_graph *g, const char *seeds, const char *targets, size_t n, int direction, int bound, float min_cov,
                  char *out_bases, uint64_t *out_f, uint64_t *out_r, float *out_count, int32_t *out_len, uint8_t *out_reason);

/* GraphUtils.greedyExtendRight / greedyExtendLeft(graph, source, lookahead, bound), batched
 * (R/util/GraphUtils.java:1961-1976, :1906-1921): up to `bound` times greedyExtend{Right,Left}Once (:501-529, :564-592) —
 * the neighbours with graph.getCount >= 1 (Kmer.getSuccessors / getPredecessors, R/graph/Kmer.java:199-255); none ends the
 * walk (out_reason 0), one is taken, several are scored with getMaxMedianCoverage{Right,Left} (:248-310, :375-438: the
 * best minimum k-mer coverage over the depth-first paths of exactly `lookahead` k-mers starting at the candidate) and
 * the highest score wins, a tie going to the strictly larger count.  out_reason 3 = bound reached, 4 = seed with a
 * base outside ACGTU.  Outputs as rb_graph_walk (out_count may be NULL).  lookahead <= 16.
 * gate (may be NULL): the `BloomFilter bf` of the gated variants (greedyExtendRight(graph, source, lookahead, bound, bf),
 * :1978-1993; Kmer.getSuccessors(k, numHash, graph, bf), R/graph/Kmer.java:257-299: a neighbour must pass bf.lookup before
 * its count is read) — another handle on the same device, same k, whose dbgbf is that filter. */
int rb_graph_greedy_extend(rb_graph *g, const rb_graph *gate, const char *seeds, size_t n, int direction, int lookahead, int bound,
                           char *out_bases, float *out_count, int32_t *out_len, uint8_t *out_reason);
/* GraphUtils.naiveExtendRight / naiveExtendLeft R/util/GraphUtils.java:6780-7112, one walk per seed k-mer (n seeds of k
 * bases): follow the only neighbour with count >= min_cov while the walk is unbranched.  mode 0 = the forms with a
 * terminator set (:6780-6833, :6959-7012): every k-mer of the walk's terminator sequence term_seq[term_off[i], term_off[i+1])
 * stops it, and so does a k-mer it added before; `cap` = room for added bases per walk (these forms have no bound; a walk
 * that fills it reports reason 6 and can be continued from its last k-mer).  mode 1 = the bounded forms (:6835-6886,
 * :7014-7065), mode 2 = naiveExtend{Right,Left}NoBackChecks (:6888-6933, :7067-7112); both write up to bound + 1 bases per
 * walk (out_bases is n * (bound + 1)).  maxTipLength does not appear: it only feeds Kmer.hasDepthLeft / hasDepthRight,
 * which never consult the graph and always return true (R/graph/Kmer.java:407-486).
 * out_reason: 0 no neighbour, 1 back branch (a variant of the current k-mer in the base about to leave exists), 2 several
 * neighbours, 3 bound, 4 seed with a base outside ACGTU, 5 terminator / already-added k-mer, 6 capacity, 7 the candidate
 * repeats the seed or the k-mer added last (mode 2). */
int rb_graph_naive_extend(rb_graph *g, const char *seeds, size_t n, int direction, int mode, int bound, int cap, float min_cov,
                          const char *term_seq, const int64_t *term_off, char *out_bases, int32_t *out_len, uint8_t *out_reason);

/* ---- filter state: popcount / FPR / raw bytes (the on-disk format of
 *      R/bloom/BloomFilter.java:113-124 is exactly these bytes,
 *      R/bloom/buffer/UnsafeByteBuffer.java:160-201) ---- */
int rb_filter_size(rb_graph *g, int which, int64_t *size /* bits or bytes */, int64_t *nbytes, int *num_hash);
/* bit filters: set bits (UnsafeByteBuffer.bitPopCount :131-150); cbf: non-zero bytes (:121-129) */
int rb_filter_popcount(rb_graph *g, int which, int64_t *out);
int rb_filter_fpr(rb_graph *g, int which, float *out); /* BloomFilter.getFPR :185-194 */
/* 64-bit digest of the filter bytes this handle holds, computed on the device: the wrapping sum, over the non-zero 32-bit
 * little-endian words, of splitmix64(global word number * 0x9E3779B97F4A7C15 + word).  The digests of the shards of a
 * distributed filter add up (mod 2^64) to the digest of the same filter on one GPU, so filters too large to export
 * (BASELINE config 4: 150 GB of counters) are compared on the device.  No reference counterpart (the reference compares
 * nothing); tests/test_gpu_config3.py, rnabloom.graph.fold_bytes is the host restatement. */
int rb_filter_fold(rb_graph *g, int which, uint64_t *out);
int rb_filter_export(rb_graph *g, int which, void *dst, size_t nbytes);
int rb_filter_import(rb_graph *g, int which, const void *src, size_t nbytes);
int64_t rb_expected_size(int64_t n, float fpr, int num_hash); /* BloomFilter.getExpectedSize :196-199 */
/* BloomFilterDeBruijnGraph.destroyDbgbf / destroyCbf / destroyRpkbf / destroyFpkbf R/graph/BloomFilterDeBruijnGraph.java:249-275
 * (BloomFilter.destroy :244-246): frees the filter's device memory; rb_filter_size then reports RB_ERR_STATE and calls
 * that need the filter fail.  rb_graph_init_fragment_pairs may create an fpkbf again (restorePkbf :341-350). */
int rb_graph_destroy_filter(rb_graph *g, int which);
/* CountingBloomFilter.incrementAndGet(long[]) R/bloom/CountingBloomFilter.java:196-222 for h0[0..n) one after the other,
 * in array order (every call sees the increments before it); out[i] = MiniFloat.toFloat(updated).  Order-exact and serial
 * on the device: meant for the subsampler's short per-read sequences, not for bulk counting (rb_graph_apply). */
int rb_filter_increment_and_get(rb_graph *g, const uint64_t *h0, size_t n, float *out);
/* CountingBloomFilter.getBloomFilter(minCov) R/bloom/CountingBloomFilter.java:328-338: bit i of filter `which` of `dst`
 * (same number of bits as src has counters, same device) := MiniFloat.toFloat(counter i of src) >= min_cov. */
int rb_cbf_to_bloom(rb_graph *src, float min_cov, rb_graph *dst, int which);

/* ---- hash-only test hook: {,Canonical,ReverseComplement}NTHashIterator over every usable
 *      segment of every read of a batch.  mode 0 fwd, 1 canonical, 2 reverse-complement.
 *      out_h0[count] in read order; out_read/out_pos optional.  Call with out_h0 == NULL to get
 *      the count. ---- */
int rb_nthash_batch(const rb_batch *b, int k, int mode, int64_t first, int64_t n, int64_t *count,
                    uint64_t *out_h0, uint32_t *out_read, uint32_t *out_pos);

/* ---- sketching over long reads (BASELINE config 5): hash-only, no shared state => plain data
 *      parallelism over reads (replicas on several GPUs, no collective).  Both hash EVERY window of a
 *      read like the reference iterators do (no segmentation; unusable bases hash as seed 0). ----
 * MinimizerHashIterator.next() R/bloom/hash/MinimizerHashIterator.java:27-128 over
 * LongRollingWindow R/util/LongRollingWindow.java:23-83: for every window of w consecutive k-mers
 * the SIGNED minimum of hVals[0] (mode 0 forward / 1 canonical / 2 reverse-complement iterator).
 * moffsets[n_reads+1]: read i yields max(0, len_i-k+1-w+1) windows.  out_pos = LongRollingWindow.getMinPos()
 * (the rolling window is replayed exactly, including which of several equal minima it reports). */
int rb_minimizers(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode,
                  int64_t *moffsets, uint64_t *out_hash, int64_t *out_pos);
/* StrobeHashIterator.getInterval(p) R/bloom/hash/StrobeHashIterator.java:133-164 (as used by
 * SeqSubsampler.strobemerBased R/util/SeqSubsampler.java:367): order-n randstrobes over FORWARD k-mer
 * hashes, unsigned argmin of combineHashValues, ties -> rightmost, equal k-mer hash -> slide right.
 * soffsets[n_reads+1]: read i yields numKmers - wMax*(n-2) - wMin strobemers if numKmers > wMax*(n-1). */
int rb_strobemers(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int n, int wmin, int wmax,
                  int64_t *soffsets, uint64_t *out_hash, int32_t *out_start, int32_t *out_end);

/* The other strobemer iterators, batched the same way (one thread per strobemer).  `count_in` (may be NULL): a handle
 * whose counting filter is asked CountingBloomFilter.getCount(long) (R/bloom/CountingBloomFilter.java:235-251) for
 * every hash without leaving the device — the "hash -> cbf.getCount" half of SeqSubsampler.strobemerBased / kmerBased
 * (R/util/SeqSubsampler.java:389-395, 176-179); out_count then receives the counts.
 * rb_randstrobes: StrobeHashIterator.next() / get() R/bloom/hash/StrobeHashIterator.java:73-131 and
 * CanonicalStrobeHashIterator.next() / get() R/bloom/hash/CanonicalStrobeHashIterator.java:79-140.  out_pos: n
 * positions per strobemer (the k-mer itself, then the strobes = getStrobes() / HashedPositions.pos). */
#define RB_STROBE_CANONICAL 1 /* CanonicalStrobeHashIterator: strobes chosen on forward hashes, SIGNED min with the reverse chain */
#define RB_STROBE_SLIDE 2     /* StrobeHashIterator.get / getInterval: the chosen strobe slides across equal k-mer hashes */
#define RB_MAX_STROBES 8
int rb_randstrobes(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int n, int wmin, int wmax, int flags,
                   rb_graph *count_in, int64_t *soffsets, uint64_t *out_hash, int32_t *out_pos, float *out_count);
/* Strobe3HashIterator / CanonicalStrobe3HashIterator next() / get() (R/bloom/hash/Strobe3HashIterator.java:78-151,
 * R/bloom/hash/CanonicalStrobe3HashIterator.java:85-225): strobemers of positions getMin()..getMax(); read i yields
 * numKmers - 2*wMin (canonical: numKmers - 2*wMax, if positive) when numKmers > 2*wMin.  out_pos: {pos1, p, pos3}. */
int rb_strobe3(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int wmin, int wmax, int canonical,
               rb_graph *count_in, int64_t *soffsets, uint64_t *out_hash, int32_t *out_pos, float *out_count);
/* SeqSubsampler.kmerBased's k-mer pair hashes R/util/SeqSubsampler.java:176-179 (stranded: combine(h[i], h[i+shift]))
 * and :266-268 (min_signed(combine(f[i], f[i+shift]), combine(r[i+shift], r[i]))): read i yields numKmers - shift. */
int rb_kmer_pair_hashes(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int shift, int canonical,
                        rb_graph *count_in, int64_t *poffsets, uint64_t *out_hash, float *out_count);
/* MinimizerHashIterator.nextMinimizer() R/bloom/hash/MinimizerHashIterator.java:97-112 while hasNext(): the first
 * window's minimizer, then one entry whenever the position of the window minimum moves (the sequence
 * SeqUtils.getMinimizerChainString consumes, R/util/SeqUtils.java:1731-1757).  out_hash / out_pos need room for one
 * entry per window (rb_minimizers' count); moffsets[n_reads+1] receives the per-read offsets of what was written. */
int rb_minimizers_next(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode,
                       int64_t *moffsets, uint64_t *out_hash, int64_t *out_pos);
/* GraphUtils.getMinimizers(seq, numKmers, itr, windowSize) R/util/GraphUtils.java:2462-2549: the distinct window
 * minimizers of each read, sorted as signed longs.  Reads with numKmers <= windowSize yield ONE value,
 * min_signed(stale[i], every k-mer hash), where stale[i] is what itr.hVals[0] held before the call (:2480-2488:
 * 0 on a fresh iterator — pass stale = NULL — else the last hash of the previous sequence).  out needs room for
 * max(1, windows) entries per read. */
int rb_minimizer_set(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode,
                     const uint64_t *stale, int64_t *moffsets, uint64_t *out);

/* ---- input formats ----
 * FASTQ text -> the buffers rb_graph_add_reads / rb_batch_create_ascii take.  FastqReader.nextWithoutName
 * R/io/FastqReader.java:140-186: four lines per record, line 1 starts with '@', line 3 with '+' (else RB_ERR_INVALID with
 * the reference's message), a record cut off by the end of the text is dropped; lines end at \n, \r\n or \r.
 * Threaded (n_threads <= 0: all hardware threads).  Call with offsets == NULL for the record count; seq / qual
 * (qual may be NULL) need offsets[n_reads] bytes — len is always enough. */
int rb_fastq_split(const char *text, size_t len, int n_threads, char *seq, char *qual, int64_t *offsets, int64_t cap_reads,
                   int64_t *n_reads);

/* FileUtils.getTextFileReader for ".gz" (R/util/FileUtils.java:50-57: GZIPInputStream, every member of a concatenated file).
 * dst == NULL: *out_len = uncompressed size.  Whatever follows a member and is not another gzip header (padding, garbage) ends the stream
 * silently, as GZIPInputStream.readTrailer does; the first member must be gzip.  BGZF input (bgzip: members of at most 64 KiB that carry their own size) is
 * inflated by n_threads threads at once (<= 0: all hardware threads); any other gzip file member after member on one thread. */
int rb_gunzip(const void *src, size_t n, int n_threads, void *dst, size_t cap, size_t *out_len);

/* The same on the GPU: the text is uploaded as it is and lines, records (same rules, same error texts) and the 2-bit
 * encoding + quality mask are found there — no host pass over the bytes, no intermediate buffers.
 * rb_batch_create_fastq: at most 4 GiB of text; final = 0: the text is a piece of a longer input (a last line without an
 * end of line belongs to the next piece); *consumed = where the records that were not complete start.
 * rb_graph_add_fastq = FastqToGraphWorker's loop (R/RNABloom.java:526-643) over a whole file's text: 1 GiB pieces, the
 * next piece is uploaded and parsed while the current one is inserted; *n_records (may be NULL) = records inserted. */
int rb_batch_create_fastq(int device, const char *text, size_t len, int final, int min_base_qual, int use_qual, rb_batch **out, size_t *consumed);
int rb_graph_add_fastq(rb_graph *g, const char *text, size_t len, int min_base_qual, unsigned flags, rb_add_stats *stats, int64_t *n_records);
/* FASTA the same way (FastaReader.next, R/io/FastaReader.java:70-104: trimmed lines, '>' opens a record, the lines after it are joined,
 * an empty line closes it and must be followed by a header — "Incorrect FASTA header format" — or by another empty line, which
 * ends the iteration: *ended; a header that is the very last line is never handed out).  rb_graph_add_fasta = FastaToGraphWorker's
 * loop (R/RNABloom.java:645-732), no quality pass.  A non-final piece leaves its last record unread. */
int rb_batch_create_fasta(int device, const char *text, size_t len, int final, rb_batch **out, size_t *consumed, int *ended);
int rb_graph_add_fasta(rb_graph *g, const char *text, size_t len, unsigned flags, rb_add_stats *stats, int64_t *n_records);
/* The same two worker loops over a FILE, streamed as FastqReader / FastaReader stream theirs (R/io/FastqReader.java:140-186,
 * R/io/FastaReader.java:70-104 over FileUtils.getTextFileReader, R/util/FileUtils.java:50-57: GZIPInputStream for gzip input — detected
 * here by the magic bytes, every member of the file).  Pieces of 256 MiB of text (RB_FASTQ_PIECE): a reader thread reads — and
 * inflates — piece c + 1 into pinned memory, uploads and parses it while the GPU inserts piece c, so the file never sits in memory
 * as a whole and a single-member .gz (which nothing can inflate in parallel) costs max(inflate, insert) instead of their sum. */
int rb_graph_add_fastq_file(rb_graph *g, const char *path, int min_base_qual, unsigned flags, rb_add_stats *stats, int64_t *n_records);
int rb_graph_add_fasta_file(rb_graph *g, const char *path, unsigned flags, rb_add_stats *stats, int64_t *n_records);
/* .nbits files (R/io/NucleotideBitsReader.java, R/util/SeqBitsUtils.java:159-161, 236-263: per sequence a 4-byte
 * big-endian length, then ceil(len/4) bytes of four 2-bit bases each, first base in the top bits, value - 128) straight
 * into a packed device batch: the bytes are uploaded as they are and permuted on the GPU (every base is usable — the
 * format has no code for N).  At most max_reads records (< 0: all); *consumed = bytes used; a truncated last record
 * is left unread, as NucleotideBitsReader.next() returns null for it. */
int rb_batch_create_nbits(int device, const void *bytes, size_t nbytes, int64_t max_reads, rb_batch **out, size_t *consumed);
/* ---- read batches packed in HOST memory ("input already resident in host memory in the build's batch format") ----
 * The build's own batch format, as a host keeps it once its reads are packed (by rb_batch_download_packed, or by a caller's packer):
 *   codes[w] u64: 32 bases, 2 bits each, base i of the word at bits 2i..2i+1 (A C G T = 0 1 2 3);  valid[w] u32: bit i = base i is usable
 *   (ACGTU and quality >= the threshold the reads were packed with);  len[r] u32: read r owns ceil(len[r] / 32) consecutive words.
 * 12 bytes per 32 bases + 4 per read over the link; the device-only columns of a batch are computed on the GPU.  The reader side of
 * FastqToGraphWorker.run (R/RNABloom.java:551-634: reads are fetched while other workers insert) is the packed stream: two device
 * batches taking turns, chunk c + 1 uploading on the stream's own HIP stream while the caller inserts chunk c. */
typedef struct rb_packed_stream rb_packed_stream;
/* pinned host memory (hipHostMalloc): uploads from it run at link speed without a registration pass per call */
int rb_host_alloc(size_t bytes, void **out);
int rb_host_free(void *p);
/* reads [first, first + n) of a device batch in the host format; *n_words = words of those reads (all three arrays NULL: size query) */
int rb_batch_download_packed(const rb_batch *b, int64_t first, int64_t n, uint64_t *codes, uint32_t *valid, uint32_t *len, int64_t *n_words);
/* a stream whose chunks hold at most max_reads reads / max_words words (the two device batches are allocated here, once) */
int rb_packed_stream_create(int device, int64_t max_reads, int64_t max_words, rb_packed_stream **out);
/* start the upload of one chunk and return at once (a helper thread drives the copies); one chunk may be in flight */
int rb_packed_stream_begin(rb_packed_stream *s, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words);
/* wait for the chunk begun last; *out is a device batch owned by the stream (never rb_batch_destroy it), valid until the begin() after
 * next reuses its buffer.  An error (the lengths do not add up to n_words, a failed copy) is reported here. */
int rb_packed_stream_finish(rb_packed_stream *s, const rb_batch **out);
int rb_packed_stream_destroy(rb_packed_stream *s);
/* FastqToGraphWorker.run over n_reads packed reads in host memory as ONE insert: the lengths go up first (word offsets and owners are computed
 * on the GPU), then codes / valid follow on the handle's copy stream in pieces of piece_reads reads (0: 2^20 reads, doubling up to 2^23) while
 * the insert pipeline already works on the pieces that have arrived — a sub-batch waits only for the pieces its words lie in.  Same filters
 * and statistics as rb_graph_add_batch of the same reads with `flags`; the caller's arrays are free again when the call returns. */
int rb_graph_add_packed(rb_graph *g, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words,
                        int64_t piece_reads, unsigned flags, rb_add_stats *stats);
/* Start the upload of a packed batch that a later rb_graph_add_packed call with THE SAME arrays and sizes will insert, and return at once (the
 * reader side of stage 1 fetches the next file while the workers insert this one, R/RNABloom.java:7123-7188).  The arrays must stay as they
 * are until that call returns.  Up to two batches may be on their way (uploads run in the order they were started). */
int rb_graph_prefetch_packed(rb_graph *g, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words,
                             int64_t piece_reads);
/* NucleotideBitsWriter.write R/io/NucleotideBitsWriter.java:24-31 for n_reads sequences; out == NULL: *written = size
 * needed.  A base outside ACGTU is an error (the reference stores a RANDOM base there, SeqBitsUtils.java:154-155). */
int rb_nbits_encode(const char *seq, const int64_t *offsets, int64_t n_reads, void *out, size_t cap, size_t *written);

/* ---- sharded engine (one process per GPU; filters split by index range across `count` shards) ----
 * Multi-GPU counterpart of rb_graph_add_batch.  The reference is a single shared-memory process, so
 * there is no reference interface for this; the phases below are what bench.py / rnabloom.sharded
 * drive, with torch.distributed (RCCL all_to_all / all_gather) moving the byte buffers between
 * ranks.  Shard s of a filter of `size` indices owns [s*span, min(size,(s+1)*span)), span =
 * roundup64(ceil(size/count)); k-mer hash space is split by log2(count) bits of hashVals[0]
 * (bits 40.., uniform even for canonical hashes)
 * (count must be a power of two).  All `dev` pointers are DEVICE pointers owned by the caller
 * (exchange buffers).  Results equal the single-GPU / sequential results bit for bit.
 *
 * per global sub-batch = a range of reads of a batch EVERY rank holds (replicated input: packed reads
 * are 1/38 of the bytes of their (hash, occurrence) records, so the reads travel, not the records):
 *   hash_group  walk all reads of the sub-batch, keep the windows whose k-mer this rank owns and the
 *             no-op prefilter lets through, group them into runs -> Bloom-bit / counter requests
 *             bucketed by filter owner; read-paired k-mers of this rank's 1/count slice of the reads
 *             -> rpkbf bit indices bucketed by owner
 *   [all_to_all requests, pair probes]
 *   serve     owner: bit tests + first-setter arbitration + bit sets, counter claims, pair bit sets
 *   [all_to_all replies back]
 *   resolve   found flags, op counts, counter updates of runs that own their counters alone; the runs that share
 *             a counter and can reach one of theirs ask those counters' owners who else can (RB_SLOT_ORD_IDX)
 *   [all_to_all counter writes + ordered-set questions]     apply_writes, order_serve (claimants per counter)
 *   [all_to_all answers]
 *   order_finish   a run that can reach a contested counter another such run claimed is in the ordered set O* (a superset of the
 *             single-GPU engine's set, without its closure rounds: csrc/rb_shard.hip) -> its (run, contested counter) edges; every
 *             other run is finished in place, its writes to other ranks' counters follow the edges as tagged records
 *   [all_gather edges + tagged writes]      apply_tagged (every rank: the writes that fall into its range)
 *   conflict_route    components of the edge graph (same on every rank); each rank sends its
 *             conflicting runs + their pending occurrence ids to the rank owning the component
 *   [all_to_all runs, ops]
 *   conflict_replay   ordered replay of whole components on a private counter table
 *   [all_to_all final counter bytes]   apply_writes
 *
 * Read-pair filter of a sharded graph: every rank ORs the pairs of its reads into a private full-size copy and the copies are merged into
 * the owners' shards at the end of an insert call; a copy KEEPS its bits afterwards (its seen-pair cache vouches for them) and merges them
 * again with the next call.  Therefore rb_graph_clear and rb_filter_import of RB_RPKBF on a sharded graph are COLLECTIVE: every rank must
 * clear (or import its shard) before any rank inserts again — a rank that does not would OR its old bits back into the peers' freshly
 * cleared or imported shards at the next flush.  (rnabloom.sharded.ShardRank.clear / the Java driver clear all ranks.)
 */
enum {
    RB_SLOT_REC_KEYS = 0,    /* split-reads mode: u64 h0 of the surviving records, bucketed by k-mer owner */
    RB_SLOT_REC_OCC = 1,     /* u32 occurrence ids (same order)          */
    RB_SLOT_PAIR_IDX = 2,    /* u64 global rpkbf bit indices, by owner   */
    RB_SLOT_DREQ_IDX = 3,    /* u64 global dbgbf bit index               */
    RB_SLOT_DREQ_PROBE = 4,  /* u64 (occ_first << 4 | probe)             */
    RB_SLOT_CREQ_IDX = 5,    /* u64 global counter index                 */
    RB_SLOT_W_IDX = 6,       /* u64 global counter index                 */
    RB_SLOT_W_VAL = 7,       /* u8 new byte, 0xFF = just drop the claim  */
    RB_SLOT_CONF_EDGES = 8,  /* 16 B {u64 counter index, u32 run id, u32 0}; run id = local number * count + rank.  Round 6: followed by the counter
                              * writes of the runs finished outside the ordered set, {u64 counter index, u32 0xFFFFFFFF, u32 byte (0xFF: drop the claim)} */
    RB_SLOT_CONF_RUNS = 9,   /* 24 B {u64 h0, u64 counter bytes, u32 component, u32 n_ops | kinds << 28}, by component owner */
    RB_SLOT_CONF_OPS = 10,   /* u32 occurrence ids of those runs, run after run */
    RB_SLOT_CW_IDX = 11,     /* u64 global counter index (replayed components) */
    RB_SLOT_CW_VAL = 12,     /* u8 final byte                            */
    RB_SLOT_Q_BIDX = 13,     /* u64 global bit indices of a query, by owner   */
    RB_SLOT_Q_CIDX = 14,     /* u64 global counter indices of a query, by owner */
    RB_SLOT_CACHE_UPD = 15,  /* split-reads mode: 16 B {u64 h0, u64 exponent} prefilter-cache entries learnt this sub-batch */
    RB_SLOT_ORD_IDX = 16,    /* u64 global counter index: the contested counters of the runs that can reach one (ordered-set question), by owner */
    RB_SLOT_COUNT = 17
};
int rb_graph_create_shard(const rb_graph_params *p, int shard_rank, int shard_count, rb_graph **out);
/* reads [first, first+n) = the global sub-batch (identical arguments on every rank); [pair_first,
 * +pair_n) = the slice of it whose read-paired k-mers this rank walks; ordinal0 = op ordinal of read
 * `first`; flags as rb_graph_add_batch.  counts: requests per destination rank. */
/* ---- split-reads mode (the default from 4 ranks up; rnabloom.sharded picks): every rank hashes only ITS slice
 * of the sub-batch and prefilters it against a REPLICA of the whole prefilter cache; the survivors
 * (12 B records) travel to the k-mer owners; what the owners learn (cache entries) is all-gathered and
 * applied to every replica.  Hashing is then 1/count per rank instead of replicated.
 *   hash   (own slice -> records bucketed by owner, pair probes)   [all_to_all records, pair probes]
 *   group  (received records, source-rank order = read order -> runs -> requests)   [all_to_all requests]
 *   serve / resolve / conflict_* as above; resolve also fills RB_SLOT_CACHE_UPD   [all_gather]   cache_apply */
int rb_shard_set_cache_replication(rb_graph *g, int on);
int rb_shard_hash(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int64_t own_first, int64_t own_n,
                  uint64_t ordinal0, uint32_t pos_bits, unsigned flags, int64_t *rec_counts /*[count]*/,
                  int64_t *pair_counts /*[count]*/, rb_add_stats *stats);
/* (round 4: the records are grouped where they lie — keys_dev / occ_dev are scratch of the call and hold garbage afterwards) */
int rb_shard_group(rb_graph *g, void *keys_dev, void *occ_dev, int64_t n, uint64_t ordinal0,
                   uint32_t pos_bits, unsigned flags, int64_t *dreq_counts, int64_t *creq_counts);
int rb_shard_cache_apply(rb_graph *g, const void *upd_dev, int64_t n);
/* split reads, look-ahead (k <= 31): the window walk + prefilter of this rank's slice [own_first, own_first + own_n) of the NEXT
 * sub-batch, enqueued on the library's producer stream without waiting for the GPU.  Call it after rb_shard_cache_apply of the current
 * sub-batch; rb_shard_hash with the same arguments then starts from its result.  Any other call in between simply discards it. */
int rb_shard_hash_begin_split(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int64_t own_first, int64_t own_n,
                              uint64_t ordinal0, uint32_t pos_bits, unsigned flags);
/* optional look-ahead (k <= 31): enqueue the window-hash/prefilter pass of the NEXT sub-batch on the
 * library's producer stream (begin), then its emit + sort + run grouping (emit; waits for the count of
 * kept records only) — both return without waiting for the GPU, so the work overlaps the serve /
 * resolve / conflict phases and the exchanges of the current sub-batch.  rb_shard_hash_group with the
 * same arguments picks the prepared sub-batch up; with different arguments it starts from scratch. */
int rb_shard_hash_begin(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, uint64_t ordinal0, uint32_t pos_bits,
                        unsigned flags);
int rb_shard_hash_emit(rb_graph *g);
int rb_shard_hash_group(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int64_t pair_first, int64_t pair_n,
                        uint64_t ordinal0, uint32_t pos_bits, unsigned flags, int64_t *dreq_counts /*[count]*/,
                        int64_t *creq_counts /*[count]*/, int64_t *pair_counts /*[count]*/, rb_add_stats *stats);
int rb_shard_serve(rb_graph *g, int mode, const void *dreq_idx_dev, const void *dreq_probe_dev, int64_t nd,
                   const void *creq_idx_dev, int64_t nc, const void *pair_idx_dev, int64_t np,
                   void *dreply_dev /* u8[nd] */, void *creply_dev /* u8[nc] */);
int rb_shard_resolve(rb_graph *g, int mode, const void *dreply_dev, const void *creply_dev, int64_t *w_counts /*[count]*/,
                     int64_t *ord_counts /*[count]: RB_SLOT_ORD_IDX entries per counter owner*/, rb_add_stats *stats);
int rb_shard_order_serve(rb_graph *g, const void *ord_idx_dev, int64_t n, void *reply_dev /* u8[n]: claimants that can reach a contested counter */);
int rb_shard_order_finish(rb_graph *g, int mode, const void *ord_reply_dev /* answers to this rank's RB_SLOT_ORD_IDX, in its order */,
                          int64_t *n_conf_runs, int64_t *n_conf_records /* 16-byte records in RB_SLOT_CONF_EDGES: edges, then tagged writes */, rb_add_stats *stats);
int rb_shard_apply_tagged(rb_graph *g, const void *records_dev /* the gathered RB_SLOT_CONF_EDGES of all ranks */, int64_t n_records);
int rb_shard_apply_writes(rb_graph *g, const void *w_idx_dev, const void *w_val_dev, int64_t n);
/* edges_dev: the edges of ALL ranks (rank order); gid_bound > every run id in them */
int rb_shard_conflict_route(rb_graph *g, const void *edges_dev, int64_t n_edges, int64_t gid_bound, int64_t *run_counts,
                            int64_t *op_counts, rb_add_stats *stats);
int rb_shard_conflict_replay(rb_graph *g, const void *runs_dev, int64_t n_runs, const void *ops_dev, int64_t n_ops,
                             int64_t *w_counts);
/* an internal slot (bucketed by destination): copy it out, or borrow the device pointer (valid until
 * the phase that fills the slot runs again) */
int rb_shard_take(rb_graph *g, int slot, void *dst_dev, int64_t nbytes);
int rb_shard_slot(rb_graph *g, int slot, void **dev_ptr, int64_t *nbytes);
int rb_shard_span(rb_graph *g, int which, int64_t *span, int64_t *lo, int64_t *hi);
/* queries on a sharded graph (BloomFilter.lookup R/bloom/BloomFilter.java:139-147,
 * CountingBloomFilter.getCount :196-210, BloomFilterDeBruijnGraph.getCount :562-570) for hashes any
 * rank holds: make (indices bucketed by owner: slots Q_BIDX / Q_CIDX) -> [all_to_all] -> serve (bit /
 * counter byte per index) -> [all_to_all back] -> finish.  what: 0 = lookup in bit filter
 * `which_bits` (RB_DBGBF / RB_RPKBF), 1 = counting-filter count, 2 = graph count (count + 1, or 0 unless in dbgbf). */
int rb_shard_query_make(rb_graph *g, int what, int which_bits, const uint64_t *h0_host, size_t n, int64_t *bit_counts,
                        int64_t *ctr_counts);
int rb_shard_query_serve(rb_graph *g, int which_bits, const void *bidx_dev, int64_t nb, const void *cidx_dev, int64_t nc,
                         void *breply_dev /* u8[nb] */, void *creply_dev /* u8[nc] */);
int rb_shard_query_finish(rb_graph *g, int which_bits, const void *breply_dev, const void *creply_dev,
                          uint8_t *out8_host /* what 0 */, float *outf_host /* what 1, 2 */);

/* ---- graph traversal on a sharded graph: rb_graph_walk / rb_graph_greedy_extend / rb_graph_naive_extend when no GPU holds the
 * filters (BASELINE configs[3]: stage 2's inner loops, R/util/GraphUtils.java:1591-1675, :1906-1996, :6780-7112, over
 * Kmer.getSuccessors / getPredecessors R/graph/Kmer.java:199-355).  Every rank drives its own walks with the SAME kernels the
 * single-GPU calls run; the counts they need come from the owners of the filter ranges through the query exchange above:
 *   begin (this rank's seeds; may be none) -> repeat { advance -> [all_to_all of slots Q_BIDX / Q_CIDX] -> rb_shard_query_serve ->
 *   [all_to_all back] -> absorb } until no rank reports an active walk -> end.
 * advance runs every walk up to the next neighbourhood whose counts it has not been told (a walk suspends at the start of its
 * current step and replays it from there when the answers are in: the step's earlier questions are then cache hits), files the
 * requests — the four neighbours of a k-mer go out together, a naive extension's back-branch variants with them — and makes the
 * query for them (bit_counts / ctr_counts per destination rank, as rb_shard_query_make).  One exchange round per step for the
 * max-coverage walk and the naive extension, one per level of the lookahead search for the greedy extension.
 * kind 0 = rb_graph_walk (targets may be NULL), 1 = rb_graph_greedy_extend (mode_or_lookahead = lookahead; gate: rb_shard_trav_set_gate),
 * 2 = rb_graph_naive_extend (mode_or_lookahead = mode; term_seq / term_off for mode 0); the other arguments, the
 * outputs and the reasons are those calls'.  answer_cap: counts one step of a walk may hold (0: 4, 8, or
 * 4 (1 + 4 (1 + 2 lookahead)) by kind); a greedy step whose search needs more ends the walk with reason 8.  Results equal the
 * single-GPU calls on the same filters (tests/test_gpu_sharded_walks.py). */
int rb_shard_trav_begin(rb_graph *g, int kind, const char *seeds, const char *targets, size_t n, int direction, int mode_or_lookahead,
                        int bound, int cap, float min_cov, const char *term_seq, const int64_t *term_off, int answer_cap);
/* the `bf` variants of the greedy extension (R/util/GraphUtils.java:1978-1993, Kmer.getSuccessors(k, numHash, graph, bf)
 * R/graph/Kmer.java:257-299): gate = this rank's shard handle of ANOTHER sharded graph (same k, strandedness, rank count) whose dbgbf is
 * that filter.  Call after begin; every round then carries a second request: advance also fills slot Q_BIDX of the GATE handle
 * (gate_bit_counts per destination rank) with the same k-mers as a lookup, to be exchanged, served by rb_shard_query_serve on the
 * owners' gate handles and handed to absorb as gate_breply_dev.  A k-mer that fails the gate counts 0. */
int rb_shard_trav_set_gate(rb_graph *g, rb_graph *gate);
int rb_shard_trav_advance(rb_graph *g, int64_t *n_active, int64_t *bit_counts, int64_t *ctr_counts, int64_t *gate_bit_counts /* NULL without a gate */);
int rb_shard_trav_absorb(rb_graph *g, const void *breply_dev, const void *creply_dev, const void *gate_breply_dev /* NULL without a gate */);
/* out_f: kinds 0 and 2 (forward hashes of the appended k-mers), out_r and out_count: kind 0, out_count: kinds 0 and 1; each may be
 * NULL.  rounds (may be NULL): exchange rounds this rank's walks took.  Frees the traversal. */
int rb_shard_trav_end(rb_graph *g, char *out_bases, uint64_t *out_f, uint64_t *out_r, float *out_count, int32_t *out_len,
                      uint8_t *out_reason, int64_t *rounds);

/* development: milliseconds for 2 x 2^28 random returning atomics on the counting filter where it lies (mode 0: OR with 0, 1: XOR pairs that
 * restore the contents) — tools/alloc_lottery.py looks for what the probe stages' allocation-dependent time follows */
int rb_debug_probe_cbf(rb_graph *g, int mode, float *ms_out);
/* development / tests: the library's own device-wide primitives (csrc/rb_sort.hip, csrc/rb_group.hip; rocPRIM until round 3) on
 * host arrays.  rb_debug_scan_u32: out[i] = in[0] + ... + in[i-1] (wrapping); the device copies start
 * (misalign & 3) [input] and (misalign >> 2) [output] words past a 16-byte boundary (0: both aligned — the vectorised path).  rb_debug_sort_pairs: stable sort of (key, value) on key bits [lo_begin, lo_end) then [hi_begin, hi_end)
 * (hi_begin < 0: one range); vals may be NULL (keys only), vals64 != 0: 64-bit values. */
int rb_debug_scan_u32(int device, const uint32_t *in, size_t n, uint32_t *out, int misalign);
int rb_debug_sort_pairs(int device, uint64_t *keys, void *vals, int vals64, size_t n, int lo_begin, int lo_end, int hi_begin, int hi_end);

/* ---- the exchange driver below the C ABI (csrc/rb_comm.hip) ----
 * rb_shard_add_range = rnabloom/sharded.py::ShardRank.add_range in the library: all sub-batches of reads [first, first + n)
 * (the same call on every rank, every rank holding the same batch), every phase above and every exchange between them, on the
 * handle's stream — nothing but per-peer byte counts visits the host, no interpreter between the phases.  ordinal0 = reads
 * inserted since the last clear (the op ordinal of read `first`).  The communicator is either
 *   RCCL (one process per GPU): rank 0 calls rb_shard_comm_unique_id, the 128 bytes travel to the other ranks by whatever the
 *     host has (torch.distributed broadcast, MPI, a file), every rank calls rb_shard_comm_create_rccl; transfers are
 *     ncclSend / ncclRecv groups in pieces of at most 256 MiB per peer (librccl.so is loaded with dlopen on first use), or
 *   a loopback hub for `world` virtual ranks of ONE process on one GPU (one host thread per rank calls rb_shard_add_range with
 *     the same hub; device-to-device copies): what the one-GPU tests drive the protocol with. */
typedef struct rb_shard_comm rb_shard_comm;
int rb_shard_comm_unique_id(void *out128);
int rb_shard_comm_create_rccl(const void *id128, int rank, int world, int device, rb_shard_comm **out);
int rb_shard_comm_create_loopback(int world, rb_shard_comm **out);
/* Read-pair filter of a sharded graph, replicated accumulation (round 4): rpkbf.add is a pure OR (R/bloom/BloomFilter.java:133-137),
 * so while a file is inserted every rank ORs the pairs of its reads into a private full-size copy and the copies are merged into the
 * owners' index ranges when the call ends: begin() hands out this rank's copy as ONE send buffer cut into `shard_count` consecutive
 * pieces (counts[r] = bytes of rank r's range; all zero when the handle uses the routed path, RB_SHARD_PAIRS=route), the caller does
 * the all-to-all, end() ORs the G received pieces (each the size of this rank's range) into the shard and clears the copy.  Every rank
 * calls both at the end of every collective insert call (rb_shard_add_range and rnabloom/sharded.py::ShardRank.add_range do). */
int rb_shard_pairs_flush_begin(rb_graph *g, void **send_dev, int64_t *counts);
int rb_shard_pairs_flush_end(rb_graph *g, const void *recv_dev, const int64_t *recv_counts);
/* a small all-to-all and all-gather of known bytes through the communicator's own transport, checked on arrival (every rank calls it;
 * big_bytes is added to every message: > 256 MiB exercises the piece-wise path).  bench.py runs it before the first step. */
int rb_shard_comm_selftest(rb_shard_comm *c, int rank, int device, int64_t big_bytes);
int rb_shard_comm_destroy(rb_shard_comm *c);
int rb_shard_add_range(rb_graph *g, rb_shard_comm *c, const rb_batch *b, int64_t first, int64_t n, unsigned flags,
                       int64_t reads_per_substep, uint32_t pos_bits, uint64_t ordinal0, rb_add_stats *stats);

/* ---- instrumentation: per-kernel-class HIP-event timing on the library's own stream ---- */
#define RB_PROF_MAX 32
typedef struct rb_profile {
    int32_t n;
    const char *name[RB_PROF_MAX];
    double ms[RB_PROF_MAX];        /* accumulated since last reset */
    int64_t launches[RB_PROF_MAX];
} rb_profile;
int rb_graph_profile_enable(rb_graph *g, int on);
int rb_graph_profile_get(rb_graph *g, rb_profile *out, int reset);

#ifdef __cplusplus
}
#endif
#endif
