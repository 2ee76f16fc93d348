/*
 * rb_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never the thing measured
 * except as bench.py's `cpu_baseline` leg).
 *
 * A plain-C sequential restatement of RNA-Bloom's k-mer hashing / Bloom-dBG hot path.  Every
 * function cites the reference file:line it follows (paths relative to
 * /root/reference/src/rnabloom/, abbreviated R/).
 *
 * PARITY PINNING: the reference is Java-only and ships no tests / golden vectors, and no JVM
 * exists in this image, so this oracle is "parity unpinned" by reference-run outputs.  It is pinned
 * by (1) the literal constant tables of R/bloom/hash/NTHash.java (tests/golden/nthash_kat.json is
 * generated from those literals by tests/golden/gen_golden.py) and (2) the algebraic identities
 * of SURVEY.md Appendix A.8 (tests/test_oracle_identities.py).
 *
 * RANDOMNESS: R/util/MiniFloat.java:31-38 draws from an UNSEEDED Math.random() once a counter byte
 * reaches 16, so the reference is not reproducible run-to-run there.  Oracle and HIP path share a
 * counter-based generator keyed on (graph seed, op ordinal, position in read) — see rbo_rng31().
 */
#ifndef RB_ORACLE_H
#define RB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ntHash (R/bloom/hash/NTHash.java) ---- */
uint64_t rbo_seed(unsigned c);                       /* seedTab[c]            :133-166 */
uint64_t rbo_mstab(unsigned c, int j);               /* msTab[c][j]           :96-131  */
uint64_t rbo_ntp64(const char *s, int k);            /* NTP64                 :318-337 */
uint64_t rbo_ntp64rc(const char *s, int k);          /* NTP64RC               :345-373 */
uint64_t rbo_ntpc64(const char *s, int k, uint64_t fr[2]); /* NTPC64          :449-475 */
uint64_t rbo_roll_f(uint64_t f, unsigned out, unsigned in, int k);   /* NTP64 roll   :388-390 */
uint64_t rbo_roll_r(uint64_t r, unsigned out, unsigned in, int k);   /* reverse roll :491-495 */
uint64_t rbo_roll_f_back(uint64_t f, unsigned out, unsigned in, int k); /* NTP64B    :400-402 */
void     rbo_ntm64(uint64_t b, uint64_t *h, int k, int m);           /* NTM64        :518-527 */
uint64_t rbo_combine(uint64_t a, uint64_t b);        /* HashFunction.combineHashValues R/bloom/hash/HashFunction.java:260-263 */
uint64_t rbo_combine3(uint64_t a, uint64_t b, uint64_t c);           /* :265-267 */

/* ---- sequence iterators (R/bloom/hash/{,Canonical,ReverseComplement}NTHashIterator.java) ---- */
enum { RBO_FWD = 0, RBO_CANON = 1, RBO_RC = 2 };
/* hash every k-mer of seq[start,end): out_h[(n)*h + j], optional out_fr[n*2] (canonical f,r).
 * returns number of k-mers (max(0,end-start-k+1)). */
int64_t rbo_hash_region(const char *seq, int64_t start, int64_t end, int k, int h, int mode,
                        uint64_t *out_h, uint64_t *out_fr);
/* paired iterator (R/bloom/hash/{,Canonical,ReverseComplement}PairedNTHashIterator.java):
 * out_p[n*h+j]; optional out_l/out_r.  returns count (max(0,end-start-k-d+1)). */
int64_t rbo_hash_pairs_region(const char *seq, int64_t start, int64_t end, int k, int h, int d,
                              int mode, uint64_t *out_p, uint64_t *out_l, uint64_t *out_r);

/* ---- neighbour / variant iterators ---- */
/* direction 0 = successors (R/bloom/hash/{,Canonical}SuccessorsNTHashIterator.java),
 * 1 = predecessors ({,Canonical}PredecessorsNTHashIterator.java).  canonical!=0 uses (f,r).
 * char_out = first base (succ) / last base (pred) of the k-mer.  Outputs 4 entries (A,C,G,T):
 * out_f[4], out_r[4] (r only when canonical), out_h[4*h]. */
void rbo_neighbors(uint64_t f, uint64_t r, unsigned char_out, int k, int h, int canonical,
                   int direction, uint64_t *out_f, uint64_t *out_r, uint64_t *out_h);
/* side 0 = left variants (swap first base), 1 = right variants (swap last base)
 * (R/bloom/hash/{,Canonical}{Left,Right}VariantsNTHashIterator.java). One alt base. */
void rbo_variant(uint64_t f, uint64_t r, unsigned char_out, unsigned char_in, int k, int h,
                 int canonical, int side, uint64_t *out_f, uint64_t *out_r, uint64_t *out_h);

/* ---- MiniFloat (R/util/MiniFloat.java:26-46) + shared RNG ---- */
uint32_t rbo_rng31(uint64_t seed, uint64_t ordinal, uint32_t pos);
uint8_t  rbo_minifloat_increment(uint8_t b, uint32_t rnd31);
float    rbo_minifloat_to_float(uint8_t b);

/* ---- filters ---- */
typedef struct rbo_bloom rbo_bloom;      /* R/bloom/BloomFilter.java + buffer/UnsafeBitBuffer.java */
typedef struct rbo_cbf rbo_cbf;          /* R/bloom/CountingBloomFilter.java + buffer/UnsafeByteBuffer.java */
int64_t  rbo_expected_size(int64_t n, float fpr, int num_hash);   /* BloomFilter.java:196-199 */
rbo_bloom *rbo_bloom_new(int64_t size_bits, int num_hash);
void     rbo_bloom_free(rbo_bloom *);
void     rbo_bloom_clear(rbo_bloom *);
void     rbo_bloom_add(rbo_bloom *, const uint64_t *h);            /* :133-137 */
int      rbo_bloom_lookup(const rbo_bloom *, const uint64_t *h);   /* :170-178 */
int      rbo_bloom_lookup_then_add(rbo_bloom *, const uint64_t *h);/* :147-155 */
int64_t  rbo_bloom_popcount(const rbo_bloom *);                    /* UnsafeByteBuffer.java:131-150 */
float    rbo_bloom_fpr(const rbo_bloom *);                         /* :185-194 */
uint8_t *rbo_bloom_bytes(rbo_bloom *, int64_t *nbytes);
uint64_t rbo_fold(const uint8_t *bytes, int64_t nbytes);   /* host twin of rb_filter_fold */
int64_t  rbo_bloom_size(const rbo_bloom *);
rbo_cbf *rbo_cbf_new(int64_t size_bytes, int num_hash);
void     rbo_cbf_free(rbo_cbf *);
void     rbo_cbf_clear(rbo_cbf *);
void     rbo_cbf_increment(rbo_cbf *, const uint64_t *h, uint32_t rnd31);   /* :170-194 */
float    rbo_cbf_increment_and_get(rbo_cbf *, const uint64_t *h, uint32_t rnd31); /* :196-222 */
float    rbo_cbf_get_count(const rbo_cbf *, const uint64_t *h);    /* :235-251 */
int64_t  rbo_cbf_popcount(const rbo_cbf *);                        /* UnsafeByteBuffer.java:121-129 */
float    rbo_cbf_fpr(const rbo_cbf *);                             /* :254-263 */
uint8_t *rbo_cbf_bytes(rbo_cbf *, int64_t *nbytes);

/* ---- graph facade (R/graph/BloomFilterDeBruijnGraph.java) ---- */
typedef struct rbo_graph rbo_graph;
rbo_graph *rbo_graph_new(int64_t dbgbf_bits, int64_t cbf_bytes, int64_t pkbf_bits, int dbg_h,
                         int cbf_h, int pk_h, int k, int stranded, int use_read_pairs,
                         uint64_t rng_seed);                       /* ctor :75-104 */
void     rbo_graph_free(rbo_graph *);
void     rbo_graph_clear(rbo_graph *);
void     rbo_graph_set_read_pair_distance(rbo_graph *, int d);     /* :375-377 */
void     rbo_graph_init_fragment_pairs(rbo_graph *, int64_t bits, int pk_h, int d); /* :352-359 */
int      rbo_graph_max_hash(const rbo_graph *);
uint64_t rbo_graph_ordinal(const rbo_graph *);
void     rbo_graph_set_ordinal(rbo_graph *, uint64_t);
/* single-op API; each call consumes one op ordinal (pos 0) for the shared RNG */
void     rbo_graph_add(rbo_graph *, const uint64_t *h);                 /* :405-412 */
void     rbo_graph_add_if_absent(rbo_graph *, const uint64_t *h);       /* :414-422 */
void     rbo_graph_add_count_if_present(rbo_graph *, const uint64_t *h);/* :424-428 */
void     rbo_graph_add_dbg_only(rbo_graph *, const uint64_t *h);        /* :430-436 */
void     rbo_graph_add_count_only(rbo_graph *, const uint64_t *h);      /* :438-440 */
void     rbo_graph_add_read_pair(rbo_graph *, const uint64_t *hp);      /* :455-457 */
void     rbo_graph_add_fragment_pair(rbo_graph *, const uint64_t *hp);  /* :459-461 */
int      rbo_graph_contains(const rbo_graph *, const uint64_t *h);      /* :538-540 */
float    rbo_graph_get_count(const rbo_graph *, const uint64_t *h);     /* :562-570 */
int      rbo_graph_lookup_read_pair(const rbo_graph *, const uint64_t *hp);     /* :529-531 */
int      rbo_graph_lookup_fragment_pair(const rbo_graph *, const uint64_t *hp); /* :525-527 */
rbo_bloom *rbo_graph_dbgbf(rbo_graph *);
rbo_cbf   *rbo_graph_cbf(rbo_graph *);
rbo_bloom *rbo_graph_rpkbf(rbo_graph *);
rbo_bloom *rbo_graph_fpkbf(rbo_graph *);

/* flags for rbo_graph_add_reads (mirror include/rb_capi.h) */
enum {
    RBO_REVCOMP = 1,          /* use the reverse-complement iterators (reverse read files) */
    RBO_COUNT_IF_PRESENT = 2, /* addCountIfPresent instead of add  (R/RNABloom.java:547) */
    RBO_STORE_READ_PAIRS = 4  /* storeReadPairedKmers               (R/RNABloom.java:587-591) */
};
typedef struct {
    int64_t reads, reads_skipped, segments, kmers, pairs;
} rbo_add_stats;
/* Stage-1 worker loop, sequential (-t 1): R/RNABloom.java:551-634 (FASTQ; qual != NULL) and
 * :672-724 (FASTA; qual == NULL).  Reads are seq[offsets[i], offsets[i+1]).  Every read consumes
 * one op ordinal. */
void rbo_graph_add_reads(rbo_graph *, const char *seq, const char *qual, const int64_t *offsets,
                         int64_t n_reads, int min_base_qual, unsigned flags, rbo_add_stats *st);
/* Same work split over T threads pulling one read at a time under a mutex with the reference's
 * NON-ATOMIC byte read-modify-writes (R/bloom/buffer/UnsafeByteBuffer.java:54-69,94-103) — the
 * cpu_baseline leg.  Not deterministic for T>1 (neither is the reference). */
void rbo_graph_add_reads_mt(rbo_graph *, const char *seq, const char *qual,
                            const int64_t *offsets, int64_t n_reads, int min_base_qual,
                            unsigned flags, int threads, rbo_add_stats *st);
/* the same from FASTQ text under the reader lock (R/io/FastqReader.java:140-149): the with-parsing CPU baseline */
void rbo_graph_add_fastq_mt(rbo_graph *g, const char *text, int64_t len, int64_t max_read_len, int min_base_qual, unsigned flags, int threads,
                            rbo_add_stats *out);
/* segmentation only: writes [start,end) pairs; returns number of segments (cap = max pairs) */
int64_t rbo_segments(const char *seq, const char *qual, int64_t len, int k, int min_base_qual,
                     int64_t *out_se, int64_t cap);

/* getKmers(seq) (R/bloom/hash/HashFunction.java:55-83 / CanonicalHashFunction.java:46-78):
 * out_f/out_r (r only when !stranded), out_count; returns number of k-mers */
int64_t rbo_graph_get_kmers(const rbo_graph *, const char *seq, int64_t len, uint64_t *out_f,
                            uint64_t *out_r, float *out_count);
/* Kmer.getSuccessors/getPredecessors counts (R/graph/Kmer.java:199-255, CanonicalKmer.java:226-270):
 * count4[i] = graph.getCount of neighbour i (A,C,G,T) — callers apply minKmerCov. */
void rbo_graph_neighbors(const rbo_graph *, uint64_t f, uint64_t r, unsigned char_out,
                         int direction, uint64_t *out_f, uint64_t *out_r, float *count4);

/* ---- minimizer / strobemer (config 5) ---- */
/* MinimizerHashIterator semantics (R/bloom/hash/MinimizerHashIterator.java:27-128 with
 * R/util/LongRollingWindow.java:23-83): window of w consecutive k-mer hashes (hVals[0] of the
 * graph-mode iterator), SIGNED minimum, leftmost on ties.  Emits one (hash,pos) per window;
 * returns number of windows (numKmers - w + 1, or 0). */
int64_t rbo_minimizers(const char *seq, int64_t len, int k, int w, int mode, uint64_t *out_hash,
                       int64_t *out_pos);
/* StrobeHashIterator.get(i) (R/bloom/hash/StrobeHashIterator.java:96-131): order n randstrobes
 * over forward k-mer hashes, unsigned argmin, rightmost on ties, then slide-right rule.
 * out_hash[i], out_end[i] (= last strobe position + k - 1 ... see HashedInterval); returns
 * number of strobemers (numKmers - wMax*(n-2) - wMin, or 0). */
int64_t rbo_strobemers(const char *seq, int64_t len, int k, int n, int wmin, int wmax,
                       uint64_t *out_hash, int32_t *out_start, int32_t *out_end);
/* rb_oracle_sketch.c: the other sketch iterators and SeqSubsampler's hashing halves (citations there) */
int64_t rbo_randstrobes(const char *seq, int64_t len, int k, int n, int wmin, int wmax, int flags /* 1 canonical, 2 slide */,
                        uint64_t *out_hash, int32_t *out_pos /* n per strobemer */);
int64_t rbo_strobe3(const char *seq, int64_t len, int k, int wmin, int wmax, int canonical, uint64_t *out_hash, int32_t *out_pos /* 3 per */);
int64_t rbo_minimizers_next(const char *seq, int64_t len, int k, int w, int mode, uint64_t *out_hash, int64_t *out_pos);
int64_t rbo_minimizer_set(const char *seq, int64_t len, int k, int w, int mode, uint64_t stale, uint64_t *out);
int64_t rbo_kmer_pair_hashes(const char *seq, int64_t len, int k, int shift, int canonical, uint64_t *out);


#ifdef __cplusplus
}
#endif
#endif
