/*
 * rb_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see rb_oracle.h header comment).
 * Sequential plain-C restatement of RNA-Bloom's ntHash / Bloom / counting-Bloom / dBG hot path.
 * Citations: R/ = /root/reference/src/rnabloom/.
 */
#include "rb_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ ntHash ---- */

/* R/bloom/hash/NTHash.java:39-43 */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
#define MULTI_SEED 0x90b45d39fb6da1faULL /* :36 */
#define MULTI_SHIFT 27                   /* :33 */
#define CP_OFF 7u                        /* :30 */

static inline uint64_t rotl64(uint64_t v, int s) { s &= 63; return s ? (v << s) | (v >> (64 - s)) : v; }
static inline uint64_t rotr64(uint64_t v, int s) { s &= 63; return s ? (v >> s) | (v << (64 - s)) : v; }

/* seedTab, R/bloom/hash/NTHash.java:133-166: rows 1,3,4,5,7 hold the COMPLEMENT seeds reached
 * through (ch & cpOff); 65/97 A, 67/99 C, 71/103 G, 84/85/116/117 T(U). Everything else 0. */
uint64_t rbo_seed(unsigned c) {
    switch (c & 0xffu) {
        case 1: return SEED_T;  case 3: return SEED_G;  case 4: return SEED_A;
        case 5: return SEED_A;  case 7: return SEED_C;
        case 'A': case 'a': return SEED_A;
        case 'C': case 'c': return SEED_C;
        case 'G': case 'g': return SEED_G;
        case 'T': case 't': case 'U': case 'u': return SEED_T;
        default: return 0;
    }
}
/* msTab[c][j] = rotl(seedTab[c], j), R/bloom/hash/NTHash.java:45-131 (literal tables) */
uint64_t rbo_mstab(unsigned c, int j) { return rotl64(rbo_seed(c), j & 63); }

/* NTP64(seq,k[,start]) :318-337 */
uint64_t rbo_ntp64(const char *s, int k) {
    uint64_t h = 0;
    for (int i = 0; i < k; ++i) h ^= rbo_mstab((unsigned char)s[i], (k - 1 - i) % 64);
    return h;
}
/* NTP64RC :345-373 */
uint64_t rbo_ntp64rc(const char *s, int k) {
    uint64_t h = 0;
    for (int i = 0; i < k; ++i) h ^= rbo_mstab((unsigned char)s[i] & CP_OFF, i % 64);
    return h;
}
/* NTPC64(seq,k,start,frhVals) :465-475 — canonical = SIGNED min(f,r) (`rhVal<fhVal`) */
uint64_t rbo_ntpc64(const char *s, int k, uint64_t fr[2]) {
    fr[0] = rbo_ntp64(s, k);
    fr[1] = rbo_ntp64rc(s, k);
    return ((int64_t)fr[1] < (int64_t)fr[0]) ? fr[1] : fr[0];
}
/* forward roll: NTP64(fhVal,out,in,k) :388-390 / NTM64 roll :584-586 */
uint64_t rbo_roll_f(uint64_t f, unsigned out, unsigned in, int k) {
    return rotl64(f, 1) ^ rbo_mstab(out, k % 64) ^ rbo_mstab(in, 0);
}
/* reverse-strand roll: NTPC64 :491-495 (table form) == NTM64RC :627-629 (rotate form) */
uint64_t rbo_roll_r(uint64_t r, unsigned out, unsigned in, int k) {
    return rotr64(r, 1) ^ rbo_mstab(out & CP_OFF, 63) ^ rbo_mstab(in & CP_OFF, (k - 1) % 64);
}
/* backward forward-strand roll NTP64B :400-402 (charOut = last base, charIn = new first base) */
uint64_t rbo_roll_f_back(uint64_t f, unsigned out, unsigned in, int k) {
    return rotr64(f, 1) ^ rbo_mstab(out, 63) ^ rbo_mstab(in, (k - 1) % 64);
}
/* NTM64(bVal,hVal,k,m) :518-527 — note `i ^ k * multiSeed` parses as i ^ (k*multiSeed) */
void rbo_ntm64(uint64_t b, uint64_t *h, int k, int m) {
    h[0] = b;
    for (int i = 1; i < m; ++i) {
        uint64_t t = b * ((uint64_t)i ^ ((uint64_t)(int64_t)k * MULTI_SEED));
        t ^= t >> MULTI_SHIFT;
        h[i] = t;
    }
}
/* HashFunction.combineHashValues R/bloom/hash/HashFunction.java:260-263: the int literal
 * 0x9e3779b9 is NEGATIVE in Java and sign-extends to 0xFFFFFFFF9E3779B9 in the long addition. */
uint64_t rbo_combine(uint64_t a, uint64_t b) {
    return a ^ (b + 0xFFFFFFFF9E3779B9ULL + (a << 6) + (b >> 2));
}
uint64_t rbo_combine3(uint64_t a, uint64_t b, uint64_t c) { return rbo_combine(rbo_combine(a, b), c); }

static inline uint64_t smin64(uint64_t a, uint64_t b) { return ((int64_t)a < (int64_t)b) ? a : b; } /* Math.min(long,long) */

/* {,Canonical,ReverseComplement}NTHashIterator: R/bloom/hash/NTHashIterator.java:45-69,
 * CanonicalNTHashIterator.java:36-48, ReverseComplementNTHashIterator.java:31-42. */
int64_t rbo_hash_region(const char *seq, int64_t start, int64_t end, int k, int h, int mode,
                        uint64_t *out_h, uint64_t *out_fr) {
    int64_t max = end - k; /* NTHashIterator.start :49-55 */
    if (max < start) return 0;
    uint64_t f = 0, r = 0, base;
    int64_t n = 0;
    for (int64_t pos = start; pos <= max; ++pos, ++n) {
        if (pos == start) { /* first k-mer hashed from scratch */
            if (mode == RBO_FWD) f = rbo_ntp64(seq + pos, k);
            else if (mode == RBO_RC) r = rbo_ntp64rc(seq + pos, k);
            else { f = rbo_ntp64(seq + pos, k); r = rbo_ntp64rc(seq + pos, k); }
        } else {
            unsigned out = (unsigned char)seq[pos - 1], in = (unsigned char)seq[pos - 1 + k];
            if (mode != RBO_RC) f = rbo_roll_f(f, out, in, k);
            if (mode != RBO_FWD) r = rbo_roll_r(r, out, in, k);
        }
        if (mode == RBO_FWD) base = f;
        else if (mode == RBO_RC) base = r;
        else base = ((int64_t)r < (int64_t)f) ? r : f; /* NTHash.java:494 */
        if (out_h) rbo_ntm64(base, out_h + n * h, k, h);
        if (out_fr) { out_fr[2 * n] = f; out_fr[2 * n + 1] = r; }
    }
    return n;
}

/* {,Canonical,ReverseComplement}PairedNTHashIterator: R/bloom/hash/PairedNTHashIterator.java:56-83,
 * CanonicalPairedNTHashIterator.java:36-60, ReverseComplementPairedNTHashIterator.java:33-56. */
int64_t rbo_hash_pairs_region(const char *seq, int64_t start, int64_t end, int k, int h, int d,
                              int mode, uint64_t *out_p, uint64_t *out_l, uint64_t *out_r) {
    int64_t max = end - k - d;
    if (max < start) return 0;
    uint64_t fl = 0, rl = 0, fr_ = 0, rr = 0;
    int64_t n = 0;
    for (int64_t pos = start; pos <= max; ++pos, ++n) {
        if (pos == start) {
            fl = rbo_ntp64(seq + pos, k);      rl = rbo_ntp64rc(seq + pos, k);
            fr_ = rbo_ntp64(seq + pos + d, k); rr = rbo_ntp64rc(seq + pos + d, k);
        } else {
            unsigned o1 = (unsigned char)seq[pos - 1], i1 = (unsigned char)seq[pos - 1 + k];
            unsigned o2 = (unsigned char)seq[pos - 1 + d], i2 = (unsigned char)seq[pos - 1 + k + d];
            fl = rbo_roll_f(fl, o1, i1, k);  rl = rbo_roll_r(rl, o1, i1, k);
            fr_ = rbo_roll_f(fr_, o2, i2, k); rr = rbo_roll_r(rr, o2, i2, k);
        }
        uint64_t L, R, P;
        if (mode == RBO_FWD) { L = fl; R = fr_; P = rbo_combine(L, R); }
        else if (mode == RBO_RC) { L = rl; R = rr; P = rbo_combine(R, L); }
        else {
            L = ((int64_t)rl < (int64_t)fl) ? rl : fl;
            R = ((int64_t)rr < (int64_t)fr_) ? rr : fr_;
            P = smin64(rbo_combine(fl, fr_), rbo_combine(rr, rl)); /* Math.min, signed */
        }
        if (out_p) rbo_ntm64(P, out_p + n * h, k, h);
        if (out_l) rbo_ntm64(L, out_l + n * h, k, h);
        if (out_r) rbo_ntm64(R, out_r + n * h, k, h);
    }
    return n;
}

static const unsigned char NUC[4] = {'A', 'C', 'G', 'T'}; /* R/util/SeqUtils.java:47 */

/* R/bloom/hash/SuccessorsNTHashIterator.java:45-52, CanonicalSuccessorsNTHashIterator.java:48-61,
 * PredecessorsNTHashIterator.java:47-54, CanonicalPredecessorsNTHashIterator.java:48-60 */
void rbo_neighbors(uint64_t f, uint64_t r, unsigned char_out, int k, int h, int canonical,
                   int direction, uint64_t *out_f, uint64_t *out_r, uint64_t *out_h) {
    int km1 = (k - 1) % 64;
    uint64_t tf, tr = 0;
    if (direction == 0) {
        tf = rotl64(f, 1) ^ rbo_mstab(char_out, k % 64);
        if (canonical) tr = rotr64(r, 1) ^ rbo_mstab(char_out & CP_OFF, 63);
    } else {
        tf = rotr64(f, 1) ^ rbo_mstab(char_out, 63);
        if (canonical) tr = rotl64(r, 1) ^ rbo_mstab(char_out & CP_OFF, k % 64);
    }
    for (int i = 0; i < 4; ++i) {
        unsigned in = NUC[i];
        uint64_t nf, nr = 0, base;
        if (direction == 0) {
            nf = tf ^ rbo_mstab(in, 0);
            if (canonical) nr = tr ^ rbo_mstab(in & CP_OFF, km1);
        } else {
            nf = tf ^ rbo_mstab(in, km1);
            if (canonical) nr = tr ^ rbo_mstab(in & CP_OFF, 0);
        }
        base = canonical ? smin64(nf, nr) : nf;
        if (out_f) out_f[i] = nf;
        if (out_r) out_r[i] = nr;
        if (out_h) rbo_ntm64(base, out_h + i * h, k, h);
    }
}

/* R/bloom/hash/{Left,Right}VariantsNTHashIterator.java, Canonical{Left,Right}Variants… */
void rbo_variant(uint64_t f, uint64_t r, unsigned char_out, unsigned char_in, int k, int h,
                 int canonical, int side, uint64_t *out_f, uint64_t *out_r, uint64_t *out_h) {
    int km1 = (k - 1) % 64;
    int jf = side == 0 ? km1 : 0, jr = side == 0 ? 0 : km1;
    uint64_t nf = f ^ rbo_mstab(char_out, jf) ^ rbo_mstab(char_in, jf);
    uint64_t nr = 0;
    if (canonical) nr = r ^ rbo_mstab(char_out & CP_OFF, jr) ^ rbo_mstab(char_in & CP_OFF, jr);
    if (out_f) *out_f = nf;
    if (out_r) *out_r = nr;
    if (out_h) rbo_ntm64(canonical ? smin64(nf, nr) : nf, out_h, k, h);
}

/* -------------------------------------------------------- MiniFloat + RNG ---- */

/* Shared counter-based generator replacing the reference's unseeded Math.random()
 * (R/util/MiniFloat.java:34): splitmix64 over (seed, op ordinal) folded to 32 bits, then murmur3
 * fmix32 with the position in the read; 31 uniform bits.  Mirrors csrc/rb_device.hpp rng31. */
uint32_t rbo_rng31(uint64_t seed, uint64_t ordinal, uint32_t pos) {
    /* per-op part: one splitmix64 round folded to 32 bits */
    uint64_t z = seed ^ (ordinal * 0x9E3779B97F4A7C15ULL);
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    uint32_t x = ((uint32_t)(z >> 32) ^ (uint32_t)z) ^ (pos * 0x9E3779B1u);
    /* per-position part: murmur3 fmix32 */
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x >> 1;
}
/* MiniFloat.increment R/util/MiniFloat.java:31-38.  Java bytes are signed; counters live in
 * 0..127.  `(int)(random()*Integer.MAX_VALUE) % (1 << ((b>>3)-1)) == 0` becomes
 * `rnd31 & ((1<<s)-1) == 0` with rnd31 uniform on [0,2^31) (probability exactly 2^-s). */
uint8_t rbo_minifloat_increment(uint8_t ub, uint32_t rnd31) {
    int8_t b = (int8_t)ub;
    if (b <= 7) return (uint8_t)(b + 1);
    if (b < 127) {
        int s = (b >> 3) - 1;
        if ((rnd31 & ((1u << s) - 1u)) == 0) return (uint8_t)(b + 1);
    }
    return ub;
}
/* MiniFloat.toFloat :40-45 */
float rbo_minifloat_to_float(uint8_t ub) {
    int8_t b = (int8_t)ub;
    if (b <= 7) return (float)b;
    return ldexpf((float)((b & 7) | 8), (b >> 3) - 1);
}

/* ------------------------------------------------------------- filters ---- */

struct rbo_bloom { uint8_t *bytes; int64_t size, nbytes; int num_hash; };
struct rbo_cbf { uint8_t *bytes; int64_t size; int num_hash; };

/* BloomFilter.getExpectedSize R/bloom/BloomFilter.java:196-199 (fpr is a Java float) */
int64_t rbo_expected_size(int64_t n, float fpr, int num_hash) {
    double r = (double)(-num_hash) / log(1 - exp(log((double)fpr) / (double)num_hash));
    return (int64_t)ceil((double)n * r);
}
/* getIndex R/bloom/BloomFilter.java:108-111 */
static inline int64_t idx_of(uint64_t h, int64_t size) { return (int64_t)((h >> 1) % (uint64_t)size); }

rbo_bloom *rbo_bloom_new(int64_t size_bits, int num_hash) {
    rbo_bloom *b = (rbo_bloom *)calloc(1, sizeof *b);
    b->size = size_bits;
    b->nbytes = size_bits / 8 + ((size_bits % 8) ? 1 : 0); /* UnsafeBitBuffer.java:34-37 */
    b->bytes = (uint8_t *)calloc((size_t)(b->nbytes ? b->nbytes : 1), 1);
    b->num_hash = num_hash;
    return b;
}
void rbo_bloom_free(rbo_bloom *b) { if (b) { free(b->bytes); free(b); } }
void rbo_bloom_clear(rbo_bloom *b) { memset(b->bytes, 0, (size_t)b->nbytes); }
/* UnsafeBitBuffer.set/get/getAndSet :42-77: bit i -> byte i/8, mask 1<<(i%8) */
void rbo_bloom_add(rbo_bloom *b, const uint64_t *h) {
    for (int j = 0; j < b->num_hash; ++j) {
        int64_t i = idx_of(h[j], b->size);
        b->bytes[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}
int rbo_bloom_lookup(const rbo_bloom *b, const uint64_t *h) {
    for (int j = 0; j < b->num_hash; ++j) {
        int64_t i = idx_of(h[j], b->size);
        if (!(b->bytes[i >> 3] & (1u << (i & 7)))) return 0;
    }
    return 1;
}
int rbo_bloom_lookup_then_add(rbo_bloom *b, const uint64_t *h) {
    int found = 1;
    for (int j = 0; j < b->num_hash; ++j) { /* no early exit: every bit is set */
        int64_t i = idx_of(h[j], b->size);
        uint8_t m = (uint8_t)(1u << (i & 7));
        int old = (b->bytes[i >> 3] & m) != 0;
        b->bytes[i >> 3] |= m;
        found = old && found;
    }
    return found;
}
int64_t rbo_bloom_popcount(const rbo_bloom *b) {
    int64_t c = 0;
    for (int64_t i = 0; i < b->nbytes; ++i) c += __builtin_popcount(b->bytes[i]);
    return c;
}
float rbo_bloom_fpr(const rbo_bloom *b) {
    return (float)pow((double)rbo_bloom_popcount(b) / (double)b->size, b->num_hash);
}
uint8_t *rbo_bloom_bytes(rbo_bloom *b, int64_t *nbytes) { if (nbytes) *nbytes = b->nbytes; return b->bytes; }
/* 64-bit digest of filter bytes, the host twin of rb_filter_fold (include/rb_capi.h): wrapping sum over the non-zero 32-bit
 * little-endian words of splitmix64(word number * 0x9E3779B97F4A7C15 + word); a last partial word is padded with zeros.  Lets
 * filters too large to copy around (the 142 GB counting filter of the config-3-size runs) be compared in place. */
uint64_t rbo_fold(const uint8_t *p, int64_t nbytes) {
    uint64_t sum = 0;
    const int64_t nw = (nbytes + 3) / 4;
    for (int64_t i = 0; i < nw; ++i) {
        uint32_t x = 0;
        const int64_t rem = nbytes - 4 * i;
        memcpy(&x, p + 4 * i, rem >= 4 ? 4 : (size_t)rem);
        if (!x) continue;
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ULL + (uint64_t)x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        sum += z ^ (z >> 31);
    }
    return sum;
}
int64_t rbo_bloom_size(const rbo_bloom *b) { return b->size; }

rbo_cbf *rbo_cbf_new(int64_t size_bytes, int num_hash) {
    rbo_cbf *c = (rbo_cbf *)calloc(1, sizeof *c);
    c->size = size_bytes;
    c->bytes = (uint8_t *)calloc((size_t)(size_bytes ? size_bytes : 1), 1);
    c->num_hash = num_hash;
    return c;
}
void rbo_cbf_free(rbo_cbf *c) { if (c) { free(c->bytes); free(c); } }
void rbo_cbf_clear(rbo_cbf *c) { memset(c->bytes, 0, (size_t)c->size); }

/* CountingBloomFilter.increment(long[]) :170-194 (incrementAndGet :196-222 is the same + toFloat) */
static uint8_t cbf_increment(rbo_cbf *c, const uint64_t *h, uint32_t rnd31) {
    int8_t min = (int8_t)c->bytes[idx_of(h[0], c->size)];
    if (min != 0) {
        for (int j = 1; j < c->num_hash; ++j) {
            int8_t v = (int8_t)c->bytes[idx_of(h[j], c->size)];
            if (v < min) { min = v; if (min == 0) break; }
        }
    }
    uint8_t updated = rbo_minifloat_increment((uint8_t)min, rnd31);
    if (updated != (uint8_t)min) {
        for (int j = 0; j < c->num_hash; ++j) { /* compareAndSwap(idx, min, updated) UnsafeByteBuffer.java:94-103 */
            int64_t i = idx_of(h[j], c->size);
            if (c->bytes[i] == (uint8_t)min) c->bytes[i] = updated;
        }
    }
    return updated;
}
void rbo_cbf_increment(rbo_cbf *c, const uint64_t *h, uint32_t rnd31) { (void)cbf_increment(c, h, rnd31); }
float rbo_cbf_increment_and_get(rbo_cbf *c, const uint64_t *h, uint32_t rnd31) {
    return rbo_minifloat_to_float(cbf_increment(c, h, rnd31));
}
/* getCount(long[]) :235-251 — the zero check sits inside the h>=1 loop */
float rbo_cbf_get_count(const rbo_cbf *c, const uint64_t *h) {
    int8_t min = (int8_t)c->bytes[idx_of(h[0], c->size)];
    for (int j = 1; j < c->num_hash; ++j) {
        int8_t v = (int8_t)c->bytes[idx_of(h[j], c->size)];
        if (v < min) min = v;
        if (min == 0) return 0;
    }
    return rbo_minifloat_to_float((uint8_t)min);
}
int64_t rbo_cbf_popcount(const rbo_cbf *c) { /* non-zero BYTES: UnsafeByteBuffer.java:121-129 */
    int64_t n = 0;
    for (int64_t i = 0; i < c->size; ++i) n += c->bytes[i] != 0;
    return n;
}
float rbo_cbf_fpr(const rbo_cbf *c) {
    return (float)pow((double)rbo_cbf_popcount(c) / (double)c->size, c->num_hash);
}
uint8_t *rbo_cbf_bytes(rbo_cbf *c, int64_t *nbytes) { if (nbytes) *nbytes = c->size; return c->bytes; }

/* --------------------------------------------------------------- graph ---- */

struct rbo_graph {
    rbo_bloom *dbgbf, *rpkbf, *fpkbf;
    rbo_cbf *cbf;
    int dbg_h, cbf_h, pk_h, max_h, k, stranded;
    int read_pair_d, frag_pair_d;
    uint64_t rng_seed, ordinal;
};

rbo_graph *rbo_graph_new(int64_t dbgbf_bits, int64_t cbf_bytes, int64_t pkbf_bits, int dbg_h,
                         int cbf_h, int pk_h, int k, int stranded, int use_read_pairs,
                         uint64_t rng_seed) {
    rbo_graph *g = (rbo_graph *)calloc(1, sizeof *g);
    g->k = k; g->stranded = stranded;
    g->dbg_h = dbg_h; g->cbf_h = cbf_h; g->pk_h = pk_h;
    g->max_h = dbg_h > cbf_h ? dbg_h : cbf_h; /* dbgbfCbfMaxNumHash :90 */
    g->dbgbf = rbo_bloom_new(dbgbf_bits, dbg_h);
    g->cbf = rbo_cbf_new(cbf_bytes, cbf_h);
    g->rpkbf = use_read_pairs ? rbo_bloom_new(pkbf_bits, pk_h) : NULL; /* :100-103 */
    g->read_pair_d = -1; g->frag_pair_d = -1;
    g->rng_seed = rng_seed;
    return g;
}
void rbo_graph_free(rbo_graph *g) {
    if (!g) return;
    rbo_bloom_free(g->dbgbf); rbo_bloom_free(g->rpkbf); rbo_bloom_free(g->fpkbf); rbo_cbf_free(g->cbf);
    free(g);
}
void rbo_graph_clear(rbo_graph *g) {
    rbo_bloom_clear(g->dbgbf); rbo_cbf_clear(g->cbf);
    if (g->rpkbf) rbo_bloom_clear(g->rpkbf);
    if (g->fpkbf) rbo_bloom_clear(g->fpkbf);
    g->ordinal = 0;
}
void rbo_graph_set_read_pair_distance(rbo_graph *g, int d) { g->read_pair_d = d; }
void rbo_graph_init_fragment_pairs(rbo_graph *g, int64_t bits, int pk_h, int d) {
    if (!g->fpkbf) g->fpkbf = rbo_bloom_new(bits, pk_h); else rbo_bloom_clear(g->fpkbf);
    g->frag_pair_d = d;
}
int rbo_graph_max_hash(const rbo_graph *g) { return g->max_h; }
uint64_t rbo_graph_ordinal(const rbo_graph *g) { return g->ordinal; }
void rbo_graph_set_ordinal(rbo_graph *g, uint64_t o) { g->ordinal = o; }
rbo_bloom *rbo_graph_dbgbf(rbo_graph *g) { return g->dbgbf; }
rbo_cbf *rbo_graph_cbf(rbo_graph *g) { return g->cbf; }
rbo_bloom *rbo_graph_rpkbf(rbo_graph *g) { return g->rpkbf; }
rbo_bloom *rbo_graph_fpkbf(rbo_graph *g) { return g->fpkbf; }

/* core ops parameterised by (ordinal,pos) so the worker loop and the single-op API share them */
static inline void g_add(rbo_graph *g, const uint64_t *h, uint64_t ord, uint32_t pos) {
    if (rbo_bloom_lookup_then_add(g->dbgbf, h)) /* :405-412 */
        rbo_cbf_increment(g->cbf, h, rbo_rng31(g->rng_seed, ord, pos));
}
static inline void g_add_count_if_present(rbo_graph *g, const uint64_t *h, uint64_t ord, uint32_t pos) {
    if (rbo_bloom_lookup(g->dbgbf, h) && rbo_cbf_get_count(g->cbf, h) > 0) /* :424-428 */
        rbo_cbf_increment(g->cbf, h, rbo_rng31(g->rng_seed, ord, pos));
}
void rbo_graph_add(rbo_graph *g, const uint64_t *h) { g_add(g, h, g->ordinal++, 0); }
void rbo_graph_add_if_absent(rbo_graph *g, const uint64_t *h) { /* :414-422 */
    uint64_t ord = g->ordinal++;
    if (!rbo_bloom_lookup(g->dbgbf, h)) {
        rbo_bloom_add(g->dbgbf, h);
        rbo_cbf_increment(g->cbf, h, rbo_rng31(g->rng_seed, ord, 0));
    } else if (rbo_cbf_get_count(g->cbf, h) == 0) {
        rbo_cbf_increment(g->cbf, h, rbo_rng31(g->rng_seed, ord, 0));
    }
}
void rbo_graph_add_count_if_present(rbo_graph *g, const uint64_t *h) { g_add_count_if_present(g, h, g->ordinal++, 0); }
void rbo_graph_add_dbg_only(rbo_graph *g, const uint64_t *h) { rbo_bloom_add(g->dbgbf, h); }
void rbo_graph_add_count_only(rbo_graph *g, const uint64_t *h) {
    rbo_cbf_increment(g->cbf, h, rbo_rng31(g->rng_seed, g->ordinal++, 0));
}
void rbo_graph_add_read_pair(rbo_graph *g, const uint64_t *hp) { rbo_bloom_add(g->rpkbf, hp); }
void rbo_graph_add_fragment_pair(rbo_graph *g, const uint64_t *hp) { rbo_bloom_add(g->fpkbf, hp); }
int rbo_graph_contains(const rbo_graph *g, const uint64_t *h) { return rbo_bloom_lookup(g->dbgbf, h); }
float rbo_graph_get_count(const rbo_graph *g, const uint64_t *h) { /* :562-570 */
    return rbo_bloom_lookup(g->dbgbf, h) ? rbo_cbf_get_count(g->cbf, h) + 1 : 0;
}
int rbo_graph_lookup_read_pair(const rbo_graph *g, const uint64_t *hp) { return rbo_bloom_lookup(g->rpkbf, hp); }
int rbo_graph_lookup_fragment_pair(const rbo_graph *g, const uint64_t *hp) { return rbo_bloom_lookup(g->fpkbf, hp); }

/* ------------------------------------------------ segmentation + workers ---- */

/* R/util/SeqUtils.java:1432-1434: class = PHRED33.substring(minQual) = chars '!'+minQual .. '~' */
static inline int qual_ok(unsigned char c, int min_q) { return (int)c >= 33 + min_q && c <= '~'; }
/* R/util/SeqUtils.java:1436-1438: [ACGTU], CASE_INSENSITIVE */
static inline int base_ok(unsigned char c) {
    switch (c) { case 'A': case 'C': case 'G': case 'T': case 'U':
                 case 'a': case 'c': case 'g': case 't': case 'u': return 1; default: return 0; }
}
/* Java regex find() over "[class]{k,}": maximal runs of length >= k, left to right; the base
 * pattern is searched only inside each quality run (mSeq.region) — R/RNABloom.java:572-577. */
int64_t rbo_segments(const char *seq, const char *qual, int64_t len, int k, int min_base_qual,
                     int64_t *out_se, int64_t cap) {
    int64_t n = 0, i = 0;
    while (i < len) {
        int64_t qs = i, qe;
        if (qual) {
            while (qs < len && !qual_ok((unsigned char)qual[qs], min_base_qual)) ++qs;
            qe = qs;
            while (qe < len && qual_ok((unsigned char)qual[qe], min_base_qual)) ++qe;
        } else { qs = 0; qe = len; }
        if (qe - qs >= k) {
            int64_t j = qs;
            while (j < qe) {
                int64_t bs = j;
                while (bs < qe && !base_ok((unsigned char)seq[bs])) ++bs;
                int64_t be = bs;
                while (be < qe && base_ok((unsigned char)seq[be])) ++be;
                if (be - bs >= k) {
                    if (n < cap && out_se) { out_se[2 * n] = bs; out_se[2 * n + 1] = be; }
                    ++n;
                }
                j = be > bs ? be : bs + 1;
            }
        }
        i = qe > qs ? qe : qs + 1;
        if (!qual) break;
    }
    return n;
}

typedef struct {
    rbo_graph *g;
    const char *seq, *qual;
    const int64_t *offsets;
    int64_t n_reads, next;
    int min_q;
    unsigned flags;
    uint64_t base_ordinal;
    pthread_mutex_t mu;
    rbo_add_stats st;
} work_t;

/* one read: R/RNABloom.java:566-593 (FASTQ) / :678-697 (FASTA) */
static void process_read(rbo_graph *g, const char *s, const char *q, int64_t len, int min_q,
                         unsigned flags, uint64_t ord, rbo_add_stats *st, uint64_t *hbuf,
                         uint64_t *pbuf, int64_t *segbuf, int64_t segcap) {
    int k = g->k;
    if (q && len < k) { st->reads_skipped++; return; } /* :567-570 (FASTQ only) */
    int mode = g->stranded ? ((flags & RBO_REVCOMP) ? RBO_RC : RBO_FWD) : RBO_CANON; /* CanonicalHashFunction.java:188-196,203-206 */
    int64_t ns = rbo_segments(s, q, len, k, min_q, segbuf, segcap);
    for (int64_t si = 0; si < ns; ++si) {
        int64_t a = segbuf[2 * si], b = segbuf[2 * si + 1];
        int64_t nk = rbo_hash_region(s, a, b, k, g->max_h, mode, hbuf, NULL);
        st->segments++;
        for (int64_t i = 0; i < nk; ++i) {
            const uint64_t *h = hbuf + i * g->max_h;
            if (flags & RBO_COUNT_IF_PRESENT) g_add_count_if_present(g, h, ord, (uint32_t)(a + i));
            else g_add(g, h, ord, (uint32_t)(a + i));
        }
        st->kmers += nk;
        if ((flags & RBO_STORE_READ_PAIRS) && nk > 0) { /* :587-591 */
            int64_t np = rbo_hash_pairs_region(s, a, b, k, g->pk_h, g->read_pair_d, mode, pbuf, NULL, NULL);
            for (int64_t i = 0; i < np; ++i) rbo_bloom_add(g->rpkbf, pbuf + i * g->pk_h);
            st->pairs += np;
        }
    }
    st->reads++;
}

static void *worker_main(void *arg) {
    work_t *w = (work_t *)arg;
    rbo_graph *g = w->g;
    int64_t maxlen = 0;
    for (int64_t i = 0; i < w->n_reads; ++i) {
        int64_t l = w->offsets[i + 1] - w->offsets[i];
        if (l > maxlen) maxlen = l;
    }
    uint64_t *hbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(maxlen + 1) * (size_t)g->max_h);
    uint64_t *pbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(maxlen + 1) * (size_t)(g->pk_h > 0 ? g->pk_h : 1));
    int64_t segcap = maxlen / (g->k > 0 ? g->k : 1) + 2;
    int64_t *segbuf = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)segcap);
    rbo_add_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        pthread_mutex_lock(&w->mu); /* FastqReader.nextWithoutName is synchronized: R/io/FastqReader.java:143-149 */
        int64_t i = w->next++;
        pthread_mutex_unlock(&w->mu);
        if (i >= w->n_reads) break;
        int64_t o = w->offsets[i], l = w->offsets[i + 1] - o;
        process_read(g, w->seq + o, w->qual ? w->qual + o : NULL, l, w->min_q, w->flags,
                     w->base_ordinal + (uint64_t)i, &st, hbuf, pbuf, segbuf, segcap);
    }
    pthread_mutex_lock(&w->mu);
    w->st.reads += st.reads; w->st.reads_skipped += st.reads_skipped; w->st.segments += st.segments;
    w->st.kmers += st.kmers; w->st.pairs += st.pairs;
    pthread_mutex_unlock(&w->mu);
    free(hbuf); free(pbuf); free(segbuf);
    return NULL;
}

void rbo_graph_add_reads_mt(rbo_graph *g, const char *seq, const char *qual,
                            const int64_t *offsets, int64_t n_reads, int min_base_qual,
                            unsigned flags, int threads, rbo_add_stats *out) {
    work_t w; memset(&w, 0, sizeof w);
    w.g = g; w.seq = seq; w.qual = qual; w.offsets = offsets; w.n_reads = n_reads;
    w.min_q = min_base_qual; w.flags = flags; w.base_ordinal = g->ordinal;
    pthread_mutex_init(&w.mu, NULL);
    if (threads <= 1) worker_main(&w);
    else {
        pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int i = 0; i < threads; ++i) pthread_create(&t[i], NULL, worker_main, &w);
        for (int i = 0; i < threads; ++i) pthread_join(t[i], NULL);
        free(t);
    }
    pthread_mutex_destroy(&w.mu);
    g->ordinal += (uint64_t)n_reads;
    if (out) *out = w.st;
}
/* The same workers fed from FASTQ TEXT: the reader half of stage 1.  FastqReader.nextWithoutName is one synchronized block
 * that pulls four lines (R/io/FastqReader.java:140-149), so T workers take turns parsing a record under the lock and hash /
 * insert it outside — what `-stage 1` times on top of the in-memory loop above ("Parsed ... sequences", R/RNABloom.java:1243). */
typedef struct {
    rbo_graph *g; const char *text; int64_t len, pos, next; int min_q; unsigned flags; uint64_t base_ordinal; int64_t max_line;
    pthread_mutex_t mu; rbo_add_stats st;
} fq_work_t;
static int64_t fq_line(const char *t, int64_t len, int64_t *pos, const char **start) {   /* one line; -1 at end of text */
    int64_t p = *pos;
    if (p >= len) return -1;
    const char *nl = (const char *)memchr(t + p, '\n', (size_t)(len - p));
    int64_t e = nl ? (int64_t)(nl - t) : len;
    *start = t + p;
    *pos = e + 1;
    int64_t l = e - p;
    if (l > 0 && t[e - 1] == '\r') --l;
    return l;
}
static void *fq_worker_main(void *arg) {
    fq_work_t *w = (fq_work_t *)arg;
    rbo_graph *g = w->g;
    int64_t maxlen = w->max_line;
    uint64_t *hbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(maxlen + 1) * (size_t)g->max_h);
    uint64_t *pbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(maxlen + 1) * (size_t)(g->pk_h > 0 ? g->pk_h : 1));
    int64_t segcap = maxlen / (g->k > 0 ? g->k : 1) + 2;
    int64_t *segbuf = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)segcap);
    rbo_add_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        const char *l1, *sq, *l3, *ql;
        pthread_mutex_lock(&w->mu);
        int64_t n1 = fq_line(w->text, w->len, &w->pos, &l1), ns = fq_line(w->text, w->len, &w->pos, &sq);
        int64_t n3 = fq_line(w->text, w->len, &w->pos, &l3), nq = fq_line(w->text, w->len, &w->pos, &ql);
        int64_t i = w->next++;
        pthread_mutex_unlock(&w->mu);
        if (n1 < 0 || ns < 0 || n3 < 0 || nq < 0) break;                  /* NoSuchElementException -> null */
        if (n1 < 1 || l1[0] != '@' || n3 < 1 || l3[0] != '+' || nq != ns || ns > maxlen) break;   /* FileFormatException */
        process_read(g, sq, ql, ns, w->min_q, w->flags, w->base_ordinal + (uint64_t)i, &st, hbuf, pbuf, segbuf, segcap);
    }
    pthread_mutex_lock(&w->mu);
    w->st.reads += st.reads; w->st.reads_skipped += st.reads_skipped; w->st.segments += st.segments;
    w->st.kmers += st.kmers; w->st.pairs += st.pairs;
    pthread_mutex_unlock(&w->mu);
    free(hbuf); free(pbuf); free(segbuf);
    return NULL;
}
void rbo_graph_add_fastq_mt(rbo_graph *g, const char *text, int64_t len, int64_t max_read_len, int min_base_qual, unsigned flags, int threads,
                            rbo_add_stats *out) {
    fq_work_t w; memset(&w, 0, sizeof w);
    w.g = g; w.text = text; w.len = len; w.min_q = min_base_qual; w.flags = flags; w.base_ordinal = g->ordinal; w.max_line = max_read_len;
    pthread_mutex_init(&w.mu, NULL);
    if (threads <= 1) fq_worker_main(&w);
    else {
        pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int i = 0; i < threads; ++i) pthread_create(&t[i], NULL, fq_worker_main, &w);
        for (int i = 0; i < threads; ++i) pthread_join(t[i], NULL);
        free(t);
    }
    pthread_mutex_destroy(&w.mu);
    g->ordinal += (uint64_t)w.st.reads + (uint64_t)w.st.reads_skipped;
    if (out) *out = w.st;
}
void rbo_graph_add_reads(rbo_graph *g, const char *seq, const char *qual, const int64_t *offsets,
                         int64_t n_reads, int min_base_qual, unsigned flags, rbo_add_stats *st) {
    rbo_graph_add_reads_mt(g, seq, qual, offsets, n_reads, min_base_qual, flags, 1, st);
}

/* ------------------------------------------------------------- queries ---- */

/* HashFunction.getKmers(seq,numHash,graph) R/bloom/hash/HashFunction.java:55-83 and the canonical
 * twin CanonicalHashFunction.java:46-78: hash EVERY window (invalid chars hash as seed 0), count
 * forced to 0 for windows containing a non-ACGTU char (SeqUtils.containsInvalidNucleotides). */
int64_t rbo_graph_get_kmers(const rbo_graph *g, const char *seq, int64_t len, uint64_t *out_f,
                            uint64_t *out_r, float *out_count) {
    int k = g->k;
    if (len < k) return 0;
    int64_t n = len - k + 1;
    int mode = g->stranded ? RBO_FWD : RBO_CANON;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n * (size_t)g->max_h);
    uint64_t *fr = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)n);
    rbo_hash_region(seq, 0, len, k, g->max_h, mode, h, fr);
    int64_t bad = 0; /* number of invalid chars in current window */
    for (int64_t i = 0; i < k - 1; ++i) bad += !base_ok((unsigned char)seq[i]);
    for (int64_t i = 0; i < n; ++i) {
        bad += !base_ok((unsigned char)seq[i + k - 1]);
        if (out_f) out_f[i] = fr[2 * i];
        if (out_r) out_r[i] = fr[2 * i + 1];
        if (out_count) out_count[i] = bad ? 0.0f : rbo_graph_get_count(g, h + i * g->max_h);
        bad -= !base_ok((unsigned char)seq[i]);
    }
    free(h); free(fr);
    return n;
}

/* Kmer.getSuccessors/getPredecessors R/graph/Kmer.java:210-255, CanonicalKmer.java:226-270 */
void rbo_graph_neighbors(const rbo_graph *g, uint64_t f, uint64_t r, unsigned char_out,
                         int direction, uint64_t *out_f, uint64_t *out_r, float *count4) {
    uint64_t h[4 * 16], nf[4], nr[4];
    rbo_neighbors(f, r, char_out, g->k, g->max_h, !g->stranded, direction, nf, nr, h);
    for (int i = 0; i < 4; ++i) {
        if (out_f) out_f[i] = nf[i];
        if (out_r) out_r[i] = nr[i];
        if (count4) count4[i] = rbo_graph_get_count(g, h + i * g->max_h);
    }
}

/* -------------------------------------------------- minimizer / strobemer ---- */

/* MinimizerHashIterator.start/next R/bloom/hash/MinimizerHashIterator.java:42-88 driving
 * LongRollingWindow R/util/LongRollingWindow.java:23-83 (circular buffer; signed `<`; a rescan
 * after the minimum is overwritten walks ARRAY order, which decides positions on hash ties). */
int64_t rbo_minimizers(const char *seq, int64_t len, int k, int w, int mode, uint64_t *out_hash,
                       int64_t *out_pos) {
    if (len < k) return 0;
    int64_t nk = len - k + 1, max = nk - w + 1;
    if (max <= 0) return 0;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk);
    rbo_hash_region(seq, 0, len, k, 1, mode, h, NULL);
    int64_t *win = (int64_t *)calloc((size_t)w, sizeof(int64_t));
    for (int i = 0; i < w - 1; ++i) win[i] = (int64_t)h[i]; /* trailing 0 placeholder :54-58 */
    int size = w, min_index = 0;
    { int64_t m = win[0]; for (int i = 1; i < size; ++i) if (win[i] < m) { m = win[i]; min_index = i; } }
    int index = w - 2; int64_t pos = w - 2; /* setIndex(w-2,w-2) */
    for (int64_t p = 0; p < max; ++p) {
        int64_t v = (int64_t)h[p + w - 1];
        ++pos; if (++index >= size) index = 0;
        win[index] = v;
        if (min_index == index) {
            min_index = 0; int64_t m = win[0];
            for (int i = 1; i < size; ++i) if (win[i] < m) { m = win[i]; min_index = i; }
        } else if (v < win[min_index]) min_index = index;
        out_hash[p] = (uint64_t)win[min_index];
        if (out_pos) out_pos[p] = (min_index > index) ? pos - index - size + min_index : pos - index + min_index;
    }
    free(h); free(win);
    return max;
}

/* StrobeHashIterator.start + getInterval R/bloom/hash/StrobeHashIterator.java:48-67,133-164 */
int64_t rbo_strobemers(const char *seq, int64_t len, int k, int n, int wmin, int wmax,
                       uint64_t *out_hash, int32_t *out_start, int32_t *out_end) {
    if (len < k) return 0;
    int64_t nk = len - k + 1;
    if (!(nk > (int64_t)wmax * (n - 1))) return 0;
    int64_t max = nk - (int64_t)wmax * (n - 2) - wmin - 1;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk);
    rbo_hash_region(seq, 0, len, k, 1, RBO_FWD, h, NULL);
    for (int64_t p = 0; p <= max; ++p) {
        uint64_t sh = h[p];
        int64_t last = p;
        for (int s = 0; s < n - 1; ++s) {
            int64_t pos2 = p + (int64_t)s * wmax + wmin;
            uint64_t pos2k = h[pos2];
            uint64_t hv = rbo_combine(sh, pos2k);
            int64_t end = p + (int64_t)s * wmax + wmax;
            if (end > nk) end = nk;
            for (int64_t i = pos2 + 1; i < end; ++i) {
                uint64_t alt = h[i];
                if (alt == pos2k) pos2 = i;
                else {
                    uint64_t h2 = rbo_combine(sh, alt);
                    if (hv >= h2) { pos2 = i; pos2k = alt; hv = h2; } /* Long.compareUnsigned(h,h2) >= 0 */
                }
            }
            sh = hv; last = pos2;
        }
        out_hash[p] = sh;
        if (out_start) out_start[p] = (int32_t)p;
        if (out_end) out_end[p] = (int32_t)(last + k - 1);
    }
    free(h);
    return max + 1;
}
