/* rb_oracle_sketch.c — CPU restatement (TEST INFRASTRUCTURE, see rb_oracle.h) of the reference's remaining sketch
 * iterators and of the hashing halves of SeqSubsampler, statement by statement:
 *   StrobeHashIterator.next / get            R/bloom/hash/StrobeHashIterator.java:73-131
 *   CanonicalStrobeHashIterator.next / get   R/bloom/hash/CanonicalStrobeHashIterator.java:79-140
 *   Strobe3HashIterator.next / get           R/bloom/hash/Strobe3HashIterator.java:78-151
 *   CanonicalStrobe3HashIterator.next / get  R/bloom/hash/CanonicalStrobe3HashIterator.java:85-225
 *   MinimizerHashIterator.nextMinimizer      R/bloom/hash/MinimizerHashIterator.java:97-112
 *   GraphUtils.getMinimizersSet / getMinimizers   R/util/GraphUtils.java:2462-2549
 *   SeqSubsampler.kmerBased pair hashes      R/util/SeqSubsampler.java:176-179, 266-268
 * Parity unpinned by reference-run outputs (no JVM in the image): pinned by reading, by the identities in
 * tests/test_oracle_sketch.py and by the inputs of the classes' own main() methods. */
#include <stdlib.h>
#include <string.h>

#include "rb_oracle.h"

/* Long.compareUnsigned(a, b) */
static int cmpu(uint64_t a, uint64_t b) { return a < b ? -1 : (a > b ? 1 : 0); }

/* flags: bit 0 canonical (CanonicalStrobeHashIterator), bit 1 slide across equal k-mer hashes
 * (StrobeHashIterator.get / getInterval :96-164; next() :73-94 and both canonical forms do not slide).
 * out_pos: n positions per strobemer (positions[0] = p, then the strobes).  Returns the number of strobemers. */
int64_t rbo_randstrobes(const char *seq, int64_t len, int k, int n, int wmin, int wmax, int flags,
                        uint64_t *out_hash, int32_t *out_pos) {
    const int canonical = flags & 1, slide = (flags & 2) && !canonical;
    if (len < k) return 0;                                    /* itr.start(seq) fails */
    int64_t nk = len - k + 1;
    if (!(nk > (int64_t)wmax * (n - 1))) return 0;            /* :54 */
    int64_t max = nk - (int64_t)wmax * (n - 2) - wmin - 1;    /* :56 */
    uint64_t *f = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk), *r = NULL;
    if (canonical) {
        uint64_t *c = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk), *fr = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)nk);
        r = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk);
        rbo_hash_region(seq, 0, len, k, 1, RBO_CANON, c, fr);  /* itr.frhval[0], itr.frhval[1] */
        for (int64_t i = 0; i < nk; ++i) { f[i] = fr[2 * i]; r[i] = fr[2 * i + 1]; }
        free(c); free(fr);
    } else
        rbo_hash_region(seq, 0, len, k, 1, RBO_FWD, f, NULL);
    int32_t *positions = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int64_t p = 0; p <= max; ++p) {
        uint64_t sh = f[p];
        positions[0] = (int32_t)p;
        for (int s = 0; s < n - 1; ++s) {
            int64_t pos2 = p + (int64_t)s * wmax + wmin;
            uint64_t pos2k = f[pos2];
            uint64_t h = rbo_combine(sh, pos2k);
            int64_t end = p + (int64_t)s * wmax + wmax;
            if (end > nk) end = nk;
            for (int64_t i = pos2 + 1; i < end; ++i) {
                uint64_t alt = f[i];
                if (slide && alt == pos2k) pos2 = i;
                else {
                    uint64_t h2 = rbo_combine(sh, alt);
                    if (cmpu(h, h2) >= 0) { pos2 = i; pos2k = alt; h = h2; }
                }
            }
            sh = h;
            positions[s + 1] = (int32_t)pos2;
        }
        if (canonical) {                                       /* :99-107 / :131-136 */
            uint64_t rs = r[positions[n - 1]];
            for (int s = n - 2; s >= 0; --s) rs = rbo_combine(r[positions[s]], rs);
            sh = ((int64_t)rs < (int64_t)sh) ? rs : sh;        /* Math.min(long,long): signed */
        }
        out_hash[p] = sh;
        if (out_pos) memcpy(out_pos + (size_t)p * (size_t)n, positions, sizeof(int32_t) * (size_t)n);
    }
    free(positions); free(f); free(r);
    return max + 1;
}

/* Strobe3HashIterator / CanonicalStrobe3HashIterator: strobemer p in [min, max], out index p - min;
 * out_pos = {pos1, p, pos3}.  Returns max + 1 - min (never negative here: 0 when there is none). */
int64_t rbo_strobe3(const char *seq, int64_t len, int k, int wmin, int wmax, int canonical, uint64_t *out_hash, int32_t *out_pos) {
    if (len < k) return 0;
    int64_t nk = len - k + 1;
    if (!(nk > (int64_t)wmin * 2)) return 0;                  /* :54 / :60 */
    const int64_t min = canonical ? wmax : wmin, max = canonical ? nk - 1 - wmax : nk - 1 - wmin;
    if (max < min) return 0;
    uint64_t *f = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk), *r = NULL;
    if (canonical) {
        uint64_t *c = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk), *fr = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)nk);
        r = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk);
        rbo_hash_region(seq, 0, len, k, 1, RBO_CANON, c, fr);
        for (int64_t i = 0; i < nk; ++i) { f[i] = fr[2 * i]; r[i] = fr[2 * i + 1]; }
        free(c); free(fr);
    } else
        rbo_hash_region(seq, 0, len, k, 1, RBO_FWD, f, NULL);
    for (int64_t pos = min; pos <= max; ++pos) {
        const uint64_t fk = f[pos];
        /* upstream strobe */
        int64_t p1 = pos - wmax + 1 > 0 ? pos - wmax + 1 : 0;
        uint64_t h1 = rbo_combine(f[p1], fk);
        int64_t end = pos - wmin + 1;
        for (int64_t i = p1 + 1; i < end; ++i) { uint64_t h = rbo_combine(f[i], fk); if (cmpu(h1, h) > 0) { p1 = i; h1 = h; } }
        /* downstream strobe */
        int64_t p3 = pos + wmin;
        uint64_t h3 = rbo_combine(h1, f[p3]);
        end = pos + wmax < nk ? pos + wmax : nk;
        for (int64_t i = p3 + 1; i < end; ++i) {
            uint64_t h = rbo_combine(h1, f[i]);
            if (canonical ? cmpu(h3, h) >= 0 : cmpu(h3, h) > 0) { p3 = i; h3 = h; }    /* canonical: >= (:113), plain: > (:103) */
        }
        uint64_t hv = h3; int64_t o1 = p1, o3 = p3;
        if (canonical) {
            const uint64_t rk = r[pos];
            int64_t q3 = pos + wmin;                                                   /* reverse strand: downstream first :123-132 */
            uint64_t rh3 = rbo_combine(r[q3], rk);
            int64_t rend = pos + wmax < nk ? pos + wmax : nk;
            for (int64_t i = q3 + 1; i < rend; ++i) { uint64_t h = rbo_combine(r[i], rk); if (cmpu(rh3, h) >= 0) { q3 = i; rh3 = h; } }
            int64_t q1 = pos - wmax + 1 > 0 ? pos - wmax + 1 : 0;                      /* upstream :135-144 */
            uint64_t rh1 = rbo_combine(rh3, r[q1]);
            rend = pos - wmin + 1;
            for (int64_t i = q1 + 1; i < rend; ++i) { uint64_t h = rbo_combine(rh3, r[i]); if (cmpu(rh1, h) > 0) { q1 = i; rh1 = h; } }
            if (cmpu(h3, rh1) > 0) { hv = rh1; o1 = q1; o3 = q3; }                     /* :146-150 */
        }
        out_hash[pos - min] = hv;
        if (out_pos) { int32_t *o = out_pos + 3 * (size_t)(pos - min); o[0] = (int32_t)o1; o[1] = (int32_t)pos; o[2] = (int32_t)o3; }
    }
    free(f); free(r);
    return max + 1 - min;
}

/* The sequence MinimizerHashIterator.nextMinimizer() returns while hasNext() (R/bloom/hash/MinimizerHashIterator.java:97-112):
 * the first window's minimizer, then one entry every time the position of the window minimum moves right. */
int64_t rbo_minimizers_next(const char *seq, int64_t len, int k, int w, int mode, uint64_t *out_hash, int64_t *out_pos) {
    if (len < k) return 0;
    int64_t nk = len - k + 1, max = nk - w + 1;
    if (max <= 0) return 0;
    uint64_t *wh = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)max);
    int64_t *wp = (int64_t *)malloc(sizeof(int64_t) * (size_t)max);
    rbo_minimizers(seq, len, k, w, mode, wh, wp);              /* window.getMin() / getMinPos() after every roll */
    int64_t n = 0, pos = -1, min_pos = 0;
    /* first call: pos < 0 -> next() */
    pos = 0; out_hash[n] = wh[0]; out_pos[n] = wp[0]; ++n;
    while (pos < max) {                                        /* caller loop: while (itr.hasNext()) itr.nextMinimizer() */
        min_pos = wp[pos];
        int found = 0;
        while (++pos < max) {
            if (wp[pos] > min_pos) { out_hash[n] = wh[pos]; out_pos[n] = wp[pos]; ++n; found = 1; break; }
        }
        if (!found) break;                                     /* returns prev again: not a new minimizer */
    }
    free(wh); free(wp);
    return n;
}

static int cmp_i64(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? -1 : (x > y ? 1 : 0); }

/* GraphUtils.getMinimizers (sorted, signed) of GraphUtils.getMinimizersSet, R/util/GraphUtils.java:2462-2549, with
 * numKmers = seq.length - k + 1 and `stale` = hvals[0] before the first next() (quirk: for numKmers <= windowSize the
 * minimum is seeded from it, :2480-2488; 0 on a fresh iterator).  mode selects the iterator class passed in. */
int64_t rbo_minimizer_set(const char *seq, int64_t len, int k, int w, int mode, uint64_t stale, uint64_t *out) {
    int64_t nk = len >= k ? len - k + 1 : 0;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nk > 0 ? nk : 1));
    if (nk > 0) rbo_hash_region(seq, 0, len, k, 1, mode, h, NULL);
    int64_t n = 0;
    if (len - k + 1 <= w) {                                    /* numKmers <= windowSize (numKmers may be <= 0) */
        int64_t m = (int64_t)stale;
        for (int64_t i = 0; i < nk; ++i) if ((int64_t)h[i] < m) m = (int64_t)h[i];
        out[n++] = (uint64_t)m;
        free(h);
        return n;
    }
    int64_t *set = (int64_t *)malloc(sizeof(int64_t) * (size_t)nk);
    int64_t minimizer = (int64_t)h[0];
    int mpos = 0;
    for (int i = 1; i < w; ++i) if ((int64_t)h[i] < minimizer) { minimizer = (int64_t)h[i]; mpos = i; }
    set[n++] = minimizer;
    for (int64_t t = w; t < nk; ++t) {                         /* window = h[t-w+1 .. t] */
        int64_t hv = (int64_t)h[t];
        if (--mpos < 0) {
            const uint64_t *win = h + (t - w + 1);
            minimizer = (int64_t)win[0]; mpos = 0;
            for (int i = 1; i < w; ++i) if ((int64_t)win[i] < minimizer) { minimizer = (int64_t)win[i]; mpos = i; }
            set[n++] = minimizer;
        } else if (hv < minimizer) { minimizer = hv; mpos = w - 1; set[n++] = minimizer; }
    }
    qsort(set, (size_t)n, sizeof(int64_t), cmp_i64);           /* HashSet -> Arrays.sort */
    int64_t u = 0;
    for (int64_t i = 0; i < n; ++i) if (i == 0 || set[i] != set[i - 1]) out[u++] = (uint64_t)set[i];
    free(set); free(h);
    return u;
}

/* SeqSubsampler.kmerBased: pair hash of k-mers i and i + shift, i in [0, numKmers - shift): stranded
 * combine(h[i], h[i+shift]) (:176-179); else min_signed(combine(f[i], f[i+shift]), combine(r[i+shift], r[i])) (:266-268) */
int64_t rbo_kmer_pair_hashes(const char *seq, int64_t len, int k, int shift, int canonical, uint64_t *out) {
    if (len < k) return 0;
    int64_t nk = len - k + 1, np = nk - shift;
    if (np <= 0) return 0;
    uint64_t *c = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk), *fr = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)nk);
    rbo_hash_region(seq, 0, len, k, 1, RBO_CANON, c, fr);
    for (int64_t i = 0; i < np; ++i) {
        uint64_t pf = rbo_combine(fr[2 * i], fr[2 * (i + shift)]);
        if (canonical) {
            uint64_t pr = rbo_combine(fr[2 * (i + shift) + 1], fr[2 * i + 1]);
            out[i] = ((int64_t)pr < (int64_t)pf) ? pr : pf;
        } else out[i] = pf;
    }
    free(c); free(fr);
    return np;
}
