// rb_graph.hip — the Bloom-filter de Bruijn graph on one MI355X: filters resident in HBM and the
// order-exact batched insert pipeline (kernels of stages A / B, the conflict path, the paired-k-mer
// walkers; run_core / add_range on two streams).  The entry points around it are rb_capi.hip, the
// batched queries and traversals rb_query.hip (one translation unit until round 5).
//
// Insert pipeline ("sorted-batch engine"): the sequential semantics of
//   for each read, for each k-mer left to right:  if (dbgbf.lookupThenAdd(h)) cbf.increment(h)
// (R/graph/BloomFilterDeBruijnGraph.java:405-412 driven by R/RNABloom.java:551-634) are reproduced
// exactly by
//   1. hashing every usable window into (h0, occurrence-id) records in read order,
//   2. a stable sort by h0 — every distinct k-mer becomes one run whose occurrences stay in order,
//   3. per DISTINCT k-mer: Bloom-bit test against the pre-batch state, first-setter arbitration
//      between new k-mers that share a bit (smallest probe id wins, as it would sequentially),
//      bit set with atomicOr, and the count of occurrences whose lookupThenAdd returned true,
//   4. counter-claim detection of k-mers that share a counting-Bloom byte with another k-mer of the
//      batch; k-mers without such a neighbour commute and are applied independently (all their
//      increments simulated in registers, in occurrence order), the rest are replayed in global
//      occurrence order.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <new>
#include <vector>

#include <functional>
#include <string>
#include <thread>

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include "rb_pipeline.hpp"

using namespace rb;

namespace rb {
static thread_local char g_err[768] = "";
const char *last_error_text() { return g_err; }
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace rb

namespace {

// ---- stage A: per distinct run — test Bloom bits against the pre-batch state, register the run's
// first probe ids for bits it may be the first to set, claim its counters (reading them) ----
__global__ void k_probe(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ counts,
                        const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals, uint32_t n_distinct,
                        int mode, Slot *ftable /* null: arbitration by k_set_bits + collision table */, uint32_t f_log2,
                        uint32_t *__restrict__ status, uint64_t *__restrict__ cvals, uint64_t *__restrict__ foreign_idx,
                        uint32_t *__restrict__ counters /* [5] = number of foreign claims */) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    const uint64_t h0 = uniq[d];
    uint32_t premask = 0, all = 1;
    if (mode != M_COUNT_ONLY) {
        uint64_t idx[RB_MAX_HASH];
        for (int j = 0; j < fv.dbg_h; ++j) {
            idx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
            if (bit_test(fv.dbg, idx[j])) premask |= 1u << j; else all = 0;
        }
        if (ftable && !all && (mode == M_ADD || mode == M_ADD_IF_ABSENT)) {
            // sequentially, the getAndSet with the smallest (occurrence, probe) id is the one that
            // finds the bit clear (R/bloom/BloomFilter.java:147-155)
            const unsigned long long v_first = vals[starts[d]];
            for (int j = 0; j < fv.dbg_h; ++j)
                if (!((premask >> j) & 1u)) {
                    Slot *s = table_insert(ftable, f_log2, idx[j]);
                    atomicMin(&s->val, (v_first << 4) | (unsigned long long)j);
                }
        }
    }
    uint32_t st = premask | (all ? ST_ALLPRE : 0u);
    uint32_t n_foreign = 0;
    uint64_t fidx[RB_MAX_HASH];                       // contested counters of this run (written only if there are any)
    for (int j = 0; j < fv.cbf_h; ++j) fidx[j] = ~0ull;
    // can this run have counting-Bloom ops?  (exact number is known in stage B)
    bool may_count = true;
    if (mode == M_COUNT_IF_PRESENT) may_count = all;
    if (mode == M_ADD && !all && counts[d] == 1u) {   // first sighting, seen once: ops = 0 unless found by arbitration
        may_count = false;                            // (sequencing-error k-mers: most runs of a real data set)
        st |= ST_LATE;
    }
    if (may_count) {
        uint64_t cidx[RB_MAX_HASH], cv = 0;
        for (int j = 0; j < fv.cbf_h; ++j) {
            cidx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
            int dup = -1;
            for (int q = 0; q < j; ++q) if (cidx[q] == cidx[j]) dup = q;
            uint32_t byte;
            if (dup >= 0) byte = (uint32_t)(cv >> (8 * dup)) & 0xFFu;
            else {
                byte = cbf_claim(fv.cbf, cidx[j]);
                if (byte & CLAIM) {                     // somebody else of this sub-batch owns it too
                    st |= ST_FOREIGN;
                    fidx[j] = cidx[j];
                    ++n_foreign;
                    byte &= 0x7Fu;
                }
            }
            cv |= (uint64_t)byte << (8 * j);
        }
        cvals[d] = cv;
        st |= ST_CLAIMED;
    }
    status[d] = st;
    if (n_foreign) {
        for (int j = 0; j < fv.cbf_h; ++j) foreign_idx[(size_t)d * fv.cbf_h + j] = fidx[j];
        atomicAdd(&counters[16 + 16 * (blockIdx.x & 31u)], n_foreign);   // 32 spread counters
    }
}
constexpr uint32_t ST_COLLIDE_SHIFT = 21;   // status bits 21..28: probe j found its bit set by another probe of the sub-batch
// after the swept Bloom-bit stage (two hash functions): bits 29..30 = probe j set its bit and another probe may have met it there, and
// ST_SWEPT_KNOWN = a probe with neither mark had its bit to itself — the arbitration (k_late_claim, stage B) need not look it up.
// (All of these live until stage B rewrites the status word.)
constexpr uint32_t ST_JOIN_SHIFT = 29, ST_SWEPT_KNOWN = 1u << 31;
__device__ __forceinline__ bool st_alone(uint32_t st, int j) {
    return (st & ST_SWEPT_KNOWN) && !(((st >> (ST_COLLIDE_SHIFT + j)) | (st >> (ST_JOIN_SHIFT + j))) & 1u);
}
// The same stage for the common filter shape (2 hash functions each, no full first-setter table), written for memory-level
// parallelism: a lane takes RUNS runs, computes all their indices, issues ALL Bloom-bit loads, then ALL counter claims
// (returning atomics), and only then consumes the answers — 2 dependent round trips per lane instead of 4 per run
// (k_probe spent 71 % of its wave cycles waiting, profiles/r01_sq_counters).  Semantics are k_probe's, line for line.
// SWEPT: the Bloom bits were tested AND set by the swept stage (rb_group.hip sweep_bits_device); st0 / st1 hold what probe 0 / 1 of each run
// found (0 set by this probe, 1 set before the sub-batch, 2 set by another probe of the sub-batch) and this kernel only assembles the
// status words (k_set_bits' collision marks and count included) and claims the counters.
template <int RUNS, bool SWEPT>
__global__ void __launch_bounds__(256) k_probe_h2(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ counts,
                                                   uint32_t n_distinct, int mode, uint32_t *__restrict__ status, uint64_t *__restrict__ cvals,
                                                   uint64_t *__restrict__ foreign_idx, uint32_t *__restrict__ counters,
                                                   const uint8_t *__restrict__ st0 = nullptr, const uint8_t *__restrict__ st1 = nullptr,
                                                   uint32_t mark_known = 0u /* every probe of the sub-batch went through the sweep */,
                                                   uint32_t plain_ok = 0u /* ... and a counter has its Bloom bit's index: an unmet run reads its counters without claiming */) {
    const uint32_t d0 = (blockIdx.x * blockDim.x + threadIdx.x) * RUNS;
    uint64_t h0[RUNS], bi[RUNS][2], ci[RUNS][2];
    uint32_t cnt[RUNS], w[RUNS][2];
    bool live[RUNS];
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
        live[r] = d0 + r < n_distinct;
        h0[r] = (live[r] && !SWEPT) ? uniq[d0 + r] : 0ull;     // (swept: only the runs that claim need their hash, below)
        cnt[r] = live[r] ? counts[d0 + r] : 0u;
    }
    if (!SWEPT) {
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {
            const uint64_t h1 = multi_hash(h0[r], 1u, fv.kmul);
            bi[r][0] = index_of(h0[r], fv.dbg_mod); bi[r][1] = index_of(h1, fv.dbg_mod);
            ci[r][0] = index_of(h0[r], fv.cbf_mod); ci[r][1] = index_of(h1, fv.cbf_mod);
        }
    }
    if (SWEPT) {
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {
            w[r][0] = live[r] ? (uint32_t)st0[d0 + r] : 1u;
            w[r][1] = live[r] ? (uint32_t)st1[d0 + r] : 1u;
        }
    } else if (mode != M_COUNT_ONLY) {
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {                      // all bit loads in flight together
            w[r][0] = live[r] ? fv.dbg[bi[r][0] >> 5] : 0u;
            w[r][1] = live[r] ? fv.dbg[bi[r][1] >> 5] : 0u;
        }
    }
    uint32_t st[RUNS];
    bool claim[RUNS], dup[RUNS], plain[RUNS];
    uint32_t n_coll_total = 0, n_allpre = 0;
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
        uint32_t premask = 0, all = 1, coll = 0;
        plain[r] = false;
        if (SWEPT) {
            // (report 4 = set before the sub-batch like 1, and another probe of the sub-batch asked for the same bit: k_sweep_bits)
            premask = ((w[r][0] == 1u || w[r][0] == 4u) ? 1u : 0u) | ((w[r][1] == 1u || w[r][1] == 4u) ? 2u : 0u);
            // Counters without claims.  With filters of equal size probe j of the counting filter has probe j's Bloom-bit index, and the sweep
            // has seen every probe of the sub-batch: a run whose two probes met NO other probe (reports 0 / 1) touches counters no other run of
            // the sub-batch touches — neither a claim mark nor the contested-counter machinery has anything to find, so it reads its counters
            // with plain loads (48.8 G/s) instead of two returning atomics (17.6 G/s).  Every run that did meet somebody (2, 3, 4 — marks go by
            // the low bits of the index, so a few more than really did) claims as before, and so do all when the sweep missed a probe.
            plain[r] = plain_ok && w[r][0] < 2u && w[r][1] < 2u;
            all = premask == 3u;
            coll = (w[r][0] == 2u ? 1u : 0u) | (w[r][1] == 2u ? 2u : 0u);
            const uint32_t join = (w[r][0] == 3u ? 1u : 0u) | (w[r][1] == 3u ? 2u : 0u);
            n_coll_total += (uint32_t)__popc(coll | (join << 2));
            if (live[r]) coll |= (join << (ST_JOIN_SHIFT - ST_COLLIDE_SHIFT)) | (mark_known ? (ST_SWEPT_KNOWN >> ST_COLLIDE_SHIFT) : 0u);
        } else if (mode != M_COUNT_ONLY) {
            if ((w[r][0] >> (uint32_t)(bi[r][0] & 31u)) & 1u) premask |= 1u; else all = 0;
            if ((w[r][1] >> (uint32_t)(bi[r][1] & 31u)) & 1u) premask |= 2u; else all = 0;
        }
        st[r] = premask | (all ? ST_ALLPRE : 0u) | (coll << ST_COLLIDE_SHIFT);
        n_allpre += (live[r] && all) ? 1u : 0u;
        bool may_count = true;
        if (mode == M_COUNT_IF_PRESENT) may_count = all;
        if (mode == M_ADD && !all && cnt[r] == 1u) { may_count = false; st[r] |= ST_LATE; }
        claim[r] = live[r] && may_count;
        if (SWEPT) {                                           // (most runs of the regime the sweep is for never claim: indices only where needed)
            ci[r][0] = ci[r][1] = 0;
            if (claim[r]) { h0[r] = uniq[d0 + r]; ci[r][0] = index_of(h0[r], fv.cbf_mod); ci[r][1] = index_of(multi_hash(h0[r], 1u, fv.kmul), fv.cbf_mod); }
        }
        dup[r] = ci[r][0] == ci[r][1];
        plain[r] = plain[r] && !dup[r];                          // (a run whose two probes are one counter met itself on the bit; belt and braces)
    }
    uint32_t b0[RUNS], b1[RUNS];
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {                          // all claims (or loads) in flight together
        b0[r] = !claim[r] ? 0u : plain[r] ? (uint32_t)fv.cbf[ci[r][0]] : cbf_claim(fv.cbf, ci[r][0]);
        b1[r] = !(claim[r] && !dup[r]) ? 0u : plain[r] ? (uint32_t)fv.cbf[ci[r][1]] : cbf_claim(fv.cbf, ci[r][1]);
    }
    uint32_t n_foreign_total = 0;
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
        if (!live[r]) continue;
        const uint32_t d = d0 + r;
        if (claim[r]) {
            uint64_t f0 = ~0ull, f1 = ~0ull;
            uint32_t nf = 0, v0 = b0[r], v1;
            if (v0 & CLAIM) { st[r] |= ST_FOREIGN; f0 = ci[r][0]; ++nf; v0 &= 0x7Fu; }
            if (dup[r]) v1 = v0;                                 // the same counter twice: one claim, the value mirrored
            else { v1 = b1[r]; if (v1 & CLAIM) { st[r] |= ST_FOREIGN; f1 = ci[r][1]; ++nf; v1 &= 0x7Fu; } }
            cvals[d] = (uint64_t)v0 | ((uint64_t)v1 << 8);
            st[r] |= ST_CLAIMED;
            if (nf) { foreign_idx[(size_t)d * 2] = f0; foreign_idx[(size_t)d * 2 + 1] = f1; n_foreign_total += nf; }
        }
        status[d] = st[r];
    }
    // one add per WORKGROUP and counter, not per lane: adds to one address go through at ~10 ns apiece whoever issues them, and there are 32
    // addresses per counter — with one add per wavefront a 390 M-run sub-batch still queued 0.57 M adds per address (6 ms of a 9 ms kernel)
    __shared__ uint32_t s_acc[3];
    if (threadIdx.x < 3u) s_acc[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int o = 32; o; o >>= 1) { n_foreign_total += __shfl_xor(n_foreign_total, o, 64); n_coll_total += __shfl_xor(n_coll_total, o, 64); n_allpre += __shfl_xor(n_allpre, o, 64); }
    if ((threadIdx.x & 63u) == 0u) {
        if (n_foreign_total) atomicAdd(&s_acc[0], n_foreign_total);
        if (n_coll_total) atomicAdd(&s_acc[1], n_coll_total);
        if (n_allpre) atomicAdd(&s_acc[2], n_allpre);
    }
    __syncthreads();
    if (threadIdx.x < 3u && s_acc[threadIdx.x] && (SWEPT || threadIdx.x != 1u))          // [16] contested claims, [17] collisions (swept stage only: k_set_bits counts them otherwise), [18] runs present before the sub-batch
        atomicAdd(&counters[16 + threadIdx.x + 16 * (blockIdx.x & 31u)], s_acc[threadIdx.x]);
}
// ---- first-setter arbitration without a table entry per new bit (the default) ----
// Two new k-mers of one sub-batch rarely share a Bloom bit (touches^2 / 2 bits: ~0.2 M of 60 M on config 2), so
// instead of registering every missing bit in a hash table the size of the sub-batch, the bits are set right after
// stage A with a returning atomicOr: a probe that finds the bit set although it was clear before the sub-batch
// (premask) COLLIDES with another probe of the sub-batch.  Only colliding bits get a table entry (min probe id over
// the colliders, then over the probe that happened to get there first — it learns about the collision from the
// table); a bit without an entry was touched by one probe only, which therefore found it clear.
// ff: bit filter in front of the collision table (k_collide_insert sets a bit per entry; nullptr: none).  The table is a few MB on
// config 2 (cache-resident: the filter measured neutral there), but where most k-mers of a sub-batch are NEW — long reads — it holds
// millions of entries and every run looks it up twice in k_collide_fixup and again in k_late_claim / stage B: 97 + 57 ms of a 0.9 s
// pass (profiles/r04_longreads.txt); the filter answers "no entry" from L2 for all but the collided bits.
struct FtFilter { const uint32_t *bits; uint32_t log2; };
__device__ __forceinline__ bool ft_maybe(const FtFilter &ff, uint64_t idx) {
    if (!ff.bits) return true;
    const uint64_t b = slot_of(idx ^ 0x5851F42D4C957F2Dull, ff.log2);
    return (ff.bits[b >> 5] >> (uint32_t)(b & 31u)) & 1u;
}
__global__ void k_set_bits(FilterView fv, const uint64_t *__restrict__ uniq, uint32_t n_distinct, uint32_t *__restrict__ status,
                           uint32_t *__restrict__ counters) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    const uint32_t st = status[d];
    if (st & ST_ALLPRE) return;
    const uint64_t h0 = uniq[d];
    uint32_t coll = 0;
    for (int j = 0; j < fv.dbg_h; ++j) {
        if ((st >> j) & 1u) continue;
        const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
        const uint32_t m = 1u << (uint32_t)(idx & 31u);
        if (atomicOr(&fv.dbg[idx >> 5], m) & m) coll |= 1u << j;
    }
    if (coll) {
        status[d] = st | (coll << ST_COLLIDE_SHIFT);
        atomicAdd(&counters[17 + 16 * (blockIdx.x & 31u)], (uint32_t)__popc(coll));
    }
}
// colliders register their probe ids ...
__global__ void k_collide_insert(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts,
                                 const uint32_t *__restrict__ vals, uint32_t n_distinct, const uint32_t *__restrict__ status,
                                 Slot *ftable, uint32_t f_log2, uint32_t *__restrict__ ffbits, uint32_t ff_log2,
                                 const uint32_t *__restrict__ list = nullptr /* the runs to look at, *list_n of them (at most n_distinct), instead of all */,
                                 const uint32_t *__restrict__ list_n = nullptr) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    if (list) { if (d >= *list_n) return; d = list[d]; }
    const uint32_t st_d = status[d];
    const uint32_t coll = ((st_d >> ST_COLLIDE_SHIFT) & 0xFFu) | ((st_d >> ST_JOIN_SHIFT) & 3u);     // (swept stage: the probe that set the bit joins here, no k_collide_fixup)
    if (!coll) return;
    const uint64_t h0 = uniq[d];
    const unsigned long long v_first = vals[starts[d]];
    for (int j = 0; j < fv.dbg_h; ++j)
        if ((coll >> j) & 1u) {
            const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
            Slot *s = table_insert(ftable, f_log2, idx);
            atomicMin(&s->val, (v_first << 4) | (unsigned long long)j);
            if (ffbits) { const uint64_t b = slot_of(idx ^ 0x5851F42D4C957F2Dull, ff_log2); atomicOr(&ffbits[b >> 5], 1u << (uint32_t)(b & 31u)); }
        }
}
// ... and the probes whose atomicOr got there first join the entries of the bits somebody collided on
__global__ void k_collide_fixup(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts,
                                const uint32_t *__restrict__ vals, uint32_t n_distinct, const uint32_t *__restrict__ status,
                                Slot *ftable, uint32_t f_log2, FtFilter ff) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    const uint32_t st = status[d];
    if (st & ST_ALLPRE) return;
    const uint32_t mine = ~(st | (st >> ST_COLLIDE_SHIFT)) & ((1u << fv.dbg_h) - 1u);   // clear before, and set by me
    if (!mine) return;
    const uint64_t h0 = uniq[d];
    unsigned long long v_first = ~0ull;
    for (int j = 0; j < fv.dbg_h; ++j)
        if ((mine >> j) & 1u) {
            const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
            if (!ft_maybe(ff, idx)) continue;
            Slot *s = const_cast<Slot *>(table_find(ftable, f_log2, idx));
            if (!s) continue;
            if (v_first == ~0ull) v_first = vals[starts[d]];
            atomicMin(&s->val, (v_first << 4) | (unsigned long long)j);
        }
}
// did an earlier probe of the sub-batch set bit idx?  (table of all missing bits, or of the collided ones only)
__device__ __forceinline__ bool set_earlier(const Slot *ftable, uint32_t f_log2, uint64_t idx, unsigned long long id, FtFilter ff = FtFilter{nullptr, 0}) {
    if (!ftable) return false;                    // no collision in this sub-batch at all
    if (!ft_maybe(ff, idx)) return false;
    const Slot *s = table_find(ftable, f_log2, idx);
    return s && s->val < id;
}
// between stage A and B: the single-occurrence runs of new k-mers learn from the first-setter table whether
// their one occurrence counts after all (every missing bit was set by an EARLIER probe of the sub-batch);
// only those claim their counters — the others never touch the counting filter
__global__ void k_late_claim(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts,
                             const uint32_t *__restrict__ vals, uint32_t n_distinct, const Slot *ftable, uint32_t f_log2,
                             uint32_t *__restrict__ status, uint64_t *__restrict__ cvals, uint64_t *__restrict__ foreign_idx,
                             uint32_t *__restrict__ counters, FtFilter ff,
                             const uint32_t *__restrict__ list = nullptr /* as in k_collide_insert */, const uint32_t *__restrict__ list_n = nullptr) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    if (list) { if (d >= *list_n) return; d = list[d]; }
    uint32_t st = status[d];
    if (!(st & ST_LATE)) return;
    for (int j = 0; j < fv.dbg_h; ++j)
        if (!((st >> j) & 1u) && st_alone(st, j)) return;      // (swept stage: this probe had its bit to itself — nobody set it earlier)
    const uint64_t h0 = uniq[d];
    const unsigned long long v_first = vals[starts[d]];
    bool found = true;
    for (int j = 0; j < fv.dbg_h && found; ++j) {
        if ((st >> j) & 1u) continue;
        if (st_alone(st, j) || !set_earlier(ftable, f_log2, index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod), (v_first << 4) | (unsigned long long)j, ff))
            found = false;
    }
    if (!found) return;
    st |= ST_LATE_FOUND | ST_CLAIMED;
    uint64_t cidx[RB_MAX_HASH], fidx[RB_MAX_HASH], cv = 0;
    uint32_t n_foreign = 0;
    for (int j = 0; j < fv.cbf_h; ++j) {
        fidx[j] = ~0ull;
        cidx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        int dup = -1;
        for (int q = 0; q < j; ++q) if (cidx[q] == cidx[j]) dup = q;
        uint32_t byte;
        if (dup >= 0) byte = (uint32_t)(cv >> (8 * dup)) & 0xFFu;
        else {
            byte = cbf_claim(fv.cbf, cidx[j]);
            if (byte & CLAIM) { st |= ST_FOREIGN; fidx[j] = cidx[j]; ++n_foreign; byte &= 0x7Fu; }
        }
        cv |= (uint64_t)byte << (8 * j);
    }
    cvals[d] = cv;
    status[d] = st;
    if (n_foreign) {
        for (int j = 0; j < fv.cbf_h; ++j) foreign_idx[(size_t)d * fv.cbf_h + j] = fidx[j];
        atomicAdd(&counters[16 + 16 * (blockIdx.x & 31u)], n_foreign);
    }
}
__global__ void k_cs_build(const uint32_t *__restrict__ status, const uint64_t *__restrict__ foreign_idx, uint32_t n_distinct, int h,
                           Slot *cs, uint32_t cs_log2, uint32_t *__restrict__ csf, uint32_t csf_log2) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct || !(status[d] & ST_FOREIGN)) return;       // only these runs have written their entries
    for (int j = 0; j < h; ++j) {
        const uint64_t idx = foreign_idx[(size_t)d * h + j];
        if (idx == ~0ull) continue;
        table_insert(cs, cs_log2, idx);
        const uint64_t b = slot_of(idx, csf_log2);                  // cache-resident bit filter in front of the table (stage B)
        atomicOr(&csf[b >> 5], 1u << (uint32_t)(b & 31u));
    }
}

// Which shared counters actually order anything?  A run's ops only ever raise its counters, and every success raises its
// minimum by exactly one; as long as nobody else writes the counters it works on, its minimum after all its m occurrences is at
// most m0 + m, so it writes a counter — and its behaviour depends on that counter's exact value — only if the counter's pre-batch
// value is within reach: c <= m0 + m - 1.  (m = occurrences in the run, an upper bound of its ops.)  The runs that need the
// ordered replay are the least set O with
//   X in O  if X can reach a shared counter that another run can reach too            (writers >= 2), or
//              X can reach a shared counter that a run of O claimed                     (that run's bound no longer holds:
//                                                                                         others raise its minimum)
// Every run outside O then works on counters nobody else writes (so its bound holds), and never writes a counter a run of O
// claimed; the runs of O see pre-batch values plus each other's writes, in occurrence order: the sequential result.  A shared
// counter a run cannot reach is left to whoever can (only the claim mark is dropped).  On config 2 two thirds of the ops that
// used to be replayed are not in O.
struct CsLookup {
    const Slot *cs; uint32_t cs_log2; const uint32_t *csf; uint32_t csf_log2;
    __device__ __forceinline__ const Slot *find(uint64_t idx) const {
        const uint64_t b = slot_of(idx, csf_log2);                  // cache-resident bit filter in front of the table
        if (csf && !((csf[b >> 5] >> (uint32_t)(b & 31u)) & 1u)) return nullptr;
        return table_find(cs, cs_log2, idx);
    }
};
// Stage B sets every run with a shared counter aside (they are few: the list `cand`); these three kernels decide which of them
// need the ordered replay and finish the others.  All of it runs behind the point where the producer is released.
// (1) per shared counter: how many candidates can reach it.  The table lookups are done HERE, once: cslot[i * h + j] = the slot of candidate i's
// counter j in the set (CS_NONE: not shared), bit 31 = the candidate can reach it — the closure rounds, the deferred resolve and (after the
// compaction) the component labelling index the table directly instead of hashing and probing again in every round (long reads, every k-mer
// re-sighted: 5 M candidates per sub-batch, 7 labelling rounds and 6 closure rounds of two lookups each were 170 of a pass's 1060 ms)
constexpr uint32_t CS_NONE = 0xFFFFFFFFu, CS_REACH = 0x80000000u;
__global__ void k_cs_writers(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ nops, const uint32_t *__restrict__ cand,
                             uint32_t n_cand, const uint64_t *__restrict__ cvals, CsLookup L, uint32_t *__restrict__ writers, uint32_t *__restrict__ cslot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    const uint32_t d = cand[i];
    const uint64_t cv0 = cvals[d], h0 = uniq[d];
    uint32_t m0 = 255;
    for (int j = 0; j < fv.cbf_h; ++j) { const uint32_t c = (uint32_t)(cv0 >> (8 * j)) & 0xFFu; m0 = c < m0 ? c : m0; }
    const uint32_t reach = m0 + nops[d] - 1u;
    uint64_t seen[RB_MAX_HASH];
    for (int j = 0; j < fv.cbf_h; ++j) {
        const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        bool dup = false;
        for (int q = 0; q < j; ++q) dup |= seen[q] == idx;
        seen[j] = idx;
        const Slot *sl = L.find(idx);
        const bool can = ((uint32_t)(cv0 >> (8 * j)) & 0xFFu) <= reach;
        cslot[(size_t)i * fv.cbf_h + j] = sl ? ((uint32_t)(sl - L.cs) | (can ? CS_REACH : 0u)) : CS_NONE;
        if (sl && can && !dup) atomicAdd(&writers[sl - L.cs], 1u);           // (the same counter twice: one run, one writer)
    }
}
// (2) repeated until nothing changes: the closure.  cflag[slot] = a run of O claimed the counter; ordered[i] = cand[i] is in O.
__global__ void k_cs_order(int h, uint32_t n_cand, const uint32_t *__restrict__ cslot, const uint32_t *__restrict__ writers,
                           uint32_t *__restrict__ cflag, uint32_t *__restrict__ ordered, uint32_t *__restrict__ changed, int everything) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand || ordered[i]) return;
    uint32_t sl[RB_MAX_HASH];
    bool ord = everything != 0;                                     // (the closure did not settle in time: every candidate is ordered)
    for (int j = 0; j < h; ++j) {
        sl[j] = cslot[(size_t)i * h + j];
        if (sl[j] != CS_NONE && (sl[j] & CS_REACH)) {
            const uint32_t q = sl[j] & ~CS_REACH;
            if (writers[q] >= 2u || __hip_atomic_load(&cflag[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ord = true;
        }
    }
    if (!ord) return;
    ordered[i] = 1u;
    for (int j = 0; j < h; ++j)
        if (sl[j] != CS_NONE) __hip_atomic_store(&cflag[sl[j] & ~CS_REACH], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *changed = 1u;
}
// the ops of one light run on the register copy of its counters, the write-back, the prefilter cache (stage B proper and (3) below).
// shared: counters another run claimed too — one this run did not change belongs to whoever can reach it: only the claim mark goes.
__device__ __forceinline__ void apply_light_run(const FilterView &fv, uint64_t h0, uint32_t occ_first, const uint64_t *idx, uint64_t cv, uint32_t shared,
                                                uint32_t kfirst, uint32_t krest, const uint8_t *__restrict__ tz, uint32_t first_op, uint32_t ops, int mode) {
    uint32_t c[RB_MAX_HASH];
    for (int j = 0; j < fv.cbf_h; ++j) c[j] = (uint32_t)(cv >> (8 * j)) & 0xFFu;
    uint32_t mn0 = c[0];
    for (int j = 1; j < fv.cbf_h; ++j) mn0 = c[j] < mn0 ? c[j] : mn0;
    run_ops(c, fv.cbf_h, kfirst, krest, tz, first_op, ops);
    for (int j = 0; j < fv.cbf_h; ++j) {
        if (((shared >> j) & 1u) && c[j] == ((uint32_t)(cv >> (8 * j)) & 0xFFu)) cbf_release(fv.cbf, idx[j]);
        else fv.cbf[idx[j]] = (uint8_t)c[j];                    // also clears the claim mark
    }
    if (cache_on(fv) && mode != M_COUNT_ONLY) {   // the k-mer is in dbgbf now; remember its counter exponent
        uint32_t mn = c[0];
        for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
        // ... unless the cache evidently knows it already: same exponent as before the sub-batch and every
        // op of the run succeeded (an up-to-date entry lets through only draws that succeed; the minimum
        // rises by one per success) — saves the bucket read + write for most runs in steady state
        const bool cached = mn0 >= 16u && (mn >> 3) == (mn0 >> 3) && mn - mn0 == ops && !(mn >= 127u && mn0 < 127u);   // (reaching 127 is news: the k-mer is saturated)
        if (mn >= 16u && !cached) cache_store(fv, h0, occ_first, cache_exp(mn));
    }
}
// (3) the candidates outside O: finished here (light) or handed to k_cbf_heavy (flag heavy2[i]); those of O keep RUN_CONFLICT
__global__ void k_resolve_deferred(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ starts,
                                   const uint32_t *__restrict__ vals, const uint32_t *__restrict__ cand, uint32_t n_cand, const uint32_t *__restrict__ ordered,
                                   int mode, uint32_t LIGHT_OPS, const uint32_t *__restrict__ cslot, uint32_t *__restrict__ status, const uint32_t *__restrict__ nops,
                                   const uint64_t *__restrict__ cvals, const uint8_t *__restrict__ tz, uint32_t *__restrict__ heavy2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    heavy2[i] = 0u;
    if (ordered[i]) return;
    const uint32_t d = cand[i], st = status[d] & ~RUN_CONFLICT, ops = nops[d];
    if (ops > LIGHT_OPS) { status[d] = st | RUN_HEAVY; heavy2[i] = 1u; return; }
    status[d] = st;
    const uint64_t h0 = uniq[d];
    uint64_t idx[RB_MAX_HASH];
    uint32_t shared = 0;
    for (int j = 0; j < fv.cbf_h; ++j) {
        idx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        if (cslot[(size_t)i * fv.cbf_h + j] != CS_NONE) shared |= 1u << j;
    }
    apply_light_run(fv, h0, vals[starts[d]], idx, cvals[d], shared, (st >> 12) & 3u, (st >> 14) & 3u, tz, starts[d] + counts[d] - ops, ops, mode);
}
// stable compaction of the flagged entries of a (short) list: pos = exclusive scan of flag
__global__ void k_list_compact(const uint32_t *__restrict__ list, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint32_t n,
                               uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = list[i];
}

// ---- stage B: resolve the found-flag of the first occurrence, set Bloom bits, then apply the
// counter updates of runs that own their counters alone; queue the rest ----
__device__ unsigned long long g_dbg_hist[96];     // RB_DEBUG: ops by (true exponent, cached exponent); uncached ops by bucket occupancy
__global__ void k_resolve_apply(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ counts,
                                const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals,
                                uint32_t n_distinct, int mode, uint32_t LIGHT_OPS, const Slot *ftable, uint32_t f_log2, FtFilter ff,
                                int bits_set /* k_set_bits ran */, const Slot *cs, uint32_t cs_log2, const uint32_t *__restrict__ csf, uint32_t csf_log2,
                                uint32_t n_foreign,
                                uint32_t *__restrict__ status, uint32_t *__restrict__ nops,
                                const uint64_t *__restrict__ cvals, const uint8_t *__restrict__ tz, float *__restrict__ dbgf) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    uint32_t st = status[d];
    // A k-mer that is new, seen once in this sub-batch, and whose first sighting did find a clear bit (ST_LATE without ST_LATE_FOUND; its bits
    // are set already): nothing to count, nothing to look up.  Five runs in six of a long-read insert are of this kind — they leave here
    // on their status word alone instead of pulling hash, count, start, counter values and first occurrence through (28 B + a gather per run).
    if (mode == M_ADD && bits_set && (st & (ST_LATE | ST_LATE_FOUND | ST_CLAIMED)) == ST_LATE) {
        nops[d] = 0u;
        status[d] = (st & 0x7FFu) | (K_INC << 12) | (K_INC << 14);
        return;
    }
    const uint64_t h0 = uniq[d];
    const uint32_t m = counts[d];
    // what the light-run path at the bottom needs, asked for up front: this kernel waits on memory for two thirds of its life
    // (profiles/r03_sq_counters), one dependent load after the other.  (Also asking for the bit-filter words and the first line
    // of strengths here measured 1 ms slower: 48.2 against 47.1 ms.)
    const uint32_t start_d = starts[d];
    const uint64_t cv_d = cvals[d];
    const uint32_t occ_d = vals[start_d];
    const bool all_pre = st & ST_ALLPRE;
    uint32_t ops = 0, kfirst = K_INC, krest = K_INC;
    if (mode == M_COUNT_ONLY) {
        ops = m;
    } else if (mode == M_COUNT_IF_PRESENT) {
        ops = all_pre ? m : 0u;                   // dbgbf never changes in this mode
        kfirst = krest = K_INC_IF_POS;
    } else {
        bool found_first = true;
        if (st & ST_LATE) {                       // arbitration already looked up by k_late_claim
            found_first = (st & ST_LATE_FOUND) != 0;
            if (!bits_set)
                for (int j = 0; j < fv.dbg_h; ++j)
                    if (!((st >> j) & 1u)) bit_set(fv.dbg, index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod));
        } else if (!all_pre) {
            const unsigned long long v_first = vals[starts[d]];
            for (int j = 0; j < fv.dbg_h; ++j) {
                if ((st >> j) & 1u) continue;
                if (bits_set && st_alone(st, j)) { found_first = false; continue; }     // (swept stage: nobody else asked for this bit)
                uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
                // old bit value seen by this probe = an earlier probe of the batch already set it
                if (!set_earlier(ftable, f_log2, idx, (v_first << 4) | (unsigned long long)j, ff)) found_first = false;
                if (!bits_set) bit_set(fv.dbg, idx);
            }
        }
        if (mode == M_ADD) {
            ops = found_first ? m : m - 1u;       // only re-sightings (or false positives) count
        } else {                                  // M_ADD_IF_ABSENT :414-422
            ops = m;
            kfirst = found_first ? K_INC_IF_ZERO : K_INC;
            krest = K_INC_IF_ZERO;
        }
    }
    nops[d] = ops;
    st = (st & 0x7FFu) | (kfirst << 12) | (krest << 14);
    status[d] = st;
    if (dbgf && ops && (st & ST_CLAIMED)) {   // debug: expected fraction of ops that are guaranteed no-ops
        const uint64_t cv0 = cvals[d];
        uint32_t mn = 255;
        for (int j = 0; j < fv.cbf_h; ++j) { uint32_t c = (uint32_t)(cv0 >> (8 * j)) & 0xFFu; mn = c < mn ? c : mn; }
        float drop = (all_pre && mn >= 16u) ? (float)ops * (1.0f - 1.0f / (float)(1u << ((mn >> 3) - 1u))) : 0.0f;
        atomicAdd(&dbgf[2 * (blockIdx.x & 63u)], drop);
        atomicAdd(&dbgf[2 * (blockIdx.x & 63u) + 1], (float)ops);
        if (all_pre && mn >= 16u && fv.mpf.tab && fv.seq_codes) {     // what did the prefilter cache know about this k-mer?
            const uint32_t occ = vals[starts[d]];
            const uint32_t r = fv.seq_first + (occ >> fv.pos_bits), p = occ & ((1u << fv.pos_bits) - 1u);
            const uint64_t bk = mpf_bucket(fv.mpf, window_min_order(fv.seq_codes + seq_word0(fv, r), p, (uint32_t)fv.k, fv.mpf.m));
            const uint32_t sc = min(mpf_match(fv.mpf.tab + (bk << 4), 1u, h0), 7u), s0 = min((mn >> 3) - 1u, 7u);
            atomicAdd(&g_dbg_hist[s0 * 8u + sc], (unsigned long long)ops);
            uint32_t occupied = 0;
            for (uint32_t q = 0; q < 16u; ++q) occupied += fv.mpf.tab[(bk << 4) + q] != 0ull;
            if (sc == 0u) atomicAdd(&g_dbg_hist[64u + occupied], (unsigned long long)ops);
        }
    }
    if (!(st & ST_CLAIMED)) return;
    uint64_t idx[RB_MAX_HASH];
    bool conflict = (st & ST_FOREIGN) != 0;
    for (int j = 0; j < fv.cbf_h; ++j) {
        idx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        if (n_foreign && !conflict) {               // did another run claim this counter after this one?  (most have not: bit filter first)
            const uint64_t b = slot_of(idx[j], csf_log2);
            if (!csf || ((csf[b >> 5] >> (uint32_t)(b & 31u)) & 1u)) conflict = table_find(cs, cs_log2, idx[j]) != nullptr;
        }
    }
    if (ops == 0) {                               // nothing to count: just drop the claim marks
        for (int j = 0; j < fv.cbf_h; ++j) cbf_release(fv.cbf, idx[j]);
        return;
    }
    // a run with a shared counter is set aside: whether it needs the ordered replay is decided behind the point where the
    // producer is released (k_cs_writers / k_cs_order / k_resolve_deferred); marks of replayed runs are dropped by k_conf_release
    if (conflict) { status[d] = st | RUN_CONFLICT; return; }
    if (ops > LIGHT_OPS) { status[d] = st | RUN_HEAVY; return; }
    apply_light_run(fv, h0, occ_d, idx, cv_d, 0u, kfirst, krest, tz, start_d + m - ops, ops, mode);
}

// drop the claim marks of the counters of conflicting runs before they are replayed in order
__global__ void k_conf_release(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ conf_kmers,
                               uint32_t n_conf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_conf) return;
    const uint64_t h0 = uniq[conf_kmers[i]];
    for (int j = 0; j < fv.cbf_h; ++j) cbf_release(fv.cbf, index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod));
}

// ---- conflict path: expand the ops of conflicting k-mers, sort by occurrence, replay in order ----
__global__ void k_conf_offsets(const uint32_t *__restrict__ conf_kmers, const uint32_t *__restrict__ nops,
                               uint32_t n_conf, uint32_t *__restrict__ sizes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_conf) sizes[i] = nops[conf_kmers[i]];
    if (i == n_conf) sizes[i] = 0;
}
// connected components of the "shares a counter" graph between conflicting k-mers, by min-label
// propagation through the claim-table slots (components are tiny: the graph is far below the
// percolation threshold for any sane filter size).  slot.lo starts as the smallest owner id.
// (the table lookups once, into lslot[i * h + j] = slot or CS_NONE; every round then indexes the table)
__global__ void k_label_init(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ conf_kmers, uint32_t n_conf,
                             const Slot *ctable, uint32_t c_log2, uint32_t *__restrict__ label, uint32_t *__restrict__ lslot) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_conf) return;
    const uint32_t d = conf_kmers[i];
    label[d] = d;
    const uint64_t h0 = uniq[d];
    for (int j = 0; j < fv.cbf_h; ++j) {
        const Slot *sl = table_find(ctable, c_log2, index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod));
        lslot[(size_t)i * fv.cbf_h + j] = sl ? (uint32_t)(sl - ctable) : CS_NONE;          // unshared counters are not in the set
    }
}
// One round of the propagation, pull and push in one kernel and in no particular order between the runs (every update is a min: any order
// ends at the component's smallest run id): a run takes the smallest label its counters have seen, follows its own label one step (label[l] is
// a run of the same component with a label at most l: pointer jumping, the long chains of a big component shrink by halves), and hands the
// result to its counters.  A fixed point has label[d] == slot value on every edge; `changed` says whether this round moved anything.
__global__ void k_label_round(int h, const uint32_t *__restrict__ conf_kmers, uint32_t n_conf, Slot *ctable, const uint32_t *__restrict__ lslot,
                              uint32_t *__restrict__ label, uint32_t *__restrict__ changed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_conf) return;
    const uint32_t d = conf_kmers[i];
    const uint32_t l0 = __hip_atomic_load(&label[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t l = l0, sl[RB_MAX_HASH];
    for (int j = 0; j < h; ++j) {
        sl[j] = lslot[(size_t)i * h + j];
        if (sl[j] == CS_NONE) continue;
        const uint32_t v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(&ctable[sl[j]].val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        l = v < l ? v : l;
    }
    const uint32_t up = __hip_atomic_load(&label[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l = up < l ? up : l;
    bool moved = l != l0;
    if (moved) __hip_atomic_store(&label[d], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j = 0; j < h; ++j)
        if (sl[j] != CS_NONE && atomicMin(reinterpret_cast<uint32_t *>(&ctable[sl[j]].val), l) > l) moved = true;
    if (moved) *changed = 1u;
}
__global__ void k_conf_kmer_keys(const uint32_t *__restrict__ conf_kmers, const uint32_t *__restrict__ label,
                                 uint32_t n_conf, uint64_t *__restrict__ keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_conf) { uint32_t d = conf_kmers[i]; keys[i] = ((uint64_t)label[d] << 32) | d; }
}
__global__ void k_conf_expand(const uint32_t *__restrict__ conf_kmers, const uint32_t *__restrict__ conf_off,
                              const uint32_t *__restrict__ counts, const uint32_t *__restrict__ starts,
                              const uint32_t *__restrict__ vals, const uint32_t *__restrict__ status,
                              const uint32_t *__restrict__ nops, const uint32_t *__restrict__ label, uint32_t n_conf,
                              uint64_t *__restrict__ op_key, uint32_t *__restrict__ op_val) {
    // eight lanes per conflicting k-mer (a run brings ~5 ops on average; heavy ones loop)
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, lane = threadIdx.x & 7u;
    if (wave >= n_conf) return;
    const uint32_t d = conf_kmers[wave];
    const uint32_t ops = nops[d], st = status[d];
    const uint64_t hi = (uint64_t)label[d] << 32;
    const uint32_t base = starts[d] + counts[d] - ops, out = conf_off[wave];
    for (uint32_t i = lane; i < ops; i += 8u) {
        op_key[out + i] = hi | vals[base + i];
        uint32_t kind = i == 0 ? (st >> 12) & 3u : (st >> 14) & 3u;
        op_val[out + i] = d | (kind << 30);
    }
}

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// ordered replay of one component's ops [os,oe) by a single lane with read-modify-write on the
// counting filter itself (nobody else touches these counters during the batch)
__device__ void replay_serial(const FilterView &fv, const uint64_t *__restrict__ uniq, const uint64_t *__restrict__ op_key,
                              const uint32_t *__restrict__ op_val, uint32_t os, uint32_t oe) {
    // consecutive ops of one run reuse its counter indices and the values this lane left there (nobody else writes
    // them); a switch to another run of the component reloads — the runs share counters
    uint64_t idx[RB_MAX_HASH];
    uint32_t c[RB_MAX_HASH], cur_d = 0xFFFFFFFFu;
    for (uint32_t i = os; i < oe; ++i) {
        const uint32_t v = (uint32_t)op_key[i];
        const uint32_t ov = op_val[i], d = ov & 0x3FFFFFFFu, kind = ov >> 30;
        if (d != cur_d) {
            const uint64_t h0 = uniq[d];
            for (int j = 0; j < fv.cbf_h; ++j) {
                idx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
                c[j] = *(volatile uint8_t *)&fv.cbf[idx[j]];
            }
            cur_d = d;
        }
        uint32_t c0[RB_MAX_HASH];
        for (int j = 0; j < fv.cbf_h; ++j) c0[j] = c[j];
        uint32_t mn = c[0];
        for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
        cbf_step(c, fv.cbf_h, kind, (mn >= 16u && mn < 127u) ? occ_rnd(fv, v) : 0u);
        for (int j = 0; j < fv.cbf_h; ++j)
            if (c[j] != c0[j]) *(volatile uint8_t *)&fv.cbf[idx[j]] = (uint8_t)c[j];
    }
}
constexpr uint32_t MAX_COMPONENT_KMERS = 8;
// one thread per conflicting k-mer (sorted by component label): component heads either replay a
// small component themselves or queue it for the wave-cooperative kernel
__global__ void k_conf_replay_small(FilterView fv, const uint64_t *__restrict__ uniq, const uint64_t *__restrict__ kmer_keys,
                                    uint32_t n_conf, const uint64_t *__restrict__ op_key, const uint32_t *__restrict__ op_val,
                                    uint32_t n_ops, uint32_t *__restrict__ big_list, uint32_t *__restrict__ n_big, int store_cache,
                                    const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals, uint32_t small_ops) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_conf) return;
    const uint32_t lab = (uint32_t)(kmer_keys[i] >> 32);
    if (i > 0 && (uint32_t)(kmer_keys[i - 1] >> 32) == lab) return;   // not a component head
    const uint32_t os = lower_bound_u64(op_key, n_ops, (uint64_t)lab << 32);
    const uint32_t oe = lower_bound_u64(op_key, n_ops, ((uint64_t)lab + 1ull) << 32);
    if (oe - os > small_ops) { big_list[atomicAdd(n_big, 1u)] = i; return; }
    replay_serial(fv, uniq, op_key, op_val, os, oe);
    if (store_cache && cache_on(fv))   // the component's k-mers are in dbgbf; remember their counter exponents
        for (uint32_t q = i; q < n_conf && (uint32_t)(kmer_keys[q] >> 32) == lab; ++q) {
            const uint32_t dq = (uint32_t)kmer_keys[q];
            const uint64_t h0 = uniq[dq];
            uint32_t mn = 255u;
            for (int j = 0; j < fv.cbf_h; ++j) {
                const uint32_t c = *(volatile uint8_t *)&fv.cbf[index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod)];
                mn = c < mn ? c : mn;
            }
            if (mn >= 16u && mn < 128u) cache_store(fv, h0, vals[starts[dq]], cache_exp(mn));
        }
}
// one wavefront per large component: the component's counters live in LDS; 64 ops are examined at
// a time against the current state and the chain hops from one state-changing op to the next
__global__ void __launch_bounds__(64) k_conf_replay_big(FilterView fv, const uint64_t *__restrict__ uniq,
                                    const uint64_t *__restrict__ kmer_keys, uint32_t n_conf,
                                    const uint64_t *__restrict__ op_key, const uint32_t *__restrict__ op_val, uint32_t n_ops,
                                    const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ n_big,
                                    uint32_t *__restrict__ dbg, int store_cache,
                                    const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals) {
    __shared__ uint32_t s_drun[MAX_COMPONENT_KMERS];                // one run carrying each distinct hash
    __shared__ uint64_t s_idx[MAX_COMPONENT_KMERS * RB_MAX_HASH];   // unique counter indices
    __shared__ uint32_t s_val[MAX_COMPONENT_KMERS * RB_MAX_HASH];   // their current bytes
    __shared__ uint32_t s_val0[MAX_COMPONENT_KMERS * RB_MAX_HASH];
    __shared__ uint32_t s_slot[MAX_COMPONENT_KMERS][RB_MAX_HASH];   // k-mer probe -> unique counter
    __shared__ uint64_t s_h0[MAX_COMPONENT_KMERS];
    __shared__ uint32_t s_nu, s_nk;
    const uint32_t lane = threadIdx.x;
    const int H = fv.cbf_h;
    for (uint32_t bi = blockIdx.x; bi < *n_big; bi += gridDim.x) {
        const uint32_t i0 = big_list[bi];
        const uint32_t lab = (uint32_t)(kmer_keys[i0] >> 32);
        const uint32_t i1 = lower_bound_u64(kmer_keys, n_conf, ((uint64_t)lab + 1ull) << 32);   // runs [i0,i1)
        const uint32_t os = lower_bound_u64(op_key, n_ops, (uint64_t)lab << 32);
        const uint32_t oe = lower_bound_u64(op_key, n_ops, ((uint64_t)lab + 1ull) << 32);
        // distinct hashes of the component (a hash split into many runs is still one k-mer)
        __syncthreads();
        if (lane == 0) s_nk = 0;
        __syncthreads();
        bool overflow = false;
        for (uint32_t rb0 = i0; rb0 < i1 && !overflow; rb0 += 64u) {
            const bool have = rb0 + lane < i1;
            const uint32_t myd = have ? (uint32_t)kmer_keys[rb0 + lane] : 0u;
            const uint64_t h = have ? uniq[myd] : 0ull;
            unsigned long long pending = __ballot(have);
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const uint64_t hl = __shfl(h, leader, 64);
                const uint32_t dl = __shfl(myd, leader, 64);
                pending &= ~__ballot(have && h == hl);
                uint32_t q = 0, nkc = s_nk;
                while (q < nkc && s_h0[q] != hl) ++q;
                if (q == nkc) {
                    if (nkc == MAX_COMPONENT_KMERS) { overflow = true; break; }
                    __syncthreads();
                    if (lane == 0) { s_h0[nkc] = hl; s_drun[nkc] = dl; s_nk = nkc + 1u; }
                    __syncthreads();
                }
            }
        }
        if (dbg && lane == 0) { atomicMax(&dbg[1], oe - os); atomicMax(&dbg[2], i1 - i0); atomicAdd(&dbg[3], oe - os); }
        if (overflow) {                          // rare: many DIFFERENT k-mers in one component -> plain ordered replay
            if (dbg && lane == 0) { atomicAdd(&dbg[0], 1u); atomicAdd(&dbg[4], oe - os); }
            if (lane == 0) replay_serial(fv, uniq, op_key, op_val, os, oe);
            continue;
        }
        const uint32_t nk = s_nk;
        __syncthreads();
        if (lane == 0) {                         // build the component's counter table (tiny)
            uint32_t nu = 0;
            for (uint32_t q = 0; q < nk; ++q) {
                const uint64_t h0 = s_h0[q];
                for (int j = 0; j < H; ++j) {
                    const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
                    uint32_t u = 0;
                    while (u < nu && s_idx[u] != idx) ++u;
                    if (u == nu) { s_idx[u] = idx; s_val0[u] = s_val[u] = fv.cbf[idx]; ++nu; }
                    s_slot[q][j] = u;
                }
            }
            s_nu = nu;
        }
        __syncthreads();
        for (uint32_t base = os; base < oe; base += 64u) {
            const uint32_t i = base + lane;
            uint32_t q = 0, kind = 0, rnd = 0;
            const bool live = i < oe;
            if (live) {
                const uint32_t ov = op_val[i];
                const uint64_t h = uniq[ov & 0x3FFFFFFFu];
                kind = ov >> 30;
                while (q < nk && s_h0[q] != h) ++q;
                rnd = occ_rnd(fv, (uint32_t)op_key[i]);
            }
            uint32_t cursor = 0;                 // ops below cursor are settled
            while (cursor < 64u) {
                bool changes = false;
                if (live && lane >= cursor) {
                    uint32_t mn = s_val[s_slot[q][0]];
                    for (int j = 1; j < H; ++j) { uint32_t c = s_val[s_slot[q][j]]; mn = c < mn ? c : mn; }
                    const bool gate = !((kind == K_INC_IF_POS && mn == 0u) || (kind == K_INC_IF_ZERO && mn != 0u));
                    changes = gate && minifloat_inc(mn, rnd) != mn;
                }
                const unsigned long long win = __ballot(changes);
                if (!win) break;
                const uint32_t first = (uint32_t)__ffsll((long long)win) - 1u;
                if (lane == first) {             // apply: every probe equal to the minimum moves up
                    uint32_t mn = s_val[s_slot[q][0]];
                    for (int j = 1; j < H; ++j) { uint32_t c = s_val[s_slot[q][j]]; mn = c < mn ? c : mn; }
                    for (int j = 0; j < H; ++j) if (s_val[s_slot[q][j]] == mn) s_val[s_slot[q][j]] = mn + 1u;
                }
                __syncthreads();
                cursor = first + 1u;
            }
            __syncthreads();
        }
        if (lane < s_nu && s_val[lane] != s_val0[lane]) fv.cbf[s_idx[lane]] = (uint8_t)s_val[lane];
        if (store_cache && cache_on(fv) && lane < nk) {   // the component's k-mers are in dbgbf; remember their exponents
            uint32_t mn = s_val[s_slot[lane][0]];
            for (int j = 1; j < H; ++j) { const uint32_t c = s_val[s_slot[lane][j]]; mn = c < mn ? c : mn; }
            if (mn >= 16u && mn < 128u) cache_store(fv, s_h0[lane], vals[starts[s_drun[lane]]], cache_exp(mn));
        }
        __syncthreads();
    }
}

// ---- paired k-mers: {,Canonical,ReverseComplement}PairedNTHashIterator + rpkbf.add ----
// (R/bloom/hash/PairedNTHashIterator.java:56-83, CanonicalPaired… :36-60, ReverseComplementPaired…
//  :33-56; R/RNABloom.java:587-591).  Pure OR => order independent => direct atomicOr.
//
// Two kernels.  k_pairs_reads (below) is the one that runs for k <= 64 and reads of at most 384 bases: one
// read per lane, both windows rolled side by side.  k_pairs_insert is the general one (any k, any read
// length, one thread per 32-base word, windows hashed from scratch and then rolled).  Whether it sets bits or
// collects bit indices (sharded engine) is a TEMPLATE parameter on purpose: with that choice made at run time
// (`if (out_idx) ... else ...` inside the roll loop) hipcc 7.2's optimised gfx950 code set ~20 % of the pair bits of a
// 200 000-read launch at wrong positions, differently from run to run, while the same source without the
// never-taken branch — and every other restructuring of the loop — is exact (tools/pairs_variants.py is the
// bisection: it rebuilds that kernel with -DRB_DIAG_PAIRS and counts the wrong bits; the unoptimised build was
// exact too, which is how round 1 shipped it, 25x slower).  Both instantiations are covered at scale
// (tests/test_gpu_parity.py::test_read_pairs_at_scale[general], tests/test_gpu_scale.py).
// first unusable base at or after p (or L) — word-wise scan of the validity bits
__device__ __forceinline__ uint32_t next_unusable(const uint32_t *__restrict__ vw, uint32_t p, uint32_t L) {
    while (p < L) {
        const uint32_t w = ~vw[p >> 5] >> (p & 31u);           // zero bits of the mask, shifted to bit 0
        if (w) { const uint32_t q = p + (uint32_t)__ffs((int)w) - 1u; return q < L ? q : L; }
        p = (p | 31u) + 1u;
    }
    return L;
}
template <int MODE, bool OUT>
__global__ void k_pairs_insert(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                               const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                               const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k, int dist,
                               uint32_t *bits, Mod mod, int num_hash, uint64_t kmul,
                               unsigned long long *__restrict__ n_pairs,
                               const uint32_t *__restrict__ chunk_off, uint64_t *__restrict__ out_idx,
                               uint32_t wpr_uniform) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nw) return;
    // uniform-length batches: chunk-major order, so that all lanes of a wavefront work on the same
    // chunk number (for 150 bp reads with d = 115 only chunk 0 of each read has paired k-mers)
    int64_t i = t;
    if (wpr_uniform) { const int64_t nreads = nw / wpr_uniform; i = (t % nreads) * wpr_uniform + t / nreads; }
    const int64_t w = w0 + i;
    const uint32_t r = word_read[w], wr = woff[r], L = len[r];
    const uint32_t b0 = (uint32_t)(w - wr) * 32u;
    const uint32_t uk = (uint32_t)k, ud = (uint32_t)dist, span = uk + ud;
    if ((uint64_t)b0 + span > L) return;
    const uint32_t pe = (b0 + 32u < L - span + 1u) ? b0 + 32u : L - span + 1u;   // pair starts handled here: [b0, pe)
    const uint64_t *cw = codes + wr;
    const uint32_t *vw = valid + wr;
    auto code_at = [&](uint32_t b) { return (uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u; };
    uint32_t cnt = 0, p = b0;
    while (p < pe) {
        const uint32_t nz = next_unusable(vw, p, L);
        if (nz < p + span) { p = nz + 1u; continue; }          // no pair can start in [p, nz]
        const uint32_t plast = (pe - 1u < nz - span) ? pe - 1u : nz - span;
        // both windows from scratch (k steps), then roll: NTHash.java:332-337,367-373 / :491-495
        uint64_t fL = 0, rL = 0, fR = 0, rR = 0;
        for (uint32_t q = 0; q < uk; ++q) {
            const uint32_t cl = code_at(p + q), cr = code_at(p + ud + q);
            fL = rotl(fL, 1) ^ seed_of(cl); rL ^= rotl(seed_of(3u - cl), q);
            fR = rotl(fR, 1) ^ seed_of(cr); rR ^= rotl(seed_of(3u - cr), q);
        }
        for (;;) {
            uint64_t P;
            if (MODE == 0) P = combine(fL, fR);                 // PairedNTHashIterator.java:69
            else if (MODE == 2) P = combine(rR, rL);            // ReverseComplementPaired… :44
            else P = smin(combine(fL, fR), combine(rR, rL));    // CanonicalPaired… :44 (signed min)
            if (OUT) {       // sharded engine: collect global bit indices instead of setting local bits
                for (int j = 0; j < num_hash; ++j)
                    out_idx[((size_t)chunk_off[i] + cnt) * (size_t)num_hash + j] = index_of(multi_hash(P, (uint32_t)j, kmul), mod);
            } else {
                for (int j = 0; j < num_hash; ++j) bit_set(bits, index_of(multi_hash(P, (uint32_t)j, kmul), mod));
            }
            ++cnt;
            if (p == plast) break;
            const uint32_t ol = code_at(p), il = code_at(p + uk), orr = code_at(p + ud), ir = code_at(p + ud + uk);
            fL = rotl(fL, 1) ^ rotl(seed_of(ol), uk) ^ seed_of(il);
            rL = rotr(rL, 1) ^ rotr(seed_of(3u - ol), 1) ^ rotl(seed_of(3u - il), uk - 1u);
            fR = rotl(fR, 1) ^ rotl(seed_of(orr), uk) ^ seed_of(ir);
            rR = rotr(rR, 1) ^ rotr(seed_of(3u - orr), 1) ^ rotl(seed_of(3u - ir), uk - 1u);
            ++p;
        }
        p = plast + 1u;
    }
    if (cnt && n_pairs) atomicAdd(n_pairs, (unsigned long long)cnt);
}


#ifdef RB_DIAG_PAIRS
// The form that miscompiles (see above): identical to k_pairs_insert<1, false> except that the out_idx choice is a run-time
// branch.  Only built with -DRB_DIAG_PAIRS, only launched by RB_PAIRS_VARIANT=9 (tools/pairs_variants.py).
__global__ void k_pairs_insert_runtime_branch(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                               const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                               const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k, int dist,
                               uint32_t *bits, Mod mod, int num_hash, uint64_t kmul,
                               unsigned long long *__restrict__ n_pairs,
                               const uint32_t *__restrict__ chunk_off, uint64_t *__restrict__ out_idx,
                               uint32_t wpr_uniform) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nw) return;
    int64_t i = t;
    if (wpr_uniform) { const int64_t nreads = nw / wpr_uniform; i = (t % nreads) * wpr_uniform + t / nreads; }
    const int64_t w = w0 + i;
    const uint32_t r = word_read[w], wr = woff[r], L = len[r];
    const uint32_t b0 = (uint32_t)(w - wr) * 32u;
    const uint32_t uk = (uint32_t)k, ud = (uint32_t)dist, span = uk + ud;
    if ((uint64_t)b0 + span > L) return;
    const uint32_t pe = (b0 + 32u < L - span + 1u) ? b0 + 32u : L - span + 1u;
    const uint64_t *cw = codes + wr;
    const uint32_t *vw = valid + wr;
    auto code_at = [&](uint32_t b) { return (uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u; };
    uint32_t cnt = 0, p = b0;
    while (p < pe) {
        const uint32_t nz = next_unusable(vw, p, L);
        if (nz < p + span) { p = nz + 1u; continue; }
        const uint32_t plast = (pe - 1u < nz - span) ? pe - 1u : nz - span;
        uint64_t fL = 0, rL = 0, fR = 0, rR = 0;
        for (uint32_t q = 0; q < uk; ++q) {
            const uint32_t cl = code_at(p + q), cr = code_at(p + ud + q);
            fL = rotl(fL, 1) ^ seed_of(cl); rL ^= rotl(seed_of(3u - cl), q);
            fR = rotl(fR, 1) ^ seed_of(cr); rR ^= rotl(seed_of(3u - cr), q);
        }
        for (;;) {
            const uint64_t P = smin(combine(fL, fR), combine(rR, rL));
            if (out_idx) {
                for (int j = 0; j < num_hash; ++j)
                    out_idx[((size_t)chunk_off[i] + cnt) * (size_t)num_hash + j] = index_of(multi_hash(P, (uint32_t)j, kmul), mod);
            } else {
                for (int j = 0; j < num_hash; ++j) bit_set(bits, index_of(multi_hash(P, (uint32_t)j, kmul), mod));
            }
            ++cnt;
            if (p == plast) break;
            const uint32_t ol = code_at(p), il = code_at(p + uk), orr = code_at(p + ud), ir = code_at(p + ud + uk);
            fL = rotl(fL, 1) ^ rotl(seed_of(ol), uk) ^ seed_of(il);
            rL = rotr(rL, 1) ^ rotr(seed_of(3u - ol), 1) ^ rotl(seed_of(3u - il), uk - 1u);
            fR = rotl(fR, 1) ^ rotl(seed_of(orr), uk) ^ seed_of(ir);
            rR = rotr(rR, 1) ^ rotr(seed_of(3u - orr), 1) ^ rotl(seed_of(3u - ir), uk - 1u);
            ++p;
        }
        p = plast + 1u;
    }
    if (cnt && n_pairs) atomicAdd(n_pairs, (unsigned long long)cnt);
}
#endif

// One read per lane (k <= 64, reads of <= 384 bases).  A pair (p, p + d) needs the window at p, the window at
// p + d and no unusable base in [p, p + k + d).  The lane rolls the two windows side by side — the right one
// over bases d.., the left one over bases 0.. — so a 150-base read with d = 115 takes L - d = 35 steps for its 11
// pairs.  Unusable bases enter and leave the rolling hashes as null bases (see k_hash_windows_fast), and the
// position of the last unusable base at or before the right window's end decides whether a pair exists: for
// bases below d it is found once from the preloaded validity words, from d on the right stream keeps it.
// The read's words are loaded into registers up front (static indices; a word boundary costs a select chain).
// Sharded engine: with out_idx the global bit indices are written instead, at the per-word offsets
// (chunk_off, relative to word w0) that launch_count_windows(k + d) + scan produced.
// PAIR_WORDS: 12 (384 bases) for the stage-1 reads, 32 (1024 bases) for fragments.  min_len: reads shorter than that
// contribute nothing (FragmentsToGraphWorker adds fragment pairs only where read pairs could start).  present: when
// set, a pair is added only if both of its k-mers are in dbgbf (PairedKmersToGraphWorker with existingKmersOnly,
// R/RNABloom.java:466-482: graph.contains(lHashVals) && graph.contains(rHashVals)).
//
// VAR — what happens to a pair's bit indices — is a template parameter (0: set the bits, 1: set them unless the SEEN-PAIR CACHE knows the
// pair, 2: write the indices out), never a run-time branch inside the roll loop (see k_pairs_insert above and
// tests/test_capi_symbols.py::test_no_kernel_picks_its_store_target_at_run_time_inside_a_loop).
//
// The seen-pair cache (VAR 1; BitFilter::seen).  At 191x coverage a step adds 943 M pairs that are sightings of ~50 M distinct ones: two
// random line requests each for bits that are set already, 1.9 G requests at the device's random-line rate (37 ms, and the kernels that
// run beside the walker pay again).  rpkbf.add is idempotent, so a pair that is KNOWN to be in the filter can be skipped: a table of
// buckets of 16 pair hashes (128 bytes), an entry = "both bits of this pair hash have been set in this filter since it was last cleared"
// (written after the bit_set calls; 64-bit compare, so a match is exact; entries stay true for ever: bits are never cleared without the
// cache — BitFilter::seen is reset with the bits).  What makes it cheaper than the probes is the bucket function: consecutive pairs of a
// read must share a bucket, and a pair must find the same bucket from whichever read it is seen in.  So the bucket is a function of the
// pair's LEFT k-mer alone: the latest ANCHOR among its m-mers (m = min(16, k); an m-mer whose mixed canonical hash has its low two bits
// clear: one in four), the k-mer's last m-mer if it has none — the m-mer's 2-bit code shifted along with the left window, one 32-bit multiply per step.  A run of ~4
// consecutive pairs shares an anchor: one 128-byte fetch into the lane's LDS image (entry e of lane l at [e][l]: conflict-free) per run
// instead of 2 line requests per pair; a pair may sit in slot P & 15 or (P >> 4) & 15.  Stale or lost entries (races between lanes, L2
// copies of another XCD) only cost the probes they would have saved.
template <int MODE, int PAIR_WORDS, int VAR>
__global__ void __launch_bounds__(64)
k_pairs_reads(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid, const uint32_t *__restrict__ woff,
              const uint32_t *__restrict__ len, int64_t r0, int64_t nr, int64_t w0, int k, int dist, uint32_t *bits, Mod mod,
              int num_hash, uint64_t kmul, unsigned long long *__restrict__ n_pairs, const uint32_t *__restrict__ chunk_off,
              uint64_t *__restrict__ out_idx, uint32_t min_len, const uint32_t *__restrict__ present, Mod present_mod, int present_h,
              PairSeen seen) {
    __shared__ uint64_t s_tf[25], s_tr[25];
    __shared__ unsigned long long s_img[VAR == 1 ? 16 * 64 : 1];
    const uint32_t uk = (uint32_t)k, ud = (uint32_t)dist, span = uk + ud;
    const uint32_t um = uk < 16u ? uk : 16u;                          // m-mer of the bucket function (VAR 1)
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;   // 0 = null, 1..4 = A,C,G,T
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    uint32_t cnt = 0;
    if (t < nr) {
        const int64_t r = r0 + t;
        const uint32_t wr = woff[r], L = len[r];
        if (L >= span && L >= min_len) {
            const uint32_t nwords = (L + 31u) >> 5;
            uint64_t carr[PAIR_WORDS];
            uint32_t varr[PAIR_WORDS];
#pragma unroll
            for (int q = 0; q < PAIR_WORDS; ++q) {
                carr[q] = 0; varr[q] = 0;
                if ((uint32_t)q < nwords) { carr[q] = codes[wr + q]; varr[q] = valid[wr + q]; }
            }
            // last unusable base below d (as position + 1, 0 = none)
            uint32_t lb = 0;
#pragma unroll
            for (int q = 0; q < PAIR_WORDS; ++q) {
                if (32u * (uint32_t)q < ud) {
                    const uint32_t hi = ud - 32u * (uint32_t)q;                       // positions of this word below d
                    const uint32_t m = ~varr[q] & (hi >= 32u ? 0xFFFFFFFFu : ((1u << hi) - 1u));
                    if (m) lb = 32u * (uint32_t)q + 32u - (uint32_t)__clz((int)m);
                }
            }
            auto word_c = [&](uint32_t wi) { uint64_t v = 0;
#pragma unroll
                for (int q = 0; q < PAIR_WORDS; ++q) v = ((uint32_t)q == wi) ? carr[q] : v;
                return v; };
            auto word_v = [&](uint32_t wi) { uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < PAIR_WORDS; ++q) v = ((uint32_t)q == wi) ? varr[q] : v;
                return v; };
            uint64_t cR = word_c(ud >> 5) >> (2u * (ud & 31u)), cL = carr[0];
            uint32_t vR = word_v(ud >> 5) >> (ud & 31u), vL = varr[0];
            // histories of the last 64 bases of each stream: codes in 128 bits (lo = the latest 32), usable bits in 64
            uint64_t fR = 0, rR = 0, fL = 0, rL = 0, hcR = 0, hcR2 = 0, hcL = 0, hcL2 = 0, hvR = 0, hvL = 0;
            const uint32_t sh_v = uk - 1u, sh_c = 2u * ((uk - 1u) & 31u);
            const bool far = uk > 32u;                                // the outgoing base sits in the older half
            const uint32_t nsteps = L - ud;
            const int64_t wrel = (int64_t)wr - w0;                  // the read's first word, relative to w0 (out_idx mode)
            uint32_t obase = 0, ocnt = 0, oword = 0xFFFFFFFFu;
            uint32_t mF = 0, mR = 0;                                 // VAR 1: the left stream's m-mer as 2-bit codes (both strands), the latest anchor
            const uint32_t mmask = um >= 16u ? 0xFFFFFFFFu : ((1u << (2u * um)) - 1u), mtop = 2u * (um - 1u);
            uint32_t akey = 0, apos1 = 0, xnew = 0, curb = 0xFFFFFFFFu;
            uint32_t dbg_miss = 0, dbg_fetch = 0;
#pragma nounroll
            for (uint32_t j = 0; j < nsteps; ++j) {
                const uint32_t e = ud + j;                            // base entering the right window
                if (j && (e & 31u) == 0u) { cR = word_c(e >> 5); vR = word_v(e >> 5); }
                if (j && (j & 31u) == 0u) { cL = word_c(j >> 5); vL = word_v(j >> 5); }
                const uint32_t codeR = (uint32_t)cR & 3u, okR = vR & 1u, codeL = (uint32_t)cL & 3u, okL = vL & 1u;
                cR >>= 2; vR >>= 1; cL >>= 2; vL >>= 1;
                {
                    const uint32_t in5 = okR ? codeR + 1u : 0u;
                    const uint32_t out5 = ((uint32_t)(hvR >> sh_v) & 1u) ? ((uint32_t)((far ? hcR2 : hcR) >> sh_c) & 3u) + 1u : 0u;
                    const uint32_t tt = out5 * 5u + in5;
                    fR = rotl(fR, 1) ^ s_tf[tt]; rR = rotr(rR, 1) ^ s_tr[tt];
                    hcR2 = (hcR2 << 2) | (hcR >> 62); hcR = (hcR << 2) | codeR; hvR = (hvR << 1) | okR;
                }
                {
                    const uint32_t in5 = okL ? codeL + 1u : 0u;
                    const uint32_t out5 = ((uint32_t)(hvL >> sh_v) & 1u) ? ((uint32_t)((far ? hcL2 : hcL) >> sh_c) & 3u) + 1u : 0u;
                    const uint32_t tt = out5 * 5u + in5;
                    fL = rotl(fL, 1) ^ s_tf[tt]; rL = rotr(rL, 1) ^ s_tr[tt];
                    if (VAR == 1) {       // the m-mer that ends at base j (starts at j + 1 - m): canonical 2-bit code, mixed; an anchor if the mix's low bits are clear
                        mF = ((mF << 2) | codeL) & mmask;                 // (an unusable base enters as whatever its code bits are: a pair whose
                        mR = (mR >> 2) | ((3u - codeL) << mtop);          //  left k-mer holds one does not exist, and nobody asks for its bucket)
                        uint32_t x = (mF < mR ? mF : mR) * 0x9E3779B1u;
                        x ^= x >> 15;
                        xnew = x >> 4;
                        if ((x & seen.amask) == 0u && j + 1u >= um) { akey = xnew; apos1 = j + 2u - um; }
                    }
                    hcL2 = (hcL2 << 2) | (hcL >> 62); hcL = (hcL << 2) | codeL; hvL = (hvL << 1) | okL;
                }
                lb = okR ? lb : e + 1u;
                if (j + 1u >= uk) {
                    const uint32_t p = j + 1u - uk;                   // pair start: windows [p, p+k) and [p+d, p+d+k)
                    bool ok = lb <= p;                                // no unusable base in [p, p + span)
                    if (ok && present) {
                        const uint64_t hl = (MODE == 0) ? fL : (MODE == 2) ? rL : smin(fL, rL);
                        const uint64_t hr = (MODE == 0) ? fR : (MODE == 2) ? rR : smin(fR, rR);
                        ok = bits_lookup(present, present_mod, present_h, kmul, hl) && bits_lookup(present, present_mod, present_h, kmul, hr);
                    }
                    if (ok) {
                        uint64_t P;
                        if (MODE == 0) P = combine(fL, fR);                 // PairedNTHashIterator.java:69
                        else if (MODE == 2) P = combine(rR, rL);            // ReverseComplementPaired… :44
                        else P = smin(combine(fL, fR), combine(rR, rL));    // CanonicalPaired… :44 (signed min)
                        if (VAR == 2) {
                            if ((p >> 5) != oword) { oword = p >> 5; obase = chunk_off[wrel + (int64_t)oword]; ocnt = 0; }
                            for (int h = 0; h < num_hash; ++h)
                                out_idx[((size_t)obase + ocnt) * (size_t)num_hash + h] = index_of(multi_hash(P, (uint32_t)h, kmul), mod);
                            ++ocnt;
                        } else if (VAR == 1) {
                            // bucket of the left k-mer [p, p + k): its latest anchor (start >= p), else its last m-mer
                            const uint32_t bk = (apos1 > p ? akey : xnew) & seen.mask;
                            if (bk != curb) {                         // fetch the bucket into this lane's column of the image
                                curb = bk; ++dbg_fetch;
                                const uint4 *src = reinterpret_cast<const uint4 *>(seen.tab + ((size_t)bk << 4));
                                uint4 v[8];
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] = src[q];
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    s_img[(2 * q) * 64 + threadIdx.x] = (unsigned long long)v[q].x | ((unsigned long long)v[q].y << 32);
                                    s_img[(2 * q + 1) * 64 + threadIdx.x] = (unsigned long long)v[q].z | ((unsigned long long)v[q].w << 32);
                                }
                            }
                            const uint32_t sa = (uint32_t)P & 15u, sb0 = ((uint32_t)P >> 4) & 15u, sb = sb0 == sa ? (sa ^ 1u) : sb0;
                            const unsigned long long ea = s_img[sa * 64u + threadIdx.x], eb = s_img[sb * 64u + threadIdx.x];
                            if (P == 0ull || (ea != P && eb != P)) {
                                ++dbg_miss;
                                for (int h = 0; h < num_hash; ++h) bit_set(bits, index_of(multi_hash(P, (uint32_t)h, kmul), mod));
                                if (P != 0ull) {                      // the entry follows the bits
                                    const uint32_t sl = ea == 0ull ? sa : (eb == 0ull ? sb : sa);
                                    s_img[sl * 64u + threadIdx.x] = P;
                                    seen.tab[((size_t)bk << 4) + sl] = P;
                                }
                            }
                        } else {
                            for (int h = 0; h < num_hash; ++h) bit_set(bits, index_of(multi_hash(P, (uint32_t)h, kmul), mod));
                        }
                        ++cnt;
                    }
                }
            }
            if (VAR == 1 && seen.dbg) { atomicAdd(seen.dbg, (unsigned long long)dbg_miss); atomicAdd(seen.dbg + 1, (unsigned long long)dbg_fetch); }
        }
    }
    if (n_pairs) {
        unsigned long long c = cnt;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (threadIdx.x == 0 && c) atomicAdd(n_pairs, c);
    }
}

}  // namespace

// one read per lane: pairs at distance `dist` of the reads of words [w0, w0 + nw) into bit filter `f`
void rb::launch_pairs_reads(rb_graph *g, const rb_batch *b, int64_t w0, int64_t nw, int mode_hash, const BitFilter &f, int dist,
                               uint32_t min_len, bool if_present, const uint32_t *chunk_off, uint64_t *out_idx, unsigned long long *pc,
                               hipStream_t st) {
    RB_REQUIRE(g->k <= 64 && b->max_len <= 1024u, "paired k-mers: k <= 64 and reads of at most 1024 bases only on this path");
    // the reads of words [w0, w0 + nw): both ends are read boundaries
    const auto &wo = b->h_woff;
    const int64_t r0 = std::lower_bound(wo.begin(), wo.end(), (uint32_t)w0) - wo.begin();
    const int64_t r1 = std::lower_bound(wo.begin(), wo.end(), (uint32_t)(w0 + nw)) - wo.begin();
    RB_REQUIRE(r0 < (int64_t)wo.size() && wo[(size_t)r0] == (uint32_t)w0 && r1 < (int64_t)wo.size() && wo[(size_t)r1] == (uint32_t)(w0 + nw),
               "launch_pairs: word range does not start and end at read boundaries");
    if (r1 <= r0) return;
    dim3 gr(blocks_for(r1 - r0, 64)), th(64);
    const uint32_t *present = if_present ? g->dbg.bits : nullptr;
    // the seen-pair cache serves plain adds into the filter it belongs to (a gated add counts the pairs that pass the gate: it probes anyway)
    const bool use_seen = f.seen && !out_idx && !present;
    static const uint32_t anchor_bits = getenv("RB_PAIR_SEEN_ANCHOR") ? (uint32_t)std::max(0, std::min(4, atoi(getenv("RB_PAIR_SEEN_ANCHOR")))) : 2u;
    // (RB_DEBUG: the main pipeline's pair counter block has room behind the pair count — add_range prints what it finds there)
    const PairSeen seen{use_seen ? f.seen : nullptr, use_seen ? (1u << f.seen_log2b) - 1u : 0u, (1u << anchor_bits) - 1u,
                        (use_seen && pc == g->pairs_ctr.as<unsigned long long>() && getenv("RB_DEBUG")) ? pc + 1 : nullptr};
#define RB_LAUNCH_PR3(M, NW, V)                                                                                          \
    hipLaunchKernelGGL((k_pairs_reads<M, NW, V>), gr, th, 0, st, b->codes, b->valid, b->woff, b->len, r0, r1 - r0, w0, g->k, \
                       dist, f.bits, f.mod, f.num_hash, kmul_of(g->k), pc, chunk_off, out_idx, min_len, present, g->dbg.mod, g->dbg.num_hash, seen)
#define RB_LAUNCH_PR(M, NW) do { if (out_idx) RB_LAUNCH_PR3(M, NW, 2); else if (use_seen) RB_LAUNCH_PR3(M, NW, 1); else RB_LAUNCH_PR3(M, NW, 0); } while (0)
    if (b->max_len <= 384u) { if (mode_hash == 0) RB_LAUNCH_PR(0, 12); else if (mode_hash == 2) RB_LAUNCH_PR(2, 12); else RB_LAUNCH_PR(1, 12); }
    else { if (mode_hash == 0) RB_LAUNCH_PR(0, 32); else if (mode_hash == 2) RB_LAUNCH_PR(2, 32); else RB_LAUNCH_PR(1, 32); }
#undef RB_LAUNCH_PR
#undef RB_LAUNCH_PR3
}

void rb::launch_pairs(rb_graph *g, const rb_batch *b, int64_t w0, int64_t nw, int mode_hash, const uint32_t *chunk_off,
                      uint64_t *out_idx, unsigned long long *pc, hipStream_t st, const BitFilter *into) {
    if (!st) st = g->stream;
    if (nw <= 0) return;
    const BitFilter &rpk = into ? *into : g->rpk;
    const bool general = g->k > 64 || b->max_len > 384u || (getenv("RB_PAIRS_GENERAL") && atoi(getenv("RB_PAIRS_GENERAL")));
    if (!general) {
        launch_pairs_reads(g, b, w0, nw, mode_hash, rpk, g->read_d, 0u, false, chunk_off, out_idx, pc, st);
        return;
    }
    dim3 gr(blocks_for(nw)), th(TPB);
#ifdef RB_DIAG_PAIRS
    if (getenv("RB_PAIRS_VARIANT") && atoi(getenv("RB_PAIRS_VARIANT")) == 9 && mode_hash == 1) {
        hipLaunchKernelGGL(k_pairs_insert_runtime_branch, gr, th, 0, st, b->codes, b->valid, b->word_read, b->woff, b->len, w0, nw, g->k, g->read_d,
                           rpk.bits, rpk.mod, rpk.num_hash, kmul_of(g->k), pc, chunk_off, out_idx,
                           (b->wpr_uniform && nw % b->wpr_uniform == 0) ? b->wpr_uniform : 0u);
        return;
    }
#endif
#define RB_LAUNCH_PAIRS(M)                                                                                          \
    if (out_idx) RB_LAUNCH_PAIRS2(M, true); else RB_LAUNCH_PAIRS2(M, false)
#define RB_LAUNCH_PAIRS2(M, O)                                                                                      \
    hipLaunchKernelGGL((k_pairs_insert<M, O>), gr, th, 0, st, b->codes, b->valid, b->word_read, b->woff, b->len, w0, nw, \
                       g->k, g->read_d, rpk.bits, rpk.mod, rpk.num_hash, kmul_of(g->k), pc, chunk_off, out_idx, \
                       (b->wpr_uniform && nw % b->wpr_uniform == 0) ? b->wpr_uniform : 0u)
    if (mode_hash == 0) { RB_LAUNCH_PAIRS(0); } else if (mode_hash == 2) { RB_LAUNCH_PAIRS(2); } else { RB_LAUNCH_PAIRS(1); }
#undef RB_LAUNCH_PAIRS
#undef RB_LAUNCH_PAIRS2
}

// is the swept Bloom-bit stage (rb_group.hip) a candidate for this graph at all?  RB_SWEEP: 0 never, 1 wherever it applies; otherwise for a whole
// (unsharded) Bloom filter too large for the caches (the sub-batch decides in run_core)
static bool sweep_wanted(const rb_graph *g) {
    if (g->shard || !g->dbg.bits || g->dbg.lo != 0 || g->dbg.hi != g->dbg.size) return false;
    if (const char *e = getenv("RB_SWEEP")) return atoi(e) != 0;
    return g->dbg.nbytes >= ((int64_t)1 << 31);
}
// Stable grouping of N (h0, occ) records sitting in keys0/vals0 into slot `slot`: sort on the top hash
// bits, draw strengths, run-length encode.  Asynchronous on `st`; group_finish reads the run count.
void rb::group_enqueue(rb_graph *g, int slot, size_t N, uint64_t ordinal0, uint32_t pos_bits, hipStream_t st, DevBuf &temp,
                       DevBuf &ctrbuf, int flags) {
    rb_graph::GroupSlot &S = g->slots[slot];
    S.N = N; S.D = 0; S.flags = flags; S.live = (uint32_t)N;
    ctrbuf.reserve(DEVCTR_BYTES);
    uint32_t *ctr = ctrbuf.as<uint32_t>();
    RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, st));
    if (N == 0) return;
    // Grouping, not ordering, is what the later stages need (rb_group.hip): equal hashes next to each other with
    // their occurrences in sequential order, except where a hash is cut into several runs — the pipeline treats
    // such runs as separate k-mers that share all their bits/counters, which the first-setter arbitration and the
    // ordered conflict replay already make exact (DESIGN.md §Pipeline "split runs").
    const int group_bits = 64 - g->sort_begin_bit;
    const int bucket_target = g->shard ? (getenv("RB_SHARD_GROUP_TARGET") ? atoi(getenv("RB_SHARD_GROUP_TARGET")) : 3072) : 0;
    S.bucket_target = bucket_target;
    temp.reserve(group_temp_bytes(N, group_bits, bucket_target, flags));
    S.keys1.reserve(N * 8); S.valsT.reserve(N * 4); S.vals1.reserve(N * 4); S.tz.reserve(N + 16);
    S.uniq.reserve(N * 8); S.counts.reserve((N + 1) * 4); S.starts.reserve((N + 1) * 4);
    // Index-keyed partition passes (rb_internal.hpp GrIdx): fine buckets are index ranges, runs come out sweeping the filters by first
    // index.  Three Barrett reductions per record and pass (invisible in the partition kernels' times: they wait on LDS and memory) buy
    // locality of the first Bloom bit / first counter of consecutive runs: config 2 probe_claim 56 -> 51 ms, the all-new-k-mers regime of
    // long reads 666 -> 574 ms with the first pass alone (profiles/r04_group_idx.txt).  RB_GROUP_IDX=0 is the hash-keyed partition of
    // rounds 2-3.
    GrIdx gidx{Mod{1, 0, 0}, 0, 0};
    bool want_idx = true;
    if (const char *e = getenv("RB_GROUP_IDX")) want_idx = atoi(e) != 0;
    if (want_idx) {
        if (g->cbf && g->cbf_size > 0) gidx = GrIdx{g->cbf_mod, (uint64_t)g->cbf_lo, (uint64_t)(g->cbf_hi - g->cbf_lo)};
        else if (g->dbg.bits && g->dbg.size > 0) gidx = GrIdx{g->dbg.mod, (uint64_t)g->dbg.lo, (uint64_t)(g->dbg.hi - g->dbg.lo)};
    }
    // the swept Bloom-bit stage (run_core) works on the grouping's own index ranges: they have to be ranges of the Bloom filter's indices
    // (the two filters have equal entries in the configurations this library is sized by; otherwise the grouping goes by the Bloom filter's
    // where the sweep is wanted), and the bucket kernel leaves each range's run slots behind
    GroupExport ex;
    S.sweep_T = 0; S.n_main = 0;
    if (want_idx && sweep_wanted(g)) {
        const GrIdx didx{g->dbg.mod, 0, (uint64_t)g->dbg.size};
        const bool same = gidx.span == didx.span && gidx.lo == 0 && gidx.mod.d == didx.mod.d;
        if (!same) gidx = didx;
        S.sweep_T = group_index_buckets(N, group_bits, bucket_target, flags, gidx);
        if (S.sweep_T) {
            S.brun.reserve(((size_t)4 << S.sweep_T) + 16); S.bnr.reserve(((size_t)4 << S.sweep_T) + 16);
            ex = GroupExport{S.brun.as<uint32_t>(), S.bnr.as<uint32_t>(), ctr + 9};
        }
    }
    uint64_t *kin = g->group_in_keys ? g->group_in_keys : g->keys0.as<uint64_t>();
    uint32_t *vin = g->group_in_vals ? g->group_in_vals : g->vals0.as<uint32_t>();
    g->group_in_keys = nullptr; g->group_in_vals = nullptr;
    group_records_device(kin, vin, S.keys1.as<uint64_t>(), S.valsT.as<uint32_t>(), N, group_bits,
                         g->p.rng_seed, ordinal0, pos_bits, temp.p, temp.cap, S.vals1.as<uint32_t>(), S.tz.as<uint8_t>(), S.uniq.as<uint64_t>(),
                         S.counts.as<uint32_t>(), S.starts.as<uint32_t>(), ctr + 8, st, g, bucket_target, flags, gidx, ex);
}
uint32_t rb::group_finish(rb_graph *g, int slot, hipStream_t st, DevBuf &temp, DevBuf &ctrbuf, hipStream_t scan_stream) {
    rb_graph::GroupSlot &S = g->slots[slot];
    if (S.N == 0) { RB_HIP(hipStreamSynchronize(st)); return 0; }
    uint32_t D = 0;
    RB_HIP(hipMemcpyAsync(&D, ctrbuf.as<uint32_t>() + 8, 4, hipMemcpyDeviceToHost, st));
    if (S.sweep_T) RB_HIP(hipMemcpyAsync(&S.n_main, ctrbuf.as<uint32_t>() + 9, 4, hipMemcpyDeviceToHost, st));
    if (S.flags & GR_FLAG_DEAD)
        RB_HIP(hipMemcpyAsync(&S.live, group_live_count(temp.p, S.N, 64 - g->sort_begin_bit, S.bucket_target, S.flags), 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    RB_REQUIRE(D < (1u << 30), "sub-batch has too many distinct k-mers (%u)", D);
    S.D = D;
    (void)scan_stream;   // the run starts come out of the grouping kernel
    if (getenv("RB_DEBUG") && temp.p) {
        uint32_t nb = 0, mx = 0; uint64_t rec = 0;
        group_debug_big(temp.p, S.N, 64 - g->sort_begin_bit, S.bucket_target, &nb, &rec, &mx, S.flags);
        fprintf(stderr, "[rb] grouping: N=%zu runs=%u; buckets that did not fit LDS: %u with %llu records, largest %u\n", S.N, D, nb, (unsigned long long)rec, mx);
    }
    return D;
}
uint32_t rb::group_records(rb_graph *g, size_t N, uint64_t ordinal0, uint32_t pos_bits, rb_add_stats *stats, uint32_t **ctr_out) {
    group_enqueue(g, g->cur, N, ordinal0, pos_bits, g->stream, g->temp, g->devctr);
    const uint32_t D = group_finish(g, g->cur, g->stream, g->temp, g->devctr, g->stream);
    if (ctr_out) *ctr_out = g->devctr.as<uint32_t>();
    if (stats) stats->distinct += D;
    return D;
}

namespace {

}  // namespace
// empty(): the filters of config 2 and the prefilter cache are 14.7 GB; the runtime's fill kernel writes them at 1.7 TB/s, plain
// 16-byte stores from every CU at about twice that
__global__ void __launch_bounds__(256) k_zero16(uint4 *__restrict__ p, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
}
void rb::fast_zero(void *p, size_t bytes, hipStream_t s) {
    if (!p || !bytes) return;
    if (bytes < ((size_t)64 << 20) || (reinterpret_cast<uintptr_t>(p) & 15u)) { RB_HIP(hipMemsetAsync(p, 0, bytes, s)); return; }
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_zero16, dim3(256 * 16), dim3(256), 0, s, static_cast<uint4 *>(p), n16);
    RB_HIP(hipGetLastError());
    if (bytes & 15u) RB_HIP(hipMemsetAsync(static_cast<char *>(p) + n16 * 16, 0, bytes & 15u, s));
}

void rb::alloc_bits(BitFilter &f, int64_t bits, int num_hash, int64_t lo, int64_t hi) {
    f.size = bits;
    f.lo = lo; f.hi = hi;
    const int64_t local = hi - lo;
    f.nbytes = local / 8 + ((local % 8) ? 1 : 0);   // UnsafeBitBuffer.java:34-37 (whole filter when lo=0,hi=bits)
    f.alloc = (((size_t)f.nbytes + 3) / 4 + 1) * 4;
    f.num_hash = num_hash;
    f.mod = make_mod((uint64_t)bits);
    RB_HIP(hipMalloc(&f.bits, f.alloc));
    RB_HIP(hipMemset(f.bits, 0, f.alloc));
    RB_HIP(hipDeviceSynchronize());   // hipMemset is asynchronous; the graph's stream is non-blocking
}   // (the bit filters do not go through alloc_best_placed: tried twice, before and after the counting filter's draw: no gain, profiles/r03_alloc_lottery.txt)
// ---- where the counting filter's pages land ----
// The stages that touch the counting filter at random (claims, counter stores) take 53-68 ms per step on config 2 depending on the
// ALLOCATION: graphs made one after the other in one process differ like that, each keeps its time for as long as it lives, and the time
// follows what a short kernel of random read-modify-write atomics measures on the fresh allocation (24.8 ... 30.6 ms for 2 x 2^28 XORs;
// random READS do not differ: tools/alloc_lottery.py, profiles/r03_alloc_lottery.txt).  Which physical pages an allocation gets is the
// driver's business; what the library can do is look: up to RB_ALLOC_TRIES (default 8; 1 = take the first) allocations are made, each
// while the earlier ones are still held so that it gets other pages, each timed with 2 x 2^26 random XOR pairs (the second pass
// restores the zeros), and the fastest is kept.  Only for filters of 1 GB and more, and only while the device has room for the copies.
namespace {
__global__ void k_alloc_probe(uint32_t *words, uint64_t n_words, uint32_t per_thread, unsigned long long *sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = t * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        acc |= atomicXor(&words[(uint64_t)(((unsigned __int128)x * n_words) >> 64)], 0x80808080u);
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);          // (keeps the returns alive)
}
}  // namespace
void *rb::alloc_best_placed(size_t bytes, const char *what) {
    int tries = getenv("RB_ALLOC_TRIES") ? atoi(getenv("RB_ALLOC_TRIES")) : 8;
    tries = std::max(1, std::min(16, tries));
    if (bytes < ((size_t)1 << 30)) tries = 1;
    // everything the trial holds is released on every way out, an exception included (several multi-GB candidates otherwise)
    struct Held {
        void *best = nullptr, *cand = nullptr;
        unsigned long long *sink = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Held() {
            if (best) (void)hipFree(best);
            if (cand) (void)hipFree(cand);
            if (sink) (void)hipFree(sink);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } H;
    std::vector<void *> losers;                               // held while later candidates are drawn (so that they get other pages)
    struct Losers { std::vector<void *> &v; ~Losers() { for (void *p : v) (void)hipFree(p); } } losers_guard{losers};
    float best_ms = 0;
    if (tries > 1) { RB_HIP(hipMalloc(&H.sink, 64)); RB_HIP(hipEventCreate(&H.e0)); RB_HIP(hipEventCreate(&H.e1)); }
    for (int t = 0; t < tries; ++t) {
        if (t) {
            size_t free_b = 0, total_b = 0;
            RB_HIP(hipMemGetInfo(&free_b, &total_b));
            if (free_b < bytes + total_b / 4) break;           // leave a quarter of the device to everything else
        }
        if (hipMalloc(&H.cand, bytes) != hipSuccess) { H.cand = nullptr; (void)hipGetLastError(); if (t) break; RB_HIP(hipErrorOutOfMemory); }
        RB_HIP(hipMemset(H.cand, 0, bytes));
        float ms = 0;
        if (tries > 1) {
            RB_HIP(hipDeviceSynchronize());
            RB_HIP(hipEventRecord(H.e0, nullptr));
            for (int pass = 0; pass < 2; ++pass)               // the same places twice: zeros again afterwards
                hipLaunchKernelGGL(k_alloc_probe, dim3(16384), dim3(256), 0, nullptr, static_cast<uint32_t *>(H.cand), (uint64_t)(bytes / 4), 16u, H.sink);
            RB_HIP(hipEventRecord(H.e1, nullptr));
            RB_HIP(hipEventSynchronize(H.e1));
            RB_HIP(hipEventElapsedTime(&ms, H.e0, H.e1));
        }
        if (getenv("RB_ALLOC_DEBUG")) fprintf(stderr, "[rb] %s allocation %d: %.3f ms for 2 x 2^26 random XORs\n", what, t, ms);
        if (!H.best || ms < best_ms) { if (H.best) losers.push_back(H.best); H.best = H.cand; best_ms = ms; }
        else losers.push_back(H.cand);
        H.cand = nullptr;
    }
    RB_HIP(hipDeviceSynchronize());
    void *best = H.best;
    H.best = nullptr;
    return best;
}

void rb::free_bits(BitFilter &f) { if (f.bits) (void)hipFree(f.bits); if (f.seen) (void)hipFree(f.seen); f = BitFilter(); }
// Seen-pair cache (k_pairs_reads): one 128-byte bucket of 16 pair hashes per 512 filter bits, 2^8 .. 2^24 buckets (config 2: 2 GB beside
// the 1.07 GB filter); RB_PAIR_SEEN=0 none, RB_PAIR_SEEN=<log2 buckets> that many.
void rb::alloc_pair_seen(BitFilter &f) {
    if (!f.bits || f.seen) return;
    uint32_t lb = std::max(8u, std::min(24u, log2_ceil((uint64_t)std::max<int64_t>(f.size / 512, 1))));
    if (const char *e = getenv("RB_PAIR_SEEN")) { const int v = atoi(e); if (v <= 0) return; lb = (uint32_t)std::max(4, std::min(26, v)); }
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && ((size_t)128 << lb) > fr / 4) return;          // a hint, never at the expense of the filters
    if (hipMalloc(&f.seen, (size_t)128 << lb) != hipSuccess) { (void)hipGetLastError(); f.seen = nullptr; return; }
    RB_HIP(hipMemset(f.seen, 0, (size_t)128 << lb));
    RB_HIP(hipDeviceSynchronize());
    f.seen_log2b = lb;
}
void rb::seen_reset(BitFilter &f, hipStream_t s) {
    if (f.seen) fast_zero(f.seen, (size_t)128 << f.seen_log2b, s);
}
namespace {


}  // namespace
BitFilter *rb::bit_filter(rb_graph *g, int which) {
    switch (which) { case RB_DBGBF: return &g->dbg; case RB_RPKBF: return &g->rpk; case RB_FPKBF: return &g->fpk; default: return nullptr; }
}
namespace {

// The order-exact pipeline over N (h0, occurrence) records already sitting in keys0/vals0.
void run_core(rb_graph *g, size_t N, uint32_t D, int mode, uint64_t ordinal0, uint32_t pos_bits, rb_add_stats *stats,
              const std::function<void()> *after_resolve = nullptr);
}  // namespace
void rb::run_pipeline(rb_graph *g, size_t N, int mode, uint64_t ordinal0, uint32_t pos_bits, rb_add_stats *stats) {
    if (N == 0) return;
    const uint32_t D = group_records(g, N, ordinal0, pos_bits, stats, nullptr);
    run_core(g, N, D, mode, ordinal0, pos_bits, stats);
}
namespace {
// stages A/B + heavy + conflict path on the grouped sub-batch in slot g->cur (consumer stream)
void run_core(rb_graph *g, size_t N, uint32_t D, int mode, uint64_t ordinal0, uint32_t pos_bits, rb_add_stats *stats,
              const std::function<void()> *after_resolve) {
    hipStream_t s = g->stream;
    if (N == 0 || D == 0) { if (after_resolve) (*after_resolve)(); return; }
    g->devctr.reserve(DEVCTR_BYTES);
    uint32_t *ctr = g->devctr.as<uint32_t>();
    RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
    FilterView fv = g->view(ordinal0, pos_bits);
    const uint64_t *uniq = g->uniq().as<uint64_t>();
    const uint32_t *counts = g->counts().as<uint32_t>(), *starts = g->starts().as<uint32_t>(), *vals = g->vals1().as<uint32_t>();
    g->status.reserve((size_t)D * 4); g->nops.reserve((size_t)D * 4); g->cvals.reserve((size_t)D * 8);
    g->heavy.reserve((size_t)D * 4); g->confk.reserve((size_t)D * 4);
    g->foreign.reserve((size_t)D * 8 * (size_t)g->cbf_h);
    uint32_t *status = g->status.as<uint32_t>(), *nops = g->nops.as<uint32_t>();
    const bool uses_dbg = (mode == M_ADD || mode == M_ADD_IF_ABSENT);
    uint32_t f_log2 = 1, c_log2 = 1, csf_log2 = 16;
    uint32_t *csf = nullptr;                     // bit filter over the contested counters, behind the table in g->ctable
    const bool full_table = getenv("RB_FIRST_SETTER_TABLE") && atoi(getenv("RB_FIRST_SETTER_TABLE"));   // the older scheme: every missing bit gets an entry
    const bool collide = uses_dbg && !full_table;
    Slot *ftab = nullptr;                        // first-setter arbitration table (null: nothing to arbitrate)
    FtFilter ffl{nullptr, 0};                    // bit filter in front of it (large tables only)
    if (uses_dbg && full_table) {
        g->prof_begin();
        f_log2 = log2_ceil(2ull * (uint64_t)D * (uint64_t)g->dbg.num_hash + 2);
        g->ftable.reserve(sizeof(Slot) << f_log2);
        RB_HIP(hipMemsetAsync(g->ftable.p, 0xFF, sizeof(Slot) << f_log2, s));
        ftab = g->ftable.as<Slot>();
        g->prof_end("table_clear");
    }
    g->prof_begin();
    // Swept Bloom-bit stage (rb_group.hip): worth the two passes over the filter it costs when the sub-batch has a run per <= 64 bytes of
    // filter — the all-new-k-mers regime of long reads, where the random second probes are three quarters of the insert.  RB_SWEEP=1 forces
    // it wherever it applies, 0 turns it off.
    const rb_graph::GroupSlot &GS = g->slots[g->cur];
    bool swept = false, swept_all = false;       // swept_all: no run of an oversized bucket among them (those probe outside the sweep)
    uint32_t *involved = nullptr, n_involved_max = 0;   // swept_all: list of the runs with a probe that met another one (device count in ctr[12]; at most n_involved_max)
    if (collide && !ftab && fv.dbg_h == 2 && fv.cbf_h == 2 && !getenv("RB_PROBE_GENERIC") && GS.sweep_T && !g->shard && sweep_wanted(g) &&
        ((uint64_t)g->dbg.size >> GS.sweep_T) < (1ull << 31)) {       // (the sweep counts a range's bits in 32 bits)
        const char *e = getenv("RB_SWEEP");
        // (and most runs new: where the sub-batch before found most of its k-mers present the sweep only rewrites set bits — the plain loads are cheaper)
        static const float present_max = getenv("RB_SWEEP_PRESENT") ? (float)atof(getenv("RB_SWEEP_PRESENT")) : 0.5f;
        swept = (e && atoi(e) == 1) || ((uint64_t)D * 64ull >= (uint64_t)g->dbg.nbytes && g->last_present_frac < present_max);
    }
    if (swept) {
        const GrIdx didx{g->dbg.mod, 0, (uint64_t)g->dbg.size};
        swept_all = GS.n_main >= D;
        const size_t st_pad = ((size_t)D + 15) & ~(size_t)15;
        g->sw_st.reserve(st_pad + D + 16);
        g->sw_temp.reserve(sweep_temp_bytes(D, GS.sweep_T));
        uint8_t *st0 = g->sw_st.as<uint8_t>(), *st1 = st0 + st_pad;
        // scratch for the (h1, run) records: arrays nobody reads before this stage is over (contested indices are written by the probe kernel below, op counts / lists in stage B)
        uint64_t *ka = g->foreign.as<uint64_t>(), *kb = ka + D;
        sweep_bits_device(g->dbg.bits, didx, GS.sweep_T, fv.kmul, uniq, D, GS.n_main, GS.brun.as<uint32_t>(), GS.bnr.as<uint32_t>(), ka, g->nops.as<uint32_t>(), kb,
                          g->heavy.as<uint32_t>(), g->sw_temp.p, g->sw_temp.cap, st0, st1, s);
        // counters without claims: "probe j of the counting filter has the index of probe j of the Bloom filter" needs the two filters to map a
        // hash the same way — same size AND the same reduction constants (index_of), not only equal sizes
        const bool same_index = (int64_t)g->dbg.size == g->cbf_size && g->cbf_lo == 0 && g->dbg.mod.d == g->cbf_mod.d && g->dbg.mod.m_lo == g->cbf_mod.m_lo &&
                                g->dbg.mod.m_hi == g->cbf_mod.m_hi && fv.dbg_h == fv.cbf_h;
        const bool plain_ok = swept_all && same_index && !(getenv("RB_SWEEP_CLAIMS") && atoi(getenv("RB_SWEEP_CLAIMS")));
        constexpr int RUNS = 2;
        hipLaunchKernelGGL((k_probe_h2<RUNS, true>), dim3(blocks_for(((int64_t)D + RUNS - 1) / RUNS)), dim3(TPB), 0, s, fv, uniq, counts, D, mode, status,
                           g->cvals.as<uint64_t>(), g->foreign.as<uint64_t>(), ctr, (const uint8_t *)st0, (const uint8_t *)st1, swept_all ? 1u : 0u,
                           plain_ok ? 1u : 0u);
    } else if (!ftab && fv.dbg_h == 2 && fv.cbf_h == 2 && !getenv("RB_PROBE_GENERIC")) {
        constexpr int RUNS = 2;
        hipLaunchKernelGGL((k_probe_h2<RUNS, false>), dim3(blocks_for(((int64_t)D + RUNS - 1) / RUNS)), dim3(TPB), 0, s, fv, uniq, counts, D, mode, status,
                           g->cvals.as<uint64_t>(), g->foreign.as<uint64_t>(), ctr);
    } else
        hipLaunchKernelGGL(k_probe, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, counts, starts, vals, D, mode,
                           ftab, f_log2, status, g->cvals.as<uint64_t>(), g->foreign.as<uint64_t>(), ctr);
    if (collide) {
        // set the new bits now; the probes that meet another probe of the sub-batch on a bit are counted
        if (!swept) hipLaunchKernelGGL(k_set_bits, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, D, status, ctr);
        uint32_t spread[16 * 32], n_collide = 0;
        RB_HIP(hipMemcpyAsync(spread, ctr + 16, sizeof spread, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < 32; ++q) n_collide += spread[16 * q + 1];
        if (n_collide) {   // entries for the collided bits only: the colliders, then the probe that got there first
            f_log2 = log2_ceil(4ull * (uint64_t)n_collide + 2);
            // a bit filter in front of it once the table outgrows the caches (<= 12 % full; config 2's 13 MB table does without: measured neutral)
            const size_t tab_bytes = sizeof(Slot) << f_log2;
            const uint32_t ff_log2 = std::max(16u, std::min(30u, log2_ceil(8ull * (uint64_t)n_collide)));
            const bool use_ff = getenv("RB_FT_FILTER") ? atoi(getenv("RB_FT_FILTER")) != 0 : tab_bytes > ((size_t)32 << 20);
            g->ftable.reserve(tab_bytes + (use_ff ? ((size_t)1 << (ff_log2 - 3)) : 0));
            RB_HIP(hipMemsetAsync(g->ftable.p, 0xFF, tab_bytes, s));
            ftab = g->ftable.as<Slot>();
            if (use_ff) {
                uint32_t *fb = reinterpret_cast<uint32_t *>(static_cast<char *>(g->ftable.p) + tab_bytes);
                RB_HIP(hipMemsetAsync(fb, 0, (size_t)1 << (ff_log2 - 3), s));
                ffl = FtFilter{fb, ff_log2};
            }
            if (swept_all) {
                // after the sweep the runs with anything to do here are marked — a few per cent, one or two lanes of most wavefronts, each a chain of
                // table atomics its whole wavefront waits for: they are listed (a scan, not an atomic counter) and the table kernels walk the list
                // (k_collide_insert took 2.7 ms of a 390 M-run sub-batch over all runs; the list is the conflict list's buffer, free until stage B's compaction)
                involved = g->confk.as<uint32_t>(); n_involved_max = n_collide;
                g->temp.reserve(select_temp_bytes(D));
                select_flagged(g->temp.p, g->temp.cap, status, (3u << ST_COLLIDE_SHIFT) | (3u << ST_JOIN_SHIFT), D, involved, ctr + 12, s);
                hipLaunchKernelGGL(k_collide_insert, dim3(blocks_for(std::min(D, n_collide))), dim3(TPB), 0, s, fv, uniq, starts, vals, std::min(D, n_collide), status, ftab, f_log2,
                                   const_cast<uint32_t *>(ffl.bits), ffl.log2, (const uint32_t *)involved, (const uint32_t *)(ctr + 12));
            } else
            hipLaunchKernelGGL(k_collide_insert, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, starts, vals, D, status, ftab, f_log2,
                               const_cast<uint32_t *>(ffl.bits), ffl.log2);
            if (!swept_all) hipLaunchKernelGGL(k_collide_fixup, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, starts, vals, D, status, ftab, f_log2, ffl);
        }
    }
    if (mode == M_ADD) {
        if (swept_all) {      // only a run with a marked probe can find its bits set earlier: the list again (nothing at all where no two probes met)
            if (involved) hipLaunchKernelGGL(k_late_claim, dim3(blocks_for(std::min(D, n_involved_max))), dim3(TPB), 0, s, fv, uniq, starts, vals, std::min(D, n_involved_max), ftab, f_log2,
                                             status, g->cvals.as<uint64_t>(), g->foreign.as<uint64_t>(), ctr, ffl, (const uint32_t *)involved, (const uint32_t *)(ctr + 12));
        } else
        hipLaunchKernelGGL(k_late_claim, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, starts, vals, D, ftab, f_log2,
                           status, g->cvals.as<uint64_t>(), g->foreign.as<uint64_t>(), ctr, ffl);
    }
    uint32_t n_foreign = 0;
    {
        uint32_t spread[16 * 32];
        RB_HIP(hipMemcpyAsync(spread, ctr + 16, sizeof spread, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        uint32_t n_present = 0;
        for (int q = 0; q < 32; ++q) { n_foreign += spread[16 * q]; n_present += spread[16 * q + 2]; }
        const bool counted = !(uses_dbg && full_table) && fv.dbg_h == 2 && fv.cbf_h == 2 && !getenv("RB_PROBE_GENERIC");
        g->last_present_frac = counted ? (float)n_present / (float)D : -1.0f;      // (-1: the generic probe kernel does not count them)
    }
    if (getenv("RB_DEBUG") && swept) fprintf(stderr, "[rb] swept stage: %u runs (%u of oversized buckets), %u contested counters\n", D, D - std::min(D, GS.n_main), n_foreign);
    g->prof_end("probe_claim");
    if (n_foreign) {   // the set of counters claimed by more than one run
        g->prof_begin();
        c_log2 = log2_ceil(2ull * (uint64_t)n_foreign + 2);
        csf_log2 = std::max(16u, std::min(25u, log2_ceil(8ull * (uint64_t)n_foreign)));   // <= 12 % full, <= 4 MB
        const size_t tab_bytes = sizeof(Slot) << c_log2;
        g->ctable.reserve(tab_bytes + ((size_t)1 << (csf_log2 - 3)));
        csf = reinterpret_cast<uint32_t *>(static_cast<char *>(g->ctable.p) + tab_bytes);
        RB_HIP(hipMemsetAsync(g->ctable.p, 0xFF, tab_bytes, s));
        RB_HIP(hipMemsetAsync(csf, 0, (size_t)1 << (csf_log2 - 3), s));
        hipLaunchKernelGGL(k_cs_build, dim3(blocks_for(D)), dim3(TPB), 0, s, status, g->foreign.as<uint64_t>(), D, g->cbf_h,
                           g->ctable.as<Slot>(), c_log2, csf, csf_log2);
        g->prof_end("conflict_set");
    }
    g->prof_begin();
    hipLaunchKernelGGL(k_resolve_apply, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, counts, starts, vals, D, mode, g->light_ops,
                       ftab, f_log2, ffl, (int)collide, g->ctable.as<Slot>(), c_log2, getenv("RB_NO_CS_FILTER") ? (uint32_t *)nullptr : csf, csf_log2, n_foreign, status, nops,
                       g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), getenv("RB_DEBUG") ? reinterpret_cast<float *>(ctr + 700) : (float *)nullptr);
    g->prof_end("resolve_apply");
    // the runs that own their counters alone have updated counters and prefilter cache: the producer may
    // filter the next sub-batch now (heavy and replayed runs follow below, overlapped with it) — unless runs with shared counters
    // were set aside: most of them are finished by k_resolve_deferred a few short kernels further down, and their cache
    // updates (hot k-mers among them) are worth waiting for (released early, the next sub-batch keeps 10 % more conflicting ops)
    bool released = false;
    auto release = [&] { if (!released && after_resolve) (*after_resolve)(); released = true; };
    const bool old_rule = getenv("RB_ORDER_ALL_SHARED") != nullptr;
    if (!n_foreign || old_rule || getenv("RB_RELEASE_EARLY")) release();
    if (getenv("RB_DEBUG")) {
        float df[128];
        RB_HIP(hipMemcpyAsync(df, ctr + 700, sizeof df, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        double dr = 0, tot = 0;
        for (int q = 0; q < 64; ++q) { dr += df[2 * q]; tot += df[2 * q + 1]; }
        unsigned long long hh[96];
        RB_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_dbg_hist), sizeof hh));
        // (dr assumes UNFILTERED ops: with the prefilter on, the ops of a correctly cached k-mer are the ones that succeed; what a
        // perfect cache would still drop are the no-op shares of the k-mers it does not hold — column 0 of the table below)
        double miss = 0;
        for (int s0 = 1; s0 < 8; ++s0) miss += (double)hh[s0 * 8] * (1.0 - 1.0 / (double)(1u << s0));
        fprintf(stderr, "[rb] ops a perfect cache would drop: %.1f%% of %.0f ops (N=%zu; %.1f%% if no occurrence had been filtered)\n",
                tot ? 100.0 * miss / tot : 0.0, tot, N, tot ? 100.0 * dr / tot : 0.0);
        for (int s0 = 1; s0 < 8; ++s0) {
            fprintf(stderr, "[rb]   true exponent %d: ops by cached exponent 0..7 (M):", s0);
            for (int sc = 0; sc < 8; ++sc) fprintf(stderr, " %.1f", hh[s0 * 8 + sc] / 1e6);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[rb]   uncached ops by occupied slots of their bucket 0..16 (M):");
        for (int q = 0; q <= 16; ++q) fprintf(stderr, " %.1f", hh[64 + q] / 1e6);
        fprintf(stderr, "\n");
        memset(hh, 0, sizeof hh);
        RB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_hist), hh, sizeof hh));
    }
    g->prof_begin();
    g->temp.reserve(select2_temp_bytes(D));
    select_flagged2(g->temp.p, g->temp.cap, status, D, RUN_HEAVY, g->heavy.as<uint32_t>(), RUN_CONFLICT, g->confk.as<uint32_t>(), ctr + 0, s);
    uint32_t hc[2] = {0, 0};
    RB_HIP(hipMemcpyAsync(hc, ctr, 8, hipMemcpyDeviceToHost, s));
    RB_HIP(hipStreamSynchronize(s));
    g->prof_end("compact_lists");
    if (hc[1] && !old_rule) {
        // The runs with a shared counter were set aside by stage B (list confk).  Which of them need the ordered replay: reach
        // counts per shared counter, then the closure (2-3 rounds over the list).  The others are finished here — light ones on
        // the spot, long ones by a second k_cbf_heavy launch.  RB_ORDER_ALL_SHARED=1: every one of them is replayed (the old rule).
        g->prof_begin();
        const uint32_t nc0 = hc[1];
        const size_t slots = (size_t)1 << c_log2;
        g->cwriters.reserve(slots * 8 + 64);
        const size_t cs_stride = ((size_t)nc0 + 1 + 3) / 4 * 4;                     // (16-byte aligned sub-arrays: the scans below take their vectorised path)
        const size_t hh = (size_t)fv.cbf_h;
        g->cshared.reserve(cs_stride * 4 * (5 + hh) + 64);
        RB_HIP(hipMemsetAsync(g->cwriters.p, 0, slots * 8 + 64, s));
        uint32_t *writers = g->cwriters.as<uint32_t>(), *cflag = writers + slots, *changed = cflag + slots;
        uint32_t *ordered = g->cshared.as<uint32_t>(), *heavy2 = ordered + cs_stride, *pos_o = heavy2 + cs_stride, *pos_h = pos_o + cs_stride,
                 *heavy2_list = pos_h + cs_stride, *cslot = heavy2_list + cs_stride;      // cslot: [nc0][h] table slots of the candidates' counters
        RB_HIP(hipMemsetAsync(ordered, 0, cs_stride * 4 * 2, s));
        const CsLookup L{g->ctable.as<Slot>(), c_log2, csf, csf_log2};
        const uint32_t *cand = g->confk.as<uint32_t>();
        hipLaunchKernelGGL(k_cs_writers, dim3(blocks_for(nc0)), dim3(TPB), 0, s, fv, uniq, nops, cand, nc0, g->cvals.as<uint64_t>(), L, writers, cslot);
        // the closure takes 2-3 rounds plus the one that finds nothing new; a round after the fixed point changes nothing, so the
        // rounds of a SHORT list go out three at a time and only the last one's flag is read back (one host round trip instead of three or
        // four); where a round is worth more than a round trip (a million candidates and more: long reads) the flag is read after every round
        const int per_look = nc0 >= (1u << 20) ? 1 : 3;
        for (int round = 0;;) {
            int everything = 0;
            for (int q = 0; q < per_look; ++q, ++round) {
                everything = round >= 16 ? 1 : 0;
                hipLaunchKernelGGL(k_cs_order, dim3(blocks_for(nc0)), dim3(TPB), 0, s, (int)fv.cbf_h, nc0, cslot, writers, cflag, ordered, changed + (round & 7), everything);
                if (everything) { ++round; break; }
            }
            const int last = round - 1;
            uint32_t ch = 0;
            RB_HIP(hipMemcpyAsync(&ch, changed + (last & 7), 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            if (!ch || everything) break;
            RB_HIP(hipMemsetAsync(changed, 0, 32, s));
        }
        hipLaunchKernelGGL(k_resolve_deferred, dim3(blocks_for(nc0)), dim3(TPB), 0, s, fv, uniq, counts, starts, vals, cand, nc0, ordered, mode, g->light_ops, cslot,
                           status, nops, g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), heavy2);
        g->prof_end("conflict_set");
        release();                                                             // (enqueues the next sub-batch's producer work; may wait on the host)
        g->prof_begin();
        g->temp.reserve(scan_temp_bytes((size_t)nc0 + 1));
        exclusive_scan_u32(g->temp.p, g->temp.cap, ordered, pos_o, (size_t)nc0 + 1, s);
        exclusive_scan_u32(g->temp.p, g->temp.cap, heavy2, pos_h, (size_t)nc0 + 1, s);
        g->kk1.reserve((size_t)nc0 * 8);                                       // (free until the conflict path proper: the replayed runs, in run order)
        uint32_t *conf2 = g->kk1.as<uint32_t>();
        hipLaunchKernelGGL(k_list_compact, dim3(blocks_for(nc0)), dim3(TPB), 0, s, cand, ordered, pos_o, nc0, conf2);
        hipLaunchKernelGGL(k_list_compact, dim3(blocks_for(nc0)), dim3(TPB), 0, s, cand, heavy2, pos_h, nc0, heavy2_list);
        uint32_t n2[2] = {0, 0};
        RB_HIP(hipMemcpyAsync(&n2[0], pos_o + nc0, 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(&n2[1], pos_h + nc0, 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        if (n2[0]) RB_HIP(hipMemcpyAsync(g->confk.p, conf2, (size_t)n2[0] * 4, hipMemcpyDeviceToDevice, s));
        if (n2[1]) {
            RB_HIP(hipMemcpyAsync(changed + 12, &n2[1], 4, hipMemcpyHostToDevice, s));   // k_cbf_heavy takes the length from the device
            hipLaunchKernelGGL(k_cbf_heavy, dim3(std::min<uint32_t>(n2[1], 262144u)), dim3(64), 0, s, fv, uniq, counts, starts,
                               vals, status, nops, g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), heavy2_list, changed + 12, (uint64_t *)nullptr);
        }
        hc[1] = n2[0];
        g->prof_end("conflict_set");
    }
    release();
    if (hc[0]) {
        g->prof_begin();
        hipLaunchKernelGGL(k_cbf_heavy, dim3(std::min<uint32_t>(hc[0], 262144u)), dim3(64), 0, s, fv, uniq, counts, starts,
                           vals, status, nops, g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), g->heavy.as<uint32_t>(), ctr, (uint64_t *)nullptr);
        g->prof_end("cbf_heavy");
    }
    if (hc[1]) {
        const uint32_t nck = hc[1];
        g->prof_begin();
        g->conf_sizes.reserve(((size_t)nck + 1) * 4); g->conf_off.reserve(((size_t)nck + 1) * 4);
        hipLaunchKernelGGL(k_conf_offsets, dim3(blocks_for(nck + 1)), dim3(TPB), 0, s, g->confk.as<uint32_t>(), nops, nck, g->conf_sizes.as<uint32_t>());
        g->temp.reserve(scan_temp_bytes((size_t)nck + 1));
        exclusive_scan_u32(g->temp.p, g->temp.cap, g->conf_sizes.as<uint32_t>(), g->conf_off.as<uint32_t>(), (size_t)nck + 1, s);
        uint32_t nco = 0;
        RB_HIP(hipMemcpyAsync(&nco, g->conf_off.as<uint32_t>() + nck, 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        g->opk0.reserve((size_t)nco * 8); g->opk1.reserve((size_t)nco * 8);
        g->opv0.reserve((size_t)nco * 4); g->opv1.reserve((size_t)nco * 4);
        g->label.reserve((size_t)D * 4); g->kk0.reserve((size_t)nck * 8); g->kk1.reserve((size_t)nck * 8);
        g->biglist.reserve((size_t)nck * 4);
        const uint32_t *confk = g->confk.as<uint32_t>();
        uint32_t *label = g->label.as<uint32_t>();
        hipLaunchKernelGGL(k_conf_release, dim3(blocks_for(nck)), dim3(TPB), 0, s, fv, uniq, confk, nck);
        // components by min-label propagation; ctr[3] = changed flag, ctr[4] = number of big components
        // (table slots of the runs' counters, looked up once; the scratch is opk1's — free until the sort of the replay keys below)
        g->opk1.reserve(std::max((size_t)nco * 8, (size_t)nck * fv.cbf_h * 4));
        uint32_t *lslot = g->opk1.as<uint32_t>();
        hipLaunchKernelGGL(k_label_init, dim3(blocks_for(nck)), dim3(TPB), 0, s, fv, uniq, confk, nck, g->ctable.as<Slot>(), c_log2, label, lslot);
        // one kernel per round (pull, one pointer jump, push); the flag is looked at after every round where a round is long (a million runs
        // and more), after every second one otherwise — most components are pairs and settle in two rounds
        const int per_look = nck >= (1u << 20) ? 1 : 2;
        for (int it = 0;; ++it) {
            RB_REQUIRE(it < 100000, "component labelling did not converge");
            uint32_t *flag = ctr + 900 + (it & 31);                          // a fresh flag per look: 32 of them zeroed by one fill
            if ((it & 31) == 0) RB_HIP(hipMemsetAsync(ctr + 900, 0, 128, s));
            for (int q = 0; q < per_look; ++q)
                hipLaunchKernelGGL(k_label_round, dim3(blocks_for(nck)), dim3(TPB), 0, s, (int)fv.cbf_h, confk, nck, g->ctable.as<Slot>(), lslot, label, flag);
            uint32_t changed = 0;
            RB_HIP(hipMemcpyAsync(&changed, flag, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            if (!changed) break;
        }
        g->prof_end("conflict_components");
        g->prof_begin();
        hipLaunchKernelGGL(k_conf_kmer_keys, dim3(blocks_for(nck)), dim3(TPB), 0, s, confk, label, nck, g->kk0.as<uint64_t>());
        g->temp.reserve(std::max(sort_pairs_temp_bytes(nco), sort_keys_temp_bytes(nck)));
        hipLaunchKernelGGL(k_conf_expand, dim3(blocks_for((int64_t)nck * 8)), dim3(TPB), 0, s, confk, g->conf_off.as<uint32_t>(),
                           counts, starts, vals, status, nops, label, nck, g->opk0.as<uint64_t>(), g->opv0.as<uint32_t>());
        // labels are run numbers (< D); the list is in run order and the sort is stable, so the label bits suffice
        const int label_end = 32 + (int)std::max(1u, log2_ceil((uint64_t)D));
        sort_keys_u64(g->temp.p, g->temp.cap, g->kk0.as<uint64_t>(), g->kk1.as<uint64_t>(), nck, 32, label_end, s);
        // op keys are (component label << 32) | occurrence id, and occurrence ids stop at occ_bits: two bit ranges
        sort_pairs_u64_u32_2r(g->temp.p, g->temp.cap, g->opk0.as<uint64_t>(), g->opk1.as<uint64_t>(),
                              g->opv0.as<uint32_t>(), g->opv1.as<uint32_t>(), nco, 0, (int)std::min(32u, std::max(1u, g->occ_bits)), 32, label_end, s);
        g->prof_end("conflict_gather_sort");
        g->prof_begin();
        RB_HIP(hipMemsetAsync(ctr + 4, 0, 4, s));
        hipLaunchKernelGGL(k_conf_replay_small, dim3(blocks_for(nck)), dim3(TPB), 0, s, fv, uniq, g->kk1.as<uint64_t>(), nck,
                           g->opk1.as<uint64_t>(), g->opv1.as<uint32_t>(), nco, g->biglist.as<uint32_t>(), ctr + 4, (int)(mode != M_COUNT_ONLY), starts, vals, g->small_ops);
        hipLaunchKernelGGL(k_conf_replay_big, dim3(std::min<uint32_t>(nck, 262144u)), dim3(64), 0, s, fv, uniq, g->kk1.as<uint64_t>(), nck,
                           g->opk1.as<uint64_t>(), g->opv1.as<uint32_t>(), nco, g->biglist.as<uint32_t>(), ctr + 4,
                           getenv("RB_DEBUG") ? ctr + 600 : (uint32_t *)nullptr, (int)(mode != M_COUNT_ONLY), starts, vals);
        g->prof_end("conflict_replay");
        if (getenv("RB_DEBUG")) {
            uint32_t nb = 0;
            RB_HIP(hipMemcpy(&nb, ctr + 4, 4, hipMemcpyDeviceToHost));
            uint32_t dv[5];
            RB_HIP(hipMemcpy(dv, ctr + 600, 20, hipMemcpyDeviceToHost));
            fprintf(stderr, "[rb] N=%zu D=%u heavy=%u conflict_runs=%u conflict_ops=%u big_components=%u n_foreign=%u | big: serial_fallback=%u (ops %u) max_ops=%u max_runs=%u ops_in_big=%u\n", N, D, hc[0], nck, nco, nb, n_foreign, dv[0], dv[4], dv[1], dv[2], dv[3]);
        }
        if (stats) stats->conflict_ops += nco;
    }
    RB_HIP(hipGetLastError());
}

}  // namespace
void rb::add_range(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, unsigned flags, rb_add_stats *stats) {
    RB_REQUIRE(!g->shard, "rb_graph_add_batch: this handle is one shard of a distributed graph; drive it with the rb_shard_* phases");
    RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, graph on %d", b->device, g->p.device);
    RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_graph_add_batch: bad read range");
    RB_HIP(hipSetDevice(g->p.device));
    hipStream_t s = g->stream;
    const int mode_hash = g->stranded ? ((flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
    const int mode = (flags & RB_ADD_COUNT_IF_PRESENT) ? M_COUNT_IF_PRESENT : M_ADD;
    const bool pairs = (flags & RB_ADD_STORE_READ_PAIRS) != 0;
    if (pairs) RB_REQUIRE(g->rpk.bits && g->read_d > 0, "STORE_READ_PAIRS needs use_read_paired_kmers and a read pair distance > 0");
    // occurrence id = read index within the sub-batch << pos_bits | window start; window starts reach max_len - k
    const uint32_t max_pos = b->max_len >= (uint32_t)g->k ? b->max_len - (uint32_t)g->k : 0u;
    uint32_t pos_bits = 1;
    while ((1u << pos_bits) <= max_pos && pos_bits < 31) ++pos_bits;
    const int64_t max_reads = ((int64_t)1 << (32 - pos_bits)) - 1;       // (- 1: the all-ones occurrence id marks a cancelled record, rb_group.hip)
    // max_batch_kmers bounds the RECORDS a sub-batch sorts.  With the prefilter most windows never
    // become records, so a sub-batch may span three times as many windows (fewer, larger runs per k-mer); a
    // sub-batch whose survivors exceed the bound after all (cold cache) is split and redone.
    const bool wide_off = getenv("RB_WIDE_PREFILTER") && atoi(getenv("RB_WIDE_PREFILTER")) == 0;   // 32 <= k <= 64 without the prefilter
    const bool npf_path = g->npf_log2 && (g->k <= 31 || (g->k <= 64 && !wide_off));
    const int wmul = getenv("RB_WINDOW_MUL") ? std::max(1, atoi(getenv("RB_WINDOW_MUL"))) : 3;
    const int64_t max_words = std::max<int64_t>(std::min<int64_t>((npf_path ? wmul : 1) * g->max_batch_kmers, (int64_t)7 << 29) / 32, 1);
    const auto &wo = b->h_woff;
    // plan the sub-batches
    struct Sub { int64_t r0, r1, w0, nw; uint32_t N; int64_t total; };
    // occurrences that provably cannot change a counter are dropped before sorting (word-per-lane walkers: k <= 64;
    // RB_WIDE_PREFILTER=0 sends 32 <= k <= 64 down the unfiltered generic path)
    const bool use_npf = npf_path && (mode == M_ADD || mode == M_COUNT_IF_PRESENT);
    // the minimizer-bucketed cache replaces the hash-bucketed one on this path (lookups here, stores by the
    // stages that retire runs, which find a k-mer's bucket from one of its occurrences in this batch)
    g->use_mpf = use_npf && g->mpf_log2b && !getenv("RB_NO_MPF") && (uint32_t)g->k >= g->mpf_m &&
                 (g->k <= RB_MPF_MAX_K ? (uint32_t)g->k - g->mpf_m + 1u <= RB_MPF_MAX_RING
                                       : filter_wide_mpf_ok(b, b->h_woff.empty() ? 0 : (int64_t)b->h_woff[(size_t)(first + n)] - (int64_t)b->h_woff[(size_t)first], g->k));
    g->seq_codes = b->codes; g->seq_woff = b->woff; g->seq_wpr = b->wpr_uniform;
    struct MpfScope { rb_graph *g; ~MpfScope() { g->use_mpf = false; g->seq_codes = nullptr; g->seq_woff = nullptr; g->seq_wpr = 0; g->occ_bits = 32; } } mpf_scope{g};
    std::vector<Sub> subs;
    {
        int64_t r0 = first;
        const int64_t rend = first + n;
        // Cold start (first insert into cleared filters): short sub-batches first, doubling up to the full
        // size — the prefilter cache learns a k-mer's counter exponent only when a sub-batch retires, and
        // with the producer running one sub-batch ahead the first full-size sub-batches would be sorted
        // almost unfiltered (measured: 2.3 G of the 3.8 G sorted records of a 12.3 G-k-mer pass).
        int64_t cur_words = max_words;
        if (use_npf && g->ordinal == 0 && !getenv("RB_NO_RAMP")) {
            const int div = getenv("RB_RAMP_DIV") ? std::max(1, atoi(getenv("RB_RAMP_DIV"))) : 64;
            cur_words = std::min(max_words, std::max<int64_t>(max_words / div, 1 << 16));
        }
        while (r0 < rend) {
            // largest r1 with words(r0..r1) <= cur_words and r1-r0 <= max_reads (at least one read)
            int64_t hi = std::min(rend, r0 + max_reads);
            int64_t lo = r0 + 1;
            while (lo < hi) {
                int64_t mid = (lo + hi + 1) >> 1;
                if ((int64_t)wo[(size_t)mid] - (int64_t)wo[(size_t)r0] <= cur_words) lo = mid; else hi = mid - 1;
            }
            cur_words = std::min(max_words, cur_words * 2);
            subs.push_back({r0, lo, (int64_t)wo[(size_t)r0], (int64_t)wo[(size_t)lo] - (int64_t)wo[(size_t)r0], 0u, 0});
            r0 = lo;
        }
    }
    hipStream_t sp = g->stream2;
    unsigned long long *pc = nullptr;
    // The paired k-mers of a sub-batch touch rpkbf and nothing else: their walker runs on a side stream of the producer, forked when
    // the sub-batch's window walk is done (beside the emit pass and the first partition pass; joined when the grouping is enqueued).
    // (Rounds 3-5 measured the other placements — beside the window walk, beside the consumer of the sub-batch before, beside the grouping
    // only, split over the emit pass and the bucket kernel: profiles/r03_pairs_side.txt, HISTORY.md; within 1 % of this one, and every
    // kernel that runs beside the walker takes about as much longer as the walker itself took.  RB_SERIAL=1: on the producer stream, after
    // the grouping.)  One counter for the whole call (a sub-batch that is halved and redone has its pairs in already: ORs, not launched again).
    const int pairs_mode = (pairs && !getenv("RB_SERIAL")) ? 3 : 0;
    const bool pairs_side = pairs_mode != 0;
    int64_t pairs_upto = first;
    bool pairs_pending = false;
    if (pairs_side) {
        g->pairs_ctr.reserve(64);
        RB_HIP(hipMemsetAsync(g->pairs_ctr.p, 0, 64, g->stream3));
    }
    auto pairs_fork = [&](size_t i, bool after_producer) {
        if (!pairs_side || i >= subs.size()) return;
        const Sub &sb = subs[i];
        if (sb.nw <= 0 || sb.r1 <= pairs_upto) return;
        if (after_producer) {
            RB_HIP(hipEventRecord(g->ev2, sp));
            RB_HIP(hipStreamWaitEvent(g->stream3, g->ev2, 0));
        }
        g->prof_begin(g->stream3);
        launch_pairs(g, b, sb.w0, sb.nw, mode_hash, nullptr, nullptr, g->pairs_ctr.as<unsigned long long>(), g->stream3);
        g->prof_end("pairs_insert", g->stream3);
        RB_HIP(hipEventRecord(g->ev3, g->stream3));
        pairs_upto = sb.r1; pairs_pending = true;
    };
    // producer: hash + group sub-batch i into slot i&1 on the producer stream (touches scratch and,
    // for the order-independent paired k-mers, rpkbf only)
    auto prepare_once = [&](size_t i) -> bool {
        Sub &sb = subs[i];
        sb.N = 0;
        const int slot = (int)(i & 1u);
        g->devctr2.reserve(DEVCTR_BYTES);
        sb.total = 0;
        if (sb.nw > 0) {
            const uint64_t ord0 = g->ordinal + (uint64_t)(sb.r0 - first);
            g->chunk_cnt.reserve(((size_t)sb.nw + 1) * 4); g->chunk_off.reserve(((size_t)sb.nw + 1) * 4);
            g->temp2.reserve(scan_temp_bytes((size_t)sb.nw + 1));
            RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + sb.nw, 0, 4, sp));
            if (g->await_words) g->await_words(sb.w0 + sb.nw, sp);      // (a batch that is still being uploaded: rb_graph_add_packed)
            // Where the cache has stopped dropping anything (two sub-batches in a row kept >= 97 % of their windows: long reads, nearly every
            // k-mer new) the window walk against it is a hashing pass for nothing: the next 15 sub-batches count their usable windows instead
            // and emit them all, then one is measured again.  RB_PF_SKIP=0: never; 2: sub-batches of any size count (tests).
            bool filt_now = use_npf;
            if (use_npf && g->pf_skip_left > 0 && !(getenv("RB_PF_SKIP") && atoi(getenv("RB_PF_SKIP")) == 0)) { filt_now = false; --g->pf_skip_left; }
            if (filt_now) {
                // pass 1 hashes every window and asks the cache (count + keep mask per word), scan, pass 2
                // re-hashes and emits the survivors.  (A one-pass kernel — hash, ask, write the survivors through an LDS stage — was
                // 12 ms faster on its own and 35 ms slower per step, rounds 2-3: HISTORY.md; removed in round 6.)
                g->prof_begin(sp);
                g->chunk_mask.reserve(((size_t)sb.nw + 1) * 4);
                g->npf_tot.reserve(2048);
                RB_HIP(hipMemsetAsync(g->npf_tot.p, 0, 2048, sp));
                FilterView fvp = g->view(ord0, pos_bits);
                void *wstate = nullptr;
                if (g->k > RB_MPF_MAX_K ? (g->use_mpf && filter_saves_state_wide(b, sb.nw, g->k)) : filter_saves_state(b, sb.nw, g->k)) { g->wstate.reserve(((size_t)sb.nw + 1) * 16); wstate = g->wstate.p; }
                launch_filter_windows(b, sb.w0, sb.nw, g->k, mode_hash, (uint32_t)sb.r0, pos_bits, g->p.rng_seed, ord0, fvp.npf,
                                      g->chunk_cnt.as<uint32_t>(), g->chunk_mask.as<uint32_t>(), g->npf_tot.as<uint32_t>(), sp, OwnRange{Mod{1, 0, 0}, 0, 0}, fvp.mpf, wstate);
                exclusive_scan_u32(g->temp2.p, g->temp2.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)sb.nw + 1, sp);
                uint32_t spread[16 * 32];
                RB_HIP(hipMemcpyAsync(&sb.N, g->chunk_off.as<uint32_t>() + sb.nw, 4, hipMemcpyDeviceToHost, sp));
                RB_HIP(hipMemcpyAsync(spread, g->npf_tot.p, sizeof spread, hipMemcpyDeviceToHost, sp));
                g->prof_end("filter_windows", sp);
                RB_HIP(hipStreamSynchronize(sp));
                if ((int64_t)sb.N > g->max_batch_kmers && sb.r1 - sb.r0 > 1) {   // too many survivors: halve and redo
                    const int64_t mid = sb.r0 + (sb.r1 - sb.r0) / 2;
                    Sub second{mid, sb.r1, (int64_t)wo[(size_t)mid], (int64_t)wo[(size_t)sb.r1] - (int64_t)wo[(size_t)mid], 0u, 0};
                    sb.r1 = mid; sb.nw = (int64_t)wo[(size_t)mid] - sb.w0;
                    subs.insert(subs.begin() + (std::ptrdiff_t)i + 1, second);   // invalidates sb
                    return false;
                }
                for (int q = 0; q < 32; ++q) sb.total += spread[16 * q];
                const bool pf_force = getenv("RB_PF_SKIP") && atoi(getenv("RB_PF_SKIP")) == 2;       // (tests: sub-batches of any size count)
                // (a small sub-batch says nothing; and a COLD cache drops nothing either — the first sub-batches of a deep short-read library keep
                // everything too, but there most k-mers are seen again at once: the consumer must have found under 0.3 of the last sub-batch's
                // k-mers present (0.16 on the long reads this is for; over 0.5 from the second sub-batch of config 2 on), or the walk stays)
                const bool mostly_new = pf_force || (g->last_present_frac >= 0.0f && g->last_present_frac < 0.3f);
                if (sb.total >= (pf_force ? 1 : ((int64_t)1 << 26)) && (double)sb.N >= 0.97 * (double)sb.total && mostly_new) {
                    if (++g->pf_streak >= 2) {
                        g->pf_skip_left = 15; g->pf_streak = 1;
                        if (getenv("RB_DEBUG")) fprintf(stderr, "[rb] prefilter: kept %u of %lld windows, %.2f of the last sub-batch's k-mers were present: the next 15 sub-batches go without the window walk\n", sb.N, (long long)sb.total, g->last_present_frac);
                    }
                } else g->pf_streak = 0;
                if (getenv("RB_WORD_STATS")) {   // development: how many words keep nothing?
                    std::vector<uint32_t> hc((size_t)sb.nw);
                    RB_HIP(hipMemcpy(hc.data(), g->chunk_cnt.p, (size_t)sb.nw * 4, hipMemcpyDeviceToHost));
                    int64_t h[5] = {0, 0, 0, 0, 0}, waves = 0, empty_waves = 0;
                    for (int64_t q = 0; q < sb.nw; ++q) { const uint32_t c = hc[(size_t)q]; ++h[c == 0 ? 0 : c <= 4 ? 1 : c <= 16 ? 2 : c < 32 ? 3 : 4]; }
                    for (int64_t q = 0; q < sb.nw; q += 64) { bool any = false; for (int64_t z = q; z < std::min(q + 64, sb.nw); ++z) any |= hc[(size_t)z] != 0; ++waves; empty_waves += !any; }
                    fprintf(stderr, "[word stats] words %lld: empty %.3f, 1-4 %.3f, 5-16 %.3f, 17-31 %.3f, 32 %.3f; empty 64-word groups %.3f\n", (long long)sb.nw,
                            (double)h[0] / sb.nw, (double)h[1] / sb.nw, (double)h[2] / sb.nw, (double)h[3] / sb.nw, (double)h[4] / sb.nw, (double)empty_waves / waves);
                }
                if (pairs_mode == 3) pairs_fork(i, true);
                if (sb.N) {
                    g->prof_begin(sp);
                    g->keys0.reserve((size_t)sb.N * 8); g->vals0.reserve((size_t)sb.N * 4);
                    launch_hash_windows_masked(b, sb.w0, sb.nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), g->chunk_mask.as<uint32_t>(),
                                               (uint32_t)sb.r0, pos_bits, g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(), sp, wstate);
                    g->prof_end("hash_windows", sp);
                }
            } else {
                g->prof_begin(sp);
                launch_count_windows(b, sb.w0, sb.nw, g->k, g->chunk_cnt.as<uint32_t>(), sp);
                exclusive_scan_u32(g->temp2.p, g->temp2.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)sb.nw + 1, sp);
                RB_HIP(hipMemcpyAsync(&sb.N, g->chunk_off.as<uint32_t>() + sb.nw, 4, hipMemcpyDeviceToHost, sp));
                g->prof_end("count_windows", sp);
                RB_HIP(hipStreamSynchronize(sp));
                if (use_npf && (int64_t)sb.N > g->max_batch_kmers && sb.r1 - sb.r0 > 1) {   // (planned for a prefilter that drops most windows: halve and redo)
                    const int64_t mid = sb.r0 + (sb.r1 - sb.r0) / 2;
                    Sub second{mid, sb.r1, (int64_t)wo[(size_t)mid], (int64_t)wo[(size_t)sb.r1] - (int64_t)wo[(size_t)mid], 0u, 0};
                    sb.r1 = mid; sb.nw = (int64_t)wo[(size_t)mid] - sb.w0;
                    subs.insert(subs.begin() + (std::ptrdiff_t)i + 1, second);   // invalidates sb
                    ++g->pf_skip_left;                                            // (the redo is not a sub-batch of its own)
                    return false;
                }
                sb.total = sb.N;
                if (sb.N) {
                    g->prof_begin(sp);
                    g->keys0.reserve((size_t)sb.N * 8); g->vals0.reserve((size_t)sb.N * 4);
                    launch_hash_windows(b, sb.w0, sb.nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), (uint32_t)sb.r0, pos_bits,
                                        g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(), nullptr, nullptr, sp);
                    g->prof_end("hash_windows", sp);
                }
            }
        }
        if (pairs_mode == 3) pairs_fork(i, true);   // (already forked on the two-pass path; this is for the paths without the two passes)
        group_enqueue(g, slot, sb.N, g->ordinal + (uint64_t)(sb.r0 - first), pos_bits, sp, g->temp2, g->devctr2);
        if (pairs_pending) { RB_HIP(hipStreamWaitEvent(sp, g->ev3, 0)); pairs_pending = false; }
        if (pairs && !pairs_side && sb.nw > 0) {   // after group_enqueue: it zeroes the producer's counter block
            g->prof_begin(sp);
            pc = reinterpret_cast<unsigned long long *>(g->devctr2.as<uint32_t>() + 12);
            launch_pairs(g, b, sb.w0, sb.nw, mode_hash, nullptr, nullptr, pc, sp);
            g->prof_end("pairs_insert", sp);
        }
        return true;
    };
    auto prepare = [&](size_t i) { while (!prepare_once(i)) {} };
    if (!subs.empty()) prepare(0);
    for (size_t i = 0; i < subs.size(); ++i) {
        const int slot = (int)(i & 1u);
        unsigned long long np = 0;
        if (pairs && !pairs_side && subs[i].nw > 0) RB_HIP(hipMemcpyAsync(&np, reinterpret_cast<unsigned long long *>(g->devctr2.as<uint32_t>() + 12), 8, hipMemcpyDeviceToHost, sp));
        const uint32_t D = group_finish(g, slot, sp, g->temp2, g->devctr2, s);  // drains the producer stream
        if (stats) { stats->pairs += (int64_t)np; stats->distinct += D; }
        const bool serial = getenv("RB_SERIAL") != nullptr;   // debugging / clean per-stage timing
        g->cur = slot;
        g->seq_first = (uint32_t)subs[i].r0;             // occurrence ids of this sub-batch are relative to its first read
        g->occ_bits = pos_bits + log2_ceil((uint64_t)std::max<int64_t>(1, subs[i].r1 - subs[i].r0));
        // sub-batch i+1 is hashed / prefiltered as soon as sub-batch i's own-counter runs have retired (their
        // cache updates are what the prefilter needs); its sort + grouping then overlap the heavy and
        // conflicting runs of sub-batch i
        const std::function<void()> next = [&]() {
            if (serial || i + 1 >= subs.size()) return;
            RB_HIP(hipEventRecord(g->ev0, s));
            RB_HIP(hipStreamWaitEvent(sp, g->ev0, 0));
            prepare(i + 1);
        };
        run_core(g, subs[i].N, D, mode, g->ordinal + (uint64_t)(subs[i].r0 - first), pos_bits, stats, &next);
        RB_HIP(hipStreamSynchronize(s));               // slot may be refilled after this
        if (serial && i + 1 < subs.size()) prepare(i + 1);
        if (stats) { stats->kmers += subs[i].total; stats->sorted_kmers += subs[i].N; stats->reads += subs[i].r1 - subs[i].r0; }
    }
    g->ordinal += (uint64_t)n;
    RB_HIP(hipStreamSynchronize(s));
    RB_HIP(hipStreamSynchronize(sp));
    if (pairs_side) {
        RB_HIP(hipStreamSynchronize(g->stream3));
        unsigned long long np = 0;
        RB_HIP(hipMemcpy(&np, g->pairs_ctr.p, 8, hipMemcpyDeviceToHost));
        if (stats) stats->pairs += (int64_t)np;
        if (getenv("RB_DEBUG") && g->rpk.seen) {
            unsigned long long d[2] = {0, 0};
            RB_HIP(hipMemcpy(d, g->pairs_ctr.as<unsigned long long>() + 1, 16, hipMemcpyDeviceToHost));
            fprintf(stderr, "[rb] seen-pair cache: %llu pairs, %llu not known (%.1f %%), %llu bucket fetches (%.2f per pair)\n", np, d[0], 100.0 * (double)d[0] / (double)std::max(np, 1ull),
                    d[1], (double)d[1] / (double)std::max(np, 1ull));
        }
    }
    g->prof_collect();
}

// upload n base hashes into a scratch buffer
uint64_t *rb::upload_h0(rb_graph *g, DevBuf &buf, const uint64_t *h0, size_t n, hipStream_t st) {
    buf.reserve(std::max<size_t>(n, 1) * 8);
    RB_HIP(hipMemcpyAsync(buf.p, h0, n * 8, hipMemcpyHostToDevice, st ? st : g->stream));
    return buf.as<uint64_t>();
}



