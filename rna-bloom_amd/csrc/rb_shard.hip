// rb_shard.hip — the sharded (multi-GPU) insert engine: one rb_graph per rank holding index range
// [lo,hi) of every filter.  The phases mirror the single-GPU pipeline of rb_graph.hip, cut where data
// has to move between ranks (see the protocol comment in include/rb_capi.h and DESIGN.md §6):
//
//   requester = owner of a k-mer (top log2 G bits of its hash): holds all occurrences of its k-mers
//               in order, decides found-flags / op counts / counter updates;
//   owner     = owner of a filter index range: tests & sets bits, arbitrates first setters, hands
//               out counter claims, stores counter bytes.
//
// Exactness argument is unchanged: first-setter arbitration happens at the bit's owner over ALL
// probes of the sub-batch; claim marks and the conflict set live at the counter's owner; runs whose
// counters nobody else claimed commute; everything else is replayed in global occurrence order —
// here by every rank redundantly on a private compact copy of the (few) contested counters.
#include "rb_pipeline.hpp"

using namespace rb;

struct ShardState {
    int G = 1, log2G = 0;
    int64_t span[4] = {0, 0, 0, 0};            // per filter (RB_DBGBF..RB_FPKBF)
    DevBuf slot[RB_SLOT_COUNT];
    size_t slot_bytes[RB_SLOT_COUNT] = {0};
    // requester-side state carried from group -> resolve
    uint32_t D = 0;
    uint64_t ordinal0 = 0;
    uint32_t pos_bits = 0;
    DevBuf dreq_pos, creq_pos;                 // [D*h] position of (run, probe) in the bucketed request order (~0 = none)
    DevBuf creq_dup;                           // [D*h] for a duplicated counter: the earlier probe it copies
    DevBuf cfinal, conf_list;
    // routing scratch
    DevBuf rkey0, rkey1, rval0, rval1, stage0, stage1, stage2, bounds;
    // owner-side scratch
    DevBuf own_f, own_cs, own_foreign;
    // conflict replay scratch
    DevBuf ck0, ck1, cv0, cv1, cuniq, ccnt, cstart, cval, oslots, olabel, okey0, okey1, oval0, oval1;
};

namespace {

inline int64_t roundup64(int64_t x) { return (x + 63) / 64 * 64; }

struct Geometry { int64_t span, lo, hi; };
Geometry geom(int64_t size, int rank, int count) {
    Geometry g;
    g.span = roundup64((size + count - 1) / count);
    g.lo = std::min<int64_t>(size, g.span * rank);
    g.hi = std::min<int64_t>(size, g.span * (rank + 1));
    return g;
}

void *slot_reserve(ShardState *S, int slot, size_t bytes) {
    S->slot[slot].reserve(std::max<size_t>(bytes, 16));
    S->slot_bytes[slot] = bytes;
    return S->slot[slot].p;
}

// ---------------------------------------------------------------- routing ----
__global__ void k_dest_keys(const uint64_t *__restrict__ idx, const uint8_t *__restrict__ drop, size_t n, uint64_t span,
                            uint32_t G, uint64_t *__restrict__ key, uint64_t *__restrict__ val) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = (drop && drop[i]) ? (uint64_t)G : idx[i] / span;
    val[i] = i;
}
// bounds[g] = first position whose key >= g, for g = 0..G (keys sorted ascending)
__global__ void k_bounds(const uint64_t *__restrict__ key, size_t n, uint32_t G, uint32_t shift, uint64_t *__restrict__ bounds) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > G) return;
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) >> 1; if ((key[mid] >> shift) < g) lo = mid + 1; else hi = mid; }
    bounds[g] = lo;
}
template <typename T>
__global__ void k_gather(const T *__restrict__ src, const uint64_t *__restrict__ perm, size_t n, T *__restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
__global__ void k_inverse(const uint64_t *__restrict__ perm, size_t n_all, size_t n_kept, uint32_t *__restrict__ pos_of) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_all) pos_of[perm[i]] = i < n_kept ? (uint32_t)i : 0xFFFFFFFFu;
}

// Sort n items by destination = idx/span (dropped items last).  Leaves the permutation in
// S->rval1 (u64 original positions) and per-destination counts in counts[G]; returns items kept.
size_t route(rb_graph *g, const uint64_t *idx, const uint8_t *drop, size_t n, int64_t span, int64_t *counts) {
    ShardState *S = g->shard;
    hipStream_t s = g->stream;
    for (int r = 0; r < S->G; ++r) counts[r] = 0;
    if (n == 0) return 0;
    S->rkey0.reserve(n * 8); S->rkey1.reserve(n * 8); S->rval0.reserve(n * 8); S->rval1.reserve(n * 8);
    S->bounds.reserve((S->G + 2) * 8);
    hipLaunchKernelGGL(k_dest_keys, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, s, idx, drop, n, (uint64_t)span, (uint32_t)S->G,
                       S->rkey0.as<uint64_t>(), S->rval0.as<uint64_t>());
    g->temp.reserve(sort_pairs32_temp_bytes(n));
    sort_pairs_u64_u64(g->temp.p, g->temp.cap, S->rkey0.as<uint64_t>(), S->rkey1.as<uint64_t>(), S->rval0.as<uint64_t>(),
                       S->rval1.as<uint64_t>(), n, 0, S->log2G + 1, s);
    hipLaunchKernelGGL(k_bounds, dim3(1), dim3(128), 0, s, S->rkey1.as<uint64_t>(), n, (uint32_t)S->G, 0u, S->bounds.as<uint64_t>());
    std::vector<uint64_t> b(S->G + 1);
    RB_HIP(hipMemcpyAsync(b.data(), S->bounds.p, (S->G + 1) * 8, hipMemcpyDeviceToHost, s));
    RB_HIP(hipStreamSynchronize(s));
    for (int r = 0; r < S->G; ++r) counts[r] = (int64_t)(b[r + 1] - b[r]);
    return (size_t)b[S->G];
}
template <typename T> void gather_to(rb_graph *g, const T *src, size_t n, T *dst) {
    if (!n) return;
    hipLaunchKernelGGL(k_gather<T>, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, g->stream, src, g->shard->rval1.as<uint64_t>(), n, dst);
}

// ------------------------------------------------------------- requester ----
// per run: one Bloom-bit request per probe, one claim request per DISTINCT counter
__global__ void k_make_requests(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts,
                                const uint32_t *__restrict__ vals, uint32_t D, int mode,
                                uint64_t *__restrict__ d_idx, uint64_t *__restrict__ d_probe, uint8_t *__restrict__ d_drop,
                                uint64_t *__restrict__ c_idx, uint8_t *__restrict__ c_drop, uint8_t *__restrict__ c_dup) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint64_t h0 = uniq[d];
    const unsigned long long v_first = vals[starts[d]];
    for (int j = 0; j < fv.dbg_h; ++j) {
        const size_t q = (size_t)d * fv.dbg_h + j;
        d_idx[q] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
        d_probe[q] = (v_first << 4) | (unsigned long long)j;
        d_drop[q] = mode == M_COUNT_ONLY;
    }
    uint64_t cidx[RB_MAX_HASH];
    for (int j = 0; j < fv.cbf_h; ++j) {
        const size_t q = (size_t)d * fv.cbf_h + j;
        cidx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        int dup = -1;
        for (int p = 0; p < j; ++p) if (cidx[p] == cidx[j] && dup < 0) dup = p;
        c_idx[q] = cidx[j];
        c_drop[q] = dup >= 0;
        c_dup[q] = (uint8_t)(dup >= 0 ? dup : j);
    }
}

__global__ void k_resolve_shard(FilterView fv, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ starts,
                                uint32_t D, int mode, uint32_t light_ops, const uint32_t *__restrict__ dreq_pos,
                                const uint8_t *__restrict__ dreply, const uint32_t *__restrict__ creq_pos,
                                const uint8_t *__restrict__ c_dup, const uint8_t *__restrict__ creply,
                                const uint8_t *__restrict__ tz, uint32_t *__restrict__ status, uint32_t *__restrict__ nops,
                                uint64_t *__restrict__ cvals, uint64_t *__restrict__ cfinal) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint32_t m = counts[d];
    uint32_t premask = 0;
    bool all_pre = true, found_first = true;
    if (mode != M_COUNT_ONLY) {
        for (int j = 0; j < fv.dbg_h; ++j) {
            const uint8_t r = dreply[dreq_pos[(size_t)d * fv.dbg_h + j]];
            if (r & 1u) premask |= 1u << j;
            else { all_pre = false; if (r & 2u) found_first = false; }   // this probe was the first setter => bit was clear
        }
    }
    uint32_t ops = 0, kfirst = K_INC, krest = K_INC;
    if (mode == M_COUNT_ONLY) ops = m;
    else if (mode == M_COUNT_IF_PRESENT) { ops = all_pre ? m : 0u; kfirst = krest = K_INC_IF_POS; }
    else if (mode == M_ADD) ops = (all_pre || found_first) ? m : m - 1u;
    else { ops = m; kfirst = (all_pre || found_first) ? K_INC_IF_ZERO : K_INC; krest = K_INC_IF_ZERO; }
    uint32_t c[RB_MAX_HASH];
    bool conflict = false;
    uint64_t cv = 0;
    for (int j = 0; j < fv.cbf_h; ++j) {
        const size_t q = (size_t)d * fv.cbf_h + j;
        const int src = c_dup[q];
        if (src != j) c[j] = c[src];
        else {
            const uint8_t r = creply[creq_pos[q]];
            c[j] = r & 0x7Fu;
            conflict |= (r & 0x80u) != 0;
        }
        cv |= (uint64_t)c[j] << (8 * j);
    }
    cvals[d] = cv;
    uint32_t st = premask | (all_pre ? ST_ALLPRE : 0u) | (kfirst << 12) | (krest << 14);
    nops[d] = ops;
    if (ops == 0) { status[d] = st | RUN_RELEASE; return; }
    if (conflict) { status[d] = st | RUN_CONFLICT; return; }
    if (ops > light_ops) { status[d] = st | RUN_WRITES | RUN_HEAVY; return; }
    status[d] = st | RUN_WRITES;
    run_ops(c, fv.cbf_h, kfirst, krest, tz, starts[d] + m - ops, ops);
    uint64_t out = 0;
    for (int j = 0; j < fv.cbf_h; ++j) out |= (uint64_t)c[j] << (8 * j);
    cfinal[d] = out;
}
__global__ void k_emit_writes(FilterView fv, const uint64_t *__restrict__ uniq, uint32_t D, const uint32_t *__restrict__ status,
                              const uint8_t *__restrict__ c_dup, const uint64_t *__restrict__ cfinal,
                              uint64_t *__restrict__ w_idx, uint8_t *__restrict__ w_val, uint8_t *__restrict__ w_drop) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint32_t st = status[d];
    const uint64_t h0 = uniq[d], cf = cfinal[d];
    for (int j = 0; j < fv.cbf_h; ++j) {
        const size_t q = (size_t)d * fv.cbf_h + j;
        const bool send = (c_dup[q] == j) && (st & (RUN_RELEASE | RUN_WRITES));
        w_idx[q] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        w_val[q] = (st & RUN_RELEASE) ? (uint8_t)0xFF : (uint8_t)(cf >> (8 * j));
        w_drop[q] = !send;
    }
}
struct ConfOp { uint32_t occ, kind; uint64_t h0; };
struct ConfCtr { uint64_t idx, val; };
__global__ void k_conf_sizes(const uint32_t *__restrict__ conf_list, const uint32_t *__restrict__ nops, uint32_t n, uint32_t *__restrict__ sizes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sizes[i] = nops[conf_list[i]];
    if (i == n) sizes[i] = 0;
}
__global__ void k_conf_export(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ counts,
                              const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals,
                              const uint32_t *__restrict__ status, const uint32_t *__restrict__ nops,
                              const uint64_t *__restrict__ cvals, const uint8_t *__restrict__ c_dup,
                              const uint32_t *__restrict__ conf_list, const uint32_t *__restrict__ conf_off, uint32_t n_conf,
                              ConfOp *__restrict__ ops_out, ConfCtr *__restrict__ ctr_out) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (wave >= n_conf) return;
    const uint32_t d = conf_list[wave];
    const uint32_t ops = nops[d], st = status[d];
    const uint64_t h0 = uniq[d];
    const uint32_t base = starts[d] + counts[d] - ops, out = conf_off[wave];
    for (uint32_t i = lane; i < ops; i += 64u) {
        ConfOp o;
        o.occ = vals[base + i];
        o.kind = i == 0 ? (st >> 12) & 3u : (st >> 14) & 3u;
        o.h0 = h0;
        ops_out[out + i] = o;
    }
    if (lane == 0) {
        const uint64_t cv = cvals[d];
        for (int j = 0; j < fv.cbf_h; ++j) {   // fixed stride; a duplicated probe leaves a sentinel (index ~0)
            ConfCtr c;
            const bool dup = c_dup[(size_t)d * fv.cbf_h + j] != j;
            c.idx = dup ? ~0ull : index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
            c.val = dup ? 0ull : ((cv >> (8 * j)) & 0xFFu);
            ctr_out[(size_t)wave * fv.cbf_h + j] = c;
        }
    }
}

// ------------------------------------------------------------------ owner ----
__global__ void k_own_dbg_test(uint32_t *bits, uint64_t lo, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ probe,
                               size_t n, int uses_f, Slot *ftable, uint32_t f_log2, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool pre = bit_test(bits, idx[i] - lo);
    reply[i] = pre ? 1u : 0u;
    if (!pre && uses_f) {
        Slot *s = table_insert(ftable, f_log2, idx[i]);
        atomicMin(&s->val, (unsigned long long)probe[i]);
    }
}
__global__ void k_own_dbg_set(uint32_t *bits, uint64_t lo, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ probe,
                              size_t n, const Slot *ftable, uint32_t f_log2, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (reply[i] & 1u)) return;
    const Slot *s = table_find(ftable, f_log2, idx[i]);
    if (s->val == (unsigned long long)probe[i]) reply[i] |= 2u;      // this probe is the sequentially first setter
    bit_set(bits, idx[i] - lo);
}
__global__ void k_own_claim(uint8_t *cbf, uint64_t lo, const uint64_t *__restrict__ idx, size_t n, uint8_t *__restrict__ reply,
                            uint32_t *__restrict__ spread /* 32 counters, 16 words apart */) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t byte = cbf_claim(cbf, idx[i] - lo);
    if (byte & CLAIM) atomicAdd(&spread[16 * (blockIdx.x & 31u)], 1u);
    reply[i] = (uint8_t)byte;                                         // bit 7 = claimed before by another run
}
__global__ void k_own_cs_build(const uint64_t *__restrict__ idx, const uint8_t *__restrict__ reply, size_t n, Slot *cs, uint32_t cs_log2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (reply[i] & 0x80u)) table_insert(cs, cs_log2, idx[i]);
}
__global__ void k_own_claim_fin(const uint64_t *__restrict__ idx, size_t n, const Slot *cs, uint32_t cs_log2, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (reply[i] & 0x80u)) return;
    if (table_find(cs, cs_log2, idx[i])) reply[i] |= 0x80u;
}
__global__ void k_own_bits(uint32_t *bits, uint64_t lo, const uint64_t *__restrict__ idx, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bit_set(bits, idx[i] - lo);
}
__global__ void k_own_writes(uint8_t *cbf, uint64_t lo, const uint64_t *__restrict__ idx, const uint8_t *__restrict__ val, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (val[i] == 0xFFu) cbf_release(cbf, idx[i] - lo); else cbf[idx[i] - lo] = val[i];
}

// -------------------------------------------------- distributed conflict replay ----
__global__ void k_split_ctr(const ConfCtr *__restrict__ c, size_t n, uint64_t *__restrict__ k, uint64_t *__restrict__ v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { k[i] = c[i].idx; v[i] = c[i].val; }
}
__global__ void k_first_vals(const uint64_t *__restrict__ vals_sorted, const uint32_t *__restrict__ starts, uint32_t M, uint32_t *__restrict__ cval) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) cval[i] = (uint32_t)vals_sorted[starts[i]];
}
__device__ __forceinline__ uint32_t find_u64(const uint64_t *a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ void k_op_slots(FilterView fv, const ConfOp *__restrict__ ops, size_t n, const uint64_t *__restrict__ cuniq, uint32_t M,
                           uint32_t *__restrict__ slots, uint32_t *__restrict__ label) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < fv.cbf_h; ++j) {
        uint32_t s = find_u64(cuniq, M, index_of(multi_hash(ops[i].h0, (uint32_t)j, fv.kmul), fv.cbf_mod));
        slots[i * fv.cbf_h + j] = s;
        label[s] = s;
    }
}
__global__ void k_op_labels(const uint32_t *__restrict__ slots, size_t n, int h, uint32_t *__restrict__ label, uint32_t *__restrict__ changed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t m = 0xFFFFFFFFu;
    for (int j = 0; j < h; ++j) { uint32_t l = __hip_atomic_load(&label[slots[i * h + j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); m = l < m ? l : m; }
    for (int j = 0; j < h; ++j) {
        uint32_t old = atomicMin(&label[slots[i * h + j]], m);
        if (old > m) *changed = 1u;
    }
}
// labels may still point at a non-root after the loop converged pairwise; chase to the root
__global__ void k_op_keys(const ConfOp *__restrict__ ops, const uint32_t *__restrict__ slots, size_t n, int h,
                          const uint32_t *__restrict__ label, uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t l = label[slots[i * h]];
    while (label[l] != l) l = label[l];
    key[i] = ((uint64_t)l << 32) | ops[i].occ;
    val[i] = (uint32_t)i;
}
__global__ void k_conf_replay_sparse(FilterView fv, const ConfOp *__restrict__ ops, const uint32_t *__restrict__ slots,
                                     const uint64_t *__restrict__ key, const uint32_t *__restrict__ order, size_t n,
                                     uint32_t *__restrict__ cval) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lab = (uint32_t)(key[i] >> 32);
    if (i > 0 && (uint32_t)(key[i - 1] >> 32) == lab) return;       // not the first op of its component
    for (size_t q = i; q < n && (uint32_t)(key[q] >> 32) == lab; ++q) {
        const uint32_t o = order[q];
        uint32_t c[RB_MAX_HASH], c0[RB_MAX_HASH];
        for (int j = 0; j < fv.cbf_h; ++j) c0[j] = c[j] = *(volatile uint32_t *)&cval[slots[(size_t)o * fv.cbf_h + j]];
        uint32_t mn = c[0];
        for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
        cbf_step(c, fv.cbf_h, ops[o].kind, (mn >= 16u && mn < 127u) ? occ_rnd(fv, ops[o].occ) : 0u);
        for (int j = 0; j < fv.cbf_h; ++j)
            if (c[j] != c0[j]) *(volatile uint32_t *)&cval[slots[(size_t)o * fv.cbf_h + j]] = c[j];
    }
}
__global__ void k_conf_writeback(uint8_t *cbf, uint64_t lo, uint64_t hi, const uint64_t *__restrict__ cuniq, const uint32_t *__restrict__ cval, uint32_t M) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const uint64_t idx = cuniq[i];
    if (idx >= lo && idx < hi) cbf[idx - lo] = (uint8_t)cval[i];     // also clears the claim mark
}

}  // namespace

// ------------------------------------------------------------------------ C ABI ----
extern "C" {

int rb_graph_create_shard(const rb_graph_params *p, int shard_rank, int shard_count, rb_graph **out) {
    rb_graph *g = nullptr;
    int rc = guarded([&] {
        RB_REQUIRE(p && out, "rb_graph_create_shard: null argument");
        RB_REQUIRE(shard_count >= 1 && shard_count <= 64 && (shard_count & (shard_count - 1)) == 0,
                   "rb_graph_create_shard: shard_count must be a power of two in [1,64]");
        RB_REQUIRE(shard_rank >= 0 && shard_rank < shard_count, "rb_graph_create_shard: bad shard_rank");
        RB_REQUIRE(p->k >= 1 && p->k <= RB_MAX_K && p->dbgbf_bits > 0 && p->cbf_bytes > 0, "rb_graph_create_shard: bad parameters");
        RB_REQUIRE(p->dbgbf_num_hash >= 1 && p->dbgbf_num_hash <= RB_MAX_HASH && p->cbf_num_hash >= 1 && p->cbf_num_hash <= RB_MAX_HASH,
                   "rb_graph_create_shard: numHash out of range");
        int ndev = 0;
        RB_HIP(hipGetDeviceCount(&ndev));
        RB_REQUIRE(p->device >= 0 && p->device < ndev, "rb_graph_create_shard: device %d not present", p->device);
        RB_HIP(hipSetDevice(p->device));
        g = new rb_graph();
        g->p = *p; g->k = p->k; g->stranded = p->stranded != 0;
        g->H = std::max(p->dbgbf_num_hash, p->cbf_num_hash);
        g->max_batch_kmers = p->max_batch_kmers > 0 ? p->max_batch_kmers : ((int64_t)1 << 30);
        if (p->group_bits) g->sort_begin_bit = 64 - p->group_bits;
        g->shard_rank = shard_rank; g->shard_count = shard_count;
        ShardState *S = g->shard = new ShardState();
        S->G = shard_count; S->log2G = (int)log2_ceil((uint64_t)shard_count);
        RB_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        RB_HIP(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
        RB_HIP(hipEventCreate(&g->ev0));
        RB_HIP(hipEventCreate(&g->ev1));
        Geometry gd = geom(p->dbgbf_bits, shard_rank, shard_count);
        S->span[RB_DBGBF] = gd.span;
        alloc_bits(g->dbg, p->dbgbf_bits, p->dbgbf_num_hash, gd.lo, gd.hi);
        Geometry gc = geom(p->cbf_bytes, shard_rank, shard_count);
        S->span[RB_CBF] = gc.span;
        g->cbf_size = p->cbf_bytes; g->cbf_lo = gc.lo; g->cbf_hi = gc.hi;
        g->cbf_alloc = (((size_t)(gc.hi - gc.lo) + 3) / 4 + 1) * 4;
        g->cbf_h = p->cbf_num_hash;
        g->cbf_mod = make_mod((uint64_t)p->cbf_bytes);
        RB_HIP(hipMalloc(&g->cbf, g->cbf_alloc));
        RB_HIP(hipMemset(g->cbf, 0, g->cbf_alloc));
        if (p->use_read_paired_kmers) {
            RB_REQUIRE(p->pkbf_bits > 0 && p->pkbf_num_hash >= 1 && p->pkbf_num_hash <= RB_MAX_HASH, "rb_graph_create_shard: pair filter parameters invalid");
            Geometry gp = geom(p->pkbf_bits, shard_rank, shard_count);
            S->span[RB_RPKBF] = gp.span;
            alloc_bits(g->rpk, p->pkbf_bits, p->pkbf_num_hash, gp.lo, gp.hi);
        }
        RB_HIP(hipDeviceSynchronize());
        *out = g;
    });
    if (rc != RB_OK && g) rb_graph_destroy(g);
    return rc;
}

int rb_shard_span(rb_graph *g, int which, int64_t *span, int64_t *lo, int64_t *hi) {
    if (!g || !g->shard) { set_error("rb_shard_span: not a sharded graph"); return RB_ERR_INVALID; }
    int64_t l, h;
    if (which == RB_CBF) { l = g->cbf_lo; h = g->cbf_hi; }
    else if (which == RB_DBGBF) { l = g->dbg.lo; h = g->dbg.hi; }
    else if (which == RB_RPKBF) { l = g->rpk.lo; h = g->rpk.hi; }
    else { set_error("rb_shard_span: unsupported filter %d", which); return RB_ERR_INVALID; }
    if (span) *span = g->shard->span[which];
    if (lo) *lo = l;
    if (hi) *hi = h;
    return RB_OK;
}

int rb_shard_take(rb_graph *g, int slot, void *dst_dev, int64_t nbytes) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && slot >= 0 && slot < RB_SLOT_COUNT, "rb_shard_take: bad argument");
        RB_REQUIRE((size_t)nbytes == g->shard->slot_bytes[slot], "rb_shard_take: slot %d holds %zu bytes, asked for %lld", slot,
                   g->shard->slot_bytes[slot], (long long)nbytes);
        RB_HIP(hipSetDevice(g->p.device));
        if (nbytes) RB_HIP(hipMemcpyAsync(dst_dev, g->shard->slot[slot].p, (size_t)nbytes, hipMemcpyDeviceToDevice, g->stream));
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_shard_hash(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, uint32_t read_rel_base, uint32_t pos_bits,
                  unsigned flags, int64_t *rec_counts, int64_t *pair_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && b && rec_counts && pair_counts, "rb_shard_hash: bad argument");
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_shard_hash: bad read range");
        RB_REQUIRE(pos_bits >= 1 && pos_bits <= 31 && ((uint64_t)b->max_len >> pos_bits) == 0, "rb_shard_hash: pos_bits too small for the reads");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) rec_counts[r] = pair_counts[r] = 0;
        S->slot_bytes[RB_SLOT_REC_KEYS] = S->slot_bytes[RB_SLOT_REC_OCC] = S->slot_bytes[RB_SLOT_PAIR_IDX] = 0;
        const int mode_hash = g->stranded ? ((flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
        const bool pairs = (flags & RB_ADD_STORE_READ_PAIRS) != 0;
        if (pairs) RB_REQUIRE(g->rpk.bits && g->read_d > 0, "STORE_READ_PAIRS needs use_read_paired_kmers and a read pair distance > 0");
        const int64_t w0 = b->h_woff[(size_t)first], nw = (int64_t)b->h_woff[(size_t)(first + n)] - w0;
        if (nw <= 0) return;
        g->chunk_cnt.reserve(((size_t)nw + 1) * 4); g->chunk_off.reserve(((size_t)nw + 1) * 4);
        g->temp.reserve(scan_temp_bytes((size_t)nw + 1));
        uint32_t N = 0;
        RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + nw, 0, 4, s));
        launch_count_windows(b, w0, nw, g->k, g->chunk_cnt.as<uint32_t>(), s);
        exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
        RB_HIP(hipMemcpyAsync(&N, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        if (N) {
            g->keys0.reserve((size_t)N * 8); g->vals0.reserve((size_t)N * 4);
            // occ = (read_rel_base + r - first) << pos_bits | pos   (u32 wrap-around arithmetic)
            launch_hash_windows(b, w0, nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), (uint32_t)first - read_rel_base, pos_bits,
                                g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(), nullptr, nullptr, s);
            uint64_t *rk = (uint64_t *)slot_reserve(S, RB_SLOT_REC_KEYS, (size_t)N * 8);
            uint32_t *ro = (uint32_t *)slot_reserve(S, RB_SLOT_REC_OCC, (size_t)N * 4);
            if (S->G == 1) {
                RB_HIP(hipMemcpyAsync(rk, g->keys0.p, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
                RB_HIP(hipMemcpyAsync(ro, g->vals0.p, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
                rec_counts[0] = N;
            } else {   // stable 1-pass bucket by k-mer owner = top log2(G) hash bits
                g->temp.reserve(sort_pairs_temp_bytes(N));
                sort_pairs_u64_u32(g->temp.p, g->temp.cap, g->keys0.as<uint64_t>(), rk, g->vals0.as<uint32_t>(), ro, N, 64 - S->log2G, 64, s);
                S->bounds.reserve((S->G + 2) * 8);
                hipLaunchKernelGGL(k_bounds, dim3(1), dim3(128), 0, s, rk, (size_t)N, (uint32_t)S->G, (uint32_t)(64 - S->log2G), S->bounds.as<uint64_t>());
                std::vector<uint64_t> bd(S->G + 1);
                RB_HIP(hipMemcpyAsync(bd.data(), S->bounds.p, (S->G + 1) * 8, hipMemcpyDeviceToHost, s));
                RB_HIP(hipStreamSynchronize(s));
                for (int r = 0; r < S->G; ++r) rec_counts[r] = (int64_t)(bd[r + 1] - bd[r]);
            }
        }
        if (pairs) {
            uint32_t P = 0;
            RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + nw, 0, 4, s));
            launch_count_windows(b, w0, nw, g->k + g->read_d, g->chunk_cnt.as<uint32_t>(), s);
            exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
            RB_HIP(hipMemcpyAsync(&P, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            const size_t np = (size_t)P * (size_t)g->rpk.num_hash;
            if (np) {
                S->stage0.reserve(np * 8);
                g->devctr.reserve(DEVCTR_BYTES);
                unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12);
                RB_HIP(hipMemsetAsync(pc, 0, 8, s));
                launch_pairs(g, b, w0, nw, mode_hash, g->chunk_off.as<uint32_t>(), S->stage0.as<uint64_t>(), pc);
                size_t kept = route(g, S->stage0.as<uint64_t>(), nullptr, np, S->span[RB_RPKBF], pair_counts);
                uint64_t *dst = (uint64_t *)slot_reserve(S, RB_SLOT_PAIR_IDX, kept * 8);
                gather_to<uint64_t>(g, S->stage0.as<uint64_t>(), kept, dst);
            }
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_group(rb_graph *g, const void *keys_dev, const void *occ_dev, int64_t n, uint64_t ordinal0, uint32_t pos_bits,
                   int mode, int64_t *dreq_counts, int64_t *creq_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && dreq_counts && creq_counts && n >= 0, "rb_shard_group: bad argument");
        RB_REQUIRE(n <= g->max_batch_kmers, "rb_shard_group: %lld records exceed max_batch_kmers", (long long)n);
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) dreq_counts[r] = creq_counts[r] = 0;
        S->D = 0; S->ordinal0 = ordinal0; S->pos_bits = pos_bits;
        S->slot_bytes[RB_SLOT_DREQ_IDX] = S->slot_bytes[RB_SLOT_DREQ_PROBE] = S->slot_bytes[RB_SLOT_CREQ_IDX] = 0;
        if (n == 0) return;
        g->keys0.reserve((size_t)n * 8); g->vals0.reserve((size_t)n * 4);
        RB_HIP(hipMemcpyAsync(g->keys0.p, keys_dev, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        RB_HIP(hipMemcpyAsync(g->vals0.p, occ_dev, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        const uint32_t D = group_records(g, (size_t)n, ordinal0, pos_bits, nullptr, nullptr);
        S->D = D;
        FilterView fv = g->view(ordinal0, pos_bits);
        const size_t nd = (size_t)D * fv.dbg_h, nc = (size_t)D * fv.cbf_h;
        S->stage0.reserve(nd * 8); S->stage1.reserve(nd * 8); S->stage2.reserve(nd + nc + 16);
        DevBuf &cidx = S->cv0;   // reuse as staging for counter indices
        cidx.reserve(nc * 8);
        S->creq_dup.reserve(nc + 16);
        uint8_t *d_drop = S->stage2.as<uint8_t>(), *c_drop = d_drop + nd;
        hipLaunchKernelGGL(k_make_requests, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, g->uniq().as<uint64_t>(), g->starts().as<uint32_t>(),
                           g->vals1().as<uint32_t>(), D, mode, S->stage0.as<uint64_t>(), S->stage1.as<uint64_t>(), d_drop,
                           cidx.as<uint64_t>(), c_drop, S->creq_dup.as<uint8_t>());
        // Bloom-bit requests
        S->dreq_pos.reserve(nd * 4 + 16); S->creq_pos.reserve(nc * 4 + 16);
        size_t kept = route(g, S->stage0.as<uint64_t>(), d_drop, nd, S->span[RB_DBGBF], dreq_counts);
        uint64_t *di = (uint64_t *)slot_reserve(S, RB_SLOT_DREQ_IDX, kept * 8);
        uint64_t *dp = (uint64_t *)slot_reserve(S, RB_SLOT_DREQ_PROBE, kept * 8);
        gather_to<uint64_t>(g, S->stage0.as<uint64_t>(), kept, di);
        gather_to<uint64_t>(g, S->stage1.as<uint64_t>(), kept, dp);
        if (nd) hipLaunchKernelGGL(k_inverse, dim3(blocks_for((int64_t)nd)), dim3(TPB), 0, s, S->rval1.as<uint64_t>(), nd, kept, S->dreq_pos.as<uint32_t>());
        // counter claims
        kept = route(g, cidx.as<uint64_t>(), c_drop, nc, S->span[RB_CBF], creq_counts);
        uint64_t *ci = (uint64_t *)slot_reserve(S, RB_SLOT_CREQ_IDX, kept * 8);
        gather_to<uint64_t>(g, cidx.as<uint64_t>(), kept, ci);
        if (nc) hipLaunchKernelGGL(k_inverse, dim3(blocks_for((int64_t)nc)), dim3(TPB), 0, s, S->rval1.as<uint64_t>(), nc, kept, S->creq_pos.as<uint32_t>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_serve(rb_graph *g, int mode, const void *dreq_idx_dev, const void *dreq_probe_dev, int64_t nd,
                   const void *creq_idx_dev, int64_t nc, const void *pair_idx_dev, int64_t np, void *dreply_dev, void *creply_dev) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && nd >= 0 && nc >= 0 && np >= 0, "rb_shard_serve: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        if (nd) {
            const int uses_f = (mode == M_ADD || mode == M_ADD_IF_ABSENT);
            uint32_t f_log2 = 1;
            if (uses_f) {
                f_log2 = log2_ceil(2ull * (uint64_t)nd + 2);
                S->own_f.reserve(sizeof(Slot) << f_log2);
                RB_HIP(hipMemsetAsync(S->own_f.p, 0xFF, sizeof(Slot) << f_log2, s));
            }
            hipLaunchKernelGGL(k_own_dbg_test, dim3(blocks_for(nd)), dim3(TPB), 0, s, g->dbg.bits, (uint64_t)g->dbg.lo,
                               (const uint64_t *)dreq_idx_dev, (const uint64_t *)dreq_probe_dev, (size_t)nd, uses_f, S->own_f.as<Slot>(), f_log2,
                               (uint8_t *)dreply_dev);
            if (uses_f)
                hipLaunchKernelGGL(k_own_dbg_set, dim3(blocks_for(nd)), dim3(TPB), 0, s, g->dbg.bits, (uint64_t)g->dbg.lo,
                                   (const uint64_t *)dreq_idx_dev, (const uint64_t *)dreq_probe_dev, (size_t)nd, S->own_f.as<Slot>(), f_log2,
                                   (uint8_t *)dreply_dev);
        }
        if (nc) {
            g->devctr.reserve(DEVCTR_BYTES);
            uint32_t *ctr = g->devctr.as<uint32_t>();
            RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
            hipLaunchKernelGGL(k_own_claim, dim3(blocks_for(nc)), dim3(TPB), 0, s, g->cbf, (uint64_t)g->cbf_lo, (const uint64_t *)creq_idx_dev,
                               (size_t)nc, (uint8_t *)creply_dev, ctr + 16);
            uint32_t nf = 0, spread[16 * 32];
            RB_HIP(hipMemcpyAsync(spread, ctr + 16, sizeof spread, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            for (int q = 0; q < 32; ++q) nf += spread[16 * q];
            if (nf) {
                const uint32_t cs_log2 = log2_ceil(2ull * nf + 2);
                S->own_cs.reserve(sizeof(Slot) << cs_log2);
                RB_HIP(hipMemsetAsync(S->own_cs.p, 0xFF, sizeof(Slot) << cs_log2, s));
                hipLaunchKernelGGL(k_own_cs_build, dim3(blocks_for(nc)), dim3(TPB), 0, s, (const uint64_t *)creq_idx_dev, (const uint8_t *)creply_dev,
                                   (size_t)nc, S->own_cs.as<Slot>(), cs_log2);
                hipLaunchKernelGGL(k_own_claim_fin, dim3(blocks_for(nc)), dim3(TPB), 0, s, (const uint64_t *)creq_idx_dev, (size_t)nc,
                                   S->own_cs.as<Slot>(), cs_log2, (uint8_t *)creply_dev);
            }
        }
        if (np) {
            RB_REQUIRE(g->rpk.bits, "rb_shard_serve: pair probes but no pair filter");
            hipLaunchKernelGGL(k_own_bits, dim3(blocks_for(np)), dim3(TPB), 0, s, g->rpk.bits, (uint64_t)g->rpk.lo, (const uint64_t *)pair_idx_dev, (size_t)np);
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_resolve(rb_graph *g, int mode, const void *dreply_dev, const void *creply_dev, int64_t *w_counts,
                     int64_t *n_conf_ops, int64_t *n_conf_ctr, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && w_counts && n_conf_ops && n_conf_ctr, "rb_shard_resolve: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) w_counts[r] = 0;
        *n_conf_ops = *n_conf_ctr = 0;
        S->slot_bytes[RB_SLOT_W_IDX] = S->slot_bytes[RB_SLOT_W_VAL] = S->slot_bytes[RB_SLOT_CONF_OPS] = S->slot_bytes[RB_SLOT_CONF_CTR] = 0;
        const uint32_t D = S->D;
        if (!D) return;
        FilterView fv = g->view(S->ordinal0, S->pos_bits);
        g->status.reserve((size_t)D * 4); g->nops.reserve((size_t)D * 4); g->cvals.reserve((size_t)D * 8);
        g->heavy.reserve((size_t)D * 4); S->conf_list.reserve((size_t)D * 4); S->cfinal.reserve((size_t)D * 8);
        g->devctr.reserve(DEVCTR_BYTES);
        uint32_t *ctr = g->devctr.as<uint32_t>();
        RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
        hipLaunchKernelGGL(k_resolve_shard, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, g->counts().as<uint32_t>(), g->starts().as<uint32_t>(), D, mode,
                           g->light_ops, S->dreq_pos.as<uint32_t>(), (const uint8_t *)dreply_dev, S->creq_pos.as<uint32_t>(),
                           S->creq_dup.as<uint8_t>(), (const uint8_t *)creply_dev, g->tz().as<uint8_t>(), g->status.as<uint32_t>(),
                           g->nops.as<uint32_t>(), g->cvals.as<uint64_t>(), S->cfinal.as<uint64_t>());
        g->temp.reserve(select_temp_bytes(D));
        select_flagged(g->temp.p, g->temp.cap, g->status.as<uint32_t>(), RUN_HEAVY, D, g->heavy.as<uint32_t>(), ctr + 0, s);
        select_flagged(g->temp.p, g->temp.cap, g->status.as<uint32_t>(), RUN_CONFLICT, D, S->conf_list.as<uint32_t>(), ctr + 1, s);
        uint32_t hc[2];
        RB_HIP(hipMemcpyAsync(hc, ctr, 8, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        if (hc[0])
            hipLaunchKernelGGL(k_cbf_heavy, dim3(std::min<uint32_t>(hc[0], 262144u)), dim3(64), 0, s, fv, g->uniq().as<uint64_t>(), g->counts().as<uint32_t>(),
                               g->starts().as<uint32_t>(), g->vals1().as<uint32_t>(), g->status.as<uint32_t>(), g->nops.as<uint32_t>(),
                               g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), g->heavy.as<uint32_t>(), ctr, S->cfinal.as<uint64_t>());
        // counter writes / releases, bucketed by counter owner
        const size_t nc = (size_t)D * fv.cbf_h;
        S->stage0.reserve(nc * 8); S->stage2.reserve(2 * nc + 32);
        uint8_t *w_val = S->stage2.as<uint8_t>(), *w_drop = w_val + nc;
        hipLaunchKernelGGL(k_emit_writes, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, g->uniq().as<uint64_t>(), D, g->status.as<uint32_t>(),
                           S->creq_dup.as<uint8_t>(), S->cfinal.as<uint64_t>(), S->stage0.as<uint64_t>(), w_val, w_drop);
        size_t kept = route(g, S->stage0.as<uint64_t>(), w_drop, nc, S->span[RB_CBF], w_counts);
        uint64_t *wi = (uint64_t *)slot_reserve(S, RB_SLOT_W_IDX, kept * 8);
        uint8_t *wv = (uint8_t *)slot_reserve(S, RB_SLOT_W_VAL, kept);
        gather_to<uint64_t>(g, S->stage0.as<uint64_t>(), kept, wi);
        gather_to<uint8_t>(g, w_val, kept, wv);
        // conflicting runs: export their ops and counters for the replicated replay
        if (hc[1]) {
            const uint32_t nck = hc[1], ncc = nck * (uint32_t)fv.cbf_h;
            g->conf_sizes.reserve(((size_t)nck + 1) * 4); g->conf_off.reserve(((size_t)nck + 1) * 4);
            hipLaunchKernelGGL(k_conf_sizes, dim3(blocks_for(nck + 1)), dim3(TPB), 0, s, S->conf_list.as<uint32_t>(), g->nops.as<uint32_t>(), nck,
                               g->conf_sizes.as<uint32_t>());
            g->temp.reserve(scan_temp_bytes((size_t)nck + 1));
            exclusive_scan_u32(g->temp.p, g->temp.cap, g->conf_sizes.as<uint32_t>(), g->conf_off.as<uint32_t>(), (size_t)nck + 1, s);
            uint32_t nco = 0;
            RB_HIP(hipMemcpyAsync(&nco, g->conf_off.as<uint32_t>() + nck, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            ConfOp *oo = (ConfOp *)slot_reserve(S, RB_SLOT_CONF_OPS, (size_t)nco * sizeof(ConfOp));
            ConfCtr *oc = (ConfCtr *)slot_reserve(S, RB_SLOT_CONF_CTR, (size_t)ncc * sizeof(ConfCtr));
            hipLaunchKernelGGL(k_conf_export, dim3(blocks_for((int64_t)nck * 64)), dim3(TPB), 0, s, fv, g->uniq().as<uint64_t>(), g->counts().as<uint32_t>(),
                               g->starts().as<uint32_t>(), g->vals1().as<uint32_t>(), g->status.as<uint32_t>(), g->nops.as<uint32_t>(),
                               g->cvals.as<uint64_t>(), S->creq_dup.as<uint8_t>(), S->conf_list.as<uint32_t>(), g->conf_off.as<uint32_t>(), nck,
                               oo, oc);
            *n_conf_ops = nco; *n_conf_ctr = ncc;
            if (stats) stats->conflict_ops += nco;
        }
        if (stats) stats->distinct += D;
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_apply_writes(rb_graph *g, const void *w_idx_dev, const void *w_val_dev, int64_t n) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n >= 0, "rb_shard_apply_writes: bad argument");
        RB_HIP(hipSetDevice(g->p.device));
        if (n) hipLaunchKernelGGL(k_own_writes, dim3(blocks_for(n)), dim3(TPB), 0, g->stream, g->cbf, (uint64_t)g->cbf_lo, (const uint64_t *)w_idx_dev,
                                  (const uint8_t *)w_val_dev, (size_t)n);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_shard_conflict_replay(rb_graph *g, const void *ops_dev, int64_t n_ops, const void *ctr_dev, int64_t n_ctr) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n_ops >= 0 && n_ctr >= 0, "rb_shard_conflict_replay: bad argument");
        if (n_ctr == 0) return;
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        FilterView fv = g->view(S->ordinal0, S->pos_bits);
        const ConfOp *ops = (const ConfOp *)ops_dev;
        // 1. the distinct contested counters and their pre-batch values
        const size_t nc = (size_t)n_ctr;
        S->ck0.reserve(nc * 8); S->ck1.reserve(nc * 8); S->cv0.reserve(nc * 8); S->cv1.reserve(nc * 8);
        S->cuniq.reserve(nc * 8); S->ccnt.reserve((nc + 1) * 4); S->cstart.reserve((nc + 1) * 4); S->cval.reserve(nc * 4);
        hipLaunchKernelGGL(k_split_ctr, dim3(blocks_for((int64_t)nc)), dim3(TPB), 0, s, (const ConfCtr *)ctr_dev, nc, S->ck0.as<uint64_t>(), S->cv0.as<uint64_t>());
        g->temp.reserve(std::max({sort_pairs32_temp_bytes(nc), rle_temp_bytes(nc), scan_temp_bytes(nc + 1), sort_pairs_temp_bytes((size_t)n_ops + 1)}));
        sort_pairs_u64_u64(g->temp.p, g->temp.cap, S->ck0.as<uint64_t>(), S->ck1.as<uint64_t>(), S->cv0.as<uint64_t>(), S->cv1.as<uint64_t>(), nc, 0, 64, s);
        g->devctr.reserve(DEVCTR_BYTES);
        uint32_t *ctr = g->devctr.as<uint32_t>();
        RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
        run_length_encode_u64(g->temp.p, g->temp.cap, S->ck1.as<uint64_t>(), nc, S->cuniq.as<uint64_t>(), S->ccnt.as<uint32_t>(), ctr + 8, s);
        uint32_t M = 0;
        RB_HIP(hipMemcpyAsync(&M, ctr + 8, 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        exclusive_scan_u32(g->temp.p, g->temp.cap, S->ccnt.as<uint32_t>(), S->cstart.as<uint32_t>(), M, s);
        hipLaunchKernelGGL(k_first_vals, dim3(blocks_for(M)), dim3(TPB), 0, s, S->cv1.as<uint64_t>(), S->cstart.as<uint32_t>(), M, S->cval.as<uint32_t>());
        if (n_ops) {
            const size_t no = (size_t)n_ops;
            const int h = fv.cbf_h;
            S->oslots.reserve(no * h * 4); S->olabel.reserve((size_t)M * 4 + 16);
            S->okey0.reserve(no * 8); S->okey1.reserve(no * 8); S->oval0.reserve(no * 4); S->oval1.reserve(no * 4);
            hipLaunchKernelGGL(k_op_slots, dim3(blocks_for((int64_t)no)), dim3(TPB), 0, s, fv, ops, no, S->cuniq.as<uint64_t>(), M,
                               S->oslots.as<uint32_t>(), S->olabel.as<uint32_t>());
            for (int it = 0;; ++it) {   // components of the shares-a-counter graph
                RB_REQUIRE(it < 100000, "conflict component labelling did not converge");
                RB_HIP(hipMemsetAsync(ctr + 3, 0, 4, s));
                hipLaunchKernelGGL(k_op_labels, dim3(blocks_for((int64_t)no)), dim3(TPB), 0, s, S->oslots.as<uint32_t>(), no, h, S->olabel.as<uint32_t>(), ctr + 3);
                uint32_t changed = 0;
                RB_HIP(hipMemcpyAsync(&changed, ctr + 3, 4, hipMemcpyDeviceToHost, s));
                RB_HIP(hipStreamSynchronize(s));
                if (!changed) break;
            }
            hipLaunchKernelGGL(k_op_keys, dim3(blocks_for((int64_t)no)), dim3(TPB), 0, s, ops, S->oslots.as<uint32_t>(), no, h, S->olabel.as<uint32_t>(),
                               S->okey0.as<uint64_t>(), S->oval0.as<uint32_t>());
            sort_pairs_u64_u32(g->temp.p, g->temp.cap, S->okey0.as<uint64_t>(), S->okey1.as<uint64_t>(), S->oval0.as<uint32_t>(), S->oval1.as<uint32_t>(), no, 0, 64, s);
            hipLaunchKernelGGL(k_conf_replay_sparse, dim3(blocks_for((int64_t)no)), dim3(TPB), 0, s, fv, ops, S->oslots.as<uint32_t>(), S->okey1.as<uint64_t>(),
                               S->oval1.as<uint32_t>(), no, S->cval.as<uint32_t>());
        }
        hipLaunchKernelGGL(k_conf_writeback, dim3(blocks_for(M)), dim3(TPB), 0, s, g->cbf, (uint64_t)g->cbf_lo, (uint64_t)g->cbf_hi, S->cuniq.as<uint64_t>(),
                           S->cval.as<uint32_t>(), M);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

}  // extern "C"

namespace rb {
void shard_free(rb_graph *g) {
    ShardState *S = g->shard;
    if (!S) return;
    for (auto &b : S->slot) b.release();
    DevBuf *bufs[] = {&S->dreq_pos, &S->creq_pos, &S->creq_dup, &S->cfinal, &S->conf_list, &S->rkey0, &S->rkey1, &S->rval0, &S->rval1,
                      &S->stage0, &S->stage1, &S->stage2, &S->bounds, &S->own_f, &S->own_cs, &S->own_foreign, &S->ck0, &S->ck1, &S->cv0,
                      &S->cv1, &S->cuniq, &S->ccnt, &S->cstart, &S->cval, &S->oslots, &S->olabel, &S->okey0, &S->okey1, &S->oval0, &S->oval1};
    for (auto *b : bufs) b->release();
    delete S;
    g->shard = nullptr;
}
}  // namespace rb
