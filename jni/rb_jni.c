/* rb_jni.c — the JNI shim between rnabloom.graph.NativeGraph (java/rnabloom/graph/NativeGraph.java) and the C ABI of
 * librb_hip.so (include/rb_capi.h).  One Java_rnabloom_graph_NativeGraph_<method> per static native method; each one
 * unpacks its arguments, calls exactly one rb_* entry point and turns a non-zero status into a Java exception carrying
 * rb_last_error() (the reference's convention is unchecked RuntimeException out of the workers,
 * src/rnabloom/RNABloom.java:903-905).
 *
 * Build where a JDK exists (none does in the image this repository was built in):
 *   gcc -shared -fPIC -O2 -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/rb_jni.c \
 *       -Lrna-bloom_amd/lib -lrb_hip -Wl,-rpath,'$ORIGIN' -o librb_jni.so
 * Without <jni.h> this file compiles to nothing. */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define RB_HAVE_JNI 1
#endif
#endif

#ifdef RB_HAVE_JNI
#define _POSIX_C_SOURCE 200809L   /* open / mmap / ftruncate under -std=c11 */
#include <jni.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <errno.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "rb_capi.h"

#define FN(name) JNIEXPORT JNICALL Java_rnabloom_graph_NativeGraph_##name
#define G(h) ((rb_graph *)(intptr_t)(h))
#define B(h) ((rb_batch *)(intptr_t)(h))

static void throw_rc(JNIEnv *e, int rc) {
    const char *cls = rc == RB_ERR_NOMEM ? "java/lang/OutOfMemoryError"
                    : rc == RB_ERR_INVALID ? "java/lang/IllegalArgumentException"
                    : rc == RB_ERR_STATE ? "java/lang/IllegalStateException" : "java/lang/RuntimeException";
    jclass c = (*e)->FindClass(e, cls);
    if (c) (*e)->ThrowNew(e, c, rb_last_error());
}
/* failures of the file plumbing in this shim (open / fstat / mmap / ftruncate): their own message, an IOException as the reference's save / load throw */
static void throw_io(JNIEnv *e, const char *what, const char *path, const char *why) {
    char msg[512];
    jclass c = (*e)->FindClass(e, "java/io/IOException");
    snprintf(msg, sizeof msg, "%s '%s': %s", what, path ? path : "(null)", why);
    if (c) (*e)->ThrowNew(e, c, msg);
}
static void *direct(JNIEnv *e, jobject buf) { return buf ? (*e)->GetDirectBufferAddress(e, buf) : NULL; }
/* pinned / copied views of primitive arrays; NULL arrays stay NULL */
static jlong *la(JNIEnv *e, jlongArray a) { return a ? (*e)->GetLongArrayElements(e, a, NULL) : NULL; }
static jbyte *ba(JNIEnv *e, jbyteArray a) { return a ? (*e)->GetByteArrayElements(e, a, NULL) : NULL; }
static jfloat *fa(JNIEnv *e, jfloatArray a) { return a ? (*e)->GetFloatArrayElements(e, a, NULL) : NULL; }
static jint *ia(JNIEnv *e, jintArray a) { return a ? (*e)->GetIntArrayElements(e, a, NULL) : NULL; }
static void lr(JNIEnv *e, jlongArray a, jlong *p, jint mode) { if (a) (*e)->ReleaseLongArrayElements(e, a, p, mode); }
static void br(JNIEnv *e, jbyteArray a, jbyte *p, jint mode) { if (a) (*e)->ReleaseByteArrayElements(e, a, p, mode); }
static void fr(JNIEnv *e, jfloatArray a, jfloat *p, jint mode) { if (a) (*e)->ReleaseFloatArrayElements(e, a, p, mode); }
static void ir(JNIEnv *e, jintArray a, jint *p, jint mode) { if (a) (*e)->ReleaseIntArrayElements(e, a, p, mode); }
static jlongArray stats_array(JNIEnv *e, const rb_add_stats *st) {
    jlong v[6] = {st->reads, st->kmers, st->pairs, st->distinct, st->conflict_ops, st->sorted_kmers};
    jlongArray a = (*e)->NewLongArray(e, 6);
    if (a) (*e)->SetLongArrayRegion(e, a, 0, 6, v);
    return a;
}

/* ---- graph lifetime ---- */
jint FN(version)(JNIEnv *e, jclass c) { (void)e; (void)c; return rb_version(); }

jlong FN(create)(JNIEnv *e, jclass c, jlong dbg, jlong cbf, jlong pk, jint dh, jint ch, jint ph, jint k, jboolean stranded,
                 jboolean pairs, jint device, jlong seed) {
    rb_graph_params p;
    rb_graph *g = NULL;
    (void)c;
    memset(&p, 0, sizeof p);
    p.dbgbf_bits = dbg; p.cbf_bytes = cbf; p.pkbf_bits = pk;
    p.dbgbf_num_hash = dh; p.cbf_num_hash = ch; p.pkbf_num_hash = ph; p.k = k;
    p.stranded = stranded; p.use_read_paired_kmers = pairs; p.device = device; p.rng_seed = (uint64_t)seed;
    int rc = rb_graph_create(&p, &g);
    if (rc) { throw_rc(e, rc); return 0; }
    return (jlong)(intptr_t)g;
}
void FN(destroy)(JNIEnv *e, jclass c, jlong h) { (void)c; int rc = rb_graph_destroy(G(h)); if (rc) throw_rc(e, rc); }
void FN(clear)(JNIEnv *e, jclass c, jlong h, jint mask) { (void)c; int rc = rb_graph_clear(G(h), (unsigned)mask); if (rc) throw_rc(e, rc); }
void FN(destroyFilter)(JNIEnv *e, jclass c, jlong h, jint which) { (void)c; int rc = rb_graph_destroy_filter(G(h), which); if (rc) throw_rc(e, rc); }
void FN(setReadPairedKmerDistance)(JNIEnv *e, jclass c, jlong h, jint d) { (void)c; int rc = rb_graph_set_read_paired_kmer_distance(G(h), d); if (rc) throw_rc(e, rc); }
void FN(setFragPairedKmerDistance)(JNIEnv *e, jclass c, jlong h, jint d) { (void)c; int rc = rb_graph_set_frag_paired_kmer_distance(G(h), d); if (rc) throw_rc(e, rc); }
void FN(initFragmentPairs)(JNIEnv *e, jclass c, jlong h, jlong bits, jint nh) { (void)c; int rc = rb_graph_init_fragment_pairs(G(h), bits, nh); if (rc) throw_rc(e, rc); }
jlong FN(getOpOrdinal)(JNIEnv *e, jclass c, jlong h) { uint64_t v = 0; (void)c; int rc = rb_graph_get_op_ordinal(G(h), &v); if (rc) throw_rc(e, rc); return (jlong)v; }
void FN(setOpOrdinal)(JNIEnv *e, jclass c, jlong h, jlong v) { (void)c; int rc = rb_graph_set_op_ordinal(G(h), (uint64_t)v); if (rc) throw_rc(e, rc); }

/* ---- batches ---- */
jlong FN(batchCreateAscii)(JNIEnv *e, jclass c, jint device, jobject seq, jobject qual, jlongArray offsets, jint n, jint min_q) {
    rb_batch *b = NULL;
    jlong *off = la(e, offsets);
    (void)c;
    int rc = rb_batch_create_ascii(device, (const char *)direct(e, seq), (const char *)direct(e, qual), (const int64_t *)off, n, min_q, &b);
    lr(e, offsets, off, JNI_ABORT);
    if (rc) { throw_rc(e, rc); return 0; }
    return (jlong)(intptr_t)b;
}
jlong FN(batchCreateNbits)(JNIEnv *e, jclass c, jint device, jobject bytes, jlong n_bytes, jlong max_reads, jlongArray consumed) {
    rb_batch *b = NULL;
    size_t used = 0;
    (void)c;
    int rc = rb_batch_create_nbits(device, direct(e, bytes), (size_t)n_bytes, max_reads, &b, &used);
    if (rc) { throw_rc(e, rc); return 0; }
    if (consumed) { jlong u = (jlong)used; (*e)->SetLongArrayRegion(e, consumed, 0, 1, &u); }
    return (jlong)(intptr_t)b;
}
/* ---- packed reads in host memory (include/rb_capi.h "read batches packed in HOST memory") ---- */
#define PS(h) ((rb_packed_stream *)(intptr_t)(h))
/* pinned host memory as a direct ByteBuffer (free it with hostFree, never let the collector do it) */
jobject FN(hostAlloc)(JNIEnv *e, jclass c, jlong bytes) {
    void *p = NULL;
    (void)c;
    int rc = rb_host_alloc((size_t)bytes, &p);
    if (rc) { throw_rc(e, rc); return NULL; }
    return (*e)->NewDirectByteBuffer(e, p, bytes);
}
void FN(hostFree)(JNIEnv *e, jclass c, jobject buf) { (void)c; (void)rb_host_free(direct(e, buf)); }
jlong FN(batchDownloadPacked)(JNIEnv *e, jclass c, jlong b, jlong first, jlong n, jobject codes, jobject valid, jobject len) {
    int64_t nw = 0;
    (void)c;
    int rc = rb_batch_download_packed(B(b), first, n, (uint64_t *)direct(e, codes), (uint32_t *)direct(e, valid), (uint32_t *)direct(e, len), &nw);
    if (rc) throw_rc(e, rc);
    return (jlong)nw;
}
jlong FN(packedStreamCreate)(JNIEnv *e, jclass c, jint device, jlong max_reads, jlong max_words) {
    rb_packed_stream *s = NULL;
    (void)c;
    int rc = rb_packed_stream_create(device, max_reads, max_words, &s);
    if (rc) { throw_rc(e, rc); return 0; }
    return (jlong)(intptr_t)s;
}
/* codes / valid / len: direct buffers; the chunk starts wordOffset words / readOffset reads into them */
void FN(packedStreamBegin)(JNIEnv *e, jclass c, jlong s, jobject codes, jobject valid, jobject len, jlong word_off, jlong read_off, jlong n_reads, jlong n_words) {
    const uint64_t *pc = (const uint64_t *)direct(e, codes);
    const uint32_t *pv = (const uint32_t *)direct(e, valid), *pl = (const uint32_t *)direct(e, len);
    (void)c;
    int rc = rb_packed_stream_begin(PS(s), pc ? pc + word_off : NULL, pv ? pv + word_off : NULL, pl ? pl + read_off : NULL, n_reads, n_words);
    if (rc) throw_rc(e, rc);
}
/* the batch handle is borrowed from the stream: pass it to addBatch, never to batchDestroy */
jlong FN(packedStreamFinish)(JNIEnv *e, jclass c, jlong s) {
    const rb_batch *b = NULL;
    (void)c;
    int rc = rb_packed_stream_finish(PS(s), &b);
    if (rc) { throw_rc(e, rc); return 0; }
    return (jlong)(intptr_t)b;
}
void FN(packedStreamDestroy)(JNIEnv *e, jclass c, jlong s) { (void)c; int rc = rb_packed_stream_destroy(PS(s)); if (rc) throw_rc(e, rc); }
jlongArray FN(addPacked)(JNIEnv *e, jclass c, jlong h, jobject codes, jobject valid, jobject len, jlong n_reads, jlong n_words, jlong piece_reads, jint flags) {
    rb_add_stats st;
    (void)c;
    int rc = rb_graph_add_packed(G(h), (const uint64_t *)direct(e, codes), (const uint32_t *)direct(e, valid), (const uint32_t *)direct(e, len), n_reads, n_words,
                                 piece_reads, (unsigned)flags, &st);
    if (rc) { throw_rc(e, rc); return NULL; }
    return stats_array(e, &st);
}
void FN(prefetchPacked)(JNIEnv *e, jclass c, jlong h, jobject codes, jobject valid, jobject len, jlong n_reads, jlong n_words, jlong piece_reads) {
    (void)c;
    int rc = rb_graph_prefetch_packed(G(h), (const uint64_t *)direct(e, codes), (const uint32_t *)direct(e, valid), (const uint32_t *)direct(e, len), n_reads, n_words, piece_reads);
    if (rc) throw_rc(e, rc);
}
jlong FN(batchCreateFastq)(JNIEnv *e, jclass c, jint device, jobject text, jlong len, jboolean final, jint min_q, jboolean use_qual, jlongArray consumed) {
    rb_batch *b = NULL;
    size_t used = 0;
    (void)c;
    int rc = rb_batch_create_fastq(device, (const char *)direct(e, text), (size_t)len, final ? 1 : 0, min_q, use_qual ? 1 : 0, &b, &used);
    if (rc) { throw_rc(e, rc); return 0; }
    if (consumed) { jlong u = (jlong)used; (*e)->SetLongArrayRegion(e, consumed, 0, 1, &u); }
    return (jlong)(intptr_t)b;
}
jlong FN(batchCreateFasta)(JNIEnv *e, jclass c, jint device, jobject text, jlong len, jboolean final, jlongArray consumed) {
    rb_batch *b = NULL;
    size_t used = 0;
    int ended = 0;
    (void)c;
    int rc = rb_batch_create_fasta(device, (const char *)direct(e, text), (size_t)len, final ? 1 : 0, &b, &used, &ended);
    if (rc) { throw_rc(e, rc); return 0; }
    if (consumed) { jlong u[2] = {(jlong)used, (jlong)ended}; (*e)->SetLongArrayRegion(e, consumed, 0, 2, u); }
    return (jlong)(intptr_t)b;
}
void FN(batchDestroy)(JNIEnv *e, jclass c, jlong b) { (void)c; int rc = rb_batch_destroy(B(b)); if (rc) throw_rc(e, rc); }
jlongArray FN(batchInfo)(JNIEnv *e, jclass c, jlong b) {
    int64_t v[3] = {0, 0, 0};
    (void)c;
    int rc = rb_batch_info(B(b), &v[0], &v[1], &v[2]);
    if (rc) { throw_rc(e, rc); return NULL; }
    jlongArray a = (*e)->NewLongArray(e, 3);
    if (a) (*e)->SetLongArrayRegion(e, a, 0, 3, (const jlong *)v);
    return a;
}

/* ---- inserts ---- */
jlongArray FN(addBatch)(JNIEnv *e, jclass c, jlong h, jlong b, jlong first, jlong n, jint flags) {
    rb_add_stats st;
    (void)c;
    int rc = rb_graph_add_batch_range(G(h), B(b), first, n, (unsigned)flags, &st);
    if (rc) { throw_rc(e, rc); return NULL; }
    return stats_array(e, &st);
}
jlongArray FN(addPairs)(JNIEnv *e, jclass c, jlong h, jlong b, jlong first, jlong n, jint which, jint flags) {
    rb_add_stats st;
    (void)c;
    int rc = rb_graph_add_pairs(G(h), B(b), first, n, which, (unsigned)flags, &st);
    if (rc) { throw_rc(e, rc); return NULL; }
    return stats_array(e, &st);
}
jlongArray FN(addFragments)(JNIEnv *e, jclass c, jlong h, jlong b, jlong first, jlong n, jboolean load_pairs) {
    rb_add_stats st;
    (void)c;
    int rc = rb_graph_add_fragments(G(h), B(b), first, n, load_pairs ? 1 : 0, &st);
    if (rc) { throw_rc(e, rc); return NULL; }
    return stats_array(e, &st);
}
jlongArray FN(addReads)(JNIEnv *e, jclass c, jlong h, jobject seq, jobject qual, jlongArray offsets, jint n, jint min_q, jint flags) {
    rb_add_stats st;
    jlong *off = la(e, offsets);
    (void)c;
    int rc = rb_graph_add_reads(G(h), (const char *)direct(e, seq), (const char *)direct(e, qual), (const int64_t *)off, n, min_q, (unsigned)flags, &st);
    lr(e, offsets, off, JNI_ABORT);
    if (rc) { throw_rc(e, rc); return NULL; }
    return stats_array(e, &st);
}
jlongArray FN(addFastq)(JNIEnv *e, jclass c, jlong h, jobject text, jlong len, jint min_q, jint flags, jlongArray n_records) {
    rb_add_stats st;
    int64_t recs = 0;
    (void)c;
    int rc = rb_graph_add_fastq(G(h), (const char *)direct(e, text), (size_t)len, min_q, (unsigned)flags, &st, &recs);
    if (rc) { throw_rc(e, rc); return NULL; }
    if (n_records) { jlong u = (jlong)recs; (*e)->SetLongArrayRegion(e, n_records, 0, 1, &u); }
    return stats_array(e, &st);
}
jlongArray FN(addFastqFile)(JNIEnv *e, jclass c, jlong h, jstring path, jint min_q, jint flags, jlongArray n_records) {
    rb_add_stats st;
    int64_t recs = 0;
    const char *p = (*e)->GetStringUTFChars(e, path, NULL);
    (void)c;
    if (!p) return NULL;      /* OutOfMemoryError pending */
    memset(&st, 0, sizeof st);
    int rc = rb_graph_add_fastq_file(G(h), p, min_q, (unsigned)flags, &st, &recs);
    if (p) (*e)->ReleaseStringUTFChars(e, path, p);
    if (rc) { throw_rc(e, rc); return NULL; }
    if (n_records) { jlong u = (jlong)recs; (*e)->SetLongArrayRegion(e, n_records, 0, 1, &u); }
    return stats_array(e, &st);
}
jlongArray FN(addFastaFile)(JNIEnv *e, jclass c, jlong h, jstring path, jint flags, jlongArray n_records) {
    rb_add_stats st;
    int64_t recs = 0;
    const char *p = (*e)->GetStringUTFChars(e, path, NULL);
    (void)c;
    if (!p) return NULL;      /* OutOfMemoryError pending */
    memset(&st, 0, sizeof st);
    int rc = rb_graph_add_fasta_file(G(h), p, (unsigned)flags, &st, &recs);
    if (p) (*e)->ReleaseStringUTFChars(e, path, p);
    if (rc) { throw_rc(e, rc); return NULL; }
    if (n_records) { jlong u = (jlong)recs; (*e)->SetLongArrayRegion(e, n_records, 0, 1, &u); }
    return stats_array(e, &st);
}
jlongArray FN(addFasta)(JNIEnv *e, jclass c, jlong h, jobject text, jlong len, jint flags, jlongArray n_records) {
    rb_add_stats st;
    int64_t recs = 0;
    (void)c;
    int rc = rb_graph_add_fasta(G(h), (const char *)direct(e, text), (size_t)len, (unsigned)flags, &st, &recs);
    if (rc) { throw_rc(e, rc); return NULL; }
    if (n_records) { jlong u = (jlong)recs; (*e)->SetLongArrayRegion(e, n_records, 0, 1, &u); }
    return stats_array(e, &st);
}
void FN(apply)(JNIEnv *e, jclass c, jlong h, jint op, jlongArray hashes, jint n) {
    jlong *p = la(e, hashes);
    (void)c;
    int rc = rb_graph_apply(G(h), op, (const uint64_t *)p, (size_t)n);
    lr(e, hashes, p, JNI_ABORT);
    if (rc) throw_rc(e, rc);
}

/* ---- queries ---- */
void FN(contains)(JNIEnv *e, jclass c, jlong h, jlongArray hashes, jint n, jbyteArray out) {
    jlong *p = la(e, hashes); jbyte *o = ba(e, out);
    (void)c;
    int rc = rb_graph_contains(G(h), (const uint64_t *)p, (size_t)n, (uint8_t *)o);
    lr(e, hashes, p, JNI_ABORT); br(e, out, o, 0);
    if (rc) throw_rc(e, rc);
}
void FN(getCount)(JNIEnv *e, jclass c, jlong h, jlongArray hashes, jint n, jfloatArray out) {
    jlong *p = la(e, hashes); jfloat *o = fa(e, out);
    (void)c;
    int rc = rb_graph_count(G(h), (const uint64_t *)p, (size_t)n, o);
    lr(e, hashes, p, JNI_ABORT); fr(e, out, o, 0);
    if (rc) throw_rc(e, rc);
}
void FN(filterLookup)(JNIEnv *e, jclass c, jlong h, jint which, jlongArray hashes, jint n, jbyteArray out) {
    jlong *p = la(e, hashes); jbyte *o = ba(e, out);
    (void)c;
    int rc = rb_filter_lookup(G(h), which, (const uint64_t *)p, (size_t)n, (uint8_t *)o);
    lr(e, hashes, p, JNI_ABORT); br(e, out, o, 0);
    if (rc) throw_rc(e, rc);
}
void FN(filterLookupThenAdd)(JNIEnv *e, jclass c, jlong h, jint which, jlongArray hashes, jint n, jbyteArray out) {
    jlong *p = la(e, hashes); jbyte *o = ba(e, out);
    (void)c;
    int rc = rb_filter_lookup_then_add(G(h), which, (const uint64_t *)p, (size_t)n, (uint8_t *)o);
    lr(e, hashes, p, JNI_ABORT); br(e, out, o, 0);
    if (rc) throw_rc(e, rc);
}
void FN(filterGetCount)(JNIEnv *e, jclass c, jlong h, jlongArray hashes, jint n, jfloatArray out) {
    jlong *p = la(e, hashes); jfloat *o = fa(e, out);
    (void)c;
    int rc = rb_filter_get_count(G(h), (const uint64_t *)p, (size_t)n, o);
    lr(e, hashes, p, JNI_ABORT); fr(e, out, o, 0);
    if (rc) throw_rc(e, rc);
}
void FN(filterIncrementAndGet)(JNIEnv *e, jclass c, jlong h, jlongArray hashes, jint n, jfloatArray out) {
    jlong *p = la(e, hashes); jfloat *o = fa(e, out);
    (void)c;
    int rc = rb_filter_increment_and_get(G(h), (const uint64_t *)p, (size_t)n, o);
    lr(e, hashes, p, JNI_ABORT); fr(e, out, o, 0);
    if (rc) throw_rc(e, rc);
}
void FN(getKmers)(JNIEnv *e, jclass c, jlong h, jobject seq, jlongArray offsets, jint n, jlongArray koffsets, jlongArray f, jlongArray r, jfloatArray count) {
    jlong *off = la(e, offsets), *ko = la(e, koffsets), *pf = la(e, f), *pr = la(e, r);
    jfloat *pc = fa(e, count);
    (void)c;
    int rc = rb_graph_kmers(G(h), (const char *)direct(e, seq), (const int64_t *)off, n, (int64_t *)ko, (uint64_t *)pf, (uint64_t *)pr, pc);
    lr(e, offsets, off, JNI_ABORT); lr(e, koffsets, ko, 0); lr(e, f, pf, 0); lr(e, r, pr, 0); fr(e, count, pc, 0);
    if (rc) throw_rc(e, rc);
}
jlong FN(batchCounts)(JNIEnv *e, jclass c, jlong h, jlong batch, jlong first, jlong n, jlongArray koffsets, jfloatArray out) {
    jlong *ko = la(e, koffsets);
    jfloat *po = fa(e, out);
    int64_t stride = 0;
    (void)c;
    int rc = rb_graph_batch_counts(G(h), B(batch), first, n, (const int64_t *)ko, po, 0, &stride);
    lr(e, koffsets, ko, JNI_ABORT); fr(e, out, po, 0);
    if (rc) throw_rc(e, rc);
    return (jlong)stride;
}
void FN(neighbors)(JNIEnv *e, jclass c, jlong h, jlongArray f, jlongArray r, jbyteArray ch, jint n, jint direction, jlongArray f4, jlongArray r4, jfloatArray c4) {
    jlong *pf = la(e, f), *pr = la(e, r), *of = la(e, f4), *orr = la(e, r4);
    jbyte *pc = ba(e, ch);
    jfloat *oc = fa(e, c4);
    (void)c;
    int rc = rb_graph_neighbors(G(h), (const uint64_t *)pf, (const uint64_t *)pr, (const uint8_t *)pc, (size_t)n, direction, (uint64_t *)of, (uint64_t *)orr, oc);
    lr(e, f, pf, JNI_ABORT); lr(e, r, pr, JNI_ABORT); br(e, ch, pc, JNI_ABORT); lr(e, f4, of, 0); lr(e, r4, orr, 0); fr(e, c4, oc, 0);
    if (rc) throw_rc(e, rc);
}
void FN(walk)(JNIEnv *e, jclass c, jlong h, jbyteArray seeds, jbyteArray targets, jint n, jint direction, jint bound, jfloat min_cov,
              jbyteArray out_bases, jlongArray out_f, jlongArray out_r, jfloatArray out_count, jintArray out_len, jbyteArray out_reason) {
    jbyte *ps = ba(e, seeds), *pt = ba(e, targets), *ob = ba(e, out_bases), *orr = ba(e, out_reason);
    jlong *of = la(e, out_f), *orv = la(e, out_r);
    jfloat *oc = fa(e, out_count);
    jint *ol = ia(e, out_len);
    (void)c;
    int rc = rb_graph_walk(G(h), (const char *)ps, (const char *)pt, (size_t)n, direction, bound, min_cov, (char *)ob, (uint64_t *)of, (uint64_t *)orv, oc,
                           (int32_t *)ol, (uint8_t *)orr);
    br(e, seeds, ps, JNI_ABORT); br(e, targets, pt, JNI_ABORT); br(e, out_bases, ob, 0); br(e, out_reason, orr, 0);
    lr(e, out_f, of, 0); lr(e, out_r, orv, 0); fr(e, out_count, oc, 0); ir(e, out_len, ol, 0);
    if (rc) throw_rc(e, rc);
}
void FN(greedyExtend)(JNIEnv *e, jclass c, jlong h, jlong gate, jbyteArray seeds, jint n, jint direction, jint lookahead, jint bound,
                      jbyteArray out_bases, jfloatArray out_count, jintArray out_len, jbyteArray out_reason) {
    jbyte *ps = ba(e, seeds), *ob = ba(e, out_bases), *orr = ba(e, out_reason);
    jfloat *oc = fa(e, out_count);
    jint *ol = ia(e, out_len);
    (void)c;
    int rc = rb_graph_greedy_extend(G(h), gate ? G(gate) : NULL, (const char *)ps, (size_t)n, direction, lookahead, bound, (char *)ob, oc, (int32_t *)ol, (uint8_t *)orr);
    br(e, seeds, ps, JNI_ABORT); br(e, out_bases, ob, 0); br(e, out_reason, orr, 0); fr(e, out_count, oc, 0); ir(e, out_len, ol, 0);
    if (rc) throw_rc(e, rc);
}

void FN(naiveExtend)(JNIEnv *e, jclass c, jlong h, jbyteArray seeds, jint n, jint direction, jint mode, jint bound, jint cap, jfloat min_cov,
                     jbyteArray term_seq, jlongArray term_off, jbyteArray out_bases, jintArray out_len, jbyteArray out_reason) {
    jbyte *ps = ba(e, seeds), *pt = ba(e, term_seq), *ob = ba(e, out_bases), *orr = ba(e, out_reason);
    jlong *po = la(e, term_off);
    jint *ol = ia(e, out_len);
    (void)c;
    int rc = rb_graph_naive_extend(G(h), (const char *)ps, (size_t)n, direction, mode, bound, cap, min_cov, (const char *)pt, (const int64_t *)po, (char *)ob,
                                   (int32_t *)ol, (uint8_t *)orr);
    br(e, seeds, ps, JNI_ABORT); br(e, term_seq, pt, JNI_ABORT); lr(e, term_off, po, JNI_ABORT); br(e, out_bases, ob, 0); br(e, out_reason, orr, 0); ir(e, out_len, ol, 0);
    if (rc) throw_rc(e, rc);
}

/* ---- filter state ---- */
jlongArray FN(filterSize)(JNIEnv *e, jclass c, jlong h, jint which) {
    int64_t size = 0, nbytes = 0;
    int nh = 0;
    (void)c;
    int rc = rb_filter_size(G(h), which, &size, &nbytes, &nh);
    if (rc) { throw_rc(e, rc); return NULL; }
    jlong v[3] = {size, nbytes, nh};
    jlongArray a = (*e)->NewLongArray(e, 3);
    if (a) (*e)->SetLongArrayRegion(e, a, 0, 3, v);
    return a;
}
jlong FN(popcount)(JNIEnv *e, jclass c, jlong h, jint which) { int64_t v = 0; (void)c; int rc = rb_filter_popcount(G(h), which, &v); if (rc) throw_rc(e, rc); return v; }
jlong FN(fold)(JNIEnv *e, jclass c, jlong h, jint which) { uint64_t v = 0; (void)c; int rc = rb_filter_fold(G(h), which, &v); if (rc) throw_rc(e, rc); return (jlong)v; }
jfloat FN(fpr)(JNIEnv *e, jclass c, jlong h, jint which) { float v = 0; (void)c; int rc = rb_filter_fpr(G(h), which, &v); if (rc) throw_rc(e, rc); return v; }
void FN(exportFilter)(JNIEnv *e, jclass c, jlong h, jint which, jobject dst, jlong n) { (void)c; int rc = rb_filter_export(G(h), which, direct(e, dst), (size_t)n); if (rc) throw_rc(e, rc); }
void FN(importFilter)(JNIEnv *e, jclass c, jlong h, jint which, jobject src, jlong n) { (void)c; int rc = rb_filter_import(G(h), which, direct(e, src), (size_t)n); if (rc) throw_rc(e, rc); }
/* filters of 2 GiB and more (a direct ByteBuffer holds less): the file is mapped here and handed to the same two calls */
void FN(importFilterFromFile)(JNIEnv *e, jclass c, jlong h, jint which, jstring path, jlong n) {
    const char *p = (*e)->GetStringUTFChars(e, path, NULL);
    int rc = RB_OK, fd, io = 0;
    struct stat sb;
    (void)c;
    if (!p) return;      /* OutOfMemoryError is pending: calling back into JNI now would be undefined; the exception propagates */
    fd = open(p, O_RDONLY);
    if (fd < 0) { throw_io(e, "cannot open filter file", p, strerror(errno)); io = 1; }
    else if (fstat(fd, &sb) != 0) { throw_io(e, "cannot stat filter file", p, strerror(errno)); io = 1; }
    else if ((int64_t)sb.st_size < (int64_t)n) {        /* a truncated file would be a SIGBUS inside the import, not an exception */
        char why[96];
        snprintf(why, sizeof why, "%lld bytes, the filter needs %lld", (long long)sb.st_size, (long long)n);
        throw_io(e, "filter file is too short", p, why); io = 1;
    } else {
        void *m = mmap(NULL, (size_t)n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { throw_io(e, "cannot map filter file", p, strerror(errno)); io = 1; }
        else { rc = rb_filter_import(G(h), which, m, (size_t)n); munmap(m, (size_t)n); }
    }
    if (fd >= 0) close(fd);
    if (p) (*e)->ReleaseStringUTFChars(e, path, p);
    if (!io && rc) throw_rc(e, rc);
}
void FN(exportFilterToFile)(JNIEnv *e, jclass c, jlong h, jint which, jstring path, jlong n) {
    const char *p = (*e)->GetStringUTFChars(e, path, NULL);
    int rc = RB_OK, fd, io = 0;
    (void)c;
    if (!p) return;      /* (as above) */
    fd = open(p, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { throw_io(e, "cannot create filter file", p, strerror(errno)); io = 1; }
    else if (ftruncate(fd, (off_t)n) != 0) { throw_io(e, "cannot size filter file", p, strerror(errno)); io = 1; }
    else {
        void *m = mmap(NULL, (size_t)n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { throw_io(e, "cannot map filter file", p, strerror(errno)); io = 1; }
        else { rc = rb_filter_export(G(h), which, m, (size_t)n); munmap(m, (size_t)n); }
    }
    if (fd >= 0) close(fd);
    if (p) (*e)->ReleaseStringUTFChars(e, path, p);
    if (!io && rc) throw_rc(e, rc);
}
jlong FN(expectedSize)(JNIEnv *e, jclass c, jlong n, jfloat fpr, jint nh) { (void)e; (void)c; return rb_expected_size(n, fpr, nh); }
void FN(cbfToBloom)(JNIEnv *e, jclass c, jlong src, jfloat min_cov, jlong dst, jint which) { (void)c; int rc = rb_cbf_to_bloom(G(src), min_cov, G(dst), which); if (rc) throw_rc(e, rc); }

/* ---- hash-only work ---- */
jlong FN(minimizers)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint n, jint k, jint w, jint mode, jlongArray offs_out,
                     jlongArray out_hash, jlongArray out_pos) {
    jlong *off = la(e, offsets), *mo = la(e, offs_out), *oh = la(e, out_hash), *op = la(e, out_pos);
    (void)c;
    int rc = rb_minimizers(device, (const char *)direct(e, seq), (const int64_t *)off, n, k, w, mode, (int64_t *)mo, (uint64_t *)oh, (int64_t *)op);
    jlong total = rc ? 0 : mo[n];
    lr(e, offsets, off, JNI_ABORT); lr(e, offs_out, mo, 0); lr(e, out_hash, oh, 0); lr(e, out_pos, op, 0);
    if (rc) throw_rc(e, rc);
    return total;
}
jlong FN(minimizersNext)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint n, jint k, jint w, jint mode, jlongArray offs_out,
                         jlongArray out_hash, jlongArray out_pos) {
    jlong *off = la(e, offsets), *mo = la(e, offs_out), *oh = la(e, out_hash), *op = la(e, out_pos);
    (void)c;
    int rc = rb_minimizers_next(device, (const char *)direct(e, seq), (const int64_t *)off, n, k, w, mode, (int64_t *)mo, (uint64_t *)oh, (int64_t *)op);
    jlong total = rc ? 0 : mo[n];
    lr(e, offsets, off, JNI_ABORT); lr(e, offs_out, mo, 0); lr(e, out_hash, oh, 0); lr(e, out_pos, op, 0);
    if (rc) throw_rc(e, rc);
    return total;
}
jlong FN(minimizerSet)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint n, jint k, jint w, jint mode, jlongArray stale,
                       jlongArray offs_out, jlongArray out) {
    jlong *off = la(e, offsets), *st = la(e, stale), *mo = la(e, offs_out), *o = la(e, out);
    (void)c;
    int rc = rb_minimizer_set(device, (const char *)direct(e, seq), (const int64_t *)off, n, k, w, mode, (const uint64_t *)st, (int64_t *)mo, (uint64_t *)o);
    jlong total = rc ? 0 : mo[n];
    lr(e, offsets, off, JNI_ABORT); lr(e, stale, st, JNI_ABORT); lr(e, offs_out, mo, 0); lr(e, out, o, 0);
    if (rc) throw_rc(e, rc);
    return total;
}
jlong FN(strobemers)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint nr, jint k, jint n, jint wmin, jint wmax, jlongArray offs_out,
                     jlongArray out_hash, jintArray out_start, jintArray out_end) {
    jlong *off = la(e, offsets), *so = la(e, offs_out), *oh = la(e, out_hash);
    jint *os = ia(e, out_start), *oe = ia(e, out_end);
    (void)c;
    int rc = rb_strobemers(device, (const char *)direct(e, seq), (const int64_t *)off, nr, k, n, wmin, wmax, (int64_t *)so, (uint64_t *)oh, (int32_t *)os, (int32_t *)oe);
    jlong total = rc ? 0 : so[nr];
    lr(e, offsets, off, JNI_ABORT); lr(e, offs_out, so, 0); lr(e, out_hash, oh, 0); ir(e, out_start, os, 0); ir(e, out_end, oe, 0);
    if (rc) throw_rc(e, rc);
    return total;
}
jlong FN(randstrobes)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint nr, jint k, jint n, jint wmin, jint wmax, jint flags, jlong count_in,
                      jlongArray offs_out, jlongArray out_hash, jintArray out_pos, jfloatArray out_count) {
    jlong *off = la(e, offsets), *so = la(e, offs_out), *oh = la(e, out_hash);
    jint *op = ia(e, out_pos);
    jfloat *oc = fa(e, out_count);
    (void)c;
    int rc = rb_randstrobes(device, (const char *)direct(e, seq), (const int64_t *)off, nr, k, n, wmin, wmax, flags, count_in ? G(count_in) : NULL, (int64_t *)so,
                            (uint64_t *)oh, (int32_t *)op, oc);
    jlong total = rc ? 0 : so[nr];
    lr(e, offsets, off, JNI_ABORT); lr(e, offs_out, so, 0); lr(e, out_hash, oh, 0); ir(e, out_pos, op, 0); fr(e, out_count, oc, 0);
    if (rc) throw_rc(e, rc);
    return total;
}
jlong FN(strobe3)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint nr, jint k, jint wmin, jint wmax, jboolean canonical, jlong count_in,
                  jlongArray offs_out, jlongArray out_hash, jintArray out_pos, jfloatArray out_count) {
    jlong *off = la(e, offsets), *so = la(e, offs_out), *oh = la(e, out_hash);
    jint *op = ia(e, out_pos);
    jfloat *oc = fa(e, out_count);
    (void)c;
    int rc = rb_strobe3(device, (const char *)direct(e, seq), (const int64_t *)off, nr, k, wmin, wmax, canonical ? 1 : 0, count_in ? G(count_in) : NULL, (int64_t *)so,
                        (uint64_t *)oh, (int32_t *)op, oc);
    jlong total = rc ? 0 : so[nr];
    lr(e, offsets, off, JNI_ABORT); lr(e, offs_out, so, 0); lr(e, out_hash, oh, 0); ir(e, out_pos, op, 0); fr(e, out_count, oc, 0);
    if (rc) throw_rc(e, rc);
    return total;
}
jlong FN(kmerPairHashes)(JNIEnv *e, jclass c, jint device, jobject seq, jlongArray offsets, jint nr, jint k, jint shift, jboolean canonical, jlong count_in,
                         jlongArray offs_out, jlongArray out_hash, jfloatArray out_count) {
    jlong *off = la(e, offsets), *po = la(e, offs_out), *oh = la(e, out_hash);
    jfloat *oc = fa(e, out_count);
    (void)c;
    int rc = rb_kmer_pair_hashes(device, (const char *)direct(e, seq), (const int64_t *)off, nr, k, shift, canonical ? 1 : 0, count_in ? G(count_in) : NULL, (int64_t *)po,
                                 (uint64_t *)oh, oc);
    jlong total = rc ? 0 : po[nr];
    lr(e, offsets, off, JNI_ABORT); lr(e, offs_out, po, 0); lr(e, out_hash, oh, 0); fr(e, out_count, oc, 0);
    if (rc) throw_rc(e, rc);
    return total;
}

/* ---- input formats ---- */
jlong FN(fastqSplit)(JNIEnv *e, jclass c, jobject text, jlong len, jint threads, jobject seq, jobject qual, jlongArray offsets) {
    int64_t n = 0;
    jlong *off = la(e, offsets);
    jsize cap = offsets ? (*e)->GetArrayLength(e, offsets) - 1 : 0;
    (void)c;
    int rc = rb_fastq_split((const char *)direct(e, text), (size_t)len, threads, (char *)direct(e, seq), (char *)direct(e, qual), (int64_t *)off, cap, &n);
    lr(e, offsets, off, 0);
    if (rc) throw_rc(e, rc);
    return n;
}
jlong FN(gunzip)(JNIEnv *e, jclass c, jobject src, jlong n, jint threads, jobject dst, jlong cap) {
    size_t out = 0;
    (void)c;
    int rc = rb_gunzip(direct(e, src), (size_t)n, threads, direct(e, dst), (size_t)cap, &out);
    if (rc) throw_rc(e, rc);
    return (jlong)out;
}
jlong FN(nbitsEncode)(JNIEnv *e, jclass c, jobject seq, jlongArray offsets, jint n, jobject out, jlong cap) {
    size_t written = 0;
    jlong *off = la(e, offsets);
    (void)c;
    int rc = rb_nbits_encode((const char *)direct(e, seq), (const int64_t *)off, n, direct(e, out), (size_t)cap, &written);
    lr(e, offsets, off, JNI_ABORT);
    if (rc) throw_rc(e, rc);
    return (jlong)written;
}
#endif /* RB_HAVE_JNI */
