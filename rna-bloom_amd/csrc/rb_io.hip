// rb_io.hip — host-side text ingest: FASTQ records -> the concatenated sequence / quality buffers + offsets that
// rb_graph_add_reads / rb_batch_create_ascii take (the 2-bit encoding and the quality segmentation happen on the GPU).
// Reference: FastqReader.nextWithoutName R/io/FastqReader.java:140-186 — four lines per record, line 1 starts with '@',
// line 3 with '+', a record cut off by the end of the input is dropped (NoSuchElementException -> null);
// BufferedReader.lines() ends a line at \n, \r\n or \r.  The reference reads under one lock (:144-149), which is what
// stops its stage 1 from scaling; here line starts are counted per chunk in parallel, and since a record is exactly four
// lines the global line number of every chunk start tells which field a line is.
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "rb_pipeline.hpp"

using namespace rb;

namespace {
struct Chunk { size_t b, e; int64_t lines_before = 0; };

// a line START is position 0 and every position after an end of line (\n, or \r not followed by \n)
inline bool eol_at(const char *t, size_t n, size_t i) { return t[i] == '\n' || (t[i] == '\r' && !(i + 1 < n && t[i + 1] == '\n')); }
}  // namespace

extern "C" {

int rb_fastq_split(const char *text, size_t len, int n_threads, char *seq, char *qual, int64_t *offsets, int64_t cap_reads, int64_t *n_reads) {
    return guarded([&] {
        RB_REQUIRE((text || len == 0) && n_reads, "rb_fastq_split: null argument");
        const int T = std::max(1, std::min(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency(), 64));
        std::vector<Chunk> ch((size_t)T);
        for (int t = 0; t < T; ++t) { ch[(size_t)t].b = len * (size_t)t / (size_t)T; ch[(size_t)t].e = len * (size_t)(t + 1) / (size_t)T; }
        // pass 1: ends of line per chunk
        std::vector<int64_t> eols((size_t)T, 0);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                int64_t c = 0;
                for (size_t i = ch[(size_t)t].b; i < ch[(size_t)t].e; ++i) c += eol_at(text, len, i);
                eols[(size_t)t] = c;
            });
            for (auto &x : th) x.join();
        }
        int64_t total_eols = 0;
        for (int t = 0; t < T; ++t) { ch[(size_t)t].lines_before = total_eols; total_eols += eols[(size_t)t]; }
        // complete lines: one per end of line, plus a last line without one
        const bool open_tail = len > 0 && !eol_at(text, len, len - 1);
        const int64_t n_lines = total_eols + (open_tail ? 1 : 0);
        const int64_t records = n_lines / 4;                                // a truncated record is dropped
        *n_reads = records;
        if (!offsets) return;                                               // count only
        RB_REQUIRE(cap_reads >= records, "rb_fastq_split: room for %lld reads, %lld present", (long long)cap_reads, (long long)records);
        // pass 2: per chunk, the start and length of every line whose number is 4r+1 (sequence) or 4r+3 (quality)
        std::vector<int64_t> seq_len((size_t)records, 0);
        std::vector<size_t> seq_pos((size_t)records, 0), qual_pos((size_t)records, 0);
        std::vector<int64_t> qual_len((size_t)records, 0);
        std::vector<int> bad((size_t)T, 0);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                // the lines that START in [b, e): a line starts at 0 and after every end of line; its number is the number
                // of ends of line before its start
                const size_t b = ch[(size_t)t].b, e = ch[(size_t)t].e;
                int64_t line = ch[(size_t)t].lines_before;
                size_t s = b;
                if (b > 0 && !eol_at(text, len, b - 1)) {                  // b is inside a line that started earlier
                    while (s < e && !eol_at(text, len, s)) ++s;
                    if (s >= e) return;                                     // no line starts in this chunk
                    ++s; ++line;
                }
                while (s < e && s < len) {
                    size_t j = s;
                    while (j < len && !eol_at(text, len, j)) ++j;           // j = end of line position (or len)
                    size_t ce = j;
                    if (j < len && text[j] == '\n' && ce > s && text[ce - 1] == '\r') --ce;   // \r\n
                    const int64_t rec = line >> 2, field = line & 3;
                    if (rec < records) {
                        if (field == 0) { if (ce == s || text[s] != '@') bad[(size_t)t] = 1; }
                        else if (field == 1) { seq_pos[(size_t)rec] = s; seq_len[(size_t)rec] = (int64_t)(ce - s); }
                        else if (field == 2) { if (ce == s || text[s] != '+') bad[(size_t)t] = 2; }
                        else { qual_pos[(size_t)rec] = s; qual_len[(size_t)rec] = (int64_t)(ce - s); }
                    }
                    s = j + 1; ++line;
                }
            });
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < T; ++t) {
            RB_REQUIRE(bad[(size_t)t] != 1, "rb_fastq_split: Line 1 of FASTQ record is expected to start with '@'");
            RB_REQUIRE(bad[(size_t)t] != 2, "rb_fastq_split: Line 3 of FASTQ record is expected to start with '+'");
        }
        offsets[0] = 0;
        for (int64_t r = 0; r < records; ++r) {
            RB_REQUIRE(!qual || qual_len[(size_t)r] == seq_len[(size_t)r], "rb_fastq_split: record %lld has %lld bases and %lld qualities", (long long)r,
                       (long long)seq_len[(size_t)r], (long long)qual_len[(size_t)r]);
            offsets[r + 1] = offsets[r] + seq_len[(size_t)r];
        }
        if (!seq) return;
        {   // pass 3: copy the fields
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                for (int64_t r = records * t / T; r < records * (t + 1) / T; ++r) {
                    memcpy(seq + offsets[r], text + seq_pos[(size_t)r], (size_t)seq_len[(size_t)r]);
                    if (qual) memcpy(qual + offsets[r], text + qual_pos[(size_t)r], (size_t)seq_len[(size_t)r]);
                }
            });
            for (auto &x : th) x.join();
        }
    });
}

}  // extern "C"
