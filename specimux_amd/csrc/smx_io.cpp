// smx_io.cpp -- host streaming helpers of libsmx.so: FASTQ/FASTA reader, window packer, output writer.
// The steps on either side of the GPU hot path; plain C++17 + zlib + std::thread,
// no device work.
//
// Reference behaviour restated (paths relative to the reference repo):
//   reader : Bio.SeqIO "fastq"/"fasta" as used by open_sequence_file (src/specimux/io_utils.py:429-450) --
//            FastqGeneralIterator rules: title after '@', sequence lines until a line starting with '+', at
//            least one quality line, further quality lines until one starts with '@' AND the quality is
//            already as long as the sequence; id = first whitespace-delimited word of the title.
//   writer : create_write_operation's orientation + slicing (demultiplex.py:74-78) and
//            OutputManager.write_sequence / _make_filename (io_utils.py:197-268).
//
// Two reader engines:
//   * general (serial): any legal FASTQ / FASTA, plain or gzip, through zlib; records are copied into the batch.
//   * fast (parallel, zero-copy): uncompressed FASTQ with strict 4-line records.  A block of the file is read
//     once, split at record boundaries ('@' line whose line+2 starts with '+') among worker threads, and every
//     worker only builds record descriptors that point into the block.  The first irregular record (wrapped
//     lines, blank lines) switches the reader permanently to the general engine at that block's offset.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "smx.h"

extern "C" int smx_set_error(int code, const char *fmt, ...);   // smx_api.cpp (thread-local message)

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// any white space in [p, e)?  Every base of the input passes through here (a wrapped or blank-padded sequence line sends
// the file to the general engine): 16 bytes per step -- no byte of a clean line is <= ' ' -- and the byte loop only for a
// block that holds one (a control character is not white space).
inline bool has_space(const char *p, const char *e) {
#if defined(__SSE2__)
    const __m128i sp = _mm_set1_epi8(0x20);
    for (; p + 16 <= e; p += 16) {
        const __m128i v = _mm_loadu_si128((const __m128i *)p);
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(v, sp), v))) {
            for (int k = 0; k < 16; k++) if (is_space((unsigned char)p[k])) return true;
        }
    }
#endif
    for (; p < e; p++) if (is_space((unsigned char)*p)) return true;
    return false;
}

int io_threads() {
    static int n = [] {
        int v = 0;
        if (const char *e = getenv("SMX_IO_THREADS")) v = atoi(e);
        if (v <= 0) {
            v = (int)std::thread::hardware_concurrency();
            // respect a cgroup CPU quota if there is one
            if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
                char q[32];
                long period = 0;
                if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                    long cpus = atol(q) / period;
                    if (cpus >= 1 && cpus < v) v = (int)cpus;
                }
                fclose(f);
            }
            // reader, packer and writer overlap and spend part of their time in page-cache system calls: half as many threads
            // again as CPUs keeps the cores busy (measured on the 16-core GPU box, 765k reads file -> tree: 16 threads 0.218 s,
            // 20 0.202 s, 24 0.177 s), more only add contention
            v = std::min(v + v / 2, 24);
        }
        return std::max(v, 1);
    }();
    return n;
}

// uninitialised byte buffer (a std::vector would zero-fill hundreds of MB per batch before they are overwritten)
struct Blob {
    char *p = nullptr;
    size_t n = 0;
    explicit Blob(size_t len) : p((char *)malloc(len ? len : 1)), n(len) {}
    ~Blob() { free(p); }
    Blob(const Blob &) = delete;
    Blob &operator=(const Blob &) = delete;
    char *data() { return p; }
    void resize(size_t m) { n = m; }   // shrink only
};

// read-only mapping of the whole input file (fast engine): batches point straight into the page cache
struct Mapping {
    const char *p = nullptr;
    size_t n = 0;
    ~Mapping() { if (p) munmap((void *)p, n); }
};

struct Rec {
    uint64_t id_off, seq_off, qual_off;   // into the segment's base; qual_off == UINT64_MAX for FASTA
    uint32_t id_len, seq_len;
    uint64_t src_off;                     // fast engine: offset of the record's '@' inside the block
};

struct Segment {
    const char *base = nullptr;           // own.data() (general engine) or a pointer into the shared block
    std::vector<char> own;
    std::vector<Rec> recs;
};

}  // namespace

struct smx_batch {
    std::vector<Segment> segs;
    std::vector<uint32_t> first;          // first[i] = global index of segs[i].recs[0]; first.back() = total
    std::shared_ptr<void> block;   // fast engine: what the segments point into (a file mapping, or an inflated / copied block)
    uint32_t n = 0;
    // mapped input: the whole pages only this batch's records lie in.  Dropping their page-table entries when the batch
    // goes (madvise, the mapping itself stays reserved) spreads the cost of unmapping a large file -- ~20 ms per GB,
    // otherwise paid in one piece after the last record is written -- over the run, on whichever thread retires batches.
    char *rel_lo = nullptr;
    size_t rel_n = 0;

    void release_pages() {
        if (rel_n) madvise(rel_lo, rel_n, MADV_DONTNEED);
        rel_lo = nullptr;
        rel_n = 0;
    }
    void clear() { segs.clear(); first.clear(); release_pages(); block.reset(); n = 0; }
    ~smx_batch() { release_pages(); }
    void finish() {
        first.assign(segs.size() + 1, 0);
        for (size_t i = 0; i < segs.size(); i++) first[i + 1] = first[i] + (uint32_t)segs[i].recs.size();
        n = first.back();
    }
    inline void locate(uint32_t i, const Segment **s, const Rec **r) const {
        size_t k = (size_t)(std::upper_bound(first.begin(), first.end(), i) - first.begin()) - 1;
        *s = &segs[k];
        *r = &segs[k].recs[i - first[k]];
    }
};

struct smx_reader {
    std::string path;
    bool fastq = true;
    // ---- general engine
    gzFile gz = nullptr;
    std::vector<char> buf;    // unconsumed bytes [pos, end)
    size_t pos = 0, end = 0;
    bool eof = false;
    uint64_t line_no = 0;
    // ---- fast engine (plain FASTQ)
    int fd = -1;
    uint64_t fsize = 0, fpos = 0;
    bool fast = false;
    std::shared_ptr<Mapping> map;   // the file, mapped read-only (nullptr: blocks are copied with pread)
    // ---- fast engine on gzip input: inflate a block, then parse it in parallel
    bool gzfast = false, gz_eof = false;
    std::vector<char> carry;
    uint64_t upos = 0;
    bool ranged = false;      // opened on a byte range: only the parallel plain-FASTQ engine can honour it
    bool gz_failed = false;   // sticky: a gzread error or a stream that ends inside a gzip member
    std::string gz_msg;

    // gzread returned <= 0: a clean end of input only if zlib reports no error AND is at end of file.  A corrupt
    // stream gives -1 (Z_DATA_ERROR); a truncated one gives 0 with Z_BUF_ERROR -- Python's gzip raises on both
    // (EOFError / BadGzipFile, what the reference's open_sequence_file sees), so neither may pass as end of file.
    bool clean_eof(int got) {
        int errnum = Z_OK;
        const char *msg = gzerror(gz, &errnum);
        if (got == 0 && (errnum == Z_OK || errnum == Z_STREAM_END) && gzeof(gz)) return true;
        gz_failed = true;
        gz_msg = (errnum == Z_BUF_ERROR || errnum == Z_OK) ? "unexpected end of compressed stream"
                                                          : (msg && *msg ? msg : "gzip stream error");
        return false;
    }

    bool next_line(const char **p, size_t *n) {
        for (;;) {
            const char *s = buf.data() + pos;
            const char *nl = (const char *)memchr(s, '\n', end - pos);
            if (nl) { *p = s; *n = (size_t)(nl - s); pos += *n + 1; line_no++; return true; }
            if (eof) {
                if (pos == end) return false;
                *p = s; *n = end - pos; pos = end; line_no++;
                return true;
            }
            if (pos > 0) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
            if (end == buf.size()) buf.resize(buf.size() * 2);
            int got = gzread(gz, buf.data() + end, (unsigned)std::min<size_t>(buf.size() - end, 1u << 30));
            if (got <= 0) { (void)clean_eof(got); eof = true; } else end += (size_t)got;
        }
    }
    int peek() {
        for (;;) {
            if (pos < end) return (unsigned char)buf[pos];
            if (eof) return -1;
            pos = end = 0;
            int got = gzread(gz, buf.data(), (unsigned)buf.size());
            if (got <= 0) { (void)clean_eof(got); eof = true; } else end = (size_t)got;
        }
    }
};

namespace {

void append_trimmed(std::vector<char> &dst, const char *p, size_t n, bool drop_inner_spaces) {
    while (n && is_space((unsigned char)p[n - 1])) n--;
    size_t a = 0;
    while (a < n && is_space((unsigned char)p[a])) a++;
    if (!drop_inner_spaces) { dst.insert(dst.end(), p + a, p + n); return; }
    for (size_t i = a; i < n; i++) if (p[i] != ' ') dst.push_back(p[i]);
}

// ---------------------------------------------------------------- general engine (serial, copies records)
int next_general(smx_reader *r, uint32_t max_reads, uint64_t max_bytes, smx_batch *b) {
    b->segs.emplace_back();
    Segment &sg = b->segs.back();
    std::vector<char> &data = sg.own;
    const char *p;
    size_t n;
    while (sg.recs.size() < max_reads && (max_bytes == 0 || data.size() < max_bytes)) {
        Rec rec;
        rec.src_off = 0;
        if (r->fastq) {
            bool have = false;
            while ((have = r->next_line(&p, &n))) {   // skip blank lines between records
                size_t k = n;
                while (k && is_space((unsigned char)p[k - 1])) k--;
                if (k) break;
            }
            if (!have) break;
            if (p[0] != '@') return smx_set_error(SMX_ERR_ARG, "line %llu: Records in Fastq files should start with '@' character", (unsigned long long)r->line_no);
            size_t a = 1;
            while (a < n && is_space((unsigned char)p[a])) a++;
            size_t e = a;
            while (e < n && !is_space((unsigned char)p[e])) e++;
            rec.id_off = data.size();
            rec.id_len = (uint32_t)(e - a);
            data.insert(data.end(), p + a, p + e);
            rec.seq_off = data.size();
            bool plus = false;
            while (r->next_line(&p, &n)) {
                if (n && p[0] == '+') { plus = true; break; }
                append_trimmed(data, p, n, false);
            }
            if (!plus) return smx_set_error(SMX_ERR_ARG, "line %llu: End of file without quality information.", (unsigned long long)r->line_no);
            uint64_t slen = data.size() - rec.seq_off;
            if (slen > 0x7FFFFFFFull) return smx_set_error(SMX_ERR_UNSUPPORTED, "read longer than 2^31-1 bases");
            rec.seq_len = (uint32_t)slen;
            rec.qual_off = data.size();
            if (r->next_line(&p, &n)) append_trimmed(data, p, n, false);
            for (;;) {
                int c = r->peek();
                if (c < 0) break;
                uint64_t qlen = data.size() - rec.qual_off;
                if (c == '@' && qlen >= slen) break;
                if (!r->next_line(&p, &n)) break;
                append_trimmed(data, p, n, false);
            }
            if (data.size() - rec.qual_off != slen)
                return smx_set_error(SMX_ERR_ARG, "line %llu: Lengths of sequence and quality values differs (%llu and %llu).",
                                     (unsigned long long)r->line_no, (unsigned long long)slen,
                                     (unsigned long long)(data.size() - rec.qual_off));
        } else {
            bool have = false;
            while ((have = r->next_line(&p, &n))) if (n && p[0] == '>') break;
            if (!have) break;
            size_t a = 1;
            while (a < n && is_space((unsigned char)p[a])) a++;
            size_t e = a;
            while (e < n && !is_space((unsigned char)p[e])) e++;
            rec.id_off = data.size();
            rec.id_len = (uint32_t)(e - a);
            data.insert(data.end(), p + a, p + e);
            rec.seq_off = data.size();
            for (;;) {
                int c = r->peek();
                if (c < 0 || c == '>') break;
                if (!r->next_line(&p, &n)) break;
                append_trimmed(data, p, n, true);
            }
            uint64_t slen = data.size() - rec.seq_off;
            if (slen > 0x7FFFFFFFull) return smx_set_error(SMX_ERR_UNSUPPORTED, "read longer than 2^31-1 bases");
            rec.seq_len = (uint32_t)slen;
            rec.qual_off = UINT64_MAX;
        }
        sg.recs.push_back(rec);
    }
    sg.base = data.data();
    return SMX_OK;
}

// ---------------------------------------------------------------- fast engine (parallel, zero-copy)
inline const char *line_end(const char *p, const char *end) {
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    return nl ? nl : end;
}

// First position >= from that starts a strict record: a line starting with '@' whose line+2 starts with '+'.
// `from` must be at a line start.  Returns end if there is none.
const char *find_record_start(const char *from, const char *end) {
    const char *p = from;
    while (p < end) {
        const char *e1 = line_end(p, end);
        if (*p == '@' && e1 < end) {
            const char *l2 = e1 + 1;
            if (l2 < end) {
                const char *e2 = line_end(l2, end);
                if (e2 < end && e2 + 1 < end && e2[1] == '+') return p;
            }
        }
        if (e1 >= end) break;
        p = e1 + 1;
    }
    return end;
}

// Parse strict 4-line records in [p, stop) (records STARTING before stop), never reading past `end`.
// Returns false on the first irregular record.  *consumed = where parsing stopped (a record start or end).
bool parse_strict(const char *base, const char *p, const char *stop, const char *end, bool last_block,
                  std::vector<Rec> &out, const char **consumed) {
    while (p < stop) {
        if (*p != '@') return false;
        const char *e1 = line_end(p, end);
        if (e1 >= end) { *consumed = p; return true; }            // incomplete record: leave it for the next block
        const char *s = e1 + 1, *e2 = line_end(s, end);
        if (e2 >= end) { *consumed = p; return true; }
        const char *pl = e2 + 1;
        if (pl >= end) { *consumed = p; return true; }
        if (*pl != '+') return false;                              // wrapped sequence or empty read: general engine
        const char *e3 = line_end(pl, end);
        if (e3 >= end) { *consumed = p; return true; }
        const char *q = e3 + 1, *e4 = line_end(q, end);
        if (e4 >= end && !last_block) { *consumed = p; return true; }   // quality line may be cut by the block end
        Rec r;
        r.src_off = (uint64_t)(p - base);
        const char *a = p + 1;
        while (a < e1 && is_space((unsigned char)*a)) a++;
        const char *ie = a;
        while (ie < e1 && !is_space((unsigned char)*ie)) ie++;
        r.id_off = (uint64_t)(a - base);
        r.id_len = (uint32_t)(ie - a);
        const char *se = e2;
        while (se > s && is_space((unsigned char)se[-1])) se--;
        const char *qe = e4;
        while (qe > q && is_space((unsigned char)qe[-1])) qe--;
        if (se == s || (size_t)(se - s) != (size_t)(qe - q)) return false;   // empty or wrapped: general engine decides
        if (has_space(s, se)) return false;
        if ((uint64_t)(se - s) > 0x7FFFFFFFull) return false;
        r.seq_off = (uint64_t)(s - base);
        r.seq_len = (uint32_t)(se - s);
        r.qual_off = (uint64_t)(q - base);
        out.push_back(r);
        p = e4 < end ? e4 + 1 : end;
    }
    *consumed = p;
    return true;
}

// Parse one in-memory block of strict FASTQ with all I/O threads.  Returns 1 = records placed in `b` and *next_off =
// offset of the first unconsumed byte; 0 = irregular input (the general engine must take over from the block start);
// 2 = no complete record in the block (caller retries with a bigger block); 3 = nothing but white space up to the
// end of the input.
int parse_block(const std::shared_ptr<void> &block, const char *base, uint64_t len, bool last_block, uint32_t max_reads, smx_batch *b,
                uint64_t *next_off_out) {
    const char *end = base + len;
    const int T = io_threads();
    std::vector<const char *> cut(T + 1);
    cut[0] = base;
    cut[T] = end;
    for (int t = 1; t < T; t++) {
        const char *nominal = base + len * (uint64_t)t / (uint64_t)T;
        const char *ls = nominal;   // back up to a line start
        while (ls > base && ls[-1] != '\n') ls--;
        cut[t] = find_record_start(ls, end);
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    std::vector<std::vector<Rec>> parts(T);
    std::vector<const char *> stopped(T, nullptr);
    std::vector<char> ok(T, 1);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            if (cut[t] >= cut[t + 1]) { stopped[t] = cut[t]; return; }
            ok[t] = parse_strict(base, cut[t], cut[t + 1], end, last_block, parts[t], &stopped[t]) ? 1 : 0;
        });
    for (auto &x : th) x.join();
    for (int t = 0; t < T; t++) if (!ok[t]) return 0;
    // parts must chain exactly (every cut is a genuine record start); only the part holding the block's last
    // record may stop early, at the start of a record the block end cuts in two
    const char *consumed = end;
    bool tail_open = false;
    for (int t = 0; t < T; t++) {
        if (cut[t] >= cut[t + 1]) continue;
        if (tail_open) { if (!parts[t].empty()) return 0; continue; }
        if (stopped[t] == cut[t + 1]) continue;
        if (stopped[t] < cut[t + 1]) { consumed = stopped[t]; tail_open = true; }
        else return 0;
    }
    uint64_t total = 0;
    for (auto &v : parts) total += v.size();
    if (total == 0) {
        if (last_block) {   // trailing blank lines / garbage: let the general engine judge it
            for (const char *c = base; c < end; c++) if (!is_space((unsigned char)*c)) return 0;
            return 3;
        }
        return 2;           // one record larger than the block
    }
    // keep at most max_reads records
    uint64_t keep = std::min<uint64_t>(total, max_reads);
    uint64_t next_off = (uint64_t)(consumed - base);
    b->block = block;
    uint64_t seen = 0;
    for (int t = 0; t < T && seen < keep; t++) {
        if (parts[t].empty()) continue;
        uint64_t take = std::min<uint64_t>(parts[t].size(), keep - seen);
        if (take < parts[t].size()) { next_off = parts[t][take].src_off; parts[t].resize(take); }
        else if (seen + take == keep && keep < total) {
            for (int u = t + 1; u < T; u++) if (!parts[u].empty()) { next_off = parts[u][0].src_off; break; }
        }
        b->segs.emplace_back();
        b->segs.back().base = base;
        b->segs.back().recs = std::move(parts[t]);
        seen += take;
    }
    *next_off_out = next_off;
    return 1;
}

// Returns 1 = batch filled by the fast engine, 0 = irregular input: caller must use the general engine from fpos.
int next_fast(smx_reader *r, uint32_t max_reads, uint64_t max_bytes, smx_batch *b) {
    if (r->fpos >= r->fsize) return 1;   // end of file: empty batch
    uint64_t want = max_bytes ? max_bytes : (256ull << 20);
    // do not read far more than max_reads records' worth (a surplus is cut off and re-read by the next call)
    want = std::min<uint64_t>(want, (uint64_t)max_reads * 4096 + (1u << 20));
    for (;;) {
        uint64_t len = std::min<uint64_t>(want, r->fsize - r->fpos);
        bool last_block = r->fpos + len >= r->fsize;
        if (r->map) {
            // zero-copy: the block IS the page cache (mapped read-only); the parser threads fault their own parts in.  What
            // the copying path below spends on moving every input byte once more is most of a reader's time on cached files.
            uint64_t next_off = 0;
            int rc = parse_block(r->map, r->map->p + r->fpos, len, last_block, max_reads, b, &next_off);
            if (rc == 0) return 0;
            if (rc == 3) { r->fpos = r->fsize; return 1; }
            if (rc == 2) { want *= 2; continue; }
            if (b->block) {   // records placed: [fpos, fpos + next_off) belongs to this batch alone, minus its two edge pages
                const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
                const uintptr_t lo = ((uintptr_t)(r->map->p + r->fpos) + page - 1) & ~(page - 1);
                const uintptr_t hi = (uintptr_t)(r->map->p + r->fpos + next_off) & ~(page - 1);
                if (hi > lo) { b->rel_lo = (char *)lo; b->rel_n = (size_t)(hi - lo); }
            }
            r->fpos += next_off;
            return 1;
        }
        auto block = std::make_shared<Blob>(len);
        if (!block->data()) { smx_set_error(SMX_ERR_ARG, "out of memory reading %s", r->path.c_str()); return -1; }
        // the block is read by all I/O threads at once (page-cache copies scale with threads)
        const int TR = io_threads();
        std::vector<uint64_t> got_t(TR, 0);
        std::vector<int> err_t(TR, 0);
        {
            std::vector<std::thread> rt;
            for (int t = 0; t < TR; t++)
                rt.emplace_back([&, t] {
                    uint64_t lo = len * (uint64_t)t / (uint64_t)TR, hi = len * (uint64_t)(t + 1) / (uint64_t)TR, g = 0;
                    while (lo + g < hi) {
                        ssize_t k = pread(r->fd, block->data() + lo + g, (size_t)std::min<uint64_t>(hi - lo - g, 1u << 30),
                                          (off_t)(r->fpos + lo + g));
                        if (k < 0) { if (errno == EINTR) continue; err_t[t] = errno; break; }
                        if (k == 0) break;
                        g += (uint64_t)k;
                    }
                    got_t[t] = g;
                });
            for (auto &x : rt) x.join();
        }
        uint64_t got = 0;
        for (int t = 0; t < TR; t++) {
            if (err_t[t]) { smx_set_error(SMX_ERR_ARG, "read %s: %s", r->path.c_str(), strerror(err_t[t])); return -1; }
            uint64_t lo = len * (uint64_t)t / (uint64_t)TR, hi = len * (uint64_t)(t + 1) / (uint64_t)TR;
            got = lo + got_t[t];
            if (got_t[t] < hi - lo) break;   // short read: the file ends here
        }
        if (got < len) { len = got; last_block = true; block->resize(len); }
        uint64_t next_off = 0;
        int rc = parse_block(block, block->data(), len, last_block, max_reads, b, &next_off);
        if (rc == 0) return 0;
        if (rc == 3) { r->fpos = r->fsize; return 1; }
        if (rc == 2) { want *= 2; continue; }
        r->fpos += next_off;
        return 1;
    }
}

// gzip input: one thread inflates a block (zlib is a single stream), then all I/O threads parse it -- the general
// engine would do both serially.  `carry` holds the inflated bytes after the last complete record of the previous block;
// upos is the uncompressed offset of carry[0] (where the general engine resumes if the input turns out irregular).
int next_fast_gz(smx_reader *r, uint32_t max_reads, uint64_t max_bytes, smx_batch *b) {
    uint64_t want = max_bytes ? std::min<uint64_t>(max_bytes, 96ull << 20) : (96ull << 20);
    want = std::min<uint64_t>(want, (uint64_t)max_reads * 4096 + (1u << 20));
    for (;;) {
        if (r->gz_eof && r->carry.empty()) return 1;   // end of input: empty batch
        auto block = std::make_shared<Blob>(r->carry.size() + want);
        if (!block->data()) { smx_set_error(SMX_ERR_ARG, "out of memory reading %s", r->path.c_str()); return -1; }
        if (!r->carry.empty()) memcpy(block->data(), r->carry.data(), r->carry.size());
        uint64_t len = r->carry.size();
        while (!r->gz_eof && len < r->carry.size() + want) {
            int got = gzread(r->gz, block->data() + len, (unsigned)std::min<uint64_t>(r->carry.size() + want - len, 1u << 30));
            if (got <= 0) {
                if (!r->clean_eof(got)) { smx_set_error(SMX_ERR_ARG, "read %s: %s", r->path.c_str(), r->gz_msg.c_str()); return -1; }
                r->gz_eof = true;
                break;
            }
            len += (uint64_t)got;
        }
        block->resize(len);
        uint64_t next_off = 0;
        int rc = parse_block(block, block->data(), len, r->gz_eof, max_reads, b, &next_off);
        if (rc == 0) return 0;
        if (rc == 3) { r->carry.clear(); r->upos += len; return 1; }
        if (rc == 2) { r->carry.assign(block->data(), block->data() + len); want *= 2; continue; }
        r->carry.assign(block->data() + next_off, block->data() + len);
        r->upos += next_off;
        return 1;
    }
}

std::vector<std::string> split_names(const char *blob, const uint32_t *off, uint32_t n) {
    std::vector<std::string> v(n);
    for (uint32_t i = 0; i < n; i++) v[i].assign(blob + off[i], off[i + 1] - off[i]);
    return v;
}

std::string safe_name(const std::string &s) {   // io_utils.py:207: chars outside [A-Za-z0-9._-$#] -> '_'
    std::string r = s;
    for (char &c : r) {
        unsigned char u = (unsigned char)c;
        bool ok = (u >= '0' && u <= '9') || (u >= 'A' && u <= 'Z') || (u >= 'a' && u <= 'z') || u == '.' || u == '_' ||
                  u == '-' || u == '$' || u == '#' || u >= 0x80;
        if (!ok) c = '_';
    }
    return r;
}

}  // namespace

// ---------------------------------------------------------------- writer
// Reversed copies of a read (half of the matched reads are written reverse-complemented, io_utils.py:289-297): out[i] =
// f(end[-1 - i]).  A byte loop runs at ~1 GB/s per core, a third of memcpy; 16 bytes per step with pshufb (the byte reversal,
// and the complement as four 16-entry tables selected by the high nibble: every letter lives in 0x40..0x7F) runs at memcpy speed.
#if defined(__x86_64__)
namespace {
__attribute__((target("ssse3"))) void reverse_copy_ssse3(char *o, const char *end, size_t n) {
    const __m128i rev = _mm_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    size_t i = 0;
    for (; i + 16 <= n; i += 16)
        _mm_storeu_si128((__m128i *)(o + i), _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(end - i - 16)), rev));
    for (; i < n; i++) o[i] = end[-(int64_t)i - 1];
}
__attribute__((target("ssse3"))) void revcomp_copy_ssse3(char *o, const char *end, size_t n, const unsigned char *comp) {
    const __m128i rev = _mm_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m128i low = _mm_set1_epi8(0x0F);
    __m128i tab[4], sel[4];
    for (int h = 0; h < 4; h++) {
        tab[h] = _mm_loadu_si128((const __m128i *)(comp + 0x40 + 16 * h));
        sel[h] = _mm_set1_epi8((char)(4 + h));
    }
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        __m128i v = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(end - i - 16)), rev);
        const __m128i lo = _mm_and_si128(v, low), hi = _mm_and_si128(_mm_srli_epi16(v, 4), low);
        for (int h = 0; h < 4; h++) {
            const __m128i m = _mm_cmpeq_epi8(hi, sel[h]);
            v = _mm_or_si128(_mm_and_si128(m, _mm_shuffle_epi8(tab[h], lo)), _mm_andnot_si128(m, v));
        }
        _mm_storeu_si128((__m128i *)(o + i), v);
    }
    for (; i < n; i++) o[i] = (char)comp[(unsigned char)end[-(int64_t)i - 1]];
}
bool have_ssse3() { static const bool ok = (__builtin_cpu_init(), __builtin_cpu_supports("ssse3")); return ok; }
}  // namespace
#else
namespace { bool have_ssse3() { return false; } }
#endif

namespace {
// `comp` must be the identity outside 0x40..0x7F (smx_writer_open builds it that way)
inline void reverse_copy(char *o, const char *end, size_t n) {
#if defined(__x86_64__)
    if (have_ssse3()) { reverse_copy_ssse3(o, end, n); return; }
#endif
    for (size_t i = 0; i < n; i++) o[i] = end[-(int64_t)i - 1];
}
inline void revcomp_copy(char *o, const char *end, size_t n, const unsigned char *comp) {
#if defined(__x86_64__)
    if (have_ssse3()) { revcomp_copy_ssse3(o, end, n, comp); return; }
#endif
    for (size_t i = 0; i < n; i++) o[i] = (char)comp[(unsigned char)end[-(int64_t)i - 1]];
}
}  // namespace

struct smx_writer {
    std::string out_dir, prefix;
    bool fastq = true;
    std::vector<std::string> specimens, pools, primers, barcodes;
    unsigned char comp[256];
    // What a file still has to receive: pieces that are either literal bytes (an offset into `lit`: record headers,
    // separators, reverse-complemented bodies) or pointers into the batch's input (forward reads are written straight
    // from the parsed input -- the mapped page cache -- by writev: no formatted copy of their bases and qualities exists).
    // Input pointers die with the batch: smx_writer_write flushes every file it touched before it returns.
    struct Piece { const char *src; size_t off, n; };   // src == nullptr: lit[off, off + n)
    struct File {
        std::string path, lit;
        std::vector<Piece> pieces;
        size_t bytes = 0;
        bool dir_made = false, dirty = false;
        void literal(const char *p, size_t n) {
            if (!n) return;
            const size_t at = lit.size();
            lit.append(p, n);
            if (!pieces.empty() && !pieces.back().src && pieces.back().off + pieces.back().n == at) pieces.back().n += n;
            else pieces.push_back(Piece{nullptr, at, n});
            bytes += n;
        }
        char *literal_space(size_t n) {   // n bytes of literal the caller fills in
            const size_t at = lit.size();
            lit.resize(at + n);
            if (!pieces.empty() && !pieces.back().src && pieces.back().off + pieces.back().n == at) pieces.back().n += n;
            else pieces.push_back(Piece{nullptr, at, n});
            bytes += n;
            return &lit[at];
        }
        void source(const char *p, size_t n) {
            if (!n) return;
            if (!pieces.empty() && pieces.back().src && pieces.back().src + pieces.back().n == p) pieces.back().n += n;
            else pieces.push_back(Piece{p, 0, n});
            bytes += n;
        }
    };
    struct Shard {   // one per writer thread: owns a disjoint set of output files
        std::vector<File> files;
        std::vector<size_t> touched;                        // files with pending pieces (flushed at the end of the batch)
        std::unordered_map<std::string, size_t> index;
        std::unordered_map<uint64_t, size_t> by_key;        // packed (class, pool, primers, sample) -> file: no string
        std::unordered_map<uint64_t, std::string> tails;    // work per record once a file has been seen
        std::vector<struct iovec> iov;
        int first_errno = 0;
        int rc = 0;
    };
    std::vector<Shard> shards;
    std::vector<std::vector<uint32_t>> route;   // [slice of the batch][shard]: the operations of that slice the shard writes
    size_t flush_bytes = 64u << 10;    // a file's pending records are appended once they exceed this (SMX_IO_FLUSH_KB; measured
                                       // on the 16-core GPU box, 765k reads: 32-64 KB 0.25 s, 256 KB 0.32 s, 1 MB 0.49 s file -> tree)

    static void mkdirs(const std::string &path) {
        for (size_t i = 1; i < path.size(); i++)
            if (path[i] == '/') { std::string d = path.substr(0, i); mkdir(d.c_str(), 0777); }
    }
    static void flush(Shard &sh, File &f) {
        f.dirty = false;
        if (f.pieces.empty()) return;
        if (!f.dir_made) { mkdirs(f.path); f.dir_made = true; }
        int fd = open(f.path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0666);
        if (fd < 0) { if (!sh.first_errno) sh.first_errno = errno; }
        else {
            sh.iov.clear();
            for (const Piece &pc : f.pieces) {
                struct iovec v;
                v.iov_base = (void *)(pc.src ? pc.src : f.lit.data() + pc.off);
                v.iov_len = pc.n;
                sh.iov.push_back(v);
            }
            size_t k = 0;
            while (k < sh.iov.size()) {
                const size_t cnt = std::min<size_t>(sh.iov.size() - k, 1024);   // IOV_MAX
                ssize_t w = writev(fd, &sh.iov[k], (int)cnt);
                if (w < 0) { if (errno == EINTR) continue; if (!sh.first_errno) sh.first_errno = errno; break; }
                size_t left = (size_t)w;
                while (k < sh.iov.size() && left >= sh.iov[k].iov_len) { left -= sh.iov[k].iov_len; k++; }
                if (left) { sh.iov[k].iov_base = (char *)sh.iov[k].iov_base + left; sh.iov[k].iov_len -= left; }
            }
            close(fd);
        }
        f.pieces.clear();
        f.lit.clear();
        f.bytes = 0;
    }
    size_t file_for(Shard &sh, const std::string &rel) {
        auto it = sh.index.find(rel);
        if (it != sh.index.end()) return it->second;
        sh.files.emplace_back();
        sh.files.back().path = out_dir + "/" + rel;
        sh.index.emplace(rel, sh.files.size() - 1);
        return sh.files.size() - 1;
    }
};

namespace {

inline uint32_t mix(uint32_t h, uint32_t v) { h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2); return h; }

// Which shard owns the primary / pool-level file of an operation (pure function of the file identity).
inline void owners(const smx_op &op, uint32_t T, uint32_t *primary, uint32_t *pool_level) {
    uint32_t cls = op.rtype == SMX_R_UNKNOWN ? 2u : ((op.rtype == SMX_R_PARTIAL_FWD || op.rtype == SMX_R_PARTIAL_REV) ? 1u : 0u);
    uint32_t skey = op.sample >= 0 ? (uint32_t)op.sample
                                   : (cls == 1 ? 0x40000000u + (uint32_t)(op.rtype == SMX_R_PARTIAL_REV) * 0x10000u + (uint32_t)(uint16_t)op.barcode
                                               : 0x7FFFFFFFu);
    uint32_t h = mix(mix(mix(mix(cls * 7919u, (uint32_t)(uint16_t)op.pool), (uint32_t)(uint16_t)op.p1), (uint32_t)(uint16_t)op.p2), skey);
    *pool_level = mix(mix(0xABCDu, (uint32_t)(uint16_t)op.pool), skey) % T;
    // a full match goes to two files (primer-pair directory and pool level, io_utils.py:256-268): one shard owns both, so the
    // record is formatted -- reverse-complemented, for half of the reads -- once and copied, not formatted twice by two threads
    *primary = cls == 0 ? *pool_level : h % T;
}

// Identity of an output file as one integer: record class, pool, primer pair, sample / partial barcode.
inline uint64_t file_key(const smx_op &op, bool pool_level) {
    uint64_t cls = op.rtype == SMX_R_UNKNOWN ? 2u : ((op.rtype == SMX_R_PARTIAL_FWD || op.rtype == SMX_R_PARTIAL_REV) ? 1u : 0u);
    uint64_t skey = op.sample >= 0 ? (uint64_t)(uint32_t)op.sample
                                   : (cls == 1 ? (1ull << 29) | ((uint64_t)(op.rtype == SMX_R_PARTIAL_REV) << 16) | (uint16_t)op.barcode
                                               : (1ull << 30) - 1);
    if (pool_level) return (3ull << 62) | ((uint64_t)(uint16_t)op.pool << 46) | skey;
    return (cls << 62) | ((uint64_t)(uint16_t)op.pool << 46) | ((uint64_t)(uint8_t)(op.p1 + 1) << 38) |
           ((uint64_t)(uint8_t)(op.p2 + 1) << 30) | skey;
}

int write_one(smx_writer *w, smx_writer::Shard &sh, uint32_t me, uint32_t T, const smx_batch *b, const smx_op &op) {
    if (op.rtype == SMX_R_FILTERED) return SMX_OK;
    uint32_t o1, o2;
    owners(op, T, &o1, &o2);
    const bool full = op.rtype == SMX_R_FULL || op.rtype == SMX_R_DEREP_FULL;
    const bool mine1 = o1 == me, mine2 = full && o2 == me;
    if (!mine1 && !mine2) return SMX_OK;
    if (op.read >= b->n) return smx_set_error(SMX_ERR_ARG, "write operation refers to read %u of %u", op.read, b->n);
    const Segment *sg;
    const Rec *rp;
    b->locate(op.read, &sg, &rp);
    const Rec &r = *rp;
    const char *base = sg->base;
    auto name = [](const std::vector<std::string> &v, int i) -> const std::string & {
        static const std::string unknown = "unknown";
        return (i >= 0 && (size_t)i < v.size()) ? v[(size_t)i] : unknown;
    };
    // names, header tail and paths are built once per distinct file; afterwards a record costs two hash lookups
    const uint64_t k1 = file_key(op, false);
    auto tit = sh.tails.find(k1);
    size_t f1 = (size_t)-1, f2 = (size_t)-1;
    if (tit == sh.tails.end() || (mine1 && sh.by_key.find(k1) == sh.by_key.end()) ||
        (mine2 && sh.by_key.find(file_key(op, true)) == sh.by_key.end())) {
        std::string sample;
        if (op.sample >= 0) sample = name(w->specimens, op.sample);
        else if (op.rtype == SMX_R_PARTIAL_FWD) sample = "barcode_fwd_" + name(w->barcodes, op.barcode);
        else if (op.rtype == SMX_R_PARTIAL_REV) sample = "barcode_rev_" + name(w->barcodes, op.barcode);
        else sample = "unknown";
        const std::string &pool = name(w->pools, op.pool), &p1 = name(w->primers, op.p1), &p2 = name(w->primers, op.p2);
        if (tit == sh.tails.end())
            tit = sh.tails.emplace(k1, " pool=" + pool + " primers=" + p1 + "+" + p2 + " " + sample + "\n").first;
        const char *top = op.rtype == SMX_R_UNKNOWN ? "unknown" : ((op.rtype == SMX_R_PARTIAL_FWD || op.rtype == SMX_R_PARTIAL_REV) ? "partial" : "full");
        const std::string ext = w->fastq ? ".fastq" : ".fasta";
        const std::string fname = w->prefix + safe_name(sample) + ext;
        if (mine1 && sh.by_key.find(k1) == sh.by_key.end())
            sh.by_key.emplace(k1, w->file_for(sh, std::string(top) + "/" + pool + "/" + p1 + "-" + p2 + "/" + fname));
        if (mine2 && sh.by_key.find(file_key(op, true)) == sh.by_key.end())   // pool-level aggregate (io_utils.py:256-268)
            sh.by_key.emplace(file_key(op, true), w->file_for(sh, "full/" + pool + "/" + fname));
    }
    if (mine1) f1 = sh.by_key.find(k1)->second;
    if (mine2) f2 = sh.by_key.find(file_key(op, true))->second;
    const std::string &tail = tit->second;
    int64_t L = r.seq_len, s = op.trim_start, e = op.trim_end;
    if (s < 0) s = 0;   // the kernel only emits 0 <= s < e <= L for non-empty reads
    if (e > L) e = L;
    if (e < s) e = s;
    const size_t n = (size_t)(e - s);
    // header: '@' id ' ' distance code, then the per-file tail (" pool=... primers=... sample\n")
    char head[512];
    size_t hn = 0;
    head[hn++] = w->fastq ? '@' : '>';
    std::string long_head;
    const bool big_id = (size_t)r.id_len + 64 > sizeof(head);
    if (!big_id) { memcpy(head + hn, base + r.id_off, r.id_len); hn += r.id_len; }
    char dist[24];
    size_t dn = 0;
    dist[dn++] = ' ';
    for (int k = 0; k < 4; k++) {
        if (k) dist[dn++] = ',';
        int d = op.dist[k];
        if (d < 0) dist[dn++] = 'X';
        else { if (d >= 100) dist[dn++] = (char)('0' + d / 100); if (d >= 10) dist[dn++] = (char)('0' + (d / 10) % 10); dist[dn++] = (char)('0' + d % 10); }
    }
    const char *seq = base + r.seq_off;
    const char *qual = r.qual_off == UINT64_MAX ? nullptr : base + r.qual_off;
    const bool rev = (op.flags & SMX_OPF_REVERSE) != 0;
    for (int dest = 0; dest < 2; dest++) {
        if (dest == 0 ? !mine1 : !mine2) continue;
        smx_writer::File &f = sh.files[dest == 0 ? f1 : f2];
        if (!f.dirty) { f.dirty = true; sh.touched.push_back(dest == 0 ? f1 : f2); }
        if (big_id) { f.literal(head, 1); f.literal(base + r.id_off, r.id_len); }
        else f.literal(head, hn);
        f.literal(dist, dn);
        f.literal(tail.data(), tail.size());
        if (!rev) {
            // forward: bases and qualities go out from where the parser found them; an untrimmed record whose input
            // lines are "SEQ\n+\nQUAL" in one piece is a single pointer
            if (w->fastq && qual && n == (size_t)L && qual == seq + L + 3 && seq[L] == '\n' && seq[L + 1] == '+' && seq[L + 2] == '\n')
                f.source(seq, 2 * n + 3);
            else {
                f.source(seq + s, n);
                if (w->fastq) {
                    f.literal("\n+\n", 3);
                    if (qual) f.source(qual + s, n);
                    else memset(f.literal_space(n), 'I', n);
                }
            }
            f.literal("\n", 1);
        } else {
            // reverse complement: formatted once (first destination) and copied to the second
            const size_t body = w->fastq ? 2 * n + 4 : n + 1;
            char *o = f.literal_space(body);
            if (dest == 1 && mine1) {
                const smx_writer::File &f0 = sh.files[f1];
                memcpy(o, f0.lit.data() + f0.lit.size() - body, body);
            } else {
                revcomp_copy(o, seq + (L - s), n, w->comp);
                o[n] = '\n';
                if (w->fastq) {
                    o[n + 1] = '+';
                    o[n + 2] = '\n';
                    if (qual) reverse_copy(o + n + 3, qual + (L - s), n);
                    else memset(o + n + 3, 'I', n);
                    o[2 * n + 3] = '\n';
                }
            }
        }
    }
    // (at most ~1000 pieces pending: a flush is then ONE writev, i.e. one append that no other process's append can land
    // inside -- ranks of a multi-GPU run append to the same files, specimux_amd/distributed.py)
    if (mine1 && (sh.files[f1].bytes > w->flush_bytes || sh.files[f1].pieces.size() > 960)) smx_writer::flush(sh, sh.files[f1]);
    if (mine2 && (sh.files[f2].bytes > w->flush_bytes || sh.files[f2].pieces.size() > 960)) smx_writer::flush(sh, sh.files[f2]);
    return SMX_OK;
}

}  // namespace

extern "C" {

int smx_reader_open(const char *path, smx_reader **out, int *is_fastq) {
    if (!path || !out) return smx_set_error(SMX_ERR_ARG, "null argument");
    // format by extension (compression suffixes stripped), then by first byte (io_utils.py:380-426)
    std::string base(path);
    size_t slash = base.find_last_of('/');
    if (slash != std::string::npos) base = base.substr(slash + 1);
    auto ends = [](const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; };
    std::string low = base;
    for (char &c : low) c = (char)tolower((unsigned char)c);
    bool compressed_name = false;
    for (bool again = true; again;) {
        again = false;
        for (const char *ext : {".gz", ".gzip", ".bz2", ".zip"})
            if (ends(low, ext)) { low.resize(low.size() - strlen(ext)); again = true; compressed_name = true; }
    }
    int fmt = 0;
    if (ends(low, ".fastq") || ends(low, ".fq")) fmt = 1;
    else if (ends(low, ".fasta") || ends(low, ".fa") || ends(low, ".fna")) fmt = 2;
    gzFile gz = gzopen(path, "rb");
    if (!gz) return smx_set_error(SMX_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
    gzbuffer(gz, 1u << 20);
    smx_reader *r = new smx_reader();
    r->path = path;
    r->gz = gz;
    r->buf.resize(8u << 20);
    if (fmt == 0) {
        int c = r->peek();
        fmt = (c == '@') ? 1 : 2;   // '>' or anything else: FASTA (the reference's default)
    }
    r->fastq = fmt == 1;
    // fast engine: uncompressed FASTQ only (gzdirect() is 1 when zlib is passing the bytes through)
    (void)r->peek();
    if (r->gz_failed) {
        int rc = smx_set_error(SMX_ERR_ARG, "read %s: %s", path, r->gz_msg.c_str());
        gzclose(gz);
        delete r;
        return rc;
    }
    if (r->fastq && !compressed_name && gzdirect(gz) == 1 && !getenv("SMX_IO_SERIAL")) {
        int fd = open(path, O_RDONLY);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            r->fd = fd;
            r->fsize = (uint64_t)st.st_size;
            r->fpos = 0;
            r->fast = true;
            if (r->fsize > 0 && !getenv("SMX_IO_NO_MMAP")) {   // (a file truncated while mapped faults: inputs are complete files)
                void *m = mmap(nullptr, (size_t)r->fsize, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, (size_t)r->fsize, MADV_SEQUENTIAL);
                    r->map = std::make_shared<Mapping>();
                    r->map->p = (const char *)m;
                    r->map->n = (size_t)r->fsize;
                }
            }
        } else if (fd >= 0) close(fd);
    }
    if (r->fastq && !r->fast && gzdirect(gz) == 0 && !getenv("SMX_IO_SERIAL")) {
        // gzip stream: what peek() has already inflated becomes the first carry
        r->carry.assign(r->buf.data() + r->pos, r->buf.data() + r->end);
        r->upos = 0;
        r->gz_eof = r->eof;
        r->gzfast = true;
    }
    if (is_fastq) *is_fastq = r->fastq ? 1 : 0;
    *out = r;
    return SMX_OK;
}

// First record start at or after byte `pos` of an uncompressed 4-line FASTQ (fd), or fsize if there is none.
static int locate_record_start(int fd, uint64_t pos, uint64_t fsize, uint64_t *out) {
    if (pos == 0) { *out = 0; return SMX_OK; }
    if (pos >= fsize) { *out = fsize; return SMX_OK; }
    for (uint64_t span = 4u << 20;; span *= 4) {
        const uint64_t a = pos - 1, len = std::min<uint64_t>(span, fsize - a);
        std::vector<char> buf(len);
        uint64_t got = 0;
        while (got < len) {
            ssize_t k = pread(fd, buf.data() + got, (size_t)(len - got), (off_t)(a + got));
            if (k < 0) { if (errno == EINTR) continue; return smx_set_error(SMX_ERR_ARG, "read: %s", strerror(errno)); }
            if (k == 0) break;
            got += (uint64_t)k;
        }
        const char *base = buf.data(), *end = base + got;
        const char *nl = (const char *)memchr(base, '\n', got);   // the line that starts at or after pos begins after this
        if (nl) {
            const char *rs = find_record_start(nl + 1, end);
            // find_record_start needs the record's three following lines in view: accept unless it ran into the buffer end
            if (rs < end && (uint64_t)(end - rs) > 0) { *out = a + (uint64_t)(rs - base); return SMX_OK; }
        }
        if (a + got >= fsize) { *out = fsize; return SMX_OK; }   // nothing but a partial tail: belongs to the previous range
    }
}

int smx_reader_open_range(const char *path, uint64_t lo, uint64_t hi, smx_reader **out, int *is_fastq) {
    int rc = smx_reader_open(path, out, is_fastq);
    if (rc) return rc;
    smx_reader *r = *out;
    if (!r->fast) {
        smx_reader_close(r);
        *out = nullptr;
        return smx_set_error(SMX_ERR_UNSUPPORTED, "%s: byte ranges need an uncompressed FASTQ file", path);
    }
    uint64_t a = 0, b = r->fsize;
    if (hi < lo) hi = lo;
    rc = locate_record_start(r->fd, lo, r->fsize, &a);
    if (!rc) rc = locate_record_start(r->fd, std::min<uint64_t>(hi, r->fsize), r->fsize, &b);
    if (rc) { smx_reader_close(r); *out = nullptr; return rc; }
    r->fpos = a;
    r->fsize = b;        // the range is read like a file that ends at the first record start at or after hi
    r->ranged = true;
    return SMX_OK;
}

void smx_reader_close(smx_reader *r) {
    if (!r) return;
    if (r->gz) gzclose(r->gz);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

smx_batch *smx_batch_new(void) { return new smx_batch(); }
void smx_batch_free(smx_batch *b) { delete b; }
uint32_t smx_batch_size(const smx_batch *b) { return b ? b->n : 0; }

int smx_batch_record(const smx_batch *b, uint32_t i, const char **id, uint32_t *id_len, const char **seq,
                     const char **qual, uint32_t *seq_len) {
    if (!b || i >= b->n) return smx_set_error(SMX_ERR_ARG, "record index out of range");
    const Segment *sg;
    const Rec *r;
    b->locate(i, &sg, &r);
    if (id) *id = sg->base + r->id_off;
    if (id_len) *id_len = r->id_len;
    if (seq) *seq = sg->base + r->seq_off;
    if (qual) *qual = r->qual_off == UINT64_MAX ? nullptr : sg->base + r->qual_off;
    if (seq_len) *seq_len = r->seq_len;
    return SMX_OK;
}

int smx_reader_next(smx_reader *r, uint32_t max_reads, uint64_t max_bytes, smx_batch *b, uint32_t *n_read) {
    if (!r || !b || !n_read) return smx_set_error(SMX_ERR_ARG, "null argument");
    b->clear();
    *n_read = 0;
    if (max_reads == 0) return SMX_OK;
    if (r->gzfast) {
        int rc = next_fast_gz(r, max_reads, max_bytes, b);
        if (rc < 0) return SMX_ERR_ARG;
        if (rc == 0) {   // irregular FASTQ: the general engine continues from the uncompressed offset of this block
            b->clear();
            r->gzfast = false;
            r->carry.clear();
            if (gzseek(r->gz, (z_off_t)r->upos, SEEK_SET) < 0)
                return smx_set_error(SMX_ERR_ARG, "cannot seek in %s", r->path.c_str());
            r->pos = r->end = 0;
            r->eof = false;
        }
    }
    if (r->fast) {
        int rc = next_fast(r, max_reads, max_bytes, b);
        if (rc < 0) return SMX_ERR_ARG;
        if (rc == 0 && r->ranged) {
            b->clear();
            return smx_set_error(SMX_ERR_UNSUPPORTED, "%s: irregular FASTQ (wrapped lines?) cannot be read by byte range", r->path.c_str());
        }
        if (rc == 0) {   // irregular FASTQ: continue with the general engine from this block's start, for good
            b->clear();
            r->fast = false;
            if (gzseek(r->gz, (z_off_t)r->fpos, SEEK_SET) < 0)
                return smx_set_error(SMX_ERR_ARG, "cannot seek in %s", r->path.c_str());
            r->pos = r->end = 0;
            r->eof = false;
        }
    }
    if (!r->fast && !r->gzfast) {
        int rc = next_general(r, max_reads, max_bytes, b);
        // a stream error wins over whatever the parser made of the truncated tail
        if (r->gz_failed) { b->clear(); return smx_set_error(SMX_ERR_ARG, "read %s: %s", r->path.c_str(), r->gz_msg.c_str()); }
        if (rc) return rc;
    }
    b->finish();
    *n_read = b->n;
    return SMX_OK;
}

int smx_pack_windows_batch(const smx_batch *b, int32_t S, uint8_t *windows, int32_t *lens) {
    if (!b || !windows || !lens || S < 1) return smx_set_error(SMX_ERR_ARG, "null argument");
    const size_t stride = ((size_t)(2 * S) + 15) & ~(size_t)15;
    const int T = (int)std::min<size_t>((size_t)io_threads(), std::max<size_t>(b->segs.size(), 1));
    std::atomic<size_t> next(0);
    auto work = [&] {
        for (size_t k = next.fetch_add(1); k < b->segs.size(); k = next.fetch_add(1)) {
            const Segment &sg = b->segs[k];
            for (size_t j = 0; j < sg.recs.size(); j++) {
                const Rec &r = sg.recs[j];
                size_t i = (size_t)b->first[k] + j;
                int L = (int)r.seq_len, Sp = L < S ? L : S;
                uint8_t *w = windows + i * stride;
                memset(w, 0, stride);
                memcpy(w, sg.base + r.seq_off, (size_t)Sp);
                memcpy(w + S, sg.base + r.seq_off + (size_t)(L - Sp), (size_t)Sp);
                lens[i] = L;
            }
        }
    };
    if (T <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    return SMX_OK;
}

// ---- 4-bit windows (transport format of the lanes; unpacked on the device by smx_pack.hip)
namespace {
struct Code4 {
    uint8_t code[256];
    uint8_t pair[65536];   // two consecutive bases (little-endian 16-bit load) -> one packed byte: 64 KB, cache resident
    Code4() {
        static const char chars[16] = {'A', 'C', 'G', 'T', 'N', 'R', 'Y', 'K', 'M', 'S', 'W', 'B', 'D', 'H', 'V', 0};   // smx_internal.h kCodeChars
        memset(code, 15, sizeof(code));
        for (int c = 0; c < 15; c++) code[(unsigned char)chars[c]] = (uint8_t)c;
        for (int v = 0; v < 65536; v++) pair[v] = (uint8_t)(code[v & 255] | (code[v >> 8] << 4));
    }
};
const Code4 *code4_tables() { static Code4 t; return &t; }

// one read: head / tail windows as nibbles (base j at byte j / 2, low nibble first), padded with code 15.
// Returns 1 if a window holds 'U': the one letter whose complement is inside the alphabet while the letter itself is not
// (Bio.Seq: U -> A), so the ASCII windows carry more than its code says -- such a batch travels as ASCII.
inline int pack4_read(const Code4 &T, const uint8_t *seq, int L, int S, uint8_t *out, size_t pstride) {
    const int Sp = L < S ? L : S, hb = (S + 1) >> 1;
    int special = 0;
    for (int end = 0; end < 2; end++) {
        const uint8_t *src = end ? seq + (L - Sp) : seq;
        uint8_t *dst = out + (end ? hb : 0);
        special |= Sp > 0 && memchr(src, 'U', (size_t)Sp) != nullptr;
        int j = 0;
        for (; j + 1 < Sp; j += 2) {
            uint16_t two;
            memcpy(&two, src + j, 2);
            dst[j >> 1] = T.pair[two];
        }
        if (j < Sp) { dst[j >> 1] = (uint8_t)(T.code[src[j]] | 0xF0u); j += 2; }
        if (j < 2 * hb) memset(dst + (j >> 1), 0xFF, (size_t)(hb - (j >> 1)));
    }
    if ((size_t)2 * hb < pstride) memset(out + 2 * hb, 0xFF, pstride - (size_t)2 * hb);
    return special;
}
}  // namespace

size_t smx_packed_stride_for(int32_t S) { return ((size_t)(2 * ((S + 1) >> 1)) + 15) & ~(size_t)15; }

int smx_pack_windows4(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int32_t S, uint8_t *packed, int32_t *lens,
                      uint32_t *n_ascii_only) {
    if (!bases || !offsets || !packed || !lens || S < 1) return smx_set_error(SMX_ERR_ARG, "null argument");
    const size_t ps = smx_packed_stride_for(S);
    const Code4 &T4 = *code4_tables();
    uint32_t special = 0;
    for (uint32_t i = 0; i < n_reads; i++) {
        const uint64_t a = offsets[i], b = offsets[i + 1];
        if (b < a || b - a > 0x7FFFFFFFull) return smx_set_error(SMX_ERR_ARG, "read %u: bad offsets", i);
        special += (uint32_t)pack4_read(T4, bases + a, (int)(b - a), S, packed + (size_t)i * ps, ps);
        lens[i] = (int32_t)(b - a);
    }
    if (n_ascii_only) *n_ascii_only = special;
    return SMX_OK;
}

int smx_pack_windows4_batch(const smx_batch *b, int32_t S, uint8_t *packed, int32_t *lens, uint32_t *n_ascii_only) {
    if (!b || !packed || !lens || S < 1) return smx_set_error(SMX_ERR_ARG, "null argument");
    const size_t ps = smx_packed_stride_for(S);
    const Code4 &T4 = *code4_tables();
    const int T = (int)std::min<size_t>((size_t)io_threads(), std::max<size_t>(b->segs.size(), 1));
    std::atomic<size_t> next(0);
    std::atomic<uint32_t> special(0);
    auto work = [&] {
        uint32_t sp = 0;
        for (size_t k = next.fetch_add(1); k < b->segs.size(); k = next.fetch_add(1)) {
            const Segment &sg = b->segs[k];
            for (size_t j = 0; j < sg.recs.size(); j++) {
                const Rec &r = sg.recs[j];
                const size_t i = (size_t)b->first[k] + j;
                sp += (uint32_t)pack4_read(T4, (const uint8_t *)sg.base + r.seq_off, (int)r.seq_len, S, packed + i * ps, ps);
                lens[i] = (int32_t)r.seq_len;
            }
        }
        special.fetch_add(sp);
    };
    if (T <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    if (n_ascii_only) *n_ascii_only = special.load();
    return SMX_OK;
}

int smx_writer_open(const char *output_dir, const char *prefix, int is_fastq, const smx_names *nm, smx_writer **out) {
    if (!output_dir || !nm || !out) return smx_set_error(SMX_ERR_ARG, "null argument");
    smx_writer *w = new smx_writer();
    w->out_dir = output_dir;
    w->prefix = prefix ? prefix : "";
    w->fastq = is_fastq != 0;
    w->specimens = split_names(nm->specimens, nm->specimen_off, nm->n_specimens);
    w->pools = split_names(nm->pools, nm->pool_off, nm->n_pools);
    w->primers = split_names(nm->primers, nm->primer_off, nm->n_primers);
    w->barcodes = split_names(nm->barcodes, nm->barcode_off, nm->n_barcodes);
    for (int c = 0; c < 256; c++) w->comp[c] = (unsigned char)c;
    const char *from = "ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu", *to = "TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna";
    for (int i = 0; from[i]; i++) w->comp[(unsigned char)from[i]] = (unsigned char)to[i];
    {   // several shards per thread (smx_writer_write hands them out heaviest first); SMX_IO_SHARDS overrides
        size_t ns = io_threads() > 1 ? (size_t)io_threads() * 8 : 1;
        if (const char *e = getenv("SMX_IO_SHARDS")) ns = (size_t)std::max(1, atoi(e));
        w->shards.resize(ns);
    }
    if (const char *e = getenv("SMX_IO_FLUSH_KB")) w->flush_bytes = (size_t)std::max(4, atoi(e)) << 10;
    mkdir(output_dir, 0777);
    *out = w;
    return SMX_OK;
}

int smx_writer_write(smx_writer *w, const smx_batch *b, const smx_op *ops, uint32_t n_reads, const smx_op *extra,
                     uint32_t n_extra) {
    if (!w || !b || !ops) return smx_set_error(SMX_ERR_ARG, "null argument");
    if (n_reads != b->n) return smx_set_error(SMX_ERR_ARG, "ops for %u reads, batch holds %u", n_reads, b->n);
    // extra records grouped by read, emission order kept (stable counting sort on the read index)
    std::vector<uint32_t> first(n_reads + 1, 0), order(n_extra);
    for (uint32_t j = 0; j < n_extra; j++) {
        if (extra[j].read >= n_reads) return smx_set_error(SMX_ERR_ARG, "extra record %u refers to read %u", j, extra[j].read);
        first[extra[j].read + 1]++;
    }
    for (uint32_t i = 0; i < n_reads; i++) first[i + 1] += first[i];
    {
        std::vector<uint32_t> fill(first.begin(), first.end() - 1);
        for (uint32_t j = 0; j < n_extra; j++) order[fill[extra[j].read]++] = j;
    }
    // Pass 1, one slice of the batch per thread: which shard(s) write each operation (bit 31 of an entry: index into
    // `extra`).  Pass 2, one shard per thread: the shard walks its entries slice by slice, i.e. in read order, and
    // formats / appends them -- per-file record order is the input order, no locks, and no thread looks at an
    // operation that is not its own.
    const uint32_t T = (uint32_t)w->shards.size();           // shards: disjoint sets of output files
    const uint32_t NT = (uint32_t)std::max(1, io_threads());   // threads
    std::vector<std::string> errs(T);
    const bool threaded = NT > 1 && n_reads >= 512;
    const uint32_t P = threaded ? NT : 1;
    w->route.resize((size_t)P * T);
    auto route = [&](uint32_t slice) {
        std::vector<uint32_t> *mine = &w->route[(size_t)slice * T];
        for (uint32_t t = 0; t < T; t++) mine[t].clear();
        const uint32_t i0 = (uint32_t)((uint64_t)n_reads * slice / P), i1 = (uint32_t)((uint64_t)n_reads * (slice + 1) / P);
        auto one = [&](const smx_op &op, uint32_t entry) {
            if (op.rtype == SMX_R_FILTERED) return;
            uint32_t o1, o2;
            owners(op, T, &o1, &o2);
            mine[o1].push_back(entry);
            if ((op.rtype == SMX_R_FULL || op.rtype == SMX_R_DEREP_FULL) && o2 != o1) mine[o2].push_back(entry);
        };
        for (uint32_t i = i0; i < i1; i++) {
            one(ops[i], i);
            for (uint32_t k = first[i]; k < first[i + 1]; k++) one(extra[order[k]], order[k] | 0x80000000u);
        }
    };
    auto work = [&](uint32_t me) {
        smx_writer::Shard &sh = w->shards[me];
        sh.rc = 0;
        for (uint32_t slice = 0; slice < P && !sh.rc; slice++)
            for (uint32_t entry : w->route[(size_t)slice * T + me]) {
                if (entry & 0x80000000u) sh.rc = write_one(w, sh, me, T, b, extra[entry & 0x7FFFFFFFu]);
                else {
                    smx_op op = ops[entry];
                    op.read = entry;
                    sh.rc = write_one(w, sh, me, T, b, op);
                }
                if (sh.rc) break;
            }
        if (sh.rc) errs[me] = smx_last_error();
        // pieces point into the batch: nothing of it may stay pending
        for (size_t fi : sh.touched) smx_writer::flush(sh, sh.files[fi]);
        sh.touched.clear();
    };
    if (!threaded) { route(0); for (uint32_t t = 0; t < T; t++) work(t); }
    else {
        static const bool dbg = getenv("SMX_IO_DEBUG") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        std::vector<double> took(NT, 0.0);
        const double t0 = now();
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < NT; t++) th.emplace_back(route, t);
        for (auto &x : th) x.join();
        th.clear();
        // One file can be a fifth of the output (unknown/unknown/unknown-unknown/unknown.fastq takes every unmatched read) and
        // a file is one thread's work: there are several shards per thread, handed out heaviest first, so that the thread
        // that draws the big file does little else.
        std::vector<std::pair<uint64_t, uint32_t>> load(T);
        for (uint32_t t = 0; t < T; t++) {
            uint64_t c = 0;
            for (uint32_t slice = 0; slice < P; slice++) c += w->route[(size_t)slice * T + t].size();
            load[t] = {c, t};
        }
        std::sort(load.begin(), load.end(), [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) {
            return x.first != y.first ? x.first > y.first : x.second < y.second; });
        std::atomic<uint32_t> next(0);
        const double t1 = now();
        for (uint32_t t = 0; t < NT; t++)
            th.emplace_back([&, t] {
                const double a = now();
                for (uint32_t k = next.fetch_add(1); k < T; k = next.fetch_add(1)) work(load[k].second);
                took[t] = now() - a;
            });
        for (auto &x : th) x.join();
        if (dbg) {
            const double t2 = now();
            std::string line;
            char tmp[64];
            double sum = 0, mx = 0;
            for (uint32_t t = 0; t < NT; t++) { sum += took[t]; mx = std::max(mx, took[t]); snprintf(tmp, sizeof(tmp), " %.1f", took[t] * 1e3); line += tmp; }
            fprintf(stderr, "[smx writer] route %.1f ms, shards %.1f ms (max %.1f, mean %.1f; heaviest shard %llu of %u entries):%s\n",
                    (t1 - t0) * 1e3, (t2 - t1) * 1e3, mx * 1e3, sum / NT * 1e3, (unsigned long long)load[0].first, n_reads + n_extra, line.c_str());
        }
    }
    for (uint32_t t = 0; t < T; t++)
        if (w->shards[t].rc) return smx_set_error(w->shards[t].rc, "%s", errs[t].c_str());
    return SMX_OK;
}

int smx_writer_close(smx_writer *w) {
    if (!w) return SMX_OK;
    int err = 0;
    {   // every shard flushes its own files: the tail of the run is as parallel as the rest
        std::vector<std::thread> th;
        std::atomic<size_t> next(0);
        auto drain = [&] {
            for (size_t t = next.fetch_add(1); t < w->shards.size(); t = next.fetch_add(1))
                for (auto &f : w->shards[t].files) smx_writer::flush(w->shards[t], f);
        };
        for (int t = 1; t < io_threads(); t++) th.emplace_back(drain);
        drain();
        for (auto &x : th) x.join();
    }
    for (auto &sh : w->shards) if (!err) err = sh.first_errno;
    delete w;
    if (err) return smx_set_error(SMX_ERR_ARG, "output write failed: %s", strerror(err));
    return SMX_OK;
}

int smx_min_pairwise_distance(const char *seqs, const uint32_t *off, uint32_t n, int32_t *out_min) {
    if (!seqs || !off || !out_min) return smx_set_error(SMX_ERR_ARG, "null argument");
    *out_min = -1;
    if (n < 2) return SMX_OK;
    // Needleman-Wunsch over two rows per pair; pairs of row i spread over threads for large panels
    const int T = n < 64 ? 1 : io_threads();
    std::vector<int32_t> best((size_t)T, INT32_MAX);
    std::atomic<uint32_t> next(0);
    auto work = [&](int me) {
        std::vector<int32_t> prev, cur;
        int32_t mine = INT32_MAX;
        for (uint32_t i = next.fetch_add(1); i + 1 < n; i = next.fetch_add(1)) {
            const char *a = seqs + off[i];
            const int32_t la = (int32_t)(off[i + 1] - off[i]);
            for (uint32_t j = i + 1; j < n; j++) {
                const char *b = seqs + off[j];
                const int32_t lb = (int32_t)(off[j + 1] - off[j]);
                prev.resize((size_t)lb + 1);
                cur.resize((size_t)lb + 1);
                for (int32_t y = 0; y <= lb; y++) prev[(size_t)y] = y;
                for (int32_t x = 1; x <= la; x++) {
                    cur[0] = x;
                    for (int32_t y = 1; y <= lb; y++) {
                        const int32_t sub = prev[(size_t)y - 1] + (a[x - 1] != b[y - 1]);
                        const int32_t gap = std::min(prev[(size_t)y], cur[(size_t)y - 1]) + 1;
                        cur[(size_t)y] = std::min(sub, gap);
                    }
                    prev.swap(cur);
                }
                mine = std::min(mine, prev[(size_t)lb]);
            }
        }
        best[(size_t)me] = mine;
    };
    if (T <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    int32_t m = INT32_MAX;
    for (int32_t v : best) m = std::min(m, v);
    *out_min = m;
    return SMX_OK;
}

}  // extern "C"
