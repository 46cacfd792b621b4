// smx_io.cpp -- host streaming helpers of libsmx.so: FASTQ/FASTA reader, window packer, output writer.
// The steps on either side of the GPU hot path; plain C++17 + zlib, no device work.
//
// Reference behaviour restated (paths relative to the reference repo):
//   reader : Bio.SeqIO "fastq"/"fasta" as used by open_sequence_file (src/specimux/io_utils.py:429-450) --
//            FastqGeneralIterator rules: title after '@', sequence lines until a line starting with '+', at
//            least one quality line, further quality lines until one starts with '@' AND the quality is
//            already as long as the sequence; id = first whitespace-delimited word of the title.
//   writer : create_write_operation's orientation + slicing (demultiplex.py:74-78) and
//            OutputManager.write_sequence / _make_filename (io_utils.py:197-268).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "smx.h"

extern "C" int smx_set_error(int code, const char *fmt, ...);   // smx_api.cpp (thread-local message)

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

struct Rec {
    uint64_t id_off, seq_off, qual_off;   // into smx_batch::data; qual_off == UINT64_MAX for FASTA
    uint32_t id_len, seq_len;
};

}  // namespace

struct smx_batch {
    std::vector<char> data;   // ids, sequences, qualities, each contiguous
    std::vector<Rec> recs;
};

struct smx_reader {
    gzFile gz = nullptr;      // zlib reads plain files transparently as well
    bool fastq = true;
    std::vector<char> buf;    // unconsumed bytes [pos, end)
    size_t pos = 0, end = 0;
    bool eof = false;
    uint64_t line_no = 0;

    // Returns the next line WITHOUT its terminator ('\n' or '\r\n'); false at end of file.  The pointer
    // stays valid until the next call.
    bool next_line(const char **p, size_t *n) {
        for (;;) {
            const char *s = buf.data() + pos;
            const char *nl = (const char *)memchr(s, '\n', end - pos);
            if (nl) {
                *p = s;
                *n = (size_t)(nl - s);
                pos += *n + 1;
                line_no++;
                return true;
            }
            if (eof) {
                if (pos == end) return false;
                *p = s;
                *n = end - pos;
                pos = end;
                line_no++;
                return true;
            }
            // refill: move the partial line to the front, grow if it fills the buffer
            if (pos > 0) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
            if (end == buf.size()) buf.resize(buf.size() * 2);
            int got = gzread(gz, buf.data() + end, (unsigned)std::min<size_t>(buf.size() - end, 1u << 30));
            if (got <= 0) eof = true; else end += (size_t)got;
        }
    }
    // peek at the first byte of the next line (0 at EOF) without consuming it
    int peek() {
        for (;;) {
            if (pos < end) return (unsigned char)buf[pos];
            if (eof) return -1;
            pos = end = 0;
            int got = gzread(gz, buf.data(), (unsigned)buf.size());
            if (got <= 0) eof = true; else end = (size_t)got;
        }
    }
};

struct smx_writer {
    std::string out_dir, prefix;
    bool fastq = true;
    std::vector<std::string> specimens, pools, primers, barcodes;
    struct File { std::string path; std::string pending; bool dir_made = false; };
    std::vector<File> files;
    std::unordered_map<std::string, size_t> index;   // relative path -> files[]
    int first_errno = 0;
    std::string scratch;
    unsigned char comp[256];

    size_t file_for(const std::string &rel) {
        auto it = index.find(rel);
        if (it != index.end()) return it->second;
        files.push_back(File{out_dir + "/" + rel, std::string(), false});
        index.emplace(rel, files.size() - 1);
        return files.size() - 1;
    }
    static void mkdirs(const std::string &path) {   // parents of `path`
        for (size_t i = 1; i < path.size(); i++)
            if (path[i] == '/') { std::string d = path.substr(0, i); mkdir(d.c_str(), 0777); }
    }
    void flush(File &f) {
        if (f.pending.empty()) return;
        if (!f.dir_made) { mkdirs(f.path); f.dir_made = true; }
        int fd = open(f.path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0666);
        if (fd < 0) { if (!first_errno) first_errno = errno; f.pending.clear(); return; }
        const char *p = f.pending.data();
        size_t left = f.pending.size();
        while (left) {
            ssize_t w = write(fd, p, left);
            if (w < 0) { if (errno == EINTR) continue; if (!first_errno) first_errno = errno; break; }
            p += w; left -= (size_t)w;
        }
        close(fd);
        f.pending.clear();
    }
};

namespace {

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

extern "C" {

int smx_reader_open(const char *path, smx_reader **out, int *is_fastq) {
    if (!path || !out) return smx_set_error(SMX_ERR_ARG, "null argument");
    // format by extension (compression suffixes stripped), then by first byte (io_utils.py:380-426)
    std::string base(path);
    size_t slash = base.find_last_of('/');
    if (slash != std::string::npos) base = base.substr(slash + 1);
    auto lower = [](std::string s) { for (char &c : s) c = (char)tolower((unsigned char)c); return s; };
    auto ends = [](const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; };
    std::string low = lower(base);
    for (bool again = true; again;) {
        again = false;
        for (const char *ext : {".gz", ".gzip", ".bz2", ".zip"})
            if (ends(low, ext)) { low.resize(low.size() - strlen(ext)); again = true; }
    }
    int fmt = 0;
    if (ends(low, ".fastq") || ends(low, ".fq")) fmt = 1;
    else if (ends(low, ".fasta") || ends(low, ".fa") || ends(low, ".fna")) fmt = 2;
    gzFile gz = gzopen(path, "rb");
    if (!gz) return smx_set_error(SMX_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
    gzbuffer(gz, 1u << 20);
    smx_reader *r = new smx_reader();
    r->gz = gz;
    r->buf.resize(8u << 20);
    if (fmt == 0) {
        int c = r->peek();
        fmt = (c == '@') ? 1 : 2;   // '>' or anything else: FASTA (the reference's default)
    }
    r->fastq = fmt == 1;
    if (is_fastq) *is_fastq = r->fastq ? 1 : 0;
    *out = r;
    return SMX_OK;
}

void smx_reader_close(smx_reader *r) {
    if (!r) return;
    if (r->gz) gzclose(r->gz);
    delete r;
}

smx_batch *smx_batch_new(void) { return new smx_batch(); }
void smx_batch_free(smx_batch *b) { delete b; }
uint32_t smx_batch_size(const smx_batch *b) { return b ? (uint32_t)b->recs.size() : 0; }

int smx_batch_record(const smx_batch *b, uint32_t i, const char **id, uint32_t *id_len, const char **seq,
                     const char **qual, uint32_t *seq_len) {
    if (!b || i >= b->recs.size()) return smx_set_error(SMX_ERR_ARG, "record index out of range");
    const Rec &r = b->recs[i];
    if (id) *id = b->data.data() + r.id_off;
    if (id_len) *id_len = r.id_len;
    if (seq) *seq = b->data.data() + r.seq_off;
    if (qual) *qual = r.qual_off == UINT64_MAX ? nullptr : b->data.data() + r.qual_off;
    if (seq_len) *seq_len = r.seq_len;
    return SMX_OK;
}

static void append_trimmed(std::vector<char> &dst, const char *p, size_t n, bool drop_inner_spaces) {
    while (n && is_space((unsigned char)p[n - 1])) n--;
    size_t a = 0;
    while (a < n && is_space((unsigned char)p[a])) a++;
    if (!drop_inner_spaces) { dst.insert(dst.end(), p + a, p + n); return; }
    for (size_t i = a; i < n; i++) if (p[i] != ' ') dst.push_back(p[i]);
}

int smx_reader_next(smx_reader *r, uint32_t max_reads, uint64_t max_bytes, smx_batch *b, uint32_t *n_read) {
    if (!r || !b || !n_read) return smx_set_error(SMX_ERR_ARG, "null argument");
    b->data.clear();
    b->recs.clear();
    const char *p;
    size_t n;
    while (b->recs.size() < max_reads && (max_bytes == 0 || b->data.size() < max_bytes)) {
        if (r->fastq) {
            // skip blank lines between records
            bool have = false;
            while ((have = r->next_line(&p, &n))) {
                size_t k = n;
                while (k && is_space((unsigned char)p[k - 1])) k--;
                if (k) break;
            }
            if (!have) break;
            if (p[0] != '@') return smx_set_error(SMX_ERR_ARG, "line %llu: Records in Fastq files should start with '@' character", (unsigned long long)r->line_no);
            Rec rec;
            // id = first whitespace-delimited word of the title
            size_t a = 1;
            while (a < n && is_space((unsigned char)p[a])) a++;
            size_t e = a;
            while (e < n && !is_space((unsigned char)p[e])) e++;
            rec.id_off = b->data.size();
            rec.id_len = (uint32_t)(e - a);
            b->data.insert(b->data.end(), p + a, p + e);
            // sequence lines until '+'
            rec.seq_off = b->data.size();
            bool plus = false;
            while (r->next_line(&p, &n)) {
                if (n && p[0] == '+') { plus = true; break; }
                append_trimmed(b->data, p, n, false);
            }
            if (!plus) return smx_set_error(SMX_ERR_ARG, "line %llu: End of file without quality information.", (unsigned long long)r->line_no);
            uint64_t slen = b->data.size() - rec.seq_off;
            if (slen > 0x7FFFFFFFull) return smx_set_error(SMX_ERR_UNSUPPORTED, "read longer than 2^31-1 bases");
            rec.seq_len = (uint32_t)slen;
            // quality: one line always, then more while the next line is not a title of a complete record
            rec.qual_off = b->data.size();
            if (r->next_line(&p, &n)) append_trimmed(b->data, p, n, false);
            for (;;) {
                int c = r->peek();
                if (c < 0) break;
                uint64_t qlen = b->data.size() - rec.qual_off;
                if (c == '@' && qlen >= slen) break;
                if (!r->next_line(&p, &n)) break;
                append_trimmed(b->data, p, n, false);
            }
            if (b->data.size() - rec.qual_off != slen)
                return smx_set_error(SMX_ERR_ARG, "line %llu: Lengths of sequence and quality values differs (%llu and %llu).",
                                     (unsigned long long)r->line_no, (unsigned long long)slen,
                                     (unsigned long long)(b->data.size() - rec.qual_off));
            b->recs.push_back(rec);
        } else {
            // FASTA: skip to the next '>' line
            bool have = false;
            while ((have = r->next_line(&p, &n))) if (n && p[0] == '>') break;
            if (!have) break;
            Rec rec;
            size_t a = 1;
            while (a < n && is_space((unsigned char)p[a])) a++;
            size_t e = a;
            while (e < n && !is_space((unsigned char)p[e])) e++;
            rec.id_off = b->data.size();
            rec.id_len = (uint32_t)(e - a);
            b->data.insert(b->data.end(), p + a, p + e);
            rec.seq_off = b->data.size();
            for (;;) {
                int c = r->peek();
                if (c < 0 || c == '>') break;
                if (!r->next_line(&p, &n)) break;
                append_trimmed(b->data, p, n, true);
            }
            uint64_t slen = b->data.size() - rec.seq_off;
            if (slen > 0x7FFFFFFFull) return smx_set_error(SMX_ERR_UNSUPPORTED, "read longer than 2^31-1 bases");
            rec.seq_len = (uint32_t)slen;
            rec.qual_off = UINT64_MAX;
            b->recs.push_back(rec);
        }
    }
    *n_read = (uint32_t)b->recs.size();
    return SMX_OK;
}

int smx_pack_windows_batch(const smx_batch *b, int32_t S, uint8_t *windows, int32_t *lens) {
    if (!b || !windows || !lens || S < 1) return smx_set_error(SMX_ERR_ARG, "null argument");
    const size_t stride = ((size_t)(2 * S) + 15) & ~(size_t)15;
    const char *base = b->data.data();
    for (size_t i = 0; i < b->recs.size(); i++) {
        const Rec &r = b->recs[i];
        int L = (int)r.seq_len, Sp = L < S ? L : S;
        uint8_t *w = windows + i * stride;
        memset(w, 0, stride);
        memcpy(w, base + r.seq_off, (size_t)Sp);
        memcpy(w + S, base + r.seq_off + (size_t)(L - Sp), (size_t)Sp);
        lens[i] = L;
    }
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
    mkdir(output_dir, 0777);
    *out = w;
    return SMX_OK;
}

static int write_one(smx_writer *w, const smx_batch *b, const smx_op &op) {
    if (op.rtype == SMX_R_FILTERED) return SMX_OK;
    if (op.read >= b->recs.size()) return smx_set_error(SMX_ERR_ARG, "write operation refers to read %u of %zu", op.read, b->recs.size());
    const Rec &r = b->recs[op.read];
    const char *base = b->data.data();
    auto name = [](const std::vector<std::string> &v, int i) -> const std::string & {
        static const std::string unknown = "unknown";
        return (i >= 0 && (size_t)i < v.size()) ? v[(size_t)i] : unknown;
    };
    std::string sample;
    if (op.sample >= 0) sample = name(w->specimens, op.sample);
    else if (op.rtype == SMX_R_PARTIAL_FWD) sample = "barcode_fwd_" + name(w->barcodes, op.barcode);
    else if (op.rtype == SMX_R_PARTIAL_REV) sample = "barcode_rev_" + name(w->barcodes, op.barcode);
    else sample = "unknown";
    const std::string &pool = name(w->pools, op.pool), &p1 = name(w->primers, op.p1), &p2 = name(w->primers, op.p2);
    int64_t L = r.seq_len, s = op.trim_start, e = op.trim_end;
    if (s < 0) s = 0;   // the kernel only emits 0 <= s < e <= L for non-empty reads
    if (e > L) e = L;
    if (e < s) e = s;
    std::string &rec = w->scratch;
    rec.clear();
    rec.push_back(w->fastq ? '@' : '>');
    rec.append(base + r.id_off, r.id_len);
    rec.push_back(' ');
    for (int k = 0; k < 4; k++) {
        if (k) rec.push_back(',');
        if (op.dist[k] < 0) rec.push_back('X'); else rec += std::to_string((int)op.dist[k]);
    }
    rec += " pool=" + pool + " primers=" + p1 + "+" + p2 + " " + sample + "\n";
    const char *seq = base + r.seq_off;
    const bool rev = (op.flags & SMX_OPF_REVERSE) != 0;
    size_t at = rec.size();
    rec.resize(at + (size_t)(e - s));
    if (!rev) memcpy(&rec[at], seq + s, (size_t)(e - s));
    else for (int64_t i = s; i < e; i++) rec[at + (size_t)(i - s)] = (char)w->comp[(unsigned char)seq[L - 1 - i]];
    rec.push_back('\n');
    if (w->fastq) {
        rec += "+\n";
        at = rec.size();
        rec.resize(at + (size_t)(e - s));
        if (r.qual_off == UINT64_MAX) memset(&rec[at], 'I', (size_t)(e - s));
        else {
            const char *q = base + r.qual_off;
            if (!rev) memcpy(&rec[at], q + s, (size_t)(e - s));
            else for (int64_t i = s; i < e; i++) rec[at + (size_t)(i - s)] = q[L - 1 - i];
        }
        rec.push_back('\n');
    }
    const char *top = op.rtype == SMX_R_UNKNOWN ? "unknown" : ((op.rtype == SMX_R_PARTIAL_FWD || op.rtype == SMX_R_PARTIAL_REV) ? "partial" : "full");
    const std::string ext = w->fastq ? ".fastq" : ".fasta";
    const std::string fname = w->prefix + safe_name(sample) + ext;
    size_t f = w->file_for(std::string(top) + "/" + pool + "/" + p1 + "-" + p2 + "/" + fname);
    w->files[f].pending += rec;
    if (w->files[f].pending.size() > (256u << 10)) w->flush(w->files[f]);
    if (op.rtype == SMX_R_FULL || op.rtype == SMX_R_DEREP_FULL) {   // pool-level aggregate (io_utils.py:256-268)
        size_t g = w->file_for("full/" + pool + "/" + fname);
        w->files[g].pending += rec;
        if (w->files[g].pending.size() > (256u << 10)) w->flush(w->files[g]);
    }
    return SMX_OK;
}

int smx_writer_write(smx_writer *w, const smx_batch *b, const smx_op *ops, uint32_t n_reads, const smx_op *extra,
                     uint32_t n_extra) {
    if (!w || !b || !ops) return smx_set_error(SMX_ERR_ARG, "null argument");
    if (n_reads != b->recs.size()) return smx_set_error(SMX_ERR_ARG, "ops for %u reads, batch holds %zu", n_reads, b->recs.size());
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
    for (uint32_t i = 0; i < n_reads; i++) {
        smx_op op = ops[i];
        op.read = i;
        int rc = write_one(w, b, op);
        if (rc) return rc;
        for (uint32_t k = first[i]; k < first[i + 1]; k++) {
            rc = write_one(w, b, extra[order[k]]);
            if (rc) return rc;
        }
    }
    return SMX_OK;
}

int smx_writer_close(smx_writer *w) {
    if (!w) return SMX_OK;
    for (auto &f : w->files) w->flush(f);
    int err = w->first_errno;
    delete w;
    if (err) return smx_set_error(SMX_ERR_ARG, "output write failed: %s", strerror(err));
    return SMX_OK;
}

}  // extern "C"
