// AddressSanitizer / UBSan driver for the host-side streaming I/O of libsmx (smx_io.cpp): readers (fast parallel and
// general engine, gzip), window packer, writer shards.  CPU only -- built and run by tests/test_native_io_asan.py with
// g++ -fsanitize=address,undefined against smx_io.cpp alone (the error sink of smx_api.cpp is stubbed here).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "smx.h"

extern "C" int smx_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}

extern "C" const char *smx_last_error(void) { return "(driver stub)"; }

static void die(const char *what) { fprintf(stderr, "driver: %s\n", what); exit(2); }

int main(int argc, char **argv) {
    if (argc < 3) die("usage: io_driver <sequence file> <output dir> [batch reads]");
    const char *path = argv[1];
    const std::string out = argv[2];
    const uint32_t batch_reads = argc > 3 ? (uint32_t)atoi(argv[3]) : 7;
    smx_reader *rd = nullptr;
    int is_fastq = 0;
    if (smx_reader_open(path, &rd, &is_fastq) != 0) die("open");
    // names: 3 specimens, 2 pools, 2 primers, 2 barcodes
    const char spec[] = "S1S2S3"; const uint32_t spec_off[] = {0, 2, 4, 6};
    const char pools[] = "ITSLSU"; const uint32_t pool_off[] = {0, 3, 6};
    const char prim[] = "ITS1FITS4"; const uint32_t prim_off[] = {0, 5, 9};
    const char bcs[] = "ACGTACGTACGTATTTTTTTTTTTTT"; const uint32_t bc_off[] = {0, 13, 26};
    smx_names nm = {spec, spec_off, 3, pools, pool_off, 2, prim, prim_off, 2, bcs, bc_off, 2};
    smx_writer *wr = nullptr;
    if (smx_writer_open(out.c_str(), "p_", is_fastq, &nm, &wr) != 0) die("writer open");
    smx_batch *b = smx_batch_new();
    uint64_t total = 0, bases = 0;
    const int S = 80, stride = 160;
    for (;;) {
        uint32_t n = 0;
        if (smx_reader_next(rd, batch_reads, 0, b, &n) != 0) die("next");
        if (n == 0) break;
        std::vector<uint8_t> win((size_t)n * stride);
        std::vector<int32_t> lens(n);
        if (smx_pack_windows_batch(b, S, win.data(), lens.data()) != 0) die("pack");
        {   // the 4-bit packer: buffers sized exactly (stride = round16(2 * ceil(S / 2))), several window lengths
            for (int S4 : {S, 81, 7, 1}) {
                const size_t ps = ((size_t)(2 * ((S4 + 1) / 2)) + 15) & ~(size_t)15;
                std::vector<uint8_t> packed((size_t)n * ps);
                std::vector<int32_t> l4(n);
                uint32_t n_ascii = 0;
                if (smx_pack_windows4_batch(b, S4, packed.data(), l4.data(), &n_ascii) != 0) die("pack4");
                for (uint32_t i = 0; i < n; i++) if (l4[i] != lens[i]) die("pack4 lens");
            }
        }
        std::vector<smx_op> ops(n), extra;
        for (uint32_t i = 0; i < n; i++) {
            const char *id, *seq, *qual;
            uint32_t idl, sl;
            if (smx_batch_record(b, i, &id, &idl, &seq, &qual, &sl) != 0) die("record");
            if ((int32_t)sl != lens[i]) die("length mismatch");
            bases += sl;
            smx_op &o = ops[i];
            memset(&o, 0, sizeof(o));
            const uint64_t k = total + i;
            o.read = i;
            o.n_ops = 1;
            o.dist[0] = (int8_t)(k % 5); o.dist[1] = -1; o.dist[2] = (int8_t)(k % 3); o.dist[3] = 12;
            o.trim_start = (int32_t)(k % 7);
            o.trim_end = (int32_t)sl - (int32_t)(k % 4);
            if (o.trim_end < o.trim_start) { o.trim_start = 0; o.trim_end = (int32_t)sl; }
            o.flags = (k & 1) ? SMX_OPF_REVERSE : 0;
            switch (k % 5) {
                case 0: o.rtype = SMX_R_DEREP_FULL; o.sample = (int32_t)(k % 3); o.pool = 0; o.p1 = 0; o.p2 = 1; o.barcode = -1; break;
                case 1: o.rtype = SMX_R_PARTIAL_FWD; o.sample = -1; o.pool = 1; o.p1 = 0; o.p2 = -1; o.barcode = 1; break;
                case 2: o.rtype = SMX_R_UNKNOWN; o.sample = -1; o.pool = -1; o.p1 = -1; o.p2 = -1; o.barcode = -1; break;
                case 3: o.rtype = SMX_R_FILTERED; o.sample = -1; o.pool = -1; o.p1 = -1; o.p2 = -1; o.barcode = -1; break;
                default: o.rtype = SMX_R_PARTIAL_REV; o.sample = -1; o.pool = 0; o.p1 = -1; o.p2 = 1; o.barcode = 0; break;
            }
            if (k % 11 == 0) {   // a second record of the same read
                smx_op e = o;
                e.rtype = SMX_R_FULL; e.sample = 2; e.pool = 1; e.p1 = 0; e.p2 = 1;
                extra.push_back(e);
            }
        }
        if (smx_writer_write(wr, b, ops.data(), n, extra.data(), (uint32_t)extra.size()) != 0) die("write");
        total += n;
    }
    smx_batch_free(b);
    if (smx_writer_close(wr) != 0) die("writer close");
    smx_reader_close(rd);
    printf("records %llu bases %llu fastq %d\n", (unsigned long long)total, (unsigned long long)bases, is_fastq);
    return 0;
}
