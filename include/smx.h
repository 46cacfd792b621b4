/*
 * smx.h -- C ABI of libsmx.so, the MI355X (gfx950) dual-barcode demultiplexing hot path.
 *
 * The reference (joshuaowalker/specimux, pure Python) has no FFI of its own; the path this
 * library replaces sits behind these reference interfaces (paths relative to the reference repo):
 *
 *   smx_batch_run / smx_batch_run_device
 *        <- process_sequences(seq_records, parameters, specimens, args, prefilter, ...)
 *           src/specimux/demultiplex.py:108-212 (one call per read batch; callers
 *           multiprocessing_utils.py:89 and orchestration.py:513), i.e. everything below it:
 *           determine_orientation :602, find_candidate_matches :668, match_one_end :748,
 *           align_seq alignment.py:21-50 (edlib HW/SHW, IUPAC equalities constants.py:13-20),
 *           BloomPrefilter.match bloom_filter.py:176, select_best_matches :216,
 *           dereplicate_matches :262 (+ :396, :480), resolve_specimen :541,
 *           create_write_operation :30 (trim extents models.py:278-319).
 *   smx_panel_create  <- the Specimens / PrimerDatabase / MatchParameters objects that
 *           process_sequences receives (databases.py:123-275, models.py:331-338,
 *           thresholds orchestration.py:548-628), flattened into arrays by the host.
 *   smx_align         <- align_seq's edlib.align call (alignment.py:42), one alignment, for unit parity.
 *   smx_pack_windows  <- the two `search_len` end slices match_one_end / determine_orientation take
 *           (demultiplex.py:757-766, :612-624): the only bases the hot path ever reads.
 *   smx_counts_*      <- the parent summing (batch_total, batch_matched) (orchestration.py:203-207).
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; no callbacks; every
 * function returns 0 on success or a negative smx_status and leaves a message for smx_last_error()
 * (thread local).  There is NO CPU fallback in this library: without a HIP device every compute
 * entry point fails with SMX_ERR_DEVICE.
 */
#ifndef SMX_H
#define SMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_ABI_VERSION 4

typedef enum {
    SMX_OK = 0,
    SMX_ERR_ARG = -1,      /* bad argument / inconsistent descriptor */
    SMX_ERR_UNSUPPORTED = -2, /* panel outside the kernel's limits (pattern > 64 nt, ...) */
    SMX_ERR_DEVICE = -3,   /* HIP error or no device */
    SMX_ERR_OVERFLOW = -4  /* the extra-record buffer was too small (retry with the reported size), or one read produced more than 65535 write operations (n_ops is 16 bits) */
} smx_status;

/* trim modes (constants.py:40-45) and dereplication strategies (:48-51) */
enum { SMX_TRIM_NONE = 0, SMX_TRIM_TAILS = 1, SMX_TRIM_BARCODES = 2, SMX_TRIM_PRIMERS = 3 };
enum { SMX_DEREP_NONE = 0, SMX_DEREP_BEST = 1 };
/* ResolutionType (constants.py:54-60); 0 = read dropped by the length filter (demultiplex.py:135-140) */
enum { SMX_R_FILTERED = 0, SMX_R_FULL = 1, SMX_R_PARTIAL_FWD = 2, SMX_R_PARTIAL_REV = 3,
       SMX_R_MULTIPLE = 4, SMX_R_UNKNOWN = 5, SMX_R_DEREP_FULL = 6 };

/* smx_op.flags */
#define SMX_OPF_REVERSE    0x01u /* output sequence is the reverse complement of the read (demultiplex.py:724) */
#define SMX_OPF_TRIM_EMPTY 0x02u /* trim would be empty: untrimmed record to unknown/unknown/unknown-unknown (:45-73) */
#define SMX_OPF_NO_SPECIMEN 0x04u /* full match without a specimen (the reference logs a warning, :571-575) */

/*
 * One WriteOperation (models.py:341-357) in index form; 32 bytes.  Every unfiltered read has exactly
 * one primary record ops[read]; reads with n_ops > 1 have their 2nd.. records appended to the
 * extra buffer (any order between reads, emission order within a read).
 */
typedef struct smx_op {
    int32_t sample;      /* specimen index (file order) for full matches, else -1 */
    int32_t trim_start;  /* output = oriented_sequence[trim_start:trim_end] */
    int32_t trim_end;
    int16_t pool;        /* pool index, -1 = "unknown" */
    int16_t p1;          /* forward primer index (registration order), -1 = "unknown" */
    int16_t p2;          /* reverse primer index, -1 = "unknown" */
    int16_t barcode;     /* partial matches: global barcode index of barcode_fwd_/barcode_rev_, else -1 */
    int8_t dist[4];      /* distance code p1,b1,b2,p2; -1 prints as 'X' (models.py:206-218) */
    uint8_t rtype;       /* SMX_R_* */
    uint8_t flags;       /* SMX_OPF_* */
    uint16_t n_ops;      /* number of write operations of this read (primary record only) */
    uint32_t read;       /* read index inside the batch */
} smx_op;

/*
 * Debug/parity dump of the per (read, primer, end) search results ("hit table"); end 0 = A = 3' end of the
 * reverse complement, end 1 = B = 3' end of the read.  Coordinates are the reference's
 * align_seq coordinates in that end's string (before AlignmentResult.reversed()).
 */
typedef struct smx_hit {
    int32_t first_start;  /* first optimal primer location (start, end), -1 when no match */
    int32_t first_end;
    int32_t tail_end;     /* max optimal end over all within-k barcodes (for --trim tails), -1 */
    int16_t pdist;        /* primer edit distance, -1 no match */
    int16_t nloc;         /* number of optimal primer end locations */
    int16_t bbest;        /* best barcode distance at this end, -1 none, -2 not searched (orientation pruned) */
    int16_t ntied;        /* barcodes tied at bbest */
    int16_t first_tied;   /* global barcode index of the first tied barcode (canonical order), -1 */
    int16_t flags;        /* bit 0: this alignment votes in determine_orientation (demultiplex.py:602-638; differs from
                             pdist >= 0 only for reads shorter than search_len - 1) */
} smx_hit;

/* Flattened panel; strings are concatenated ASCII with n+1 offsets. */
typedef struct smx_panel_desc {
    uint32_t abi_version;        /* SMX_ABI_VERSION */
    uint32_t n_primers;          /* Specimens._primers registration order (Q5, databases.py:151-165) */
    uint32_t n_barcodes;         /* distinct barcode strings, global list */
    uint32_t n_specimens;        /* file order */
    uint32_t n_pools;
    uint32_t n_pairs;            /* (fwd, rev) in find_candidate_matches order (demultiplex.py:699-700) */

    const char *primer_rc;       /* reverse complements (the searched patterns, models.py:27) */
    const uint32_t *primer_rc_off;   /* n_primers + 1 */
    const uint8_t *primer_dir;   /* 0 forward, 1 reverse */
    const int32_t *primer_k;     /* max_dist_primers (orchestration.py:605-614) */
    const int32_t *primer_file_index; /* order in primers.fasta (models.py:31) */
    const uint32_t *primer_bc_off;   /* n_primers + 1: CSR into primer_bc */
    const uint32_t *primer_bc;   /* global barcode indices, canonical order (Q4) */

    const char *barcode_rc;      /* reverse complements of the barcodes (demultiplex.py:782) */
    const uint32_t *barcode_rc_off;  /* n_barcodes + 1 */

    const uint32_t *pair_fwd;    /* n_pairs primer indices */
    const uint32_t *pair_rev;
    const int32_t *pair_pool;    /* get_pool_from_primers (demultiplex.py:640-665), -1 none */

    const uint32_t *spec_b1;     /* n_specimens global barcode indices */
    const uint32_t *spec_b2;
    const uint64_t *spec_p1mask; /* bit p set: primer p in the specimen's p1 list (wildcards expand) */
    const uint64_t *spec_p2mask;
    const int32_t *spec_pool;

    int32_t k_index;             /* max_dist_index */
    int32_t search_len;          /* -l */
    int32_t barcode_len_max;     /* Specimens.b_length() */
    int32_t prefilter_min_len;   /* BloomPrefilter.min_length = L - k; 0 = prefilter disabled */
    int32_t preorient;           /* 0/1 (--disable-preorient) */
    int32_t trim;                /* SMX_TRIM_* */
    int32_t dereplicate;         /* SMX_DEREP_* */
    int32_t min_length;          /* -1 off */
    int32_t max_length;          /* -1 off */
    int32_t want_starts;         /* 1: report first_start even when the trim mode does not need it (trace, --color) */
} smx_panel_desc;

typedef struct smx_panel smx_panel; /* opaque, immutable after create; holds host + device copies */

/* counts vector layout (uint64): fixed slots then one per specimen */
enum { SMX_CNT_TOTAL = 0, SMX_CNT_MATCHED = 1, SMX_CNT_FILTERED = 2, SMX_CNT_OPS_FULL = 3,
       SMX_CNT_OPS_PARTIAL = 4, SMX_CNT_OPS_UNKNOWN = 5, SMX_CNT_MULTI_OP_READS = 6,
       SMX_CNT_OVERFLOW = 7, SMX_CNT_SPECIMEN0 = 8 };

int smx_abi_version(void);
const char *smx_last_error(void);

/* device selection for the calling thread; returns the device count in *n_devices if non-NULL */
int smx_device_init(int device, int *n_devices);

int smx_panel_create(const smx_panel_desc *desc, smx_panel **out);
void smx_panel_destroy(smx_panel *panel);
size_t smx_counts_len(const smx_panel *panel);      /* SMX_CNT_SPECIMEN0 + n_specimens */
size_t smx_window_stride(const smx_panel *panel);   /* bytes per read in the window buffer: round16(2*search_len) */
size_t smx_hits_per_read(const smx_panel *panel);   /* 2 * n_primers */
size_t smx_bdist_per_read(const smx_panel *panel);  /* 2 * n_primers * max barcodes per primer */

/*
 * Cut the two end windows of each read (host, no device work): for read i with bases
 * bases[offsets[i] .. offsets[i+1]) and S' = min(search_len, len):
 *   windows[i*stride .. +S')            = first S' bases   (zero padded to search_len)
 *   windows[i*stride+search_len .. +S') = last  S' bases   (zero padded)
 * lens[i] = len.
 */
int smx_pack_windows(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int32_t search_len,
                     uint8_t *windows, int32_t *lens);

/*
 * The same two windows as 4-bit text codes (the transport format of the lanes: half the bytes across PCIe).  Code of a
 * base = its index in "ACGTNRYKMSWBDHV", 15 for anything else (such characters match no pattern character in either
 * format); per read: ceil(search_len / 2) bytes of head window, the same of tail window, base j in byte j / 2 (low nibble
 * first), unused nibbles 15; stride = smx_packed_stride(panel) = round16(2 * ceil(search_len / 2)).
 * *n_ascii_only (optional) counts the reads with a 'U' inside a window: the one letter the 4-bit alphabet cannot carry
 * faithfully (Bio.Seq complements U to A); a batch with any must be sent as ASCII windows.
 * smx_unpack_windows_device turns packed windows resident in device memory into the ASCII layout smx_batch_run_device takes.
 */
size_t smx_packed_stride(const smx_panel *panel);
int smx_pack_windows4(const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads, int32_t search_len,
                      uint8_t *packed, int32_t *lens, uint32_t *n_ascii_only);
int smx_unpack_windows_device(const smx_panel *panel, void *stream, const uint8_t *d_packed, uint32_t n_reads,
                              uint8_t *d_windows);

/*
 * Run the hot path on windows already resident in device memory.  All pointers are DEVICE pointers;
 * `stream` is a hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing.
 *   d_ops          n_reads records
 *   d_extra        extra_cap records; d_n_extra one uint32, WRITTEN by the call (number of extra records produced,
 *                  may exceed extra_cap: then only the first extra_cap were stored); no need to clear it
 *   d_counts       smx_counts_len() uint64, accumulated into (not cleared)
 *   d_hits/d_bdist optional parity dumps (NULL to skip): n_reads*smx_hits_per_read() smx_hit and
 *                  n_reads*smx_bdist_per_read() int8 best distance per (read, primer, end, barcode slot), -1 none
 *                  d_bdist selects the per-barcode ("slots") kernel; d_hits alone is filled by whichever kernel the
 *                  panel's flags select (tail_end is then defined only under --trim tails)
 * Launches on one stream are ordered; a panel keeps one tile queue per stream it is launched on (at most 16
 * streams), so double-buffered callers may overlap launches of one panel on different streams.
 */
int smx_batch_run_device(const smx_panel *panel, void *stream, const uint8_t *d_windows, const int32_t *d_lens,
                         uint32_t n_reads, smx_op *d_ops, smx_op *d_extra, uint32_t extra_cap,
                         uint32_t *d_n_extra, uint64_t *d_counts, smx_hit *d_hits, int8_t *d_bdist);

/*
 * Tell the panel how many batches the caller keeps in flight on as many streams (default 1).  The demux kernel is a
 * persistent launch sized to the CUs' resident workgroup slots; with n batches in flight each launch takes 1/n of them, so
 * that kernels of different batches run side by side (the next batch's prescan beside this batch's demux kernel) instead
 * of queueing behind a launch that fills the machine.  Takes effect from the next launch; call it between batches.
 */
int smx_panel_set_streams(smx_panel *panel, int n_streams);

/*
 * Diagnostic: per-kernel device times of smx_batch_run_device on this panel.  enable != 0 makes every following launch
 * record HIP events around its kernels (transpose, primer DP, demux) on the launch stream; with ms != NULL the call waits
 * for the most recent instrumented launch and writes its three durations in milliseconds (0 for a kernel that did not
 * run; panels with many primers launch the demux kernel twice per batch -- compact tiles, then the reads of the tiles
 * that did not fit: the third duration covers both).  Off by default: the events cost a few microseconds per launch.
 */
int smx_debug_kernel_times(smx_panel *panel, int enable, float ms[3]);

/*
 * Convenience wrapper over host buffers: allocates device scratch, copies, runs, copies back, synchronises.
 * counts is accumulated into (host, smx_counts_len() uint64).  Returns SMX_ERR_OVERFLOW if extra_cap was too
 * small (n_extra then holds the required capacity) or a read exceeded the per-read operation limit.
 */
int smx_batch_run(const smx_panel *panel, const uint8_t *windows, const int32_t *lens, uint32_t n_reads,
                  smx_op *ops, smx_op *extra, uint32_t extra_cap, uint32_t *n_extra, uint64_t *counts,
                  smx_hit *hits, int8_t *bdist);

/*
 * Lanes: the asynchronous host-buffer path (pinned, double-buffered hand-off).  A lane owns a HIP
 * stream, page-locked host staging for one batch (windows + lengths in, records + counts out) and its device buffers.
 * Pipelines keep two or three lanes in flight: while lane A's kernels run, lane B's windows cross PCIe and lane C's
 * records are written out -- the replacement of the reference parent's pickled 1000-read batches
 * (io_utils.py:429-450, orchestration.py:447-456).
 *   smx_lane_create    max_reads = capacity of one batch
 *   smx_lane_windows / smx_lane_lens   pinned staging the packer fills (smx_pack_windows_batch writes there directly)
 *   smx_lane_submit    enqueue H2D copy, prescan + demux kernels, D2H copy on the lane's stream; returns at once
 *   smx_lane_submit_packed   the same for a staging filled with 4-bit windows (84 instead of 164 bytes per read at -l 80)
 *   smx_lane_wait      block until the lane's batch is done; *ops / *extra point into the lane's pinned result buffers
 *                      (valid until the next submit on this lane); counts (host, smx_counts_len() uint64) is accumulated
 *                      into.  SMX_ERR_OVERFLOW as for smx_batch_run (the extra buffer of a lane holds max_reads records).
 */
typedef struct smx_lane smx_lane;
int smx_lane_create(const smx_panel *panel, uint32_t max_reads, smx_lane **out);
void smx_lane_destroy(smx_lane *lane);
uint8_t *smx_lane_windows(smx_lane *lane);
int32_t *smx_lane_lens(smx_lane *lane);
int smx_lane_submit(smx_lane *lane, uint32_t n_reads);
/* the staging behind smx_lane_windows holds 4-bit windows (smx_pack_windows4_batch): H2D of half the bytes + unpack kernel */
int smx_lane_submit_packed(smx_lane *lane, uint32_t n_reads);
int smx_lane_wait(smx_lane *lane, const smx_op **ops, const smx_op **extra, uint32_t *n_extra, uint64_t *counts);

/*
 * One edlib-equivalent alignment on the device (Myers bit-vector kernel), for unit parity with the oracle.
 * mode 0 = HW (infix), 1 = SHW (prefix); IUPAC equalities always on; qlen <= 64.
 * *dist = -1 if the best distance exceeds k.  Up to cap (start,end) pairs are written, *nloc = true count.
 */
int smx_align(const char *query, int qlen, const char *target, int tlen, int k, int mode,
              int *dist, int *starts, int *ends, int cap, int *nloc);

/*
 * N alignments in one launch (trace level 3 and --color ask for thousands per batch; device workspace is cached).
 *   queries / qoff      n_queries distinct query strings (concatenated, n_queries + 1 offsets), each 1..64 letters
 *   targets / toff      n targets (concatenated, n + 1 offsets), each >= 1 letter
 *   qidx, k, mode       per alignment: query index, max distance, 0 = HW / 1 = SHW
 *   dist, nloc          per alignment: best distance (-1: above k) and number of optimal locations
 *   starts, ends        n x cap; the first min(nloc, cap) locations of alignment i at [i * cap ..)
 */
int smx_align_batch(const char *queries, const uint32_t *qoff, uint32_t n_queries, const char *targets,
                    const uint64_t *toff, const uint32_t *qidx, const int32_t *k, const uint8_t *mode, uint32_t n,
                    int32_t *dist, int32_t *nloc, int32_t *starts, int32_t *ends, uint32_t cap);

/*
 * RCCL reduction of the per-specimen counts over xGMI (one communicator per process/GPU).
 * smx_comm_unique_id fills a 128-byte id on rank 0; broadcast it by any means, then every rank calls
 * smx_comm_init.  smx_counts_allreduce sums d_counts (device pointer) in place across ranks.
 */
int smx_comm_unique_id(uint8_t id[128]);
int smx_comm_init(const uint8_t id[128], int n_ranks, int rank, void **comm_out);
int smx_counts_allreduce(uint64_t *d_counts, size_t n, void *comm, void *stream);
void smx_comm_destroy(void *comm);


/* ------------------------------------------------------------------------------------------------
 * Host streaming helpers: the steps on either side of the hot path.
 *
 *   smx_reader_*            <- open_sequence_file / SeqIO.parse + iter_batches
 *                              (src/specimux/io_utils.py:380-450, orchestration.py:447-456): FASTQ / FASTA,
 *                              plain or gzip; id = first whitespace-delimited word of the title; wrapped
 *                              sequence / quality lines accepted (Biopython FastqGeneralIterator rules)
 *   smx_pack_windows_batch  <- the end slices of match_one_end, straight from a parsed batch
 *   smx_writer_*            <- create_write_operation's slicing + OutputManager.write_sequence
 *                              (demultiplex.py:74-78, io_utils.py:197-268): orientation, trim, header
 *                              "{id} {p1d,b1d,b2d,p2d} pool={pool} primers={p1}+{p2} {sample}", path
 *                              {full|partial|unknown}/{pool}/{p1}-{p2}/{prefix}{sample}.{fastq|fasta} and the
 *                              pool-level copy of full matches.  Append-only, buffered per file.
 * Pure host code (no device work); a batch owns its memory, so reading batch i+1 may overlap the GPU run
 * of batch i and the writing of batch i-1 from different threads.
 */
typedef struct smx_reader smx_reader;
typedef struct smx_batch smx_batch;
typedef struct smx_writer smx_writer;

int smx_reader_open(const char *path, smx_reader **out, int *is_fastq);
/* the records that START inside bytes [lo, hi) of an uncompressed 4-line FASTQ (multi-GPU sharding of one file: the
 * ranges of all ranks partition the records); SMX_ERR_UNSUPPORTED for compressed, FASTA or irregular input */
int smx_reader_open_range(const char *path, uint64_t lo, uint64_t hi, smx_reader **out, int *is_fastq);
void smx_reader_close(smx_reader *reader);
smx_batch *smx_batch_new(void);
void smx_batch_free(smx_batch *batch);
/* parse up to max_reads records (and at most ~max_bytes of sequence data, 0 = no limit) into `batch`;
 * *n_read = 0 at end of file */
int smx_reader_next(smx_reader *reader, uint32_t max_reads, uint64_t max_bytes, smx_batch *batch, uint32_t *n_read);
uint32_t smx_batch_size(const smx_batch *batch);
/* record i: pointers into the batch (valid until the batch is refilled or freed); qual == NULL for FASTA */
int smx_batch_record(const smx_batch *batch, uint32_t i, const char **id, uint32_t *id_len, const char **seq,
                     const char **qual, uint32_t *seq_len);
int smx_pack_windows_batch(const smx_batch *batch, int32_t search_len, uint8_t *windows, int32_t *lens);
int smx_pack_windows4_batch(const smx_batch *batch, int32_t search_len, uint8_t *packed, int32_t *lens, uint32_t *n_ascii_only);

/* index -> name tables for the record headers and paths: concatenated strings with n+1 offsets each */
typedef struct smx_names {
    const char *specimens; const uint32_t *specimen_off; uint32_t n_specimens;
    const char *pools;     const uint32_t *pool_off;     uint32_t n_pools;
    const char *primers;   const uint32_t *primer_off;   uint32_t n_primers;
    const char *barcodes;  const uint32_t *barcode_off;  uint32_t n_barcodes;
} smx_names;

int smx_writer_open(const char *output_dir, const char *prefix, int is_fastq, const smx_names *names,
                    smx_writer **out);
/* format and append every write operation of the batch: ops[n_reads] primary records + extra[n_extra] */
int smx_writer_write(smx_writer *writer, const smx_batch *batch, const smx_op *ops, uint32_t n_reads,
                     const smx_op *extra, uint32_t n_extra);
int smx_writer_close(smx_writer *writer);   /* flushes; returns the first I/O error seen, if any */

/* ---- run setup helper (host only)
 * Minimum global edit distance (exact character equality) over all pairs of the n strings seqs[off[i]..off[i+1]):
 * what setup_match_parameters derives the default barcode threshold from (reference orchestration.py:548-575,
 * edlib.align(a, b, task="distance") over itertools.combinations).  *out_min = -1 for n < 2. */
int smx_min_pairwise_distance(const char *seqs, const uint32_t *off, uint32_t n, int32_t *out_min);

#ifdef __cplusplus
}
#endif
#endif /* SMX_H */
