/* include/elector_split.h -- C ABI of the window splitter and the window merger,
 * the two stages either side of the POA hot path.
 *
 *   elector_split_reads   replaces the per-read work of `bin/masterSplitter`
 *                         (reference: src/split/Master_Splitter.cpp:175-332
 *                          split/best_split, :396-446 the read loop;
 *                          spawned from elector/alignment.py:99-101)
 *   elector_merge_windows replaces `bin/Donatello`
 *                         (reference: src/split/Donatello.cpp:13-93;
 *                          spawned from elector/alignment.py:120-122)
 *
 * Both are host-side byte/integer work (the reference's are single-threaded
 * CPU programs too); the splitter runs on `nthreads` host threads.
 * Plain C: pointers + sizes; results live in library-owned buffers released by
 * the matching *_free call.  Return 0 or a negative ELECTOR_E_* code
 * (include/elector_poa.h).
 */
#ifndef ELECTOR_SPLIT_H
#define ELECTOR_SPLIT_H

#include <stdint.h>
#include "elector_poa.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Windows of a batch of reads, already in the layout elector_poa_batch takes. */
typedef struct elector_windows {
  int64_t n_reads;       /* reads that produced output (reference length > 2)      */
  int64_t n_windows;
  uint8_t *bases;        /* per window: reference, corrected, uncorrected           */
  int64_t *off;          /* 3*n_windows + 1 byte offsets into bases                 */
  int64_t *read_first;   /* n_reads + 1: windows of emitted read r are
                            [read_first[r], read_first[r+1])                        */
  int64_t *read_index;   /* n_reads: index of the emitted read in the input batch   */
  int64_t small_reads;   /* corrected shorter than threshold * reference (:425-431) */
  int64_t wrong_reads;   /* no usable anchor chain (:417-423)                       */
} elector_windows;

/* reads: concatenated sequences, per read in the order reference, uncorrected,
 * corrected (masterSplitter's argument order, Master_Splitter.cpp:354-356);
 * read_off: 3*n_reads_in + 1 offsets; hdr_len: length of each read's header
 * line including '>' (it takes part in best_split's fragment-size comparison,
 * Master_Splitter.cpp:158-169,313-331).
 * size_threshold: SIZE_CORRECTED_READ_THRESHOLD (argv[10]). */
int  elector_split_reads(int64_t n_reads_in, const uint8_t *reads, const int64_t *read_off,
                         const int32_t *hdr_len, double size_threshold, int nthreads,
                         elector_windows *out);
void elector_windows_free(elector_windows *w);

/* The same splitter on the GPU (elector_amd/csrc/split_dev.hip): one workgroup per read, the k-mer tables in
 * HBM scratch, anchor chaining in LDS.  Same inputs as elector_split_reads (HOST buffers: the reads go to the
 * device in one copy), same windows -- but their bases stay in device memory, already in the layout
 * elector_poa_batch_device takes, so the windows never cross PCIe:
 *   d_bases    device pointer, owned by the context, valid until its next splitter call
 *   off, read_first, read_index   host arrays as in elector_windows (malloc'd; elector_windows_dev_free)
 * Reads of any length are taken (the anchor arrays move from LDS to HBM beyond 3,000 possible anchors, i.e. for
 * reads beyond ~60 kb).  Returns ELECTOR_E_LIMIT when a read's window list outgrows the room reserved for it (one
 * window per 16 reference bases): elector_split_reads takes such a batch. */
typedef struct elector_windows_dev {
  int64_t n_reads, n_windows;
  uint8_t *d_bases;
  int64_t *off, *read_first, *read_index;
  int64_t small_reads, wrong_reads;
  int64_t *d_off;          /* the same 3 n_windows + 1 offsets in DEVICE memory (valid as long as d_bases is): what
                              elector_poa_batch_device_offsets takes, so that no per-window array crosses to the host
                              on the way from the splitter to the POA kernels */
} elector_windows_dev;

int  elector_split_reads_device(elector_ctx *ctx, int64_t n_reads_in, const uint8_t *reads, const int64_t *read_off,
                                const int32_t *hdr_len, double size_threshold, int nthreads, elector_windows_dev *out);
void elector_windows_dev_free(elector_windows_dev *w);
/* d_bases stays valid until the context's next splitter call: elector_ctx_copy moves it (device to device, or
 * to the host) after the work queued on the context */
int  elector_ctx_copy(elector_ctx *ctx, const void *src, void *dst, int64_t bytes);
/* copies device memory of the context's device to the host (tests, debugging) */
int  elector_ctx_copy_to_host(elector_ctx *ctx, const void *d_src, void *h_dst, int64_t bytes);

/* ---- the file ends of call site #1 (elector_amd/csrc/io_host.cpp) -------------------------------------------
 * The three sorted FASTA files as masterSplitter reads them (Master_Splitter.cpp:396-446: one header line and one
 * sequence line per record; :414 records whose reference has fewer than 3 bases are skipped without counting),
 * handed out in processing batches: at least min_records kept records, extended to the end of the last read (run
 * of records whose msa.fa header line -- fasta_format.c:33-37, lpo_format.c:410, Donatello.cpp:71-73 -- is the
 * same).  start / stop: range of kept-record indices to hand out (stop < 0: to the end).  A batch with n == 0
 * is the end.  seq holds per record the reference, uncorrected and corrected sequence (the order
 * elector_split_reads takes), hdr the reference records' header lines; the buffers belong to the handle and stay
 * valid until ELECTOR_READ_SETS - 1 more calls have been made on the handle (that many sets are used in turn). */
#define ELECTOR_READ_SETS 4
typedef struct elector_reads {
  int64_t n, first_index;
  uint8_t *seq;  int64_t *seq_off;     /* 3n + 1 offsets */
  uint8_t *hdr;  int64_t *hdr_off;     /*  n + 1 offsets */
} elector_reads;
int  elector_reads_open(const char *reference, const char *uncorrected, const char *corrected, void **handle);
/* The device whose runtime the readers opened from now on register their batch buffers with (page-locked memory: the
 * device splitter copies a batch's reads out of it in one DMA); -1, the default, is the calling thread's current device.
 * Without a device the buffers are plain memory. */
void elector_reads_set_device(int device);
int  elector_reads_next(void *handle, int64_t min_records, int64_t start, int64_t stop, elector_reads *out);
void elector_reads_close(void *handle);

/* One pass over the same three files without keeping the sequences: per kept record (same skipping rule as
 * elector_reads_next) the lengths of its reference, uncorrected and corrected sequence and whether its msa.fa header
 * line differs from the previous kept record's (a new read).  What a multi-GPU run needs to cut the records into
 * per-rank ranges at read boundaries (elector/alignment.py:117-119 is the fan-out this replaces): one rank scans,
 * the bounds are broadcast.  Arrays are malloc'd; elector_reads_index_free releases them. */
typedef struct elector_reads_index {
  int64_t n;
  int64_t *len;          /* 3n: reference, uncorrected, corrected */
  uint8_t *new_read;     /*  n */
} elector_reads_index;
int  elector_reads_scan(const char *reference, const char *uncorrected, const char *corrected, elector_reads_index *out);
void elector_reads_index_free(elector_reads_index *ix);

/* Donatello's records (Donatello.cpp:61-93): per piece "header\nrow\n" three times (reference, corrected,
 * uncorrected row; rows = per piece the three rows of piece_cols[p] bytes back to back).  out == NULL returns the
 * size; pieces with drop[p] != 0 (drop may be NULL) are left out.  Returns the bytes written or a negative code. */
int64_t elector_msa_format(int64_t n_pieces, const uint8_t *rows, const int64_t *piece_cols, const uint8_t *hdr,
                           const int64_t *hdr_off, const uint8_t *drop, uint8_t *out, int64_t out_cap, int nthreads);
/* ... of the context's last collected statistics job (include/elector_stats.h), device rows -> pinned host
 * memory -> records -> one write() to the file descriptor */
int64_t elector_msa_records_write(elector_ctx *ctx, int64_t n_pieces, const int64_t *piece_cols, const uint8_t *hdr,
                                  const int64_t *hdr_off, const uint8_t *drop, int fd, int nthreads);

/* ... at byte `offset` of a descriptor opened without O_APPEND (the caller keeps the end-of-file position), the copy
 * into the page cache spread over `nthreads` threads (one pwrite each) */
int64_t elector_msa_records_pwrite(elector_ctx *ctx, int64_t n_pieces, const int64_t *piece_cols, const uint8_t *hdr,
                                   const int64_t *hdr_off, const uint8_t *drop, int fd, int64_t offset, int nthreads);

/* Merged per-read MSA (what Donatello appends to msa.fa): for emitted read r the
 * three rows each have read_cols[r] columns; columns whose corrected letter is
 * 'n' are dropped (Donatello.cpp:13-31). */
typedef struct elector_msa {
  int64_t n_reads;
  uint8_t *rows;         /* per read: reference row, corrected row, uncorrected row */
  int64_t *row_off;      /* n_reads + 1 offsets (3 * cols bytes per read)           */
  int64_t *cols;         /* n_reads                                                 */
} elector_msa;

/* window_rows / row_off / ncol: the output of elector_poa_batch for the windows
 * described by `read_first` (n_reads + 1 entries). */
int  elector_merge_windows(int64_t n_reads, const int64_t *read_first,
                           const uint8_t *window_rows, const int64_t *row_off, const int32_t *ncol,
                           elector_msa *out);
void elector_msa_free(elector_msa *m);

#ifdef __cplusplus
}
#endif
#endif
