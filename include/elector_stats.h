/* include/elector_stats.h -- C ABI of the per-read MSA statistics, the second
 * call site of the hot path.
 *
 * Replaces the per-column loops of the reference's elector/computeStats.py
 * (computeMetrics :519-675 and its helpers nbLeftGaps/nbRightGaps :61-98,
 * findGapStretches :104-189, getCorrectedPositions :712-752, indels :291-328,
 * getCorrectionAtEachPosition :371-393; called from elector/__main__.py:141
 * through outputRecallPrecision :196).  The library returns INTEGERS only;
 * every ratio, mean and round() stays with the caller so that floating-point
 * results are computed on the host in read order and do not depend on how reads
 * were sharded over GPUs.
 *
 * A *piece* is one 6-line record of msa.fa (reference row, corrected row,
 * uncorrected row of equal length `cols`); a *read* is a run of consecutive
 * pieces with the same header (getSplit, computeStats.py:45-56).
 */
#ifndef ELECTOR_STATS_H
#define ELECTOR_STATS_H

#include <stdint.h>
#include "elector_poa.h"

#ifdef __cplusplus
extern "C" {
#endif

/* counters per piece (int64 each) */
enum {
  ES_TP = 0, ES_FP, ES_FN,            /* true/false positives, false negatives        */
  ES_COR, ES_UNC,                     /* corBases / uncorBases (corrected read)       */
  ES_UCOR, ES_UUNC,                   /* uncorCorBases / uncorUncorBases              */
  ES_GC_REF, ES_GC_COR,               /* G/C letters in the reference / corrected row */
  ES_INS_U, ES_DEL_U, ES_SUB_U,       /* uncorrected vs reference                     */
  ES_INS_C, ES_DEL_C, ES_SUB_C,       /* corrected vs reference                       */
  ES_LEN_REF, ES_LEN_COR, ES_LEN_UNC, /* non-gap lengths of the three rows            */
  ES_GAPS_LEFT, ES_GAPS_RIGHT,        /* min(reference, uncorrected) end-gap extents  */
  ES_EXT_LEFT, ES_EXT_RIGHT,          /* extended bases left/right, -1 = none         */
  ES_MISSING,                         /* running missing size after this piece        */
  ES_MISSING_LAST,                    /* split reads, last piece: recount (:596-599); -1 otherwise */
  ES_PROCESSED,                       /* 1 if the piece was long enough (cols > 10)   */
  ES_NCOUNTERS
};

/* n_reads runs of pieces: read r owns pieces [read_first[r], read_first[r+1]).
 * rows: per piece three rows of cols[p] bytes at rows[row_off[p]] (reference,
 * corrected, uncorrected).  clips: optional 2*n_pieces soft-clip counts
 * (left, right) per piece, NULL = none (computeStats.py:718-741).
 * counters: out, n_pieces * ES_NCOUNTERS.
 * last_mask: optional out, sum of cols over the LAST read's pieces: 1 where the
 * column takes part in the statistics (existingCorrectedPositions), used by the
 * caller for the homopolymer ratio of the last read (computeStats.py:671-674). */
int elector_stats_batch(elector_ctx *ctx, int64_t n_reads, const int64_t *read_first,
                        int64_t n_pieces, const uint8_t *rows, const int64_t *row_off, const int64_t *cols,
                        const int32_t *clips, int64_t *counters, uint8_t *last_mask);

/* The same statistics computed where the MSAs already are: merges the windows of
 * the context's LAST elector_poa_batch_device call into one record per piece on
 * the device (what `Donatello` does on files: src/split/Donatello.cpp:13-31
 * drops the columns whose corrected letter is 'n', :48-93 concatenates the
 * windows of a read; spawned from elector/alignment.py:120-122) and counts.
 * Nothing but the integer counters crosses PCIe.
 *
 * n_windows, d_cols, d_ncol, d_status: the arguments/outputs of that POA call
 * (device pointers).  Piece p owns windows [piece_first[p], piece_first[p+1]),
 * read r owns pieces [read_first[r], read_first[r+1]) (host arrays); a window
 * whose status is not 0 contributes no column.
 * counters: out (host), n_pieces * ES_NCOUNTERS.  piece_cols: optional out
 * (host), surviving columns per piece.  last_rows / last_mask: optional out
 * (host), the LAST read's pieces back to back: per piece 3 * cols bytes
 * (reference, corrected, uncorrected row) resp. cols mask bytes; last_cap =
 * room in columns (the call fails with ELECTOR_E_INVAL when it is too small). */
int elector_msa_stats_device(elector_ctx *ctx, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                             const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                             int64_t n_reads, const int64_t *read_first, const int32_t *clips,
                             int64_t *counters, int64_t *piece_cols, uint8_t *last_rows, uint8_t *last_mask,
                             int64_t last_cap);

/* The same call in two halves, for callers that keep the GPU busy across batches:
 * _enqueue validates, uploads the two small index arrays and queues the merge
 * and statistics kernels and the copy of their results behind the POA kernels
 * on the context's stream, without waiting; _collect waits for the OLDEST
 * queued job and hands out its results (arguments as above; n_pieces must be
 * the job's).  At most two jobs may be in flight per context, so a caller can
 * prepare and queue batch i+1 (elector_poa_batch_device + _enqueue) while the
 * device still works on batch i, then collect batch i.  d_cols/d_ncol/d_status
 * are read by the queued kernels in stream order: a later POA call on the same
 * context may reuse them. */
int elector_msa_stats_enqueue(elector_ctx *ctx, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                              const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                              int64_t n_reads, const int64_t *read_first, const int32_t *clips);
int elector_msa_stats_collect(elector_ctx *ctx, int64_t n_pieces, int64_t *counters, int64_t *piece_cols,
                              uint8_t *last_rows, uint8_t *last_mask, int64_t last_cap);
/* _enqueue that also delivers the merged records -- the body of msa.fa (Donatello.cpp:86-91): piece p's three rows
 * (3 * piece_cols[p] bytes) back to back in piece order, what elector_msa_rows_fetch returns -- to rows_out, without a
 * call of its own in between: rows_out is page-locked host memory (hipHostMalloc / hipHostRegister; torch's
 * pin_memory()) or device memory, rows_cap its size, at least 3 bytes per base of the batch.  A kernel packs the rows
 * in the same queue; a device destination is complete when _collect returns for this job.  For a host destination
 * _collect (which learns the byte count with the counters) starts ONE copy of exactly that size on the context's copy
 * stream and returns; the copy engine moves the rows while the kernels of the following batches run, and
 * elector_msa_rows_wait() returns when every such copy of the context has arrived (call it before reading the rows
 * or handing the same destination to another job).  _collect's piece_cols say where each piece starts. */
int elector_msa_stats_enqueue_rows(elector_ctx *ctx, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                                   const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                                   int64_t n_reads, const int64_t *read_first, const int32_t *clips,
                                   uint8_t *rows_out, int64_t rows_cap);
int elector_msa_rows_wait(elector_ctx *ctx);

/* The merged records of the context's last COLLECTED statistics job, copied
 * to the host (elector_msa_stats_device collects its own job): piece p's three rows (3 * piece_cols[p] bytes) back to back in
 * piece order -- the body of msa.fa (Donatello.cpp:86-91).  piece_cols as
 * returned by that call. */
int elector_msa_rows_fetch(elector_ctx *ctx, int64_t n_pieces, const int64_t *piece_cols, uint8_t *rows);

/* Homopolymer size pairs of ONE read (host side, integer output): walks the
 * pieces' columns with the reference's state machine (computeStats.py:291-365,
 * :427-432) and writes (corrected_size, reference_size) pairs.
 * Returns the number of pairs (<= cap pairs written) or a negative code. */
int64_t elector_homopolymer_pairs(int64_t n_pieces, const uint8_t *rows, const int64_t *row_off,
                                  const int64_t *cols, const uint8_t *mask, int32_t threshold,
                                  int32_t *pairs, int64_t cap);

/* The host half of call site #2 (elector_amd/csrc/report_host.cpp): what computeMetrics / outputMetrics
 * (elector/computeStats.py:519-675, 444-468) derive from the per-piece counters -- per-read ratios, their means
 * (sequential double sums in read order), the totals, the three lists the report needs and the text of
 * `per_read_metrics.txt` behind its header line (three lines per assessed read, floats printed as Python's repr,
 * an int 0 where the reference has one).  counters: n_pieces x ES_NCOUNTERS as the statistics entries return them;
 * read r owns pieces [read_first[r], read_first[r+1]).  flags: bit 0 -- no read was assessed (the reference divides
 * by zero); bit 1 -- a processed piece without reference or corrected letters (likewise); bit 2 -- nb_reads is 0 (the
 * four means are the integer 0).  The arrays and the text are malloc'd; elector_report_free releases them. */
typedef struct elector_report {
  int64_t nb_reads, throughput, uncor_throughput;
  double precision, recall, cor_bases_rate, error_rate, uncor_cor_bases_rate, uncor_error_rate;
  double gc_ref, gc_cor;                       /* round(mean, 3) */
  int64_t indelsubs_unc[3], indelsubs_cor[3];
  int64_t count_split, count_trimmed, count_extended;
  int64_t n_missing, n_len_cor, n_extended;
  int64_t *missing_size, *len_corrected, *extended_bases;
  char *per_read_text;
  int64_t per_read_bytes;
  int32_t flags, pad;
} elector_report;
int  elector_report_aggregate(int64_t n_reads, const int64_t *read_first, int64_t n_pieces, const int64_t *counters,
                              elector_report *out);
void elector_report_free(elector_report *r);
/* The corrected-read lines of the read-size file (outputReadSizeDistribution, computeStats.py:279-285): one line
 * "<n> sequences\n" per record of the corrected FASTA file, n = length of the record's second line less its last
 * character, written to `fd`.  Returns the number of records or a negative code. */
int64_t elector_read_size_lines(const char *corrected_fasta, int fd);
/* The other lines of that file (computeStats.py:276-278): "<value><suffix>\n" per value (the lengths of the
 * corrected reads the report counted, suffix " reads"), written to `fd`.  Returns n or a negative code. */
int64_t elector_write_count_lines(const int64_t *values, int64_t n, const char *suffix, int fd);

#ifdef __cplusplus
}
#endif
#endif
