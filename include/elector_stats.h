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

/* Homopolymer size pairs of ONE read (host side, integer output): walks the
 * pieces' columns with the reference's state machine (computeStats.py:291-365,
 * :427-432) and writes (corrected_size, reference_size) pairs.
 * Returns the number of pairs (<= cap pairs written) or a negative code. */
int64_t elector_homopolymer_pairs(int64_t n_pieces, const uint8_t *rows, const int64_t *row_off,
                                  const int64_t *cols, const uint8_t *mask, int32_t threshold,
                                  int32_t *pairs, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
