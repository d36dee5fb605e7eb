/* include/elector_poa.h -- C ABI of the MI355X-native triplet-MSA engine.
 *
 * This is the drop-in boundary for the one hot path of kamimrcht/ELECTOR that
 * this library replaces: the per-window (reference, corrected, uncorrected)
 * partial-order alignment that ELECTOR obtains today by spawning its embedded
 * poaV2 binary,
 *
 *     bin/poa -pir OUT -corrected_reads_fasta F3 -reference_reads_fasta F1
 *             -uncorrected_reads_fasta F2 -pathMatrix blosum80.mat
 *                                   (reference: elector/alignment.py:59-63,
 *                                    src/poa-graph/main.c:241-287)
 *
 * Every entry point is plain C (pointers + sizes, no torch / HIP types), is
 * re-entrant per context, returns 0 or a negative ELECTOR_E_* code and never
 * calls exit()/abort().  There is NO CPU fallback: without a usable gfx950
 * device every compute entry returns ELECTOR_E_NODEVICE.
 *
 * Vocabulary: a *window* is one (reference, corrected, uncorrected) triple of
 * short sequences as the splitter emits them; its result is a 3-row MSA of
 * `ncol` columns (rows in the order reference, corrected, uncorrected, gap
 * character '.', lower-case letters; unknown symbols print as 'A').
 */
#ifndef ELECTOR_POA_H
#define ELECTOR_POA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELECTOR_MAX_SYMBOL 32     /* device alphabet limit (shipped matrix: 31) */
#define ELECTOR_MAX_GAPTAB 64     /* max_gap_length + 2 must fit               */
#define ELECTOR_MAX_SEQ    524000 /* longest sequence of a window (bases): with the largest penalty (500) scores of
                                     three such sequences still fit the packed DP cell; windows whose moves scratch
                                     would not fit device memory are refused per batch (ELECTOR_W_TOOLONG) */

/* error codes (negative) */
#define ELECTOR_OK            0
#define ELECTOR_E_INVAL      (-1)   /* bad argument / malformed offsets            */
#define ELECTOR_E_NODEVICE   (-2)   /* no HIP device, or not gfx950                */
#define ELECTOR_E_NOMEM      (-3)   /* device or host allocation failed            */
#define ELECTOR_E_HIP        (-4)   /* a HIP runtime call failed                   */
#define ELECTOR_E_PARAMS     (-5)   /* scoring parameters outside device limits    */
#define ELECTOR_E_IO         (-6)   /* matrix file unreadable / malformed          */
#define ELECTOR_E_WINDOW     (-7)   /* at least one window failed: see status[]    */
#define ELECTOR_E_LIMIT      (-8)   /* an input exceeds an on-chip limit of a device entry; its host twin takes it */
/* per-window status values */
#define ELECTOR_W_OK          0
#define ELECTOR_W_EMPTY       1     /* a sequence of the window is empty           */
#define ELECTOR_W_TOOLONG     2     /* a sequence exceeds ELECTOR_MAX_SEQ, scores could leave the packed-cell range,
                                       the moves of the window (4 bits per DP cell) exceed the scratch budget, or a
                                       graph node has a predecessor more than 65,535 nodes back */
#define ELECTOR_W_INTERNAL    3     /* graph invariant violated on device          */

/* Scoring parameters = what the reference's read_score_matrix() leaves behind
 * (src/poa-graph/seq_util.c:82-217): alphabet, substitution scores and the two
 * gap-penalty arrays indexed by the per-cell gap tag 0..max_gap_length+1. */
typedef struct elector_params {
  int32_t nsymbol;
  char    symbol[ELECTOR_MAX_SYMBOL + 4];
  int32_t score[ELECTOR_MAX_SYMBOL][ELECTOR_MAX_SYMBOL];
  int32_t max_gap_length;                         /* M = trunc + decay */
  int32_t gap_penalty_x[ELECTOR_MAX_GAPTAB];      /* [0..M+1]          */
  int32_t gap_penalty_y[ELECTOR_MAX_GAPTAB];
} elector_params;

typedef struct elector_ctx elector_ctx;

/* library / device probes (no GPU needed) */
const char *elector_version(void);
const char *elector_strerror(int code);
int  elector_device_count(void);                  /* number of usable gfx950 devices */

/* a1: scoring parameters.
 * elector_params_default = the values ELECTOR ships in src/poa-graph/blosum80.mat;
 * elector_params_read replaces read_score_matrix (seq_util.c:82-217). */
void elector_params_default(elector_params *p);
int  elector_params_read(const char *matrix_path, elector_params *p);

/* one context per GPU (stream + workspace); thread-safe per context */
int  elector_ctx_create(int device, const elector_params *p, elector_ctx **out);
void elector_ctx_destroy(elector_ctx *ctx);
/* last error text of this context (HIP error strings etc.) */
const char *elector_ctx_last_error(const elector_ctx *ctx);

/* a2-a11 for a batch of n windows, HOST buffers (PCIe-inclusive path).
 *   bases : concatenated raw ASCII sequences (no whitespace), order per window
 *           reference, corrected, uncorrected
 *   off   : 3n+1 byte offsets into bases (off[0] = 0)
 *   rows  : out, capacity rows_cap bytes; window w occupies
 *           rows[row_off[w] .. row_off[w+1]) = 3 rows of ncol[w] bytes each
 *           (reference row, corrected row, uncorrected row), no terminators
 *   row_off : out, n+1 entries       ncol : out, n entries
 *   status  : out, n entries (ELECTOR_W_*)
 *   scores  : optional out, 2n entries: best score of alignment #1 and #2
 * Replaces one `poa` process over n records (main.c:265-284). */
int elector_poa_batch(elector_ctx *ctx, int64_t n,
                      const uint8_t *bases, const int64_t *off,
                      uint8_t *rows, int64_t rows_cap, int64_t *row_off,
                      int32_t *ncol, int32_t *status, int32_t *scores);

/* Same computation with the bulk data resident in device memory (the hot-path
 * entry bench.py times).  `off` stays a HOST array (metadata); d_bases is a
 * device pointer to off[3n] bytes.  Results stay on the device:
 *   d_cols : device buffer of 3*off[3n] bytes; window w's MSA is stored
 *            column-interleaved at d_cols[3*off[3w] + 3*c + r]
 *            (r = 0 reference, 1 corrected, 2 uncorrected), c < ncol[w]
 *   d_ncol, d_status : device int32[n];  d_scores: optional device int32[2n]
 * Work is enqueued on the context's stream; elector_ctx_sync() waits for it. */
int elector_poa_batch_device(elector_ctx *ctx, int64_t n,
                             const uint8_t *d_bases, const int64_t *off,
                             uint8_t *d_cols, int32_t *d_ncol, int32_t *d_status,
                             int32_t *d_scores);
/* The same with the offsets in device memory too: d_off = 3n+1 int64 byte offsets (d_off[0] = 0, d_off[3n] = total,
 * checked on the device), e.g. the array the device splitter leaves behind (include/elector_split.h:
 * elector_windows_dev.d_off).  No per-window work happens on the host: window status, launch class and the class
 * lists are computed by kernels, the host reads back per-class totals (29 KB) and queues the launches.  This is
 * the entry the pipeline and bench.py use; the reference has no counterpart (its poa reads the windows from three
 * FASTA files in file order, main.c:265-284).
 * The call is not fully asynchronous: it waits once for the classification kernel's totals on the context's stream
 * -- that is, for whatever the context still had queued (the previous batch's merge / statistics) -- before it queues
 * the alignment launches.  A caller that wants batch i + 1 queued while batch i runs uses several contexts in turn (the
 * pipeline: three, bench.py: four); with one context the GPU idles while the host decides the launches.
 * A call that fails leaves the context without a "last batch" (elector_msa_stats_enqueue / elector_poa_bundles refuse). */
int elector_poa_batch_device_offsets(elector_ctx *ctx, int64_t n,
                                     const uint8_t *d_bases, const int64_t *d_off, int64_t total,
                                     uint8_t *d_cols, int32_t *d_ncol, int32_t *d_status,
                                     int32_t *d_scores);
int elector_ctx_sync(elector_ctx *ctx);

/* measurement hooks for bench.py: HIP-event time (ms) and span count of a
 * kernel class accumulated since the last reset, measured with events on the
 * stream each kernel is launched on.  kernel: 0 = alignment #1 stage
 * (k_fused_a<G,R> launches; k_dp1 on the generic path), 1 = alignment #2 stage
 * (k_fused_b<G,R,D>; k_dp2), 2 = everything else of the POA stage (symbolize, the
 * trivial-window pass and list sort, generic leftovers, tiled long windows),
 * 3 = merge + statistics kernels (include/elector_stats.h), 4 = k_poa (the whole window in one kernel,
 * poa_pack.hip: when it is in use, kinds 0 and 1 only see the windows it handed back; its far-edge instance,
 * four launches of a few hundred wavefronts per batch, is kind 6), 5 = k_bundle (a12).
 * The geometry classes run on two concurrent launch chains: the sums overlap in wall time. */
int elector_ctx_timing_enable(elector_ctx *ctx, int on);
int elector_ctx_timing_read(elector_ctx *ctx, int kernel, double *ms, int64_t *launches);
int elector_ctx_timing_reset(elector_ctx *ctx);
/* tuning / measurement knobs.  "chains" = number of concurrent launch chains the
 * geometry classes of the fused kernels are dealt to (1..4; 0 = the default, 2):
 * with 1 every kernel of a batch runs alone on the chip, which is what an
 * un-overlapped per-kernel measurement needs (bench.py's serial pass).
 * "priority" = -1 / 0 / +1: the context's streams at the device's highest / default / lowest priority (before the
 * context's first call): a pipeline gives its splitter context the lowest, so that the alignment kernels' wavefronts
 * go first where both want the chip (measured: no gain).
 * "cus" = lo * 1000 + hi: the context's streams on the compute units lo .. hi - 1 of the 256-bit queue mask
 * (hipExtStreamCreateWithCUMask; before the context's first call; the runtime deals the bits out to the XCDs in turn,
 * so 8 k bits are k units on each).  Measured on the end-to-end pipeline: a splitter held to a part of the chip halves
 * the rate; off unless asked for. */
int elector_ctx_option(elector_ctx *ctx, const char *name, int64_t value);
/* diagnostics: |PO| (nodes after fusion #1) of the first n windows of the last
 * batch, so callers can count the DP cells of alignment #2 (|PO| x Lu). */
int elector_ctx_last_po_sizes(elector_ctx *ctx, int64_t n, int32_t *po_nodes);

/* a12 -- heaviest-bundle consensus (OPTIONAL output: the reference compiles
 * generate_lpo_bundles but its main() never calls it, src/poa-graph/main.c:345-347;
 * heaviest_bundle.c:16-78 heaviest_bundle, :80-110 assign_sequence_bundle_id,
 * :144-172 generate_lpo_bundles, lpo.c:762-781 add_path_sequence).
 *
 * elector_ctx_keep_graph(ctx, 1) makes the following POA batches keep what the
 * search needs in device memory (letters and flags of the graph after the first
 * fusion, its ring ids, the x -> y map of the second alignment: 14 bytes per node;
 * the kernels otherwise keep them on chip).  The batches run through the same
 * kernels as without it (round 5; earlier rounds took them off the one-kernel
 * path).  elector_poa_bundles then runs the bundle
 * search on every window of the LAST batch of this context (n must equal that
 * batch's n; its ncol / status device arrays must still be alive) exactly as
 * generate_lpo_bundles(lpo, minimum_fraction) would on the window's final graph:
 *   info      : out, 8 x int32 per window: [0] number of consensus sequences
 *               CONSENS0.. (0..3), [1..3] the "containing %d seqs" count of each,
 *               [4..6] bundle id assigned to the reference / corrected /
 *               uncorrected sequence (-1 = none), [7] ncol
 *   cons_rows : out, window w's info[8w] consensus rows of ncol[w] bytes each at
 *               cons_rows[cons_off[w] ..), as write_lpo_bundle_as_fasta would
 *               print them after the three sequences ('.' = not on the path)
 *   cons_off  : out, n+1 entries;  cons_cap: capacity of cons_rows in bytes
 *               (3 * off[3n] always suffices)
 * The reference's default minimum_fraction is 0.9 (main.c:30). */
int elector_ctx_keep_graph(elector_ctx *ctx, int on);
/* the search alone, queued on the context's stream, its results left in the context's device buffers (what
 * elector_poa_bundles then fetches): the entry bench.py --bundles times; timing kind 5 (elector_ctx_timing_read).
 * The search waits twice for the context's stream (the scratch size, the class counts); behind a whole alignment those
 * waits would hold up a thread that feeds several contexts.  So the call only NOTES the search (and refuses at once what
 * can be refused: no graph kept, n not the last batch's); the context queues it at its next call that waits for it
 * anyway -- elector_msa_stats_collect, elector_ctx_sync, elector_poa_bundles, or the next batch, in front of everything
 * that batch queues.  The last batch's ncol / status arrays must stay alive until then.
 * elector_ctx_option(ctx, "bundles_now", 1) queues the search inside the call instead. */
int elector_poa_bundles_enqueue(elector_ctx *ctx, int64_t n, float minimum_fraction);
int elector_poa_bundles(elector_ctx *ctx, int64_t n, float minimum_fraction,
                        uint8_t *cons_rows, int64_t cons_cap, int64_t *cons_off,
                        int32_t *info);

#ifdef __cplusplus
}
#endif
#endif
