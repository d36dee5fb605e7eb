// elector_amd/csrc/stats.hip -- per-read MSA statistics.
//
// k_stats: one lane per READ walks the read's pieces and their columns and
// produces the integer counters of include/elector_stats.h.  It replaces the
// per-column Python loops of the reference (elector/computeStats.py:61-189,
// 291-328, 371-440, 472-498, 712-752).  Byte-per-column work, HBM/latency bound;
// rows are read three bytes per column, the per-column "takes part" mask is a
// one-byte-per-column scratch array.  Floats never appear here.
//
// elector_homopolymer_pairs: host-side integer state machine for the one read
// whose homopolymer ratio the reference reports (computeStats.py:671-674).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ctx.h"
#include "elector_stats.h"

namespace elector {

constexpr int kThresh = 5;     // THRESH  computeStats.py:40
constexpr int kThresh2 = 20;   // THRESH2 computeStats.py:41

struct StatsArgs {
  int64_t n_reads;
  const int64_t *read_first;
  const uint8_t *rows;
  const int64_t *row_off;
  const int64_t *cols;
  const int32_t *clips;      // may be null
  int64_t *counters;
  uint8_t *mask;             // 1 byte per column, indexed like one row: mask_off[p]
  const int64_t *mask_off;   // n_pieces + 1
  int32_t *scratch;          // per read: interval lists + union bytes
  const int64_t *scr_off;    // n_reads + 1 (in int32 units)
};

// computeStats.py:61-77
__device__ int left_gaps(const uint8_t *row, int n)
{
  int gaps = 0, nts = 0, total = 0, i = 0;
  while (i < n && nts <= kThresh) {
    if (row[i] == '.') { ++gaps; nts = 0; }
    else { if (gaps >= kThresh) total = i; gaps = 0; ++nts; }
    ++i;
  }
  return total;
}

// computeStats.py:82-98
__device__ int right_gaps(const uint8_t *row, int n)
{
  int gaps = 0, nts = 0, total = 0, i = n - 1;
  while (i >= 0 && nts <= kThresh) {
    if (row[i] == '.') { ++gaps; nts = 0; }
    else { if (gaps >= kThresh) total = n - i; gaps = 0; ++nts; }
    --i;
  }
  return total;
}

// findGapStretches (computeStats.py:104-189).  Interval lists live in `scr`
// (pairs of ints): runs | tmp | merged | dict.  Returns the number of kept
// intervals, written as pairs to dict_out.
__device__ int gap_stretches(const uint8_t *cor, const uint8_t *ref, int n, int32_t *scr, int cap_pairs,
                             int32_t **dict_out)
{
  int32_t *runs = scr, *tmp = runs + 2 * cap_pairs, *mrg = tmp + 4 * cap_pairs, *dict = mrg + 4 * cap_pairs;
  int n_total = 0, n_ne = 0;      // list length incl. empty entries / non-empty entries
  bool last_empty = false, have_prev = false, prev_gap = false;
  int cg = 0, cgr = 0;
  for (int pos = 0; pos < n; ++pos) {
    const bool cgap = cor[pos] == '.', rgap = ref[pos] == '.';
    if (have_prev && prev_gap) {
      if (cgap) cg = (cg > 0) ? cg + 1 : 2;
      if (rgap) cgr = (cgr > 0) ? cgr + 1 : 2;
    }
    if (!have_prev) { if (cgap) ++cg; if (rgap) ++cgr; }
    if (!cgap) { if (cg > 0) { ++n_total; last_empty = true; } cg = 0; }
    if (!rgap) cgr = 0;
    if (cg >= kThresh && cgr < kThresh2) {
      if (n_total == 0) { runs[0] = pos - kThresh + 1; runs[1] = pos; n_total = 1; n_ne = 1; last_empty = false; }
      else {
        if (last_empty) {
          if (n_ne < cap_pairs) { runs[2 * n_ne] = pos - kThresh + 1; runs[2 * n_ne + 1] = pos; ++n_ne; }
          last_empty = false;
        }
        runs[2 * (n_ne - 1) + 1] = pos;
      }
    }
    have_prev = true;
    prev_gap = cgap;
  }
  // borders (:146-162)
  int nt = 0;
  for (int k = 0; k < n_ne; ++k) {
    const int s0 = runs[2 * k], s1 = runs[2 * k + 1];
    if (n_total > 1) {
      if (s0 <= kThresh2) { tmp[2 * nt] = 0; tmp[2 * nt + 1] = s1; ++nt; }
      if (n - s1 <= kThresh2) { tmp[2 * nt] = s0; tmp[2 * nt + 1] = n - 1; ++nt; }
      else { tmp[2 * nt] = s0; tmp[2 * nt + 1] = s1; ++nt; }
    } else {
      if (s0 <= kThresh2) { tmp[2 * nt] = 0; tmp[2 * nt + 1] = s1; ++nt; }
      else { tmp[2 * nt] = s0; tmp[2 * nt + 1] = s1; ++nt; }
      if (n - s1 <= kThresh2) tmp[2 * (nt - 1) + 1] = n - 1;
    }
  }
  // merge neighbours (:165-177)
  int nm = 0;
  bool merge = false;
  for (int i = 0; i + 1 < nt; ++i) {
    if (tmp[2 * (i + 1)] - tmp[2 * i + 1] <= kThresh) { mrg[2 * nm] = tmp[2 * i]; mrg[2 * nm + 1] = tmp[2 * (i + 1) + 1]; merge = true; }
    else { mrg[2 * nm] = tmp[2 * i]; mrg[2 * nm + 1] = tmp[2 * i + 1]; merge = false; }
    ++nm;
  }
  if (!merge && nt > 0) { mrg[2 * nm] = tmp[2 * (nt - 1)]; mrg[2 * nm + 1] = tmp[2 * (nt - 1) + 1]; ++nm; }
  // keep only long stretches touching an end; dict semantics: same start overwrites (:182-188)
  int nd = 0;
  for (int i = 0; i < nm; ++i) {
    const int s0 = mrg[2 * i], s1 = mrg[2 * i + 1];
    if ((s0 == 0 || s1 == n - 1) && s1 - s0 > kThresh2) {
      int j = 0;
      while (j < nd && dict[2 * j] != s0) ++j;
      dict[2 * j] = s0; dict[2 * j + 1] = s1;
      if (j == nd) ++nd;
    }
  }
  *dict_out = dict;
  return nd;
}

__device__ inline bool is_gc(uint8_t c) { return c == 'g' || c == 'c' || c == 'G' || c == 'C'; }

__global__ void __launch_bounds__(64) k_stats(StatsArgs a)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_reads) return;
  const int64_t p0 = a.read_first[r], p1 = a.read_first[r + 1];
  const int nfrag = (int)(p1 - p0);
  int32_t *scr = a.scratch + a.scr_off[r];
  const int64_t scr_len = a.scr_off[r + 1] - a.scr_off[r];
  // scratch layout: [union bytes: ucap] [interval lists: 14*cap_pairs ints]
  int64_t maxc = 0;
  for (int64_t p = p0; p < p1; ++p) maxc = a.cols[p] > maxc ? a.cols[p] : maxc;
  const int ucap = (int)maxc;
  uint8_t *uni = reinterpret_cast<uint8_t *>(scr);
  int32_t *lists = scr + (ucap + 3) / 4;
  const int cap_pairs = (int)((scr_len - (ucap + 3) / 4) / 14) - 1;
  if (nfrag > 1) for (int i = 0; i < ucap; ++i) uni[i] = 0;

  int64_t missing = 0;
  for (int64_t p = p0; p < p1; ++p) {
    int64_t *out = a.counters + p * ES_NCOUNTERS;
    for (int k = 0; k < ES_NCOUNTERS; ++k) out[k] = 0;
    out[ES_EXT_LEFT] = out[ES_EXT_RIGHT] = -1;
    out[ES_MISSING_LAST] = -1;
    const int n = (int)a.cols[p];
    if (n <= 10) { out[ES_MISSING] = missing; continue; }          // computeStats.py:577,624
    const uint8_t *ref = a.rows + a.row_off[p], *cor = ref + n, *unc = cor + n;
    uint8_t *mask = a.mask + a.mask_off[p];
    out[ES_PROCESSED] = 1;

    // gapsAndExtensions (:472-498)
    const int gl = min(left_gaps(ref, n), left_gaps(unc, n));
    const int gr = min(right_gaps(ref, n), right_gaps(unc, n));
    if (gl >= kThresh && gl >= kThresh2) {
      int dots = 0;
      for (int i = 0; i < gl; ++i) dots += cor[i] == '.';
      out[ES_EXT_LEFT] = gl - dots;
    }
    if (gr >= kThresh && gr >= kThresh2) {
      int dots = 0;
      for (int i = n - gr + 1; i < n; ++i) dots += cor[i] == '.';
      out[ES_EXT_RIGHT] = gr - dots;
    }
    int32_t *dict = nullptr;
    const int nd = gap_stretches(cor, ref, n, lists, cap_pairs, &dict);
    for (int k = 0; k < nd; ++k) {
      const int s0 = dict[2 * k], s1 = dict[2 * k + 1];
      int dots = 0;
      for (int i = s0; i <= s1; ++i) dots += ref[i] == '.';
      missing += s1 - s0 - dots;
    }
    missing -= gl + gr;
    if (missing < 0) missing = 0;
    out[ES_MISSING] = missing;
    out[ES_GAPS_LEFT] = gl;
    out[ES_GAPS_RIGHT] = gr;

    // getCorrectedPositions (:712-752)
    for (int i = 0; i < n; ++i) mask[i] = 1;
    if (a.clips) {
      const int lc = a.clips[2 * p], rc = a.clips[2 * p + 1];
      int i = 0, j = 0;
      while (j < lc && i < n) { if (cor[i] != '.') ++j; mask[i] = 0; ++i; }
      if (lc != 0 || rc != 0) {
        const int right_clip = n - rc;
        i = n - 1; j = n - 1;
        while (j >= right_clip && i >= 0) { if (cor[i] != '.') --j; mask[i] = 0; --i; }
      }
    }
    for (int k = 0; k < nd; ++k)
      for (int i = dict[2 * k]; i <= dict[2 * k + 1]; ++i) mask[i] = 0;
    if (gl >= kThresh) for (int i = 0; i < gl; ++i) mask[i] = 0;
    if (gr >= kThresh) for (int i = n - 1; i > n - gr; --i) mask[i] = 0;

    // per-column counters (:399-440 with :291-328 and :371-393)
    int64_t tp = 0, fp = 0, fn = 0, cb = 0, ub = 0, ucb = 0, uub = 0, gcr = 0, gcc = 0;
    int64_t insu = 0, delu = 0, subu = 0, insc = 0, delc = 0, subc = 0, lr = 0, lcn = 0, lu = 0;
    for (int i = 0; i < n; ++i) {
      const uint8_t x = ref[i], c = cor[i], u = unc[i];
      gcr += is_gc(x); gcc += is_gc(c);
      lr += x != '.'; lcn += c != '.'; lu += u != '.';
      if (!mask[i]) continue;
      if (c != x) { if (x == '.') ++insc; else if (c != '.') ++subc; else ++delc; }
      if (u != x) { if (x == '.') ++insu; else if (u != '.') ++subu; else ++delu; }
      if (x == u) { if (u != c) { ++fp; ++ub; } else { ++tp; ++cb; } ++ucb; }
      else { if (x == c) { ++tp; ++cb; } else { if (u == c) { ++fn; ++fp; } ++ub; } ++uub; }
    }
    out[ES_TP] = tp; out[ES_FP] = fp; out[ES_FN] = fn; out[ES_COR] = cb; out[ES_UNC] = ub;
    out[ES_UCOR] = ucb; out[ES_UUNC] = uub; out[ES_GC_REF] = gcr; out[ES_GC_COR] = gcc;
    out[ES_INS_U] = insu; out[ES_DEL_U] = delu; out[ES_SUB_U] = subu;
    out[ES_INS_C] = insc; out[ES_DEL_C] = delc; out[ES_SUB_C] = subc;
    out[ES_LEN_REF] = lr; out[ES_LEN_COR] = lcn; out[ES_LEN_UNC] = lu;

    if (nfrag > 1) {
      for (int i = 0; i < n; ++i) uni[i] |= mask[i];               // realNotMissing (:589-591)
      if (p == p1 - 1) {                                            // last piece (:595-599)
        int64_t miss = 0;
        for (int i = 0; i < n; ++i) miss += (!uni[i] && ref[i] != '.');
        out[ES_MISSING_LAST] = miss;
      }
    }
  }
}

}  // namespace elector

using namespace elector;

extern "C" int elector_stats_batch(elector_ctx *c, int64_t n_reads, const int64_t *read_first, int64_t n_pieces,
                                   const uint8_t *rows, const int64_t *row_off, const int64_t *cols,
                                   const int32_t *clips, int64_t *counters, uint8_t *last_mask)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n_reads < 0 || n_pieces < 0 || !read_first || (n_pieces > 0 && (!rows || !row_off || !cols || !counters)))
    return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  if (n_reads == 0 || n_pieces == 0) return ELECTOR_OK;
  if (read_first[0] != 0 || read_first[n_reads] != n_pieces) return elector_fail(c, ELECTOR_E_INVAL, "read_first must cover all pieces");
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  // host metadata: mask offsets (one byte per column) and per-read scratch offsets
  std::vector<int64_t> mask_off((size_t)n_pieces + 1), scr_off((size_t)n_reads + 1);
  mask_off[0] = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    if (cols[p] < 0 || cols[p] > 0x3fffffff || row_off[p + 1] - row_off[p] != 3 * cols[p])
      return elector_fail(c, ELECTOR_E_INVAL, "row_off/cols mismatch");
    mask_off[p + 1] = mask_off[p] + cols[p];
  }
  scr_off[0] = 0;
  for (int64_t r = 0; r < n_reads; ++r) {
    if (read_first[r + 1] < read_first[r]) return elector_fail(c, ELECTOR_E_INVAL, "read_first must be non-decreasing");
    int64_t maxc = 0;
    for (int64_t p = read_first[r]; p < read_first[r + 1]; ++p) maxc = std::max(maxc, cols[p]);
    const int64_t pairs = maxc / kThresh + 4;                       // a run needs >= 5 gap columns
    scr_off[r + 1] = scr_off[r] + (maxc + 3) / 4 + 14 * (pairs + 2) + 8;
  }
  const int64_t total_rows = row_off[n_pieces];
  int rc = c->d_st_rows.ensure((size_t)total_rows + 64) | c->d_st_rowoff.ensure((size_t)(n_pieces + 1) * 8) |
           c->d_st_cols.ensure((size_t)n_pieces * 8) | c->d_st_first.ensure((size_t)(n_reads + 1) * 8) |
           c->d_st_cnt.ensure((size_t)n_pieces * ES_NCOUNTERS * 8) | c->d_st_mask.ensure((size_t)mask_off[n_pieces] + 64) |
           c->d_st_scr.ensure((size_t)scr_off[n_reads] * 4 + 64) |
           c->d_st_scroff.ensure((size_t)(n_reads + 1 + n_pieces + 1) * 8) |
           (clips ? c->d_st_clips.ensure((size_t)n_pieces * 8) : 0);
  if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "statistics workspace");
  hipStream_t st = c->stream;
  int64_t *d_scroff = c->d_st_scroff.as<int64_t>();
  int64_t *d_maskoff = d_scroff + (n_reads + 1);
  HIPCHK(c, hipMemcpyAsync(c->d_st_rows.p, rows, (size_t)total_rows, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_st_rowoff.p, row_off, (size_t)(n_pieces + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_st_cols.p, cols, (size_t)n_pieces * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_st_first.p, read_first, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_scroff, scr_off.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_maskoff, mask_off.data(), (size_t)(n_pieces + 1) * 8, hipMemcpyHostToDevice, st));
  if (clips) HIPCHK(c, hipMemcpyAsync(c->d_st_clips.p, clips, (size_t)n_pieces * 8, hipMemcpyHostToDevice, st));
  StatsArgs a;
  a.n_reads = n_reads;
  a.read_first = c->d_st_first.as<int64_t>();
  a.rows = c->d_st_rows.as<uint8_t>();
  a.row_off = c->d_st_rowoff.as<int64_t>();
  a.cols = c->d_st_cols.as<int64_t>();
  a.clips = clips ? c->d_st_clips.as<int32_t>() : nullptr;
  a.counters = c->d_st_cnt.as<int64_t>();
  a.mask = c->d_st_mask.as<uint8_t>();
  a.mask_off = d_maskoff;
  a.scratch = c->d_st_scr.as<int32_t>();
  a.scr_off = d_scroff;
  hipLaunchKernelGGL(k_stats, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, st, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(counters, c->d_st_cnt.p, (size_t)n_pieces * ES_NCOUNTERS * 8, hipMemcpyDeviceToHost, st));
  if (last_mask) {
    const int64_t pl = read_first[n_reads - 1];
    const int64_t nb = mask_off[n_pieces] - mask_off[pl];
    if (nb > 0)
      HIPCHK(c, hipMemcpyAsync(last_mask, c->d_st_mask.as<uint8_t>() + mask_off[pl], (size_t)nb, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  return ELECTOR_OK;
}

// ------------------------------------------------------- homopolymers (host) ---
// The reference keeps two growing lists (`reported`) and, when a homopolymer
// ends, picks the most frequent reference letter and measures its longest run
// in both lists (computeStats.py:344-363).  Here the lists are never stored:
// per candidate letter we keep its count, first position, current run and
// longest run in each list, which is all those lines read.  Ties in "most
// frequent" go to the letter seen first (the reference's tie order depends on
// Python's per-process string hashing).

namespace {

struct HomoState {
  // one entry per distinct byte seen in either list since the last reset
  struct Cand { uint8_t ch; int64_t count_r, first_r; int64_t cur_r, max_r, cur_c, max_c; };
  std::vector<Cand> cands;
  int64_t len = 0;
  uint8_t last_r = 0, last_c = 0;
  void reset(uint8_t r, uint8_t c) { cands.clear(); len = 0; push(r, c); }
  Cand &find(uint8_t ch)
  {
    for (auto &k : cands) if (k.ch == ch) return k;
    // first appearance in either list: every earlier non-gap letter differed, so all runs are 0
    cands.push_back({ch, 0, -1, 0, 0, 0, 0});
    return cands.back();
  }
  void push(uint8_t r, uint8_t c)
  {
    find(r); find(c);
    for (auto &k : cands) {
      if (k.ch == r) { if (k.count_r++ == 0) k.first_r = len; ++k.cur_r; if (k.cur_r > k.max_r) k.max_r = k.cur_r; }
      else if (r != '.') k.cur_r = 0;
      if (k.ch == c) { ++k.cur_c; if (k.cur_c > k.max_c) k.max_c = k.cur_c; }
      else if (c != '.') k.cur_c = 0;
    }
    ++len;
    last_r = r; last_c = c;
  }
};

}  // namespace

extern "C" int64_t elector_homopolymer_pairs(int64_t n_pieces, const uint8_t *rows, const int64_t *row_off,
                                             const int64_t *cols, const uint8_t *mask, int32_t threshold,
                                             int32_t *pairs, int64_t cap)
{
  if (n_pieces < 0 || (n_pieces > 0 && (!rows || !row_off || !cols || !mask))) return ELECTOR_E_INVAL;
  int64_t npairs = 0, moff = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    const int64_t n = cols[p];
    if (n <= 10) { moff += n; continue; }
    const uint8_t *ref = rows + row_off[p], *cor = ref + n;
    const uint8_t *mk = mask + moff;
    HomoState st;
    st.reset('x', 'x');                                   // reported = [['x'],['x']] (:419)
    bool ok_to_report = false, end_ref = false;
    for (int64_t i = 0; i < n; ++i) {
      const uint8_t r = ref[i], c = cor[i];
      bool app_r = false, app_c = false, end_cor = true;
      if (mk[i]) {
        if (r != '.') {
          if (r == st.last_r) { app_r = true; if (st.len + 1 >= threshold) ok_to_report = true; }
          else if (ok_to_report) end_ref = true;
        }
        if (c != '.' && c == st.last_c) { app_c = true; end_cor = false; }
      }
      if (app_c || app_r) st.push(r, c);
      else if (!(end_ref && end_cor) && !end_ref && r != '.') st.reset(r, c);
      if (end_ref && end_cor) {
        // most frequent letter of the reference list, gaps only if nothing else is there
        const HomoState::Cand *best = nullptr, *best_ng = nullptr;
        for (auto &k : st.cands) {
          if (k.count_r == 0) continue;
          if (!best || k.count_r > best->count_r || (k.count_r == best->count_r && k.first_r < best->first_r)) best = &k;
          if (k.ch != '.' && (!best_ng || k.count_r > best_ng->count_r ||
                              (k.count_r == best_ng->count_r && k.first_r < best_ng->first_r)))
            best_ng = &k;
        }
        const HomoState::Cand *pick = (best && best->ch == '.') ? best_ng : best;
        if (pick) {
          if (npairs < cap && pairs) { pairs[2 * npairs] = (int32_t)pick->max_c; pairs[2 * npairs + 1] = (int32_t)pick->max_r; }
          ++npairs;
        }
        ok_to_report = false; end_ref = false;
        st.reset(r, c);
      }
    }
    moff += n;
  }
  return npairs;
}
