// elector_amd/csrc/splitter.cpp -- window splitter and window merger (host side).
//
// Restates, with flat buffers, open-addressing k-mer tables and a thread pool,
// what the reference's two helper programs do around the POA engine:
//   src/split/Master_Splitter.cpp  (split :175-308, best_split :310-332, read loop :396-446)
//   src/split/Donatello.cpp        (clean_msa :13-31, per-read concatenation :50-84)
// Integer/unsigned conversion quirks of the reference are kept where they
// decide which windows come out (each one is marked "ref:").
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "elector_poa.h"
#include "elector_split.h"

namespace {

typedef uint32_t kmer_t;
struct Seq { const uint8_t *p; uint32_t n; };
struct Window { Seq ref, s1, s2; bool n_filler; };   // s2 == "N" filler when n_filler

// ---- 2-bit k-mer coding exactly as the reference does it -------------------
// ref: str2num maps every non-ACG byte to 3 (:26-38) while the rolling update
// maps every non-CGT byte to 0 (:41-49); both are kept.
inline kmer_t first_kmer(Seq s, int k)
{
  kmer_t r = 0;
  const uint32_t m = std::min<uint32_t>(s.n, (uint32_t)k);
  for (uint32_t i = 0; i < m; ++i) {
    r <<= 2;
    switch (s.p[i]) { case 'A': break; case 'C': r += 1; break; case 'G': r += 2; break; default: r += 3; break; }
  }
  return r;
}
inline kmer_t roll(kmer_t v, uint8_t c, int k)
{
  v <<= 2;
  switch (c) { case 'C': v += 1; break; case 'G': v += 2; break; case 'T': v += 3; break; default: break; }
  return v & (kmer_t)((1u << (2 * k)) - 1u);
}

// ---- k-mer -> position table (open addressing, per thread, reused) ---------
// One 12-byte entry per slot, occupied when its stamp is the table's current epoch: clearing the table is
// bumping the epoch, and a lookup touches one cache line.
struct KmerTable {
  struct Entry { kmer_t key; int32_t val; uint32_t stamp; };   // val: position, or -1 = repeated
  std::vector<Entry> e;
  uint32_t mask = 0, epoch = 0;
  void reset(size_t n)
  {
    size_t cap = 64;
    while (cap < 2 * n + 2) cap <<= 1;
    if (e.size() < cap) { e.assign(cap, Entry{0, 0, 0}); epoch = 0; }
    mask = (uint32_t)e.size() - 1;
    if (++epoch == 0) { for (auto &x : e) x.stamp = 0; epoch = 1; }
  }
  // the entry of k, or the free slot where it would go
  inline Entry *slot(kmer_t k)
  {
    uint32_t h = (k * 2654435761u) & mask;
    while (e[h].stamp == epoch && e[h].key != k) h = (h + 1) & mask;
    return &e[h];
  }
  inline const Entry *find(kmer_t k) const
  {
    uint32_t h = (k * 2654435761u) & mask;
    while (e[h].stamp == epoch && e[h].key != k) h = (h + 1) & mask;
    return e[h].stamp == epoch ? &e[h] : nullptr;
  }
  inline bool has(kmer_t k) const { return find(k) != nullptr; }
  inline int32_t get(kmer_t k) const { const Entry *x = find(k); return x ? x->val : 0; }
  // first occurrence keeps its position, any later one marks the k-mer as repeated
  inline void add(kmer_t k, int32_t pos)
  {
    Entry *x = slot(k);
    if (x->stamp == epoch) x->val = -1;
    else { x->key = k; x->val = pos; x->stamp = epoch; }
  }
  inline void put(kmer_t k, int32_t v) { Entry *x = slot(k); x->key = k; x->val = v; x->stamp = epoch; }
};

struct Anchor { int32_t r, a, b; };

struct Scratch {
  KmerTable t_ref, t_s1, t_sh;
  std::vector<Anchor> anchors;
  std::vector<int32_t> chain_len, chain_next, chain;
};

// longest chain of anchors increasing in all three reads with each step < 1000
// (ref: best_chain_from_anchor / _list, :79-126; the memoised recursion is
// evaluated back to front, ties keep the earliest successor / earliest start)
void best_chain(Scratch &sc)
{
  const std::vector<Anchor> &A = sc.anchors;
  const int n = (int)A.size();
  sc.chain_len.assign(n, 0);
  sc.chain_next.assign(n, -1);
  sc.chain.clear();
  for (int i = n - 1; i >= 0; --i) {
    int best = -1, nxt = -1;
    for (int j = i + 1; j < n; ++j) {
      if (A[j].r - A[i].r < 1000 && A[j].r > A[i].r) {
        if (A[j].a - A[i].a < 1000 && A[j].a > A[i].a && A[j].b - A[i].b < 1000 && A[j].b > A[i].b)
          if (sc.chain_len[j] > best) { best = sc.chain_len[j]; nxt = j; }
      } else break;                                         // ref: "TOO FAR NOW" (:98-101)
    }
    sc.chain_len[i] = 1 + best;
    sc.chain_next[i] = nxt;
  }
  int best = -1, start = -1;
  for (int i = 0; i < n; ++i)
    if (sc.chain_len[i] > best) { best = sc.chain_len[i]; start = i; }
  for (int i = start; i != -1; i = sc.chain_next[i]) sc.chain.push_back(i);
}

inline Seq sub(Seq s, uint32_t pos, uint32_t len)           // std::string::substr semantics (pos <= n)
{
  if (pos > s.n) pos = s.n;
  return Seq{s.p + pos, std::min<uint32_t>(len, s.n - pos)};
}
inline Seq sub(Seq s, uint32_t pos) { return sub(s, pos, 0xFFFFFFFFu); }

const uint8_t kN[1] = {'N'};

// ref: split (:175-308).  Appends windows to `out`.
void split(Seq ref, Seq S1, Seq S2, std::vector<Window> &out, bool first_call, int k, uint32_t minSize,
           Scratch &sc)
{
  KmerTable &kref = sc.t_ref, &kin1 = sc.t_s1, &ksh = sc.t_sh;
  kref.reset(ref.n); kin1.reset(S1.n); ksh.reset(S2.n);

  auto unique_in = [](const KmerTable &t, kmer_t q) { const KmerTable::Entry *x = t.find(q); return x && x->val != -1; };
  kmer_t seq = first_kmer(ref, k);
  kref.put(seq, 0);
  for (uint32_t j = 0; (uint64_t)j + k < ref.n; ++j) {
    seq = roll(seq, ref.p[j + k], k);
    kref.add(seq, (int32_t)(j + 1));                        // a second occurrence: repeated in the reference read
  }
  seq = first_kmer(S1, k);
  if (unique_in(kref, seq)) kin1.put(seq, 0);
  for (uint32_t j = 0; (uint64_t)j + k < S1.n; ++j) {
    seq = roll(seq, S1.p[j + k], k);
    if (unique_in(kref, seq)) kin1.add(seq, (int32_t)(j + 1));
  }
  seq = first_kmer(S2, k);
  if (unique_in(kin1, seq)) ksh.put(seq, 0);
  for (uint32_t j = 0; (uint64_t)j + k < S2.n; ++j) {
    seq = roll(seq, S2.p[j + k], k);
    if (unique_in(kin1, seq)) ksh.add(seq, (int32_t)(j + 1));
  }

  // anchors: k-mers unique in all three reads, at least minSize apart on the reference (:234-251)
  sc.anchors.clear();
  seq = first_kmer(ref, k);
  if (unique_in(ksh, seq)) sc.anchors.push_back({kref.get(seq), kin1.get(seq), ksh.get(seq)});
  uint32_t last_indexed = 0;
  for (uint32_t j = 0; (uint64_t)j + k < ref.n; ++j) {
    seq = roll(seq, ref.p[j + k], k);
    if ((uint32_t)(j - last_indexed) > minSize && unique_in(ksh, seq)) {
      sc.anchors.push_back({kref.get(seq), kin1.get(seq), ksh.get(seq)});
      last_indexed = j;
    }
  }
  best_chain(sc);
  // the recursive calls below reuse the scratch: keep what this level still needs
  const std::vector<Anchor> A = sc.anchors;
  const std::vector<int32_t> BL = sc.chain;

  if (BL.empty()) { out.push_back({ref, S1, S2, false}); return; }      // :256-261

  int i = 0;
  uint32_t pred_ref = 0, pred_S1 = 0, pred_S2 = 0;
  {
    const Seq start_ref = sub(ref, 0, (uint32_t)(A[BL[0]].r + k));
    const Seq start_S1 = sub(S1, 0, (uint32_t)(A[BL[0]].a + k));
    const Seq start_S2 = sub(S2, 0, (uint32_t)(A[BL[0]].b + k));
    // corrected read misses the start (trimmed/split): re-split reference vs uncorrected
    // and pad the corrected side with 'N' windows (:268-277)
    if ((uint64_t)start_S2.n * 2 < start_ref.n && start_ref.n - start_S2.n > 200 && first_call) {
      std::vector<Window> tmp;
      split(start_ref, start_S1, start_ref, tmp, false, k, (uint32_t)(1.2 * start_S2.n), sc);
      const size_t nf = tmp.size();
      for (size_t f = 0; f < nf; ++f) {
        Window w = tmp[f];
        if (f + 1 < nf || start_S2.n == 0) { w.s2 = Seq{kN, 1}; w.n_filler = true; }   // generate_dumb_str :139-154
        else { w.s2 = start_S2; w.n_filler = false; }
        out.push_back(w);
      }
      pred_S1 = (uint32_t)(A[BL[0]].a + k);
      pred_ref = (uint32_t)(A[BL[0]].r + k);
      pred_S2 = (uint32_t)(A[BL[0]].b + k);
      ++i;
    }
  }
  for (; i < (int)BL.size() - 1; ++i) {
    const Anchor &an = A[BL[i]];
    // ref: ints computed from uint arithmetic, then compared against the uint minSize (:281-282)
    const int size_R = (int)((uint32_t)an.r - pred_ref), size_S1 = (int)((uint32_t)an.a - pred_S1),
              size_S2 = (int)((uint32_t)an.b - pred_S2);
    if ((uint32_t)size_R > minSize && (uint32_t)size_S1 > minSize && (uint32_t)size_S2 > minSize &&
        std::abs(size_S1 - size_R) < size_R * 0.5 && std::abs(size_S2 - size_R) < size_R * 0.5) {
      out.push_back({sub(ref, pred_ref, (uint32_t)(an.r - (int32_t)pred_ref + k)),
                     sub(S1, pred_S1, (uint32_t)(an.a - (int32_t)pred_S1 + k)),
                     sub(S2, pred_S2, (uint32_t)(an.b - (int32_t)pred_S2 + k)), false});
      pred_S1 = (uint32_t)(an.a + k);
      pred_ref = (uint32_t)(an.r + k);
      pred_S2 = (uint32_t)(an.b + k);
    }
  }
  const Seq end_ref = sub(ref, pred_ref), end_S1 = sub(S1, pred_S1), end_S2 = sub(S2, pred_S2);
  if ((uint64_t)end_S2.n * 2 < end_ref.n && end_ref.n - end_S2.n > 200 && first_call) {     // :295-301
    std::vector<Window> tmp;
    split(end_ref, end_S1, end_ref, tmp, false, k, (uint32_t)(1.2 * end_S2.n), sc);
    const size_t nf = tmp.size();
    for (size_t f = 0; f < nf; ++f) {
      Window w = tmp[f];
      if (f == 0 && end_S2.n > 0) { w.s2 = end_S2; w.n_filler = false; }
      else { w.s2 = Seq{kN, 1}; w.n_filler = true; }
      out.push_back(w);
    }
  } else {
    out.push_back({end_ref, end_S1, end_S2, false});
  }
}

// ref: largest_fragment (:158-169) measures every LINE of the "header\nseq\n"
// text, header lines included, so the header length takes part.
uint32_t largest_fragment(const std::vector<Window> &ws, uint32_t hdr_len)
{
  uint32_t res = 0;
  bool first = true;
  for (const Window &w : ws) {
    res = std::max(res, first ? hdr_len : hdr_len + 1);
    first = false;
    res = std::max(res, w.ref.n + 1);
  }
  return res;
}

// ref: best_split (:310-332)
void best_split(Seq ref, Seq S1, Seq S2, uint32_t hdr_len, std::vector<Window> &best, Scratch &sc)
{
  int k = 15;
  best.clear();
  split(ref, S1, S2, best, true, k, 20, sc);
  uint32_t largest = largest_fragment(best, hdr_len);
  std::vector<Window> aux;
  for (;;) {
    k -= 2;
    if (k < 9) return;
    aux.clear();
    split(ref, S1, S2, aux, true, k, 20, sc);
    const uint32_t la = largest_fragment(aux, hdr_len);
    if (la < largest) { largest = la; best.swap(aux); }
    else return;
  }
}

const uint8_t kAAA[3] = {'A', 'A', 'A'};

struct ReadOut { std::vector<Window> ws; int kind; };   // kind 0 windows, 1 small, 2 wrong, -1 skipped

}  // namespace

extern "C" int elector_split_reads(int64_t n_in, const uint8_t *reads, const int64_t *read_off,
                                   const int32_t *hdr_len, double size_threshold, int nthreads,
                                   elector_windows *out)
{
  if (!out) return ELECTOR_E_INVAL;
  std::memset(out, 0, sizeof *out);
  if (n_in < 0 || (n_in > 0 && (!reads || !read_off || !hdr_len))) return ELECTOR_E_INVAL;
  for (int64_t r = 0; r < 3 * n_in; ++r)
    if (read_off[r + 1] < read_off[r] || read_off[r + 1] - read_off[r] > 0x7fffffff) return ELECTOR_E_INVAL;
  std::vector<ReadOut> res((size_t)n_in);
  if (nthreads < 1) nthreads = 1;
  std::atomic<int64_t> next(0);
  std::atomic<int> oom(0);
  auto worker = [&]() {
    try {
      Scratch sc;
      for (;;) {
        const int64_t r = next.fetch_add(1);
        if (r >= n_in) break;
        const Seq ref{reads + read_off[3 * r], (uint32_t)(read_off[3 * r + 1] - read_off[3 * r])};
        const Seq S1{reads + read_off[3 * r + 1], (uint32_t)(read_off[3 * r + 2] - read_off[3 * r + 1])};
        const Seq S2{reads + read_off[3 * r + 2], (uint32_t)(read_off[3 * r + 3] - read_off[3 * r + 2])};
        ReadOut &ro = res[(size_t)r];
        if (ref.n <= 2) { ro.kind = -1; continue; }                                   // :414
        if ((double)S2.n / ref.n >= size_threshold) {                                // :415
          best_split(ref, S1, S2, (uint32_t)hdr_len[r], ro.ws, sc);
          if (ro.ws.size() <= 1) { ro.ws.clear(); ro.kind = 2; }                      // :417-423
          else ro.kind = 0;
        } else ro.kind = 1;                                                           // :425-431
        if (ro.kind != 0) ro.ws.push_back({Seq{kAAA, 3}, Seq{kAAA, 3}, Seq{kAAA, 3}, false});
      }
    } catch (const std::bad_alloc &) { oom.store(1); }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
  }
  if (oom.load()) return ELECTOR_E_NOMEM;

  int64_t nreads = 0, nwin = 0, nb = 0;
  for (auto &ro : res) {
    if (ro.kind < 0) continue;
    ++nreads;
    nwin += (int64_t)ro.ws.size();
    for (auto &w : ro.ws) nb += (int64_t)w.ref.n + w.s1.n + w.s2.n;
    if (ro.kind == 1) out->small_reads++;
    if (ro.kind == 2) out->wrong_reads++;
  }
  out->bases = (uint8_t *)std::malloc((size_t)nb + 16);
  out->off = (int64_t *)std::malloc((size_t)(3 * nwin + 1) * sizeof(int64_t));
  out->read_first = (int64_t *)std::malloc((size_t)(nreads + 1) * sizeof(int64_t));
  out->read_index = (int64_t *)std::malloc((size_t)(nreads + 1) * sizeof(int64_t));
  if (!out->bases || !out->off || !out->read_first || !out->read_index) { elector_windows_free(out); return ELECTOR_E_NOMEM; }
  int64_t pos = 0, wi = 0, ri = 0;
  out->off[0] = 0;
  for (int64_t r = 0; r < n_in; ++r) {
    ReadOut &ro = res[(size_t)r];
    if (ro.kind < 0) continue;
    out->read_first[ri] = wi;
    out->read_index[ri] = r;
    ++ri;
    for (auto &w : ro.ws) {
      // C-ABI window order: reference, corrected (S2), uncorrected (S1)
      std::memcpy(out->bases + pos, w.ref.p, w.ref.n); pos += w.ref.n; out->off[3 * wi + 1] = pos;
      std::memcpy(out->bases + pos, w.s2.p, w.s2.n);   pos += w.s2.n;  out->off[3 * wi + 2] = pos;
      std::memcpy(out->bases + pos, w.s1.p, w.s1.n);   pos += w.s1.n;  out->off[3 * wi + 3] = pos;
      ++wi;
    }
  }
  out->read_first[ri] = wi;
  out->n_reads = nreads;
  out->n_windows = nwin;
  return ELECTOR_OK;
}

extern "C" void elector_windows_free(elector_windows *w)
{
  if (!w) return;
  std::free(w->bases); std::free(w->off); std::free(w->read_first); std::free(w->read_index);
  std::memset(w, 0, sizeof *w);
}

// ---------------------------------------------------------------- merger ---

extern "C" int elector_merge_windows(int64_t n_reads, const int64_t *read_first, const uint8_t *rows,
                                     const int64_t *row_off, const int32_t *ncol, elector_msa *out)
{
  if (!out) return ELECTOR_E_INVAL;
  std::memset(out, 0, sizeof *out);
  if (n_reads < 0 || (n_reads > 0 && (!read_first || !rows || !row_off || !ncol))) return ELECTOR_E_INVAL;
  out->row_off = (int64_t *)std::malloc((size_t)(n_reads + 1) * sizeof(int64_t));
  out->cols = (int64_t *)std::malloc((size_t)(n_reads + 1) * sizeof(int64_t));
  if (!out->row_off || !out->cols) { elector_msa_free(out); return ELECTOR_E_NOMEM; }
  // pass 1: surviving columns per read (Donatello.cpp:13-31 drops columns whose
  // corrected letter is 'n')
  int64_t total = 0;
  out->row_off[0] = 0;
  for (int64_t r = 0; r < n_reads; ++r) {
    int64_t keep = 0;
    for (int64_t w = read_first[r]; w < read_first[r + 1]; ++w) {
      const uint8_t *cor = rows + row_off[w] + ncol[w];
      for (int32_t c = 0; c < ncol[w]; ++c) keep += (cor[c] != 'n');
    }
    out->cols[r] = keep;
    total += 3 * keep;
    out->row_off[r + 1] = total;
  }
  out->rows = (uint8_t *)std::malloc((size_t)total + 16);
  if (!out->rows) { elector_msa_free(out); return ELECTOR_E_NOMEM; }
  for (int64_t r = 0; r < n_reads; ++r) {
    const int64_t nc = out->cols[r];
    uint8_t *d0 = out->rows + out->row_off[r], *d1 = d0 + nc, *d2 = d1 + nc;
    int64_t k = 0;
    for (int64_t w = read_first[r]; w < read_first[r + 1]; ++w) {
      const uint8_t *s0 = rows + row_off[w], *s1 = s0 + ncol[w], *s2 = s1 + ncol[w];
      for (int32_t c = 0; c < ncol[w]; ++c)
        if (s1[c] != 'n') { d0[k] = s0[c]; d1[k] = s1[c]; d2[k] = s2[c]; ++k; }
    }
  }
  out->n_reads = n_reads;
  return ELECTOR_OK;
}

extern "C" void elector_msa_free(elector_msa *m)
{
  if (!m) return;
  std::free(m->rows); std::free(m->row_off); std::free(m->cols);
  std::memset(m, 0, sizeof *m);
}
