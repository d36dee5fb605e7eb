// report_host.cpp -- the host half of call site #2 in native code: per-read ratios, means and the text of
// `per_read_metrics.txt` from the per-piece integer counters the device computed (include/elector_stats.h).
//
// Restates the bookkeeping of the reference's computeMetrics / outputMetrics (elector/computeStats.py:519-675,
// 444-468) and outputReadSizeDistribution (:273-286): which pieces of a read count, in which order the lists grow,
// where an int 0 stands instead of a float, sequential double sums in read order.  Everything here is a function of
// integers gathered on rank 0, so the report does not depend on how the reads were sharded over GPUs.
// The Python mirror (elector_amd/computeStats.py) prints the report from what this returns; tests/agg_ref.py holds
// the same logic as plain Python loops, and the tests compare the two on random counters.
//
// Host-only, no GPU: plain C ABI.
#include "elector_stats.h"

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int kThresh = 5;      // computeStats.py:40

// repr(float) of CPython (the shortest string that reads back as the same double; exponent form when the decimal
// exponent is below -4 or at least 16, ".0" behind an integral mantissa in fixed form)
int py_repr(double x, char *out)
{
  if (x == 0.0) { std::memcpy(out, std::signbit(x) ? "-0.0" : "0.0", std::signbit(x) ? 4 : 3); return std::signbit(x) ? 4 : 3; }
  if (x != x) { std::memcpy(out, "nan", 3); return 3; }
  char sci[40];
  const auto r = std::to_chars(sci, sci + sizeof sci - 1, x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
  const int len = (int)(r.ptr - sci);
  sci[len] = 0;
  int i = 0, o = 0;
  if (sci[0] == '-') { out[o++] = '-'; i = 1; }
  if (len - i >= 3 && sci[i] == 'i') { std::memcpy(out + o, "inf", 3); return o + 3; }
  char digits[24];
  int nd = 0;
  digits[nd++] = sci[i++];
  if (sci[i] == '.') { ++i; while (sci[i] != 'e') digits[nd++] = sci[i++]; }
  ++i;                                                     // 'e'
  const int e10 = std::atoi(sci + i);
  if (e10 < -4 || e10 >= 16) {
    out[o++] = digits[0];
    if (nd > 1) { out[o++] = '.'; std::memcpy(out + o, digits + 1, (size_t)nd - 1); o += nd - 1; }
    out[o++] = 'e';
    out[o++] = e10 < 0 ? '-' : '+';
    const int a = e10 < 0 ? -e10 : e10;
    o += std::snprintf(out + o, 8, "%02d", a);
    return o;
  }
  if (e10 < 0) {                                           // 0.000ddd
    out[o++] = '0'; out[o++] = '.';
    for (int k = 0; k < -e10 - 1; ++k) out[o++] = '0';
    std::memcpy(out + o, digits, (size_t)nd); o += nd;
    return o;
  }
  // e10 + 1 digits in front of the point
  for (int k = 0; k <= e10; ++k) out[o++] = k < nd ? digits[k] : '0';
  out[o++] = '.';
  if (nd > e10 + 1) { std::memcpy(out + o, digits + e10 + 1, (size_t)(nd - e10 - 1)); o += nd - e10 - 1; }
  else out[o++] = '0';
  return o;
}

// round(x, 3) of CPython: the correctly rounded three-decimal string of the exact binary value, read back
double py_round3(double x)
{
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.3f", x);
  return std::strtod(buf, nullptr);
}

template <class T>
T *to_malloc(const std::vector<T> &v)
{
  T *p = static_cast<T *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

}  // namespace

extern "C" int elector_report_aggregate(int64_t n_reads, const int64_t *read_first, int64_t n_pieces, const int64_t *counters,
                                        elector_report *out)
{
  if (!out || n_reads < 0 || n_pieces < 0 || !read_first || (n_pieces > 0 && !counters)) return ELECTOR_E_INVAL;
  std::memset(out, 0, sizeof *out);
  if (read_first[0] != 0 || read_first[n_reads] != n_pieces) return ELECTOR_E_INVAL;
  // The reads are independent except for the six floating-point sums, which the reference accumulates in read order:
  // ranges of reads go to threads (integer sums, the lists and the text per range, the per-read ratios into arrays),
  // and one short pass then adds the ratios up in the reference's order.
  struct Part {
    std::vector<int64_t> missing, len_cor, ext;
    std::string text;
    int64_t nb = 0, n_split = 0, n_ext = 0, n_trim = 0, total_cor = 0, total_unc = 0, len_unc_sum = 0, len_cor_sum = 0, n_gc = 0;
    int64_t isu[3] = {0, 0, 0}, isc[3] = {0, 0, 0};
    int flags = 0;
    bool bad = false;
  };
  struct Ratios { double rec, prec, cbr, ucbr, gcr, gcc; uint8_t emitted, any; };
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(32, (int64_t)std::thread::hardware_concurrency()), n_reads >> 12));
  std::vector<Part> parts((size_t)nt);
  std::vector<Ratios> ratios((size_t)n_reads);
  auto work = [&](int t) {
    Part &P = parts[(size_t)t];
    const int64_t ra = n_reads * t / nt, rb = n_reads * (t + 1) / nt;
    P.text.reserve((size_t)(rb - ra) * 64);
    char num[40];
    auto put = [&](bool is_int0, double v, const char *label) {
      if (is_int0) P.text.push_back('0');
      else P.text.append(num, (size_t)py_repr(v, num));
      P.text.append(label);
    };
    for (int64_t r = ra; r < rb; ++r) {
      Ratios &R = ratios[(size_t)r];
      R = Ratios{0, 0, 0, 0, 0, 0, 0, 0};
      const int64_t p0 = read_first[r], p1 = read_first[r + 1];
      if (p1 < p0) { P.bad = true; return; }
      const bool split = p1 - p0 > 1;
      if (split) ++P.n_split;
      bool extended = false, trimmed = false, any = false, emitted = false;
      int64_t TP = 0, FP = 0, FN = 0, cors = 0, uncs = 0, ucors = 0, uuncs = 0, miss = 0;
      double gcr = 0, gcc = 0;
      for (int64_t p = p0; p < p1; ++p) {
        const int64_t *c = counters + p * ES_NCOUNTERS;
        if (!c[ES_PROCESSED]) continue;
        any = true;
        if (p == p0 || !split) P.len_unc_sum += c[ES_LEN_UNC];
        if (c[ES_EXT_LEFT] >= 0) { extended = true; P.ext.push_back(c[ES_EXT_LEFT]); }
        if (c[ES_EXT_RIGHT] >= 0) { extended = true; P.ext.push_back(c[ES_EXT_RIGHT]); }
        miss = c[ES_MISSING];
        if (miss > kThresh) trimmed = true;
        P.isc[0] += c[ES_INS_C]; P.isc[1] += c[ES_DEL_C]; P.isc[2] += c[ES_SUB_C];
        P.isu[0] += c[ES_INS_U]; P.isu[1] += c[ES_DEL_U]; P.isu[2] += c[ES_SUB_U];
        TP += c[ES_TP]; FP += c[ES_FP]; FN += c[ES_FN];
        cors += c[ES_COR]; uncs += c[ES_UNC]; ucors += c[ES_UCOR]; uuncs += c[ES_UUNC];
        P.len_cor.push_back(c[ES_LEN_COR]);
        P.len_cor_sum += c[ES_LEN_COR];
        if (c[ES_LEN_REF] == 0 || c[ES_LEN_COR] == 0) { P.flags |= 2; continue; }     // the reference divides by zero here
        gcr = py_round3((double)c[ES_GC_REF] * 1.0 / (double)c[ES_LEN_REF]);
        gcc = py_round3((double)c[ES_GC_COR] * 1.0 / (double)c[ES_LEN_COR]);
        if (split && p == p1 - 1) { miss = c[ES_MISSING_LAST]; emitted = true; }
        else if (!split) emitted = true;
      }
      if (!emitted) continue;
      R.emitted = 1;
      if (any) {
        const bool r0 = TP + FN == 0, q0 = TP + FP == 0, c0 = cors + uncs == 0, u0 = ucors + uuncs == 0;
        const double rec = r0 ? 0.0 : (double)TP / (double)(TP + FN), prec = q0 ? 0.0 : (double)TP / (double)(TP + FP);
        const double cbr = c0 ? 0.0 : (double)cors / (double)(cors + uncs), ucbr = u0 ? 0.0 : (double)ucors / (double)(ucors + uuncs);
        if (miss != 0) P.missing.push_back(miss);
        put(r0, rec, " recall\n"); put(q0, prec, " precision\n"); put(c0, cbr, " correct_rate\n");
        R.any = 1; R.rec = rec; R.prec = prec; R.cbr = cbr; R.ucbr = ucbr;
        P.total_cor += cors; P.total_unc += uncs;
      }
      R.gcr = gcr; R.gcc = gcc; ++P.n_gc;
      if (extended) ++P.n_ext;
      if (trimmed && !split) ++P.n_trim;
      ++P.nb;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  std::vector<int64_t> missing, len_cor, ext;
  std::string text;
  int64_t nb = 0, n_split = 0, n_ext = 0, n_trim = 0, total_cor = 0, total_unc = 0, len_unc_sum = 0, len_cor_sum = 0, n_gc = 0;
  int64_t isu[3] = {0, 0, 0}, isc[3] = {0, 0, 0};
  {
    size_t tb = 0, nm = 0, nl = 0, ne = 0;
    for (const Part &P : parts) { if (P.bad) return ELECTOR_E_INVAL; tb += P.text.size(); nm += P.missing.size(); nl += P.len_cor.size(); ne += P.ext.size(); }
    text.reserve(tb); missing.reserve(nm); len_cor.reserve(nl); ext.reserve(ne);
  }
  for (const Part &P : parts) {
    text.append(P.text);
    missing.insert(missing.end(), P.missing.begin(), P.missing.end());
    len_cor.insert(len_cor.end(), P.len_cor.begin(), P.len_cor.end());
    ext.insert(ext.end(), P.ext.begin(), P.ext.end());
    nb += P.nb; n_split += P.n_split; n_ext += P.n_ext; n_trim += P.n_trim; total_cor += P.total_cor; total_unc += P.total_unc;
    len_unc_sum += P.len_unc_sum; len_cor_sum += P.len_cor_sum; n_gc += P.n_gc;
    for (int k = 0; k < 3; ++k) { isu[k] += P.isu[k]; isc[k] += P.isc[k]; }
    out->flags |= P.flags;
  }
  double s_rec = 0, s_prec = 0, s_cbr = 0, s_ucbr = 0, s_gcr = 0, s_gcc = 0;
  for (int64_t r = 0; r < n_reads; ++r) {                      // in read order, as the reference adds them
    const Ratios &R = ratios[(size_t)r];
    if (!R.emitted) continue;
    if (R.any) { s_rec += R.rec; s_prec += R.prec; s_cbr += R.cbr; s_ucbr += R.ucbr; }
    s_gcr += R.gcr; s_gcc += R.gcc;
  }
  out->nb_reads = nb;
  out->throughput = len_cor_sum;
  out->uncor_throughput = len_unc_sum;
  out->count_split = n_split; out->count_trimmed = n_trim; out->count_extended = n_ext;
  for (int k = 0; k < 3; ++k) { out->indelsubs_unc[k] = isu[k]; out->indelsubs_cor[k] = isc[k]; }
  if (n_gc == 0) out->flags |= 1;                          // round(sum([]) / len([]), 3): ZeroDivisionError in the reference
  else { out->gc_ref = py_round3(s_gcr / (double)n_gc); out->gc_cor = py_round3(s_gcc / (double)n_gc); }
  if (nb != 0) {
    out->recall = s_rec * 1.0 / (double)nb; out->precision = s_prec * 1.0 / (double)nb;
    out->cor_bases_rate = s_cbr * 1.0 / (double)nb; out->uncor_cor_bases_rate = s_ucbr * 1.0 / (double)nb;
  } else out->flags |= 4;                                  // the four means are the int 0
  if (total_cor + total_unc == 0) out->flags |= 1;
  else {
    out->error_rate = 1 - ((double)total_cor / (double)(total_cor + total_unc));
    out->uncor_error_rate = 1 - ((double)total_unc / (double)(total_cor + total_unc));
  }
  out->n_missing = (int64_t)missing.size(); out->n_len_cor = (int64_t)len_cor.size(); out->n_extended = (int64_t)ext.size();
  out->missing_size = to_malloc(missing); out->len_corrected = to_malloc(len_cor); out->extended_bases = to_malloc(ext);
  out->per_read_bytes = (int64_t)text.size();
  out->per_read_text = static_cast<char *>(std::malloc(std::max<size_t>(1, text.size())));
  if (!out->missing_size || !out->len_corrected || !out->extended_bases || !out->per_read_text) {
    elector_report_free(out);
    return ELECTOR_E_NOMEM;
  }
  std::memcpy(out->per_read_text, text.data(), text.size());
  return ELECTOR_OK;
}

extern "C" void elector_report_free(elector_report *r)
{
  if (!r) return;
  std::free(r->missing_size); std::free(r->len_corrected); std::free(r->extended_bases); std::free(r->per_read_text);
  r->missing_size = r->len_corrected = r->extended_bases = nullptr;
  r->per_read_text = nullptr;
}

// The first half of outputReadSizeDistribution (computeStats.py:276-278): one line "<n><suffix>\n" per value.
extern "C" int64_t elector_write_count_lines(const int64_t *values, int64_t n, const char *suffix, int fd)
{
  if (n < 0 || (n > 0 && !values) || !suffix || fd < 0) return ELECTOR_E_INVAL;
  const size_t sl = std::strlen(suffix);
  std::string out;
  out.reserve((size_t)n * (sl + 8));
  char num[24];
  for (int64_t i = 0; i < n; ++i) {
    const auto r = std::to_chars(num, num + sizeof num, (long long)values[i]);
    out.append(num, (size_t)(r.ptr - num));
    out.append(suffix, sl);
    out.push_back('\n');
  }
  size_t at = 0;
  while (at < out.size()) {
    const ssize_t w = ::write(fd, out.data() + at, out.size() - at);
    if (w < 0) return ELECTOR_E_IO;
    at += (size_t)w;
  }
  return n;
}

// The second half of outputReadSizeDistribution (computeStats.py:279-285): the corrected FASTA file again, one line
// "<n> sequences\n" per record, n = the length of the record's second line less its last character (the newline --
// or the last base of a file that ends without one); a record whose second line is missing counts 0.  The text goes
// to `fd`.  The file is mapped and cut into ranges for a handful of threads: a first pass counts the line ends of
// every range (which line of a record a range starts in follows from the counts before it), a second pass prints the
// sequence lines that END in the range.  Returns the records seen or a negative code.
extern "C" int64_t elector_read_size_lines(const char *corrected_fasta, int fd)
{
  if (!corrected_fasta || fd < 0) return ELECTOR_E_INVAL;
  const int in = ::open(corrected_fasta, O_RDONLY);
  if (in < 0) return ELECTOR_E_IO;
  struct stat st;
  if (::fstat(in, &st) != 0) { ::close(in); return ELECTOR_E_IO; }
  const int64_t size = (int64_t)st.st_size;
  if (size == 0) { ::close(in); return 0; }
  // Only the positions of the line ends matter.  Every thread reads its range of the file through a small buffer of
  // its own (pread: no page tables to build and tear down for a file of gigabytes, which was most of the time when
  // the file was mapped) and keeps the offsets of the line ends it sees; the lengths follow from the offsets.
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64, (int64_t)std::thread::hardware_concurrency()), size >> 23));
  std::vector<int64_t> cut((size_t)nt + 1);
  std::vector<std::vector<int64_t>> ends((size_t)nt);
  std::vector<int> bad((size_t)nt, 0);
  for (int t = 0; t <= nt; ++t) cut[(size_t)t] = size * t / nt;
  auto run = [&](auto &&fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
  };
  run([&](int t) {
    std::vector<int64_t> &v = ends[(size_t)t];
    std::vector<char> buf((size_t)4 << 20);
    int64_t at = cut[(size_t)t];
    const int64_t e = cut[(size_t)t + 1];
    while (at < e) {
      const ssize_t got = ::pread(in, buf.data(), (size_t)std::min<int64_t>((int64_t)buf.size(), e - at), (off_t)at);
      if (got < 0) { if (errno == EINTR) continue; bad[(size_t)t] = 1; return; }
      if (got == 0) break;                                   // the file shrank under us: what is there is what counts
      const char *p = buf.data(), *pe = p + got;
      while (p < pe) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(pe - p)));
        if (!nl) break;
        v.push_back(at + (int64_t)(nl - buf.data()));
        p = nl + 1;
      }
      at += got;
    }
  });
  ::close(in);
  for (int t = 0; t < nt; ++t) if (bad[(size_t)t]) return ELECTOR_E_IO;
  std::vector<int64_t> before((size_t)nt + 1, 0), prev_end((size_t)nt + 1, -1);   // line ends in front of a range, the last of them
  for (int t = 0; t < nt; ++t) {
    before[(size_t)t + 1] = before[(size_t)t] + (int64_t)ends[(size_t)t].size();
    prev_end[(size_t)t + 1] = ends[(size_t)t].empty() ? prev_end[(size_t)t] : ends[(size_t)t].back();
  }
  std::vector<std::string> parts((size_t)nt);
  run([&](int t) {
    std::string &out = parts[(size_t)t];
    out.reserve(ends[(size_t)t].size() * 10 + 64);
    int64_t line = before[(size_t)t];                      // 0-based index of the line the range's first line end closes
    int64_t ls = prev_end[(size_t)t] + 1;                  // where that line starts
    char num[32];
    for (const int64_t nl : ends[(size_t)t]) {
      if (line & 1) {                                      // a record's second line: its length with the newline, less one
        const auto r = std::to_chars(num, num + sizeof num, (long long)(nl - ls));
        out.append(num, (size_t)(r.ptr - num));
        out.append(" sequences\n");
      }
      ++line;
      ls = nl + 1;
    }
  });
  int64_t records = before[(size_t)nt] / 2;
  // the file's tail behind its last line end: a sequence line without a newline loses its last base instead; a header
  // line without a sequence line behind it counts 0
  {
    const int64_t lines = before[(size_t)nt];
    const int64_t tail = size - (prev_end[(size_t)nt] + 1);
    char num[32];
    std::string &out = parts[(size_t)nt - 1];
    if (tail > 0) {
      if (lines & 1) { out.append(num, (size_t)std::snprintf(num, sizeof num, "%lld sequences\n", (long long)(tail - 1))); ++records; }
      else { out.append("0 sequences\n"); ++records; }   // a header, nothing behind it
    } else if (lines & 1) { out.append("0 sequences\n"); ++records; }   // the last line was a header line
  }
  for (const std::string &o : parts) {
    size_t at = 0;
    while (at < o.size()) {
      const ssize_t w = ::write(fd, o.data() + at, o.size() - at);
      if (w < 0) return ELECTOR_E_IO;
      at += (size_t)w;
    }
  }
  return records;
}
