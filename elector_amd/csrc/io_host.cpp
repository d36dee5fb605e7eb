// io_host.cpp -- the file ends of call site #1 in native code: the three FASTA files in, msa.fa records out.
//
//   elector_reads_open / _next / _close   the record loop of masterSplitter (Master_Splitter.cpp:396-446: two
//                                         getline calls per record and file, :414 records whose reference has
//                                         fewer than 3 bases are skipped without counting) cut into processing
//                                         batches by the rule of elector_amd/alignment.py (at least min_records
//                                         records, extended to the end of the last read = run of records with one
//                                         msa.fa header line, Donatello.cpp:61-84 / computeStats.py:45-56)
//   elector_msa_format                    the records Donatello appends to msa.fa (Donatello.cpp:61-93: header
//                                         line and row, three times per piece)
//
// Host-only, no GPU: plain C ABI (include/elector_split.h).
#include "elector_split.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// growable byte buffer for a batch's sequences: anonymous mapping that asks for huge pages (a fresh 250 MB batch
// buffer otherwise costs 60,000 page faults, more than reading the files) and grows in place where it can
static std::atomic<int> g_pin_device{-1};      // elector_reads_set_device

struct BigBuf {
  char *p = nullptr;
  size_t n = 0, cap = 0;
  // pin: the buffer is registered with the HIP runtime (page-locked) whenever it (re)grows, so that the device
  // splitter's copy of a batch's reads is ONE DMA out of this buffer -- staged through pinned memory by host threads it
  // took 5 ms when the host was idle and up to 37 ms beside the parser's, the writers' and the page cache's copies.
  // Without a device (CPU tests) the registration fails and the buffer is plain memory.
  bool pin = false, pinned = false;
  ~BigBuf() { if (p) { unpin(); ::munmap(p, cap); } }
  void unpin() { if (pinned) { (void)hipHostUnregister(p); pinned = false; } }
  bool reserve(size_t want)
  {
    if (want <= cap) return true;
    size_t nc = std::max<size_t>(std::max(want, cap * 2), (size_t)32 << 20);
    if (pin) nc = std::max(nc, want + want / 8);                  // (a registration costs tens of milliseconds: grow rarely)
    nc = (nc + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    unpin();
    void *q = cap ? ::mremap(p, cap, nc, MREMAP_MAYMOVE)
                  : ::mmap(nullptr, nc, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q == MAP_FAILED) return false;
    (void)::madvise(q, nc, MADV_HUGEPAGE);
    p = static_cast<char *>(q);
    cap = nc;
    if (pin) {
      const int dev = g_pin_device.load();
      if (dev >= 0) (void)hipSetDevice(dev);                       // (this thread's current device from here on)
      pinned = hipHostRegister(p, cap, hipHostRegisterDefault) == hipSuccess;
      if (!pinned) (void)hipGetLastError();
    }
    return true;
  }
  bool append(const char *a, size_t len)
  {
    if (n + len > cap && !reserve(n + len)) return false;
    std::memcpy(p + n, a, len);
    n += len;
    return true;
  }
  size_t size() const { return n; }
  void resize(size_t m) { n = m; }              // shrink only
  void resize_to(size_t m) { n = m; }           // after reserve(m): the bytes are written by the caller
  void clear() { n = 0; }
  char *data() { return p; }
};

struct LineFile {
  FILE *f = nullptr;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false, oom = false;
  bool open(const char *path)
  {
    f = std::fopen(path, "rb");
    buf.resize(8u << 20);
    return f != nullptr;
  }
  void close() { if (f) std::fclose(f); f = nullptr; }
  // one line without its '\n' appended to out; false at end of file (python's readline() returning b"")
  template <class Buf>
  bool append_line(Buf &out)
  {
    bool any = false;
    for (;;) {
      if (pos == end) {
        if (eof) return any;
        end = std::fread(buf.data(), 1, buf.size(), f);
        pos = 0;
        if (end == 0) { eof = true; return any; }
      }
      const char *p = buf.data() + pos;
      const char *nl = static_cast<const char *>(std::memchr(p, '\n', end - pos));
      if (nl) {
        if (!out.append(p, (size_t)(nl - p))) { oom = true; return false; }
        pos = (size_t)(nl - buf.data()) + 1;
        return true;
      }
      if (!out.append(p, end - pos)) { oom = true; return false; }
      any = true;
      pos = end;
    }
  }
  bool skip_line()
  {
    bool any = false;
    for (;;) {
      if (pos == end) {
        if (eof) return any;
        end = std::fread(buf.data(), 1, buf.size(), f);
        pos = 0;
        if (end == 0) { eof = true; return any; }
      }
      const char *p = buf.data() + pos;
      const char *nl = static_cast<const char *>(std::memchr(p, '\n', end - pos));
      if (nl) { pos = (size_t)(nl - buf.data()) + 1; return true; }
      any = true;
      pos = end;
    }
  }
};

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// the header line Donatello keeps for a window whose FASTA header line is h: `poa` prints '>name title' with
// "untitled" for an empty title (fasta_format.c:33-37, lpo_format.c:410), Donatello drops the last 11 bytes
// when there are that many and appends a blank (Donatello.cpp:71-73).  elector_amd/alignment.py:_poa_header,
// _donatello_header.
void record_key(const std::string &h, std::string &key)
{
  size_t i = h.empty() ? 0 : 1;
  while (i < h.size() && is_space((unsigned char)h[i])) ++i;
  size_t j = i;
  while (j < h.size() && !is_space((unsigned char)h[j])) ++j;
  size_t r = j;
  while (r < h.size() && is_space((unsigned char)h[r])) ++r;
  key.assign(">");
  key.append(h, i, j - i);
  key.push_back(' ');
  if (r < h.size()) key.append(h, r, std::string::npos); else key.append("untitled");
  if (key.size() >= 11) key.resize(key.size() - 11);
  key.push_back(' ');
}

// key of a header line held in a buffer
void record_key(const char *h, size_t n, std::string &key)
{
  const std::string tmp(h, n);
  record_key(tmp, key);
}

// a file mapped read-only and walked line by line: the lines are handed out as spans of the mapping, nothing is copied
struct MapFile {
  const char *data = nullptr;
  size_t size = 0, pos = 0;
  bool open(const char *path)
  {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); return false; }
    size = (size_t)st.st_size;
    if (size) {
      void *m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { ::close(fd); return false; }
      (void)::madvise(m, size, MADV_SEQUENTIAL);
      data = static_cast<const char *>(m);
    }
    ::close(fd);
    return true;
  }
  void close() { if (data) ::munmap(const_cast<char *>(data), size); data = nullptr; size = pos = released = 0; ahead.clear(); ahead_head = 0; }
  // The bytes before `upto` will not be read again: their pages leave the mapping now, a batch's worth at a time.
  // Left to the final munmap, the page tables of several GB were torn down in one go under the exclusive mmap lock:
  // every other thread of the process that took a page fault meanwhile -- the host half of the last batches -- stood
  // still for some 200 ms.  (MADV_DONTNEED takes the lock shared; the pages stay in the page cache.)
  size_t released = 0;
  void release_to(const char *upto)
  {
    if (!data || upto <= data) return;
    const size_t page = 4096;
    const size_t end = std::min((size_t)(upto - data), size) & ~(page - 1);
    if (end > released) { (void)::madvise(const_cast<char *>(data) + released, end - released, MADV_DONTNEED); released = end; }
  }
  // Lines found ahead of the reader (scan_ahead, one thread per file while a batch is put together): line() hands
  // them out before it looks for more itself.  pos is the end of the last line FOUND, here() the reader's position.
  struct Span { const char *p; size_t n; };
  std::vector<Span> ahead;
  size_t ahead_head = 0;
  void scan_ahead(size_t lines)
  {
    if (ahead_head == ahead.size()) { ahead.clear(); ahead_head = 0; }
    else if (ahead_head >= 4096) {                      // the spans handed out are dropped: the vector stays a batch long
      ahead.erase(ahead.begin(), ahead.begin() + (std::ptrdiff_t)ahead_head);
      ahead_head = 0;
    }
    const size_t have = ahead.size() - ahead_head;
    for (size_t i = have; i < lines && pos < size; ++i) {
      const char *b = data + pos;
      const char *nl = static_cast<const char *>(std::memchr(b, '\n', size - pos));
      if (nl) { ahead.push_back({b, (size_t)(nl - b)}); pos = (size_t)(nl - data) + 1; }
      else { ahead.push_back({b, size - pos}); pos = size; }
    }
  }
  const char *here() const { return ahead_head < ahead.size() ? ahead[ahead_head].p : data + pos; }
  // the next line without its '\n'; false at the end of the file (python's readline() returning b"")
  bool line(const char *&p, size_t &n)
  {
    if (ahead_head < ahead.size()) { p = ahead[ahead_head].p; n = ahead[ahead_head].n; ++ahead_head; return true; }
    if (pos >= size) { p = data + size; n = 0; return false; }
    const char *b = data + pos;
    const char *nl = static_cast<const char *>(std::memchr(b, '\n', size - pos));
    p = b;
    if (nl) { n = (size_t)(nl - b); pos = (size_t)(nl - data) + 1; }
    else { n = size - pos; pos = size; }
    return true;
  }
  bool skip_line() { const char *p; size_t n; return line(p, n); }
};

struct Reader {
  MapFile ref, unc, cor;
  int64_t k = 0;              // kept records read so far
  bool done = false;
  // a kept record: its reference header line and the three sequence lines, as spans of the mappings
  struct Rec { const char *h, *s[3]; size_t hl, sl[3]; };
  // The batch handed out lives in one of kReadSets buffer sets, used in turn: the batch handed out by a call stays
  // valid until kReadSets - 1 more calls have been made, so a caller can read ahead while other threads still work on
  // the batches before (two sets, as up to round 3, tied the parser to the splitters: batch i + 2 could only be read
  // when batch i had been split completely, and a splitter then waited a whole parse for its next batch).
  // A record read ahead of a batch's end opens the next batch.
  struct Set {
    BigBuf seq, hdr;
    std::vector<int64_t> seq_off, hdr_off;
  } set[ELECTOR_READ_SETS];
  int cur = 0;
  bool has_pending = false;
  Rec pending;
  std::thread releaser;       // gives the pages of the batch before back (MapFile::release_to)
  std::vector<Rec> recs;      // the batch under construction
  std::string last_key, key;
  // the next kept record; false at the end of any of the three files
  bool next(Rec &r)
  {
    for (;;) {
      if (!ref.line(r.h, r.hl)) return false;
      if (!ref.line(r.s[0], r.sl[0])) { r.s[0] = ref.data + ref.size; r.sl[0] = 0; }   // python: a header line without a sequence line gives an empty sequence
      if (r.sl[0] <= 2) {                             // Master_Splitter.cpp:414 -- skipped without counting; the other
        if (!unc.skip_line()) return false;           // two files advance by one record all the same
        unc.skip_line();
        if (!cor.skip_line()) return false;
        cor.skip_line();
        continue;
      }
      if (!unc.skip_line()) return false;
      if (!unc.line(r.s[1], r.sl[1])) { r.s[1] = unc.data + unc.size; r.sl[1] = 0; }
      if (!cor.skip_line()) return false;
      if (!cor.line(r.s[2], r.sl[2])) { r.s[2] = cor.data + cor.size; r.sl[2] = 0; }
      return true;
    }
  }
};

}  // namespace

extern "C" void elector_reads_set_device(int device) { g_pin_device.store(device); }

extern "C" int elector_reads_open(const char *reference, const char *uncorrected, const char *corrected, void **handle)
{
  if (!reference || !uncorrected || !corrected || !handle) return ELECTOR_E_INVAL;
  Reader *rd = new (std::nothrow) Reader();
  if (!rd) return ELECTOR_E_NOMEM;
  if (!std::getenv("ELECTOR_NO_PINNED_READS"))
    for (auto &s : rd->set) s.seq.pin = true;
  if (!rd->ref.open(reference) || !rd->unc.open(uncorrected) || !rd->cor.open(corrected)) {
    rd->ref.close(); rd->unc.close(); rd->cor.close();
    delete rd;
    return ELECTOR_E_INVAL;
  }
  *handle = rd;
  return ELECTOR_OK;
}

extern "C" void elector_reads_close(void *handle)
{
  Reader *rd = static_cast<Reader *>(handle);
  if (!rd) return;
  if (rd->releaser.joinable()) rd->releaser.join();
  rd->ref.close(); rd->unc.close(); rd->cor.close();
  delete rd;
}

extern "C" int elector_reads_next(void *handle, int64_t min_records, int64_t start, int64_t stop, elector_reads *out)
{
  Reader *rd = static_cast<Reader *>(handle);
  if (!rd || !out || min_records < 1) return ELECTOR_E_INVAL;
  std::memset(out, 0, sizeof *out);
  static const bool prof = std::getenv("ELECTOR_DEBUG_HOST") != nullptr;
  auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
  const double t0 = now_ms();
  rd->cur = (rd->cur + 1) % ELECTOR_READ_SETS;     // the next buffer set takes the new batch
  Reader::Set &S = rd->set[rd->cur];
  S.seq.clear(); S.hdr.clear(); S.seq_off.assign(1, 0); S.hdr_off.assign(1, 0);
  if (rd->done && !rd->has_pending) return ELECTOR_OK;
  // ---- which records: the sequential part, on spans (no byte is copied here).  Finding the line ends is most of it
  // (a pass over the batch's 250 MB): the three files are scanned side by side, a few records beyond what the batch
  // is likely to take; what is left over stays queued for the next batch ----
  if (!rd->done) {
    const size_t want = 2 * ((size_t)min_records + 64);
    std::thread tu([&] { rd->unc.scan_ahead(want); }), tc([&] { rd->cor.scan_ahead(want); });
    rd->ref.scan_ahead(want);
    tu.join(); tc.join();
  }
  std::vector<Reader::Rec> &recs = rd->recs;
  recs.clear();
  int64_t first = -1;
  bool ended = rd->done;
  for (;;) {
    Reader::Rec r;
    int64_t index;
    if (rd->has_pending) { rd->has_pending = false; r = rd->pending; index = rd->k - 1; }
    else {
      if (ended || !rd->next(r)) { ended = true; break; }
      index = rd->k++;
    }
    if (stop >= 0 && index >= stop) { ended = true; break; }
    if (index < start) continue;
    record_key(r.h, r.hl, rd->key);
    if ((int64_t)recs.size() >= min_records && rd->key != rd->last_key) {       // the batch ends before this record
      rd->has_pending = true;
      rd->pending = r;
      break;
    }
    if (first < 0) first = index;
    recs.push_back(r);
    rd->last_key = rd->key;
  }
  if (ended) rd->done = true;
  const double t1 = now_ms();
  // ---- the bytes: offsets by a prefix sum, the copies on several threads (equal shares of the bytes) ----
  const size_t n = recs.size();
  S.seq_off.resize(3 * n + 1);
  S.hdr_off.resize(n + 1);
  size_t at = 0, hat = 0;
  for (size_t i = 0; i < n; ++i) {
    const Reader::Rec &r = recs[i];
    S.hdr_off[i] = (int64_t)hat; hat += r.hl;
    for (int q = 0; q < 3; ++q) { S.seq_off[3 * i + q] = (int64_t)at; at += r.sl[q]; }
  }
  S.seq_off[3 * n] = (int64_t)at;
  S.hdr_off[n] = (int64_t)hat;
  if (!S.seq.reserve(at + 1) || !S.hdr.reserve(hat + 1)) return ELECTOR_E_NOMEM;
  S.seq.resize_to(at); S.hdr.resize_to(hat);
  char *sd = S.seq.data(), *hd = S.hdr.data();
  const int nt = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, std::thread::hardware_concurrency()), at >> 22));
  auto copy = [&](int t) {
    // records whose sequence bytes start inside this thread's share
    const size_t b0 = at * (size_t)t / (size_t)nt, b1 = at * ((size_t)t + 1) / (size_t)nt;
    size_t i = (size_t)(std::lower_bound(S.seq_off.begin(), S.seq_off.begin() + (ptrdiff_t)(3 * n), (int64_t)b0) - S.seq_off.begin());
    i = (i + 2) / 3;                                  // first record that starts at or behind b0
    for (; i < n && (size_t)S.seq_off[3 * i] < b1; ++i) {
      const Reader::Rec &r = recs[i];
      for (int q = 0; q < 3; ++q) std::memcpy(sd + S.seq_off[3 * i + q], r.s[q], r.sl[q]);
      std::memcpy(hd + S.hdr_off[i], r.h, r.hl);
    }
  };
  if (nt <= 1) copy(0);
  else {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(copy, t);
    copy(0);
    for (auto &x : th) x.join();
  }
  const double t2 = now_ms();
  // everything before the read position has been copied, except the record read ahead (if any)
  {
    const char *ur = rd->has_pending ? rd->pending.h : rd->ref.here(), *uu = rd->has_pending ? rd->pending.s[1] : rd->unc.here(),
               *uc = rd->has_pending ? rd->pending.s[2] : rd->cor.here();
    if (rd->releaser.joinable()) rd->releaser.join();
    rd->releaser = std::thread([rd, ur, uu, uc] { rd->ref.release_to(ur); rd->unc.release_to(uu); rd->cor.release_to(uc); });   // beside the next batch's scan
  }
  if (prof) std::fprintf(stderr, "[elector] reader: lines %.1f ms, copy x%d %.1f ms, pages released %.1f ms (%zu records, %zu bytes)\n", t1 - t0, nt, t2 - t1, now_ms() - t2, n, at);
  out->n = (int64_t)n;
  out->first_index = first < 0 ? 0 : first;
  out->seq = reinterpret_cast<uint8_t *>(S.seq.data());
  out->seq_off = S.seq_off.data();
  out->hdr = reinterpret_cast<uint8_t *>(S.hdr.data());
  out->hdr_off = S.hdr_off.data();
  return ELECTOR_OK;
}

// One pass over the three files for the multi-GPU shard bounds (elector_amd/alignment.py: rank 0 scans, the bounds
// are broadcast): per kept record the three sequence lengths and whether its msa.fa header line differs from the
// previous record's (= a new read).  Nothing but the reference's header lines is copied.
namespace {
struct CountBuf {                      // a sink that only measures
  size_t n = 0;
  bool append(const char *, size_t len) { n += len; return true; }
};
struct StrBuf {
  std::string s;
  bool append(const char *a, size_t len) { s.append(a, len); return true; }
};
}  // namespace

extern "C" int elector_reads_scan(const char *reference, const char *uncorrected, const char *corrected, elector_reads_index *out)
{
  if (!reference || !uncorrected || !corrected || !out) return ELECTOR_E_INVAL;
  std::memset(out, 0, sizeof *out);
  LineFile ref, unc, cor;
  if (!ref.open(reference) || !unc.open(uncorrected) || !cor.open(corrected)) {
    ref.close(); unc.close(); cor.close();
    return ELECTOR_E_INVAL;
  }
  std::vector<int64_t> len;
  std::vector<uint8_t> fresh;
  StrBuf hdr;
  std::string key, last;
  bool first = true;
  for (;;) {
    hdr.s.clear();
    if (!ref.append_line(hdr)) break;
    CountBuf r, u, c;
    ref.append_line(r);
    // the other two files advance by one record whether the reference record is kept or not (Master_Splitter.cpp:407-414)
    if (!unc.skip_line()) break;
    unc.append_line(u);
    if (!cor.skip_line()) break;
    cor.append_line(c);
    if (r.n <= 2) continue;
    record_key(hdr.s, key);
    len.push_back((int64_t)r.n); len.push_back((int64_t)u.n); len.push_back((int64_t)c.n);
    fresh.push_back(first || key != last ? 1 : 0);
    first = false;
    last.swap(key);
  }
  ref.close(); unc.close(); cor.close();
  const size_t n = fresh.size();
  out->n = (int64_t)n;
  out->len = static_cast<int64_t *>(std::malloc(std::max<size_t>(1, 3 * n) * sizeof(int64_t)));
  out->new_read = static_cast<uint8_t *>(std::malloc(std::max<size_t>(1, n)));
  if (!out->len || !out->new_read) { std::free(out->len); std::free(out->new_read); std::memset(out, 0, sizeof *out); return ELECTOR_E_NOMEM; }
  if (n) { std::memcpy(out->len, len.data(), 3 * n * sizeof(int64_t)); std::memcpy(out->new_read, fresh.data(), n); }
  return ELECTOR_OK;
}

extern "C" void elector_reads_index_free(elector_reads_index *ix)
{
  if (!ix) return;
  std::free(ix->len); std::free(ix->new_read);
  std::memset(ix, 0, sizeof *ix);
}

// bytes of the records of the pieces that are not dropped
static int64_t records_bytes(int64_t n, const int64_t *cols, const int64_t *hdr_off, const uint8_t *drop)
{
  int64_t t = 0;
  for (int64_t p = 0; p < n; ++p)
    if (!drop || !drop[p]) t += 3 * ((hdr_off[p + 1] - hdr_off[p]) + 1 + cols[p] + 1);
  return t;
}

extern "C" int64_t elector_msa_format(int64_t n_pieces, const uint8_t *rows, const int64_t *piece_cols, const uint8_t *hdr,
                                       const int64_t *hdr_off, const uint8_t *drop, uint8_t *out, int64_t out_cap, int nthreads)
{
  if (n_pieces < 0 || (n_pieces > 0 && (!rows || !piece_cols || !hdr || !hdr_off))) return ELECTOR_E_INVAL;
  for (int64_t p = 0; p < n_pieces; ++p) if (piece_cols[p] < 0 || hdr_off[p + 1] < hdr_off[p]) return ELECTOR_E_INVAL;
  const int64_t total = records_bytes(n_pieces, piece_cols, hdr_off, drop);
  if (!out) return total;                                   // size query
  if (total > out_cap) return ELECTOR_E_INVAL;
  std::vector<int64_t> in_at((size_t)n_pieces + 1), out_at((size_t)n_pieces + 1);
  in_at[0] = out_at[0] = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    in_at[(size_t)p + 1] = in_at[(size_t)p] + 3 * piece_cols[p];
    const bool keep = !drop || !drop[p];
    out_at[(size_t)p + 1] = out_at[(size_t)p] + (keep ? 3 * ((hdr_off[p + 1] - hdr_off[p]) + 1 + piece_cols[p] + 1) : 0);
  }
  auto work = [&](int64_t p0, int64_t p1) {
    for (int64_t p = p0; p < p1; ++p) {
      if (drop && drop[p]) continue;
      uint8_t *o = out + out_at[(size_t)p];
      const int64_t hl = hdr_off[p + 1] - hdr_off[p], nc = piece_cols[p];
      const uint8_t *h = hdr + hdr_off[p], *r = rows + in_at[(size_t)p];
      for (int row = 0; row < 3; ++row) {
        std::memcpy(o, h, (size_t)hl); o += hl; *o++ = '\n';
        std::memcpy(o, r + row * nc, (size_t)nc); o += nc; *o++ = '\n';
      }
    }
  };
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n_pieces / 64 + 1));
  if (nt == 1) work(0, n_pieces);
  else {
    // equal output bytes per thread
    std::vector<std::thread> th;
    int64_t p0 = 0;
    for (int t = 0; t < nt; ++t) {
      const int64_t want = total * (t + 1) / nt;
      int64_t p1 = (t == nt - 1) ? n_pieces
                                 : (int64_t)(std::upper_bound(out_at.begin(), out_at.end(), want) - out_at.begin()) - 1;
      p1 = std::max(p0, std::min(p1, n_pieces));
      th.emplace_back(work, p0, p1);
      p0 = p1;
    }
    for (auto &x : th) x.join();
  }
  return total;
}
