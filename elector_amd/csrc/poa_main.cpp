// poa_main.cpp -- a `poa`-compatible executable on top of the C ABI (optional): the one
// command line ELECTOR issues (elector/alignment.py:60),
//
//   poa -pir OUT -preserve_seqorder -corrected_reads_fasta F3 -reference_reads_fasta F1
//       -uncorrected_reads_fasta F2 -preserve_seqorder -threads 1 -pathMatrix MAT
//
// with the live flags of the reference's main() (src/poa-graph/main.c:95,108-111: -pir, the three FASTA flags,
// -pathMatrix; everything else is accepted and ignored).  Record i of the three files is one window triple
// (main.c:241-287: the count comes from the reference file); the output is what write_lpo_bundle_as_fasta prints
// (lpo_format.c:398-426): per triple `>name title` + row for the reference, the corrected and the uncorrected read.
// FASTA reading as the reference's read_fasta (fasta_format.c:10-66): name = first token behind '>', title = the
// rest of the line or "untitled"; sequence = the following lines without white space; '#' and '*' lines are not
// sequence; a record without sequence letters is skipped.  The alignment itself runs on the GPU behind
// elector_poa_batch; there is no CPU path here either (exit code 2 without a gfx950 device).
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "elector_poa.h"

namespace {

struct Rec { std::string name, title, seq; };

bool read_fasta(const char *path, std::vector<Rec> &out)
{
  std::FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  std::string data;
  char buf[1 << 16];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
  std::fclose(f);
  Rec cur;
  bool have = false;
  auto flush = [&]() {
    if (have && !cur.name.empty() && !cur.seq.empty()) out.push_back(cur);
    cur = Rec();
    have = false;
  };
  size_t pos = 0;
  while (pos < data.size()) {
    size_t nl = data.find('\n', pos);
    if (nl == std::string::npos) nl = data.size();
    const char *l = data.data() + pos;
    const size_t n = nl - pos;
    pos = nl + 1;
    if (n == 0) continue;
    if (l[0] == '#') { if (!out.empty() || (have && !cur.seq.empty())) { flush(); break; } continue; }
    if (l[0] == '*') continue;
    if (l[0] == '>') {
      flush();
      size_t i = 1;
      while (i < n && std::isspace((unsigned char)l[i])) ++i;
      size_t j = i;
      while (j < n && !std::isspace((unsigned char)l[j])) ++j;
      cur.name.assign(l + i, j - i);
      while (j < n && std::isspace((unsigned char)l[j])) ++j;
      cur.title = j < n ? std::string(l + j, n - j) : std::string("untitled");
      have = true;
      continue;
    }
    if (have && !cur.name.empty())
      for (size_t i = 0; i < n; ++i) if (!std::isspace((unsigned char)l[i])) cur.seq.push_back(l[i]);
  }
  flush();
  return true;
}

}  // namespace

int main(int argc, char **argv)
{
  const char *out_path = "default_output_msa.fasta", *cor = nullptr, *unc = nullptr, *ref = nullptr, *mat = nullptr;
  for (int i = 1; i < argc; ++i) {
    auto val = [&](const char *flag, const char *&dst) {
      if (std::strcmp(argv[i], flag) == 0 && i + 1 < argc) { dst = argv[++i]; return true; }
      return false;
    };
    if (val("-pir", out_path) || val("-corrected_reads_fasta", cor) || val("-uncorrected_reads_fasta", unc) ||
        val("-reference_reads_fasta", ref) || val("-pathMatrix", mat))
      continue;
    if (std::strcmp(argv[i], "-threads") == 0 && i + 1 < argc) ++i;      // accepted, unused (as in the reference)
  }
  if (!cor || !unc || !ref) {
    std::fprintf(stderr, "usage: %s -pir OUT -corrected_reads_fasta F3 -reference_reads_fasta F1 -uncorrected_reads_fasta F2 [-pathMatrix MAT]\n", argv[0]);
    return 1;
  }
  elector_params p;
  if (mat) {
    if (elector_params_read(mat, &p) != ELECTOR_OK) { std::fprintf(stderr, "%s: cannot read the scoring matrix %s\n", argv[0], mat); return 1; }
  } else elector_params_default(&p);
  std::vector<Rec> R, C, U;
  if (!read_fasta(ref, R) || !read_fasta(cor, C) || !read_fasta(unc, U)) {
    std::fprintf(stderr, "%s: cannot read the sequence files\n", argv[0]);
    return 1;
  }
  const size_t n = R.size();
  if (C.size() < n || U.size() < n) {
    std::fprintf(stderr, "%s: %zu reference records but %zu corrected and %zu uncorrected ones\n", argv[0], n, C.size(), U.size());
    return 1;
  }
  const char *dev_env = std::getenv("ELECTOR_DEVICE") ? std::getenv("ELECTOR_DEVICE") : std::getenv("LOCAL_RANK");
  elector_ctx *ctx = nullptr;
  int rc = elector_ctx_create(dev_env ? std::atoi(dev_env) : 0, &p, &ctx);
  if (rc) { std::fprintf(stderr, "%s: %s\n", argv[0], elector_strerror(rc)); return 2; }
  std::vector<int64_t> off(3 * n + 1, 0);
  std::string bases;
  for (size_t i = 0; i < n; ++i) {
    bases += R[i].seq; off[3 * i + 1] = (int64_t)bases.size();
    bases += C[i].seq; off[3 * i + 2] = (int64_t)bases.size();
    bases += U[i].seq; off[3 * i + 3] = (int64_t)bases.size();
  }
  const int64_t cap = 3 * (int64_t)bases.size() + 16;
  std::vector<uint8_t> rows((size_t)cap);
  std::vector<int64_t> row_off(n + 1, 0);
  std::vector<int32_t> ncol(n, 0), status(n, 0);
  rc = n ? elector_poa_batch(ctx, (int64_t)n, reinterpret_cast<const uint8_t *>(bases.data()), off.data(), rows.data(), cap,
                             row_off.data(), ncol.data(), status.data(), nullptr) : ELECTOR_OK;
  if (rc && rc != ELECTOR_E_WINDOW) {
    std::fprintf(stderr, "%s: %s (%s)\n", argv[0], elector_strerror(rc), elector_ctx_last_error(ctx));
    elector_ctx_destroy(ctx);
    return 2;
  }
  std::FILE *o = std::fopen(out_path, "wb");
  if (!o) { std::fprintf(stderr, "%s: cannot write %s\n", argv[0], out_path); elector_ctx_destroy(ctx); return 1; }
  int failed = 0;
  for (size_t i = 0; i < n; ++i) {
    if (status[i]) { ++failed; continue; }
    const Rec *src[3] = {&R[i], &C[i], &U[i]};
    const uint8_t *r = rows.data() + row_off[i];
    for (int k = 0; k < 3; ++k) {
      std::fprintf(o, ">%s %s\n", src[k]->name.c_str(), src[k]->title.c_str());
      std::fwrite(r + (size_t)k * (size_t)ncol[i], 1, (size_t)ncol[i], o);
      std::fputc('\n', o);
    }
  }
  std::fclose(o);
  std::printf("0 1 2 \n");                               // what buildup_progressive_lpo prints (buildup_lpo.c:545)
  elector_ctx_destroy(ctx);
  if (failed) std::fprintf(stderr, "%s: %d of %zu windows could not be aligned (left out of %s)\n", argv[0], failed, n, out_path);
  return failed ? 3 : 0;
}
