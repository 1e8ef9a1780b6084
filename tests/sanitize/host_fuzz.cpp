// host_fuzz.cpp -- TEST INFRASTRUCTURE.  The other file readers of the product's host side under ASan + UBSan
// (g++ -fsanitize=address,undefined -fno-sanitize-recover=all; tests/test_sanitize_nc.py, `make asan-host`):
//   rcm_rtab_load           on mutated .rtab blobs            (rcm_host.cpp)
//   rcm_atm_load            on mutated atmosphere files       (rcm_host.cpp; main.cpp:396-430 semantics)
//   rcm_ascii_file2xy2D     on mutated / random ASCII tables  (rcm_host.cpp; ASCII_file2xy2D semantics)
//   rcm_lbl_cache_read      on mutated LBLTAB1 caches         (rcm_host.cpp)
// Every case must return RCM_OK or a documented negative code -- never a sanitizer report or a crash -- and what a
// successful call returns must be readable to the extent it claims.
//   usage: host_fuzz RTAB ATM NCASES SCRATCH
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rcm.h"

static uint64_t s_rng = 0x243F6A8885A308D3ull;
static uint64_t rnd() {
  uint64_t z = (s_rng += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static std::vector<unsigned char> slurp(const char *p) {
  std::vector<unsigned char> b;
  FILE *f = std::fopen(p, "rb");
  if (!f) return b;
  unsigned char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
  std::fclose(f);
  return b;
}
static bool spit(const char *p, const std::vector<unsigned char> &b) {
  FILE *f = std::fopen(p, "wb");
  if (!f) return false;
  const bool ok = b.empty() || std::fwrite(b.data(), 1, b.size(), f) == b.size();
  return (std::fclose(f) == 0) && ok;
}

static void mutate(std::vector<unsigned char> &b, size_t head, bool text) {
  if (b.empty()) return;
  static const char *tokens[] = {"nan", "inf", "-", "1e999", "1e-999", "%", "#", "\n\n", "\t", "0x1p3", "abc", " ", "1.5.2", ",", "\r\n"};
  switch (rnd() % 6) {
    case 0:
      for (int k = 1 + (int)(rnd() % 8); k > 0; --k) b[rnd() % (rnd() % 2 ? b.size() : (head < b.size() ? head : b.size()))] = (unsigned char)rnd();
      break;
    case 1: {  // a field goes huge / zero / all-ones
      const size_t at = rnd() % (head < b.size() ? head : b.size());
      const unsigned char v = (unsigned char)(rnd() % 3 == 0 ? 0xff : rnd() % 2 ? 0x00 : 0x7f);
      for (size_t k = 0; k < 8 && at + k < b.size(); ++k) b[at + k] = v;
      break;
    }
    case 2: b.resize(rnd() % b.size()); break;
    case 3: b.resize(rnd() % (head < b.size() ? head + 1 : b.size())); break;
    case 4: {  // splice
      const size_t len = 1 + rnd() % 256;
      if (len < b.size()) {
        const size_t from = rnd() % (b.size() - len), to = rnd() % (b.size() - len);
        std::memmove(&b[to], &b[from], len);
      }
      break;
    }
    default:
      if (text) {  // insert a token
        const char *t = tokens[rnd() % (sizeof tokens / sizeof *tokens)];
        const size_t at = rnd() % b.size();
        b.insert(b.begin() + at, t, t + std::strlen(t));
      } else {
        b.insert(b.begin() + rnd() % b.size(), (size_t)(1 + rnd() % 64), (unsigned char)rnd());
      }
  }
}

static int g_bad = 0;
static void check(const char *what, int c, int rc, std::initializer_list<int> allowed) {
  for (int a : allowed)
    if (rc == a) {
      if (rc != 0 && !rcm_last_error()[0] && rc != -5) {  // (ASCII_file2xy2D's own codes carry no message by contract)
        std::printf("%s case %d: rc %d without a message\n", what, c, rc);
        ++g_bad;
      }
      return;
    }
  std::printf("%s case %d: unexpected rc %d (%s)\n", what, c, rc, rcm_last_error());
  ++g_bad;
}

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const std::vector<unsigned char> rtab = slurp(argv[1]), atm = slurp(argv[2]);
  const int ncases = std::atoi(argv[3]);
  const char *scratch = argv[4];
  if (rtab.empty() || atm.empty()) return 2;
  int ok[4] = {0, 0, 0, 0};
  volatile double sink = 0;
  // an LBLTAB1 cache and an ASCII table to start from, written by the product itself
  std::vector<unsigned char> cache, ascii;
  {
    const int n = 37, nl = 20;
    std::vector<double> wvl(n), tab((size_t)n * nl);
    for (int i = 0; i < n; ++i) wvl[i] = 3000.0 + 7.5 * i;
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = 1e-3 * (double)(i % 97);
    const double *tabs[RCM_NSPEC_LBL];
    for (int s = 0; s < RCM_NSPEC_LBL; ++s) tabs[s] = tab.data();
    if (rcm_lbl_cache_write(scratch, n, nl, wvl.data(), tabs) != RCM_OK) return 2;
    cache = slurp(scratch);
    std::string t = "% lbl.xxx.asc\n# comment\n\n";
    for (int r = 0; r < 25; ++r) {
      t += std::to_string(3000.0 + r);
      for (int c = 0; c < 20; ++c) t += (c % 3 ? " " : "\t") + std::to_string(1e-4 * (r + 1) * (c + 1));
      t += r % 6 ? "\n" : "  % tail 1 2\n";
    }
    ascii.assign(t.begin(), t.end());
  }
  for (int c = 0; c < ncases; ++c) {
    {  // .rtab
      std::vector<unsigned char> b = rtab;
      mutate(b, 4096, false);
      if (!spit(scratch, b)) return 2;
      long long dims[4] = {0, 0, 0, 0};
      double *blob = nullptr;
      const int rc = rcm_rtab_load(scratch, dims, &blob);
      check("rtab", c, rc, {RCM_OK, RCM_ERR_FORMAT, RCM_ERR_IO, RCM_ERR_NOMEM});
      if (rc == RCM_OK) {
        ++ok[0];
        const size_t nt = dims[0], ns = dims[1], nw = dims[2], np = dims[3], n = 2 * np + nt + ns * np + 2 * nw + nt * ns * nw * np;
        for (size_t i = 0; i < n; i += 1 + n / 4096) sink += blob[i];
        if (n) sink += blob[n - 1];
      }
      rcm_free(blob);
    }
    {  // .atm
      std::vector<unsigned char> b = atm;
      mutate(b, b.size(), true);
      if (!spit(scratch, b)) return 2;
      const int nmax = 1 + (int)(rnd() % 40);
      std::vector<double> cols((size_t)9 * nmax, 0.0);
      int nlev = -1, ncols = -1;
      const int rc = rcm_atm_load(scratch, nmax, cols.data(), &nlev, &ncols);
      check("atm", c, rc, {RCM_OK, RCM_ERR_FORMAT, RCM_ERR_IO, RCM_ERR_ARG});
      if (rc == RCM_OK) {
        ++ok[1];
        if (nlev < 0 || nlev > nmax) std::printf("atm case %d: nlevel %d of %d\n", c, nlev, nmax), ++g_bad;
      }
    }
    {  // ASCII matrix
      std::vector<unsigned char> b = ascii;
      for (int k = 1 + (int)(rnd() % 3); k > 0; --k) mutate(b, b.size(), true);
      if (!spit(scratch, b)) return 2;
      int nx = -1, ny = -1;
      double *x = nullptr, *y = nullptr;
      const int rc = rcm_ascii_file2xy2D(scratch, &nx, &ny, &x, &y);
      check("ascii", c, rc, {0, -1, -2, -5});
      if (rc == 0) {
        ++ok[2];
        for (int i = 0; i < nx; ++i) sink += x[i];
        for (long i = 0; i < (long)nx * ny; ++i) sink += y[i];
      }
      rcm_free(x), rcm_free(y);
    }
    {  // lbl cache
      std::vector<unsigned char> b = cache;
      mutate(b, 64, false);
      if (!spit(scratch, b)) return 2;
      int nn = -1, nl = -1;
      double *wvl = nullptr, *tabs = nullptr;
      const int rc = rcm_lbl_cache_read(scratch, &nn, &nl, &wvl, &tabs);
      check("cache", c, rc, {RCM_OK, RCM_ERR_FORMAT, RCM_ERR_IO, RCM_ERR_NOMEM, RCM_ERR_ARG});
      if (rc == RCM_OK) {
        ++ok[3];
        for (int i = 0; i < nn; ++i) sink += wvl[i];
        for (long i = 0; i < (long)RCM_NSPEC_LBL * nn * nl; ++i) sink += tabs[i];
      }
      rcm_free(wvl), rcm_free(tabs);
    }
  }
  std::printf("host_fuzz: %d cases each: loaded rtab %d, atm %d, ascii %d, cache %d; %d unexpected\n", ncases, ok[0], ok[1],
              ok[2], ok[3], g_bad);
  return g_bad ? 1 : 0;
}
