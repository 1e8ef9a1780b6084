// nc_fuzz.cpp -- TEST INFRASTRUCTURE.  Mutation harness for the product's hand-written HDF5-subset reader
// (our_first_climate_model_amd/csrc/rcm_nc.cpp behind rcm_rtab_load, rcm_host.cpp), built with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all
// by tests/test_sanitize_nc.py (and `make asan-nc`).  It parses an untrusted binary format with hand-rolled
// offsets, so: every seeded byte mutation, splice and truncation of a real NetCDF-4 table must come back as RCM_OK
// or RCM_ERR_FORMAT / RCM_ERR_IO with a message -- never a sanitizer report, a crash, a hang or another code.
//
//   usage: nc_fuzz SEED_FILE.nc NCASES SCRATCH_PATH [MORE_FILES_TO_LOAD_UNMUTATED ...]
// Prints one summary line; exit 0 = every case behaved.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rcm.h"

static uint64_t s_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // splitmix64: seeded, the same cases on every run
  uint64_t z = (s_rng += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static bool write_file(const char *path, const std::vector<unsigned char> &b) {
  FILE *f = std::fopen(path, "wb");
  if (!f) return false;
  const bool ok = std::fwrite(b.data(), 1, b.size(), f) == b.size();
  return (std::fclose(f) == 0) && ok;
}

static int load(const char *path, long long dims[4]) {
  double *blob = nullptr;
  const int rc = rcm_rtab_load(path, dims, &blob);
  if (rc == RCM_OK) {
    // touch every value the reader says it returned: a short allocation shows up here under ASan
    const size_t nt = dims[0], ns = dims[1], nw = dims[2], np = dims[3];
    const size_t n = np + np + nt + ns * np + nw + nw + nt * ns * nw * np;
    volatile double acc = 0;
    for (size_t i = 0; i < n; ++i) acc += blob[i];
    (void)acc;
  }
  rcm_free(blob);
  return rc;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const char *seed = argv[1], *scratch = argv[3];
  const int ncases = std::atoi(argv[2]);
  std::vector<unsigned char> orig;
  {
    FILE *f = std::fopen(seed, "rb");
    if (!f) return 2;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) orig.insert(orig.end(), buf, buf + n);
    std::fclose(f);
  }
  long long dims[4];
  int ok = 0, fmt = 0, io = 0, other = 0;
  for (int i = 4; i <= argc; ++i) {  // the unmutated files first (the seed itself last)
    const char *p = i < argc ? argv[i] : seed;
    const int rc = load(p, dims);
    if (rc != RCM_OK) {
      std::printf("unmutated %s: rc %d (%s)\n", p, rc, rcm_last_error());
      return 1;
    }
  }
  // where the structures are: mutations concentrate on the first 64 KiB (superblock, object headers, heaps) and on
  // every 4-byte signature the reader follows; the rest of the file is the xsec payload
  std::vector<size_t> sigs;
  for (size_t i = 0; i + 4 <= orig.size(); ++i)
    if (!std::memcmp(&orig[i], "OHDR", 4) || !std::memcmp(&orig[i], "FRHP", 4) || !std::memcmp(&orig[i], "FHDB", 4) ||
        !std::memcmp(&orig[i], "BTHD", 4) || !std::memcmp(&orig[i], "OCHK", 4) || !std::memcmp(&orig[i], "FHIB", 4))
      sigs.push_back(i);
  for (int c = 0; c < ncases; ++c) {
    std::vector<unsigned char> b = orig;
    const int kind = (int)(rnd() % 6);
    auto near_structure = [&]() -> size_t {
      if (!sigs.empty() && rnd() % 3) return (sigs[rnd() % sigs.size()] + rnd() % 96) % b.size();
      return rnd() % (b.size() < 65536 ? b.size() : 65536);
    };
    if (kind == 0) {  // flip 1..8 bytes near a structure
      for (int k = 1 + (int)(rnd() % 8); k > 0; --k) b[near_structure()] = (unsigned char)rnd();
    } else if (kind == 1) {  // a length/offset field goes huge, zero or all-ones
      const size_t at = near_structure();
      static const unsigned char pat[3] = {0xff, 0x00, 0x7f};
      const unsigned char v = pat[rnd() % 3];
      for (size_t k = 0; k < 8 && at + k < b.size(); ++k) b[at + k] = v;
    } else if (kind == 2) {  // truncation anywhere
      b.resize(rnd() % b.size());
    } else if (kind == 3) {  // truncation inside the metadata
      b.resize(rnd() % (b.size() < 70000 ? b.size() : 70000));
    } else if (kind == 4) {  // splice: a block of the file copied over another place
      const size_t len = 1 + rnd() % 512, from = rnd() % (b.size() - len), to = near_structure();
      for (size_t k = 0; k < len && to + k < b.size(); ++k) b[to + k] = orig[from + k];
    } else {  // random bytes anywhere (mostly payload: must still load, values differ)
      for (int k = 1 + (int)(rnd() % 4); k > 0; --k) b[rnd() % b.size()] = (unsigned char)rnd();
    }
    if (!write_file(scratch, b)) return 2;
    const int rc = load(scratch, dims);
    if (rc == RCM_OK) ++ok;
    else if (rc == RCM_ERR_FORMAT) ++fmt;
    else if (rc == RCM_ERR_IO) ++io;
    else {
      ++other;
      std::printf("case %d kind %d: unexpected rc %d (%s)\n", c, kind, rc, rcm_last_error());
    }
    if (rc != RCM_OK && !rcm_last_error()[0]) {
      std::printf("case %d: rc %d without a message\n", c, rc);
      ++other;
    }
  }
  std::printf("nc_fuzz: %d cases: %d loaded, %d RCM_ERR_FORMAT, %d RCM_ERR_IO, %d unexpected\n", ncases, ok, fmt, io, other);
  return other ? 1 : 0;
}
