// rcm_internal.hpp -- shared between the host-only and the HIP translation units.
#pragma once
#include <string>

namespace rcm {
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);  // records msg, returns code
// rcm_nc.cpp: the repwvl table straight from a NetCDF-4 (= HDF5) file, same outputs as rcm_rtab_load
int nc_table_load(const char *path, long long dims[4], double **blob);
}  // namespace rcm

// Closes a FILE on every way out of an entry point, an exception included.
#include <cstdio>
namespace rcm {
struct File {
  FILE *f;
  explicit File(FILE *fp) : f(fp) {}
  File(const File &) = delete;
  File &operator=(const File &) = delete;
  ~File() {
    if (f) std::fclose(f);
  }
  int close() {  // fclose's own result (a failed flush of a written file must be reported)
    FILE *g = f;
    f = nullptr;
    return g ? std::fclose(g) : 0;
  }
  operator FILE *() const { return f; }
};
}  // namespace rcm

// No exception may cross the C ABI (include/rcm.h): every `int rcm_*` entry point is a function-try-block that ends in
// this handler.  Host-side containers (std::vector, std::string) can throw std::bad_alloc under memory pressure.
#include <exception>
#include <new>
#define RCM_ABI_CATCH                                                                                    \
  catch (const std::bad_alloc &) { return rcm::fail(RCM_ERR_NOMEM, "out of memory (host)"); }            \
  catch (const std::exception &e) { return rcm::fail(RCM_ERR_STATE, std::string("unexpected exception: ") + e.what()); }
