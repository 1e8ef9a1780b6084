// rcm_internal.hpp -- shared between the host-only and the HIP translation units.
#pragma once
#include <string>

namespace rcm {
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);  // records msg, returns code
// rcm_nc.cpp: the repwvl table straight from a NetCDF-4 (= HDF5) file, same outputs as rcm_rtab_load
int nc_table_load(const char *path, long long dims[4], double **blob);
}  // namespace rcm
