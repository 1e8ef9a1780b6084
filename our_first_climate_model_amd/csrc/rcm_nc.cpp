// rcm_nc.cpp -- the seven datasets of a repwvl Reduced*Forcing.nc table, read straight from the file.
//
// The reference opens these files through netcdf-cxx4 (repwvl_thermal.cpp:113-176, 208): NcFile, getDim,
// getVar.  A NetCDF-4 file IS an HDF5 file; neither library is assumed here (the image has no netcdf and
// no HDF5 development files, the GPU box no h5dump), so this is a reader for exactly the HDF5 subset that
// netcdf-c writes for such a table and nothing more:
//   superblock version 0..3; version-2 object headers with continuation chunks; links of the root group
//   either in its header (compact) or in a fractal heap (dense); datasets that are IEEE little-endian
//   f64, CONTIGUOUS and unfiltered.
// Anything else (chunked or compressed variables, version-1 object headers, other types) is refused with
// RCM_ERR_FORMAT and a message naming the feature; tools/flatten_repwvl.py (h5dump) converts such a file
// to the .rtab blob instead.  Layout knowledge: "HDF5 File Format Specification Version 3.0" (public).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "rcm.h"
#include "rcm_internal.hpp"

namespace {

using rcm::fail;

struct FormatError {
  std::string what;
};

struct H5 {
  std::vector<unsigned char> buf;
  int so = 8, sl = 8;  // size of offsets / lengths
  uint64_t base = 0;

  [[noreturn]] void bad(const std::string &w) const { throw FormatError{w}; }
  void need(uint64_t pos, uint64_t n) const {
    if (pos > buf.size() || n > buf.size() - pos) bad("structure runs past the end of the file");
  }
  uint64_t le(uint64_t pos, int n) const {
    need(pos, n);
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; --i) v = (v << 8) | buf[pos + i];
    return v;
  }
  bool sig(uint64_t pos, const char *s) const {  // (pos may be any 64-bit value from the file: no pos + 4, it can wrap)
    return pos <= buf.size() && buf.size() - pos >= 4 && std::memcmp(&buf[pos], s, 4) == 0;
  }
  bool undef(uint64_t a) const { return so == 8 ? a == ~0ull : a == ((1ull << (8 * so)) - 1); }
};

struct Msg {
  int type;
  uint64_t pos, size;  // body
};

// All messages of a version-2 object header, continuation chunks followed.
std::vector<Msg> object_messages(const H5 &f, uint64_t addr) {
  std::vector<Msg> out;
  addr += f.base;
  if (!f.sig(addr, "OHDR")) {
    if (addr < f.buf.size() && f.buf[addr] == 1) f.bad("version-1 object header (file written in the pre-1.8 format)");
    f.bad("object header signature not found");
  }
  if (f.le(addr + 4, 1) != 2) f.bad("unknown object header version");
  const unsigned flags = (unsigned)f.le(addr + 5, 1);
  uint64_t p = addr + 6;
  if (flags & 0x20) p += 16;  // access, modification, change, birth times
  if (flags & 0x10) p += 4;   // attribute storage phase change values
  const int szf = 1 << (flags & 3);
  const uint64_t chunk0 = f.le(p, szf);
  p += szf;
  const int mh = 4 + ((flags & 0x04) ? 2 : 0);  // message header: type, size(2), flags, [creation order]
  std::vector<std::pair<uint64_t, uint64_t>> chunks{{p, chunk0}};
  for (size_t c = 0; c < chunks.size(); ++c) {
    uint64_t q = chunks[c].first;
    const uint64_t end = q + chunks[c].second;
    f.need(q, chunks[c].second);
    while (q + mh <= end) {
      const int type = (int)f.le(q, 1);
      const uint64_t size = f.le(q + 1, 2);
      q += mh;
      if (q + size > end) f.bad("object header message runs past its chunk");
      if (type == 0x10) {  // continuation: offset, length of an OCHK block (signature .. checksum)
        const uint64_t off = f.le(q, f.so) + f.base, len = f.le(q + f.so, f.sl);
        if (!f.sig(off, "OCHK") || len < 8) f.bad("object header continuation block not found");
        chunks.push_back({off + 4, len - 8});
        if (chunks.size() > 4096) f.bad("object header continuation loop");
      } else if (type != 0) {
        out.push_back({type, q, size});
      }
      q += size;
    }
  }
  return out;
}

// One link message body (also the objects of the link heap): name -> object header address.
// Returns the number of bytes consumed, 0 if this is not a link message.
uint64_t parse_link(const H5 &f, uint64_t p, uint64_t end, std::map<std::string, uint64_t> &links) {
  if (p + 2 > end || f.buf[p] != 1) return 0;
  const unsigned flags = f.buf[p + 1];
  uint64_t q = p + 2;
  int ltype = 0;
  if (flags & 0x08) {
    if (q + 1 > end) return 0;
    ltype = f.buf[q++];
  }
  if (flags & 0x04) q += 8;  // creation order
  if (flags & 0x10) q += 1;  // character set
  const int nlen_sz = 1 << (flags & 3);
  if (q + nlen_sz > end) return 0;
  const uint64_t nlen = f.le(q, nlen_sz);
  q += nlen_sz;
  if (nlen == 0 || nlen > 1024 || q + nlen > end) return 0;
  const std::string name((const char *)&f.buf[q], (size_t)nlen);
  q += nlen;
  if (ltype == 0) {  // hard link
    if (q + f.so > end) return 0;
    links[name] = f.le(q, f.so);
    q += f.so;
  } else if (ltype == 1) {  // soft link: length + path, skipped
    if (q + 2 > end) return 0;
    q += 2 + f.le(q, 2);
  } else {
    return 0;
  }
  return q - p;
}

void heap_direct_block(const H5 &f, uint64_t addr, uint64_t size, int off_bytes, bool checksummed,
                       std::map<std::string, uint64_t> &links) {
  addr += f.base;
  if (!f.sig(addr, "FHDB")) f.bad("fractal heap direct block not found");
  f.need(addr, size);
  uint64_t p = addr + 5 + f.so + off_bytes + (checksummed ? 4 : 0);
  const uint64_t end = addr + size;
  while (p < end) {  // objects are packed from the front; the unused tail is zero
    const uint64_t used = parse_link(f, p, end, links);
    if (!used) break;
    p += used;
  }
}

// Links of a group in dense storage: walk the fractal heap's blocks.
void heap_links(const H5 &f, uint64_t heap_addr, std::map<std::string, uint64_t> &links) {
  uint64_t p = heap_addr + f.base;
  if (!f.sig(p, "FRHP") || f.le(p + 4, 1) != 0) f.bad("fractal heap header not found");
  p += 5;
  p += 2;  // heap ID length
  const uint64_t filter_len = f.le(p, 2);
  p += 2;
  const unsigned flags = (unsigned)f.le(p, 1);
  p += 1;
  if (filter_len) f.bad("filtered fractal heap");
  p += 4;                   // max size of managed objects
  p += f.sl + f.so;         // next huge ID, huge-object B-tree
  p += f.sl + f.so;         // free space, free space manager
  p += 4 * (uint64_t)f.sl;  // managed space, allocated space, iterator offset, number of managed objects
  p += 4 * (uint64_t)f.sl;  // huge size/count, tiny size/count
  const uint64_t width = f.le(p, 2);
  p += 2;
  const uint64_t start = f.le(p, f.sl);
  p += f.sl;
  const uint64_t max_direct = f.le(p, f.sl);
  p += f.sl;
  const int heap_bits = (int)f.le(p, 2);
  p += 2;
  p += 2;  // starting # of rows
  const uint64_t root = f.le(p, f.so);
  p += f.so;
  const uint64_t nrows = f.le(p, 2);
  const int off_bytes = (heap_bits + 7) / 8;
  const bool cks = flags & 0x02;
  if (f.undef(root)) return;  // empty group
  if (!start || !width || (start & (start - 1)) || (max_direct & (max_direct - 1)) || max_direct < start)
    f.bad("fractal heap doubling table");
  if (nrows == 0) {
    heap_direct_block(f, root, start, off_bytes, cks, links);
    return;
  }
  int max_drows = 2;
  for (uint64_t s = start; s < max_direct; s <<= 1) ++max_drows;
  if (nrows > (uint64_t)max_drows) f.bad("fractal heap with nested indirect blocks");  // > width*max_direct bytes of links
  uint64_t q = root + f.base;
  if (!f.sig(q, "FHIB")) f.bad("fractal heap indirect block not found");
  q += 5 + f.so + off_bytes;
  for (uint64_t r = 0; r < nrows; ++r) {
    const uint64_t bsize = r < 2 ? start : start << (r - 1);
    for (uint64_t c = 0; c < width; ++c, q += f.so) {
      const uint64_t a = f.le(q, f.so);
      if (!f.undef(a)) heap_direct_block(f, a, bsize, off_bytes, cks, links);
    }
  }
}

struct Dataset {
  std::vector<uint64_t> dims;
  uint64_t addr = 0, bytes = 0;
};

Dataset dataset_info(const H5 &f, const std::string &name, uint64_t header) {
  Dataset d;
  bool have_space = false, have_type = false, have_layout = false;
  for (const Msg &m : object_messages(f, header)) {
    const uint64_t p = m.pos;
    if (m.type == 0x01) {  // dataspace
      const int ver = (int)f.le(p, 1), rank = (int)f.le(p + 1, 1);
      uint64_t q;
      if (ver == 1) q = p + 8;
      else if (ver == 2) q = p + 4;
      else f.bad(name + ": unknown dataspace version");
      for (int i = 0; i < rank; ++i) d.dims.push_back(f.le(q + (uint64_t)i * f.sl, f.sl));
      have_space = true;
    } else if (m.type == 0x03) {  // datatype: class 1 (floating point), little endian, 8 bytes
      const unsigned cv = (unsigned)f.le(p, 1), bits0 = (unsigned)f.le(p + 1, 1);
      if ((cv & 0x0f) != 1 || (bits0 & 0x41) != 0 || f.le(p + 4, 4) != 8)
        f.bad(name + ": not a little-endian IEEE f64 variable");
      have_type = true;
    } else if (m.type == 0x08) {  // data layout
      const int ver = (int)f.le(p, 1);
      if (ver != 3 && ver != 4) f.bad(name + ": data layout message version " + std::to_string(ver));
      const int cls = (int)f.le(p + 1, 1);
      if (cls != 1) f.bad(name + (cls == 2 ? ": chunked storage" : ": compact storage"));
      d.addr = f.le(p + 2, f.so);
      d.bytes = f.le(p + 2 + f.so, f.sl);
      have_layout = true;
    } else if (m.type == 0x0B) {
      f.bad(name + ": filter pipeline (compressed variable)");
    }
  }
  if (!have_space || !have_type || !have_layout) f.bad(name + ": dataspace, datatype or layout message missing");
  uint64_t n = 1;
  for (uint64_t v : d.dims) n *= v;
  if (f.undef(d.addr) || d.bytes != n * 8) f.bad(name + ": storage not allocated or of unexpected size");
  f.need(d.addr + f.base, d.bytes);
  return d;
}

}  // namespace

namespace rcm {

int nc_table_load(const char *path, long long dims[4], double **blob);

// dims = {n_tpert, n_species, n_wvl, n_p}; blob = p_grid, t_ref, t_pert, vmrs_ref, wvl, weight, xsec (the .rtab order).
int nc_table_load(const char *path, long long dims[4], double **blob) {
  H5 f;
  {
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(RCM_ERR_IO, std::string("rcm_rtab_load: cannot open ") + path);
    std::fseek(fp, 0, SEEK_END);
    const long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (n < 64 || n > (1L << 31)) {
      std::fclose(fp);
      return fail(RCM_ERR_FORMAT, std::string("rcm_rtab_load: implausible file size: ") + path);
    }
    try {
      f.buf.resize((size_t)n);
    } catch (const std::bad_alloc &) {
      std::fclose(fp);
      return fail(RCM_ERR_NOMEM, "rcm_rtab_load: out of memory");
    }
    const bool ok = std::fread(f.buf.data(), 1, (size_t)n, fp) == (size_t)n;
    std::fclose(fp);
    if (!ok) return fail(RCM_ERR_IO, std::string("rcm_rtab_load: read failed: ") + path);
  }
  try {
    const int ver = f.buf[8];
    uint64_t root;
    if (ver == 0 || ver == 1) {
      f.so = f.buf[13], f.sl = f.buf[14];
      uint64_t p = 24 + (ver == 1 ? 4 : 0);
      if ((f.so != 4 && f.so != 8) || (f.sl != 4 && f.sl != 8)) f.bad("size of offsets/lengths");
      f.base = f.le(p, f.so);
      p += 4 * (uint64_t)f.so;       // base, free-space info, end of file, driver info
      root = f.le(p + f.so, f.so);   // root symbol table entry: link name offset, OBJECT HEADER ADDRESS
    } else if (ver == 2 || ver == 3) {
      f.so = f.buf[9], f.sl = f.buf[10];
      if ((f.so != 4 && f.so != 8) || (f.sl != 4 && f.sl != 8)) f.bad("size of offsets/lengths");
      f.base = f.le(12, f.so);
      root = f.le(12 + 3 * (uint64_t)f.so, f.so);
    } else {
      f.bad("superblock version " + std::to_string(ver));
    }
    // links of the root group: in its header (compact) or in a fractal heap (dense)
    std::map<std::string, uint64_t> links;
    for (const Msg &m : object_messages(f, root)) {
      if (m.type == 0x06) {
        parse_link(f, m.pos, m.pos + m.size, links);
      } else if (m.type == 0x02) {  // link info: version, flags, [max creation index], heap address, ...
        if (f.le(m.pos, 1) != 0) f.bad("link info message version");
        const unsigned lf = (unsigned)f.le(m.pos + 1, 1);
        const uint64_t heap = f.le(m.pos + 2 + ((lf & 1) ? 8 : 0), f.so);
        if (!f.undef(heap)) heap_links(f, heap, links);
      } else if (m.type == 0x11) {
        f.bad("old-style group (symbol table): file written in the pre-1.8 format");
      }
    }
    // repwvl_thermal.cpp:113-176: the variables the reference reads
    const char *names[7] = {"p_grid", "t_ref", "t_pert", "vmrs_ref", "ChosenWvls", "ChosenWeights", "xsec"};
    Dataset ds[7];
    for (int i = 0; i < 7; ++i) {
      auto it = links.find(names[i]);
      if (it == links.end()) f.bad(std::string("variable ") + names[i] + " not found");
      ds[i] = dataset_info(f, names[i], it->second);
    }
    if (ds[6].dims.size() != 4) f.bad("xsec is not four-dimensional");
    const uint64_t nt = ds[6].dims[0], ns = ds[6].dims[1], nw = ds[6].dims[2], np = ds[6].dims[3];
    auto is = [&](int i, std::initializer_list<uint64_t> want) { return ds[i].dims == std::vector<uint64_t>(want); };
    if (!is(0, {np}) || !is(1, {np}) || !is(2, {nt}) || !is(3, {ns, np}) || !is(4, {nw}) || !is(5, {nw}))
      f.bad("variable shapes do not fit xsec[n_tpert][n_species][n_wvl][n_p]");
    if (nt < 2 || ns < 1 || nw < 1 || np < 2 || nt > 4096 || ns > 4096 || nw > (1u << 24) || np > 4096)
      f.bad("implausible table dimensions");
    size_t total = 0;
    for (int i = 0; i < 7; ++i) total += (size_t)ds[i].bytes;
    double *b = (double *)std::malloc(total);
    if (!b) return fail(RCM_ERR_NOMEM, "rcm_rtab_load: out of memory");
    size_t at = 0;
    for (int i = 0; i < 7; ++i) {
      std::memcpy((char *)b + at, &f.buf[ds[i].addr + f.base], (size_t)ds[i].bytes);
      at += (size_t)ds[i].bytes;
    }
    dims[0] = (long long)nt, dims[1] = (long long)ns, dims[2] = (long long)nw, dims[3] = (long long)np;
    *blob = b;
    return RCM_OK;
  } catch (const FormatError &e) {
    return fail(RCM_ERR_FORMAT, std::string("rcm_rtab_load: ") + path + ": unsupported NetCDF-4/HDF5 layout (" + e.what +
                                    "); convert it with tools/flatten_repwvl.py");
  } catch (const std::bad_alloc &) {  // (no exception may cross the C ABI)
    return fail(RCM_ERR_NOMEM, "rcm_rtab_load: out of memory");
  }
}

}  // namespace rcm
