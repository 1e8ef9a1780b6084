"""CPU: the product's HDF5-subset reader (csrc/rcm_nc.cpp behind rcm_rtab_load) under AddressSanitizer and
UndefinedBehaviorSanitizer.  It parses an untrusted binary format with hand-rolled offsets: tests/sanitize/nc_fuzz.cpp
feeds it the reference's tables and seeded byte mutations, splices and truncations of tests/golden/Reduced10Forcing.nc;
every case must return RCM_OK or RCM_ERR_FORMAT / RCM_ERR_IO with a message -- no sanitizer report, no other code.
Host code only (the GPU box refuses device sanitizers); `make asan-nc` runs the same harness with more cases."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "our_first_climate_model_amd", "csrc")
REF_NC_DIR = "/root/reference/repwvl_V2.01_cpp"  # the two larger tables, when the reference tree is there


def test_nc_reader_mutation_fuzz_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "nc_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", exe,
                        os.path.join(ROOT, "tests", "sanitize", "nc_fuzz.cpp"), os.path.join(CSRC, "rcm_nc.cpp"),
                        os.path.join(CSRC, "rcm_host.cpp")], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find" in r.stderr and "asan" in r.stderr:
        pytest.skip("libasan is not installed here")
    assert r.returncode == 0, r.stderr[-3000:]
    extra = [p for p in (os.path.join(REF_NC_DIR, f"Reduced{n}Forcing.nc") for n in (20, 100)) if os.path.exists(p)]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "Reduced10Forcing.nc"), "4000",
                        str(tmp_path / "mutated.nc")] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("nc_fuzz: 4000 cases:") and last.endswith(" 0 unexpected"), last
    m = re.match(r"nc_fuzz: 4000 cases: (\d+) loaded, (\d+) RCM_ERR_FORMAT, (\d+) RCM_ERR_IO, 0 unexpected$", last)
    assert m, last
    loaded, refused = int(m.group(1)), int(m.group(2))
    assert loaded >= 500 and refused >= 500, last   # both outcomes are really exercised


def test_other_host_readers_mutation_fuzz_under_asan_ubsan(tmp_path):
    """The same for the other file readers of the host side (tests/sanitize/host_fuzz.cpp): rcm_rtab_load on mutated
    .rtab blobs, rcm_atm_load on mutated atmosphere files, rcm_ascii_file2xy2D on mutated ASCII tables (1 MiB line
    buffer semantics of the reference), rcm_lbl_cache_read on mutated caches: RCM_OK or a documented code, never a
    sanitizer report; whatever a successful call returns is read to the extent it claims."""
    exe = str(tmp_path / "host_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", exe,
                        os.path.join(ROOT, "tests", "sanitize", "host_fuzz.cpp"), os.path.join(CSRC, "rcm_nc.cpp"),
                        os.path.join(CSRC, "rcm_host.cpp")], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find" in r.stderr and "asan" in r.stderr:
        pytest.skip("libasan is not installed here")
    assert r.returncode == 0, r.stderr[-3000:]
    golden = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([exe, os.path.join(golden, "Reduced10Forcing.rtab"), os.path.join(golden, "test.atm"), "500",
                        str(tmp_path / "mutated.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    m = re.match(r"host_fuzz: 500 cases each: loaded rtab (\d+), atm (\d+), ascii (\d+), cache (\d+); 0 unexpected$",
                 r.stdout.strip().splitlines()[-1])
    assert m and all(30 <= int(g) <= 470 for g in m.groups()), r.stdout[-500:]   # both outcomes exercised, each reader
