"""ctypes binding of librcm_hip.so -- one thin method per C-ABI entry point (include/rcm.h)."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCM_LIB_PATH") or os.path.join(PKG, "librcm_hip.so")  # override: A/B builds only

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)

NSPEC_REPWVL = 9
NSPEC_LBL = 5

_lib = None


class RcmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rcm error {code}: {msg}")
        self.code = code


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _take(ptr, shape):
    """A numpy copy of a malloc'd double block the library handed over (np.ctypeslib.as_array builds a ctypes array
    type per call and needs seconds for 10^7 elements; a buffer view of the address does not)."""
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape)
    buf = (C.c_double * n).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape).copy()


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {a.shape}")
    return a


def _configure(L):
    """Declares the argument types of every entry point of include/rcm.h on a loaded library."""
    vp = C.c_void_p
    L.rcm_abi_version.restype = C.c_int
    L.rcm_last_error.restype = C.c_char_p
    L.rcm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.rcm_destroy.argtypes = [vp]
    L.rcm_set_grid.argtypes = [vp, _dp]
    L.rcm_set_forcing.argtypes = [vp, C.c_double, C.c_int, C.c_double]
    L.rcm_set_column_forcing.argtypes = [vp, _dp, _ip, _dp]
    L.rcm_solar_setup_columns.argtypes = [vp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, _dp]
    L.rcm_run_to_equilibrium.argtypes = [vp, C.c_double, C.c_int, C.c_int, _ip, _dp]
    L.rcm_get_residual.argtypes = [vp, _dp]
    L.rcm_checkpoint_save.argtypes = [vp, C.c_char_p]
    L.rcm_checkpoint_load.argtypes = [vp, C.c_char_p]
    L.rcm_lbl_cache_write.argtypes = [C.c_char_p, C.c_int, C.c_int, _dp, C.POINTER(_dp)]
    L.rcm_lbl_cache_read.argtypes = [C.c_char_p, _ip, _ip, C.POINTER(_dp), C.POINTER(_dp)]
    L.rcm_set_repwvl_tables.argtypes = [vp] + [C.c_int] * 4 + [_dp] * 6
    L.rcm_set_lbl_tables.argtypes = [vp, C.c_int, _dp, C.POINTER(_dp), _dp]
    L.rcm_set_columns.argtypes = [vp] + [_dp] * 4
    L.rcm_set_species_merging.argtypes = [vp, C.c_int]
    L.rcm_step.argtypes = [vp, C.c_int]
    L.rcm_sync.argtypes = [vp]
    L.rcm_get_state.argtypes = [vp, _dp, _dp, _fp, _dp]
    L.rcm_get_fluxes.argtypes = [vp, _dp, _dp, _dp]
    L.rcm_get_profile.argtypes = [vp, C.c_int, _dp, _dp, _fp]
    L.rcm_get_tau.argtypes = [vp, C.c_int, _dp]
    L.rcm_get_tau_production.argtypes = [vp, C.c_int, _dp]
    L.rcm_gather.argtypes = [vp, vp, vp, C.c_int]
    L.rcm_get_gas.argtypes = [vp, _dp]
    L.rcm_nspectral.argtypes = [vp]
    L.rcm_pack_state.argtypes = [vp, vp]
    L.rcm_set_spectral_chunks.argtypes = [vp, C.c_int]
    L.rcm_get_spectral_chunks.argtypes = [vp]
    L.rcm_set_rt_timing.argtypes = [vp, C.c_int]
    L.rcm_get_rt_timing.argtypes = [vp, _dp, _ip]
    L.rcm_measure_hbm_bandwidth.argtypes = [C.c_int, C.c_size_t, C.c_int, _dp, _dp]
    L.rcm_atm_load.argtypes = [C.c_char_p, C.c_int, _dp, _ip, _ip]
    L.rcm_column_init.argtypes = [C.c_int, _dp, C.c_double, C.c_double] + [_dp] * 6
    L.rcm_solar_setup.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                                  C.c_double, _dp]
    L.rcm_default_solar.argtypes = [_dp]
    L.rcm_ascii_file2xy2D.argtypes = [C.c_char_p, _ip, _ip, C.POINTER(_dp), C.POINTER(_dp)]
    L.rcm_rtab_load.argtypes = [C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(_dp)]
    L.rcm_free.argtypes = [vp]
    L.rcm_free.restype = None
    L.rcm_write_banner.argtypes = [vp, C.c_double, C.c_double, _dp]
    L.rcm_write_profile.argtypes = [vp, C.c_int, _dp, _dp, _dp, C.c_float]
    return L




def measure_hbm_bandwidth(device=0, nbytes=1 << 30, reps=5, library=None):
    """rcm_measure_hbm_bandwidth: (copy GB/s counting read + written bytes, read-only GB/s) of two `nbytes` buffers."""
    L = library or lib()
    c, r = C.c_double(0.0), C.c_double(0.0)
    rc = L.rcm_measure_hbm_bandwidth(device, nbytes, reps, C.byref(c), C.byref(r))
    if rc != 0:
        raise RcmError(rc, L.rcm_last_error().decode())
    return c.value, r.value


def load_library(path):
    """dlopens a build of the C-ABI library (RTLD_LOCAL) and declares its entry points.  The product is
    LIB_PATH; tests load the variants build (tests/_variants, -DRCM_WITH_VARIANTS) beside it."""
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 / libhsa-runtime64; if this
    # library were loaded first it would pull /opt/rocm's copy in and the second runtime to initialise
    # finds no device (observed on the MI355X box).  Importing torch first makes the loader resolve our
    # DT_NEEDED libamdhip64.so.7 to the copy torch already mapped.  Without torch (C++ driver) the
    # system runtime is the only one and is used as is.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -m our_first_climate_model_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    return _configure(C.CDLL(path))


def lib():
    """Loads the HIP library; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        _lib = load_library(LIB_PATH)
    return _lib


def _ck(rc, L=None):
    if rc != 0:
        raise RcmError(rc, (L or lib()).rcm_last_error().decode(errors="replace"))


# ------------------------------------------------------------------------------------------------
# host-side pieces of the reference driver
# ------------------------------------------------------------------------------------------------
def atm_load(path, nlevel_max=21):
    """main.cpp:396-430.  Returns (cols[9][nlevel], ncols_found)."""
    cols = np.zeros((9, nlevel_max))
    n, nc = C.c_int(0), C.c_int(0)
    _ck(lib().rcm_atm_load(os.fsencode(path), nlevel_max, _p(cols), C.byref(n), C.byref(nc)))
    return cols[:, :n.value].copy(), nc.value


def column_init(atm_cols, co2_factor=1.0, rh_scale=1.0):
    """main.cpp:439-485 for one column."""
    a = _f64(atm_cols)
    nlevel = a.shape[1]
    nl = nlevel - 1
    out = dict(plevel=np.zeros(nlevel), player=np.zeros(nl), Tlayer=np.zeros(nl), conv=np.zeros(nl),
               vmr9=np.zeros((9, nl)), rel_hum=np.zeros(nl))
    _ck(lib().rcm_column_init(nlevel, _p(a), co2_factor, rh_scale, _p(out["plevel"]), _p(out["player"]),
                              _p(out["Tlayer"]), _p(out["conv"]), _p(out["vmr9"]), _p(out["rel_hum"])))
    return out


def solar_setup(tau_s, mu_s, doublings=20, g_asym=0.85, albedo=0.12, daytime=0.5, E_0=1361.0):
    """main.cpp:214-264 -> dict(r_dir, s_dir, t_dir, r, t, r_total, solar_irr)."""
    out = np.zeros(7)
    _ck(lib().rcm_solar_setup(tau_s, mu_s, doublings, g_asym, albedo, daytime, E_0, _p(out)))
    return dict(zip(("r_dir", "s_dir", "t_dir", "r", "t", "r_total", "solar_irr"), out.tolist()))


def default_solar():
    out = np.zeros(7)
    _ck(lib().rcm_default_solar(_p(out)))
    return dict(zip(("r_dir", "s_dir", "t_dir", "r", "t", "r_total", "solar_irr"), out.tolist()))


def ascii_file2xy2D(path):
    """ASCII_file2xy2D (lbl.arts/ascii.cpp:1631-1691).  Returns (status, x[nx], y[nx][ny])."""
    nx, ny = C.c_int(0), C.c_int(0)
    x, y = _dp(), _dp()
    st = lib().rcm_ascii_file2xy2D(os.fsencode(path), C.byref(nx), C.byref(ny), C.byref(x), C.byref(y))
    if st != 0:
        return st, None, None
    xs = _take(x, (nx.value,))
    ys = _take(y, (nx.value, max(ny.value, 1)))[:, :ny.value]
    lib().rcm_free(x)
    lib().rcm_free(y)
    return 0, xs, ys


def lbl_cache_write(path, wvl, tabs):
    """One-time binary cache of the five lbl tables (rcm_lbl_cache_write)."""
    wvl = _f64(wvl)
    tabs = [_f64(t) for t in tabs]
    arr = (_dp * NSPEC_LBL)(*[_p(t) for t in tabs])
    _ck(lib().rcm_lbl_cache_write(os.fsencode(path), wvl.size, tabs[0].shape[1], _p(wvl), arr))


def lbl_cache_read(path):
    """Returns (wvl[n], [5 tables of shape (n, nlayer)])."""
    n, nl = C.c_int(0), C.c_int(0)
    w, t = _dp(), _dp()
    _ck(lib().rcm_lbl_cache_read(os.fsencode(path), C.byref(n), C.byref(nl), C.byref(w), C.byref(t)))
    wvl = _take(w, (n.value,))
    tabs = _take(t, (NSPEC_LBL, n.value, nl.value))
    lib().rcm_free(w)
    lib().rcm_free(t)
    return wvl, [tabs[s] for s in range(NSPEC_LBL)]


def read_rtab(path):
    """The datasets of a flattened Reduced*Forcing.nc (tools/flatten_repwvl.py) via rcm_rtab_load."""
    dims = (C.c_longlong * 4)()
    blob = _dp()
    _ck(lib().rcm_rtab_load(os.fsencode(path), dims, C.byref(blob)))
    nt, ns, nw, n_p = (int(v) for v in dims)
    n = n_p + n_p + nt + ns * n_p + nw + nw + nt * ns * nw * n_p
    a = _take(blob, (n,))
    lib().rcm_free(blob)
    o = 0

    def take(k, shape=None):
        nonlocal o
        v = a[o:o + k].copy()
        o += k
        return v if shape is None else v.reshape(shape)

    t = dict(n_tpert=nt, n_species=ns, n_wvl=nw, n_p=n_p)
    t["p_grid"], t["t_ref"], t["t_pert"] = take(n_p), take(n_p), take(nt)
    t["vmrs_ref"] = take(ns * n_p, (ns, n_p))
    t["wvl"], t["weight"] = take(nw), take(nw)
    t["xsec"] = take(nt * ns * nw * n_p, (nt, ns, nw, n_p))
    return t


# ------------------------------------------------------------------------------------------------
# the ensemble handle
# ------------------------------------------------------------------------------------------------
class Ensemble:
    """An ensemble of independent columns on one MI355X (rcm_ensemble, include/rcm.h)."""

    def __init__(self, ncol, nlayer=20, nangle=30, device=0, stream=None, library=None):
        """`library`: a build from load_library() other than the product (tests: the variants build)."""
        self.ncol, self.nlayer, self.nangle, self.device = ncol, nlayer, nangle, device
        self.ngas = None
        self._L = library if library is not None else lib()
        self.h = C.c_void_p()
        self._ck(self._L.rcm_create(ncol, nlayer, nangle, device, C.c_void_p(stream) if stream else None,
                                    C.byref(self.h)))

    def _ck(self, rc):
        _ck(rc, self._L)

    def close(self):
        if getattr(self, "h", None):
            self._L.rcm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_grid(self, plevel):
        self._ck(self._L.rcm_set_grid(self.h, _p(_f64(plevel, (self.nlayer + 1,), "plevel"))))

    def set_forcing(self, solar_irr, cloud_layer=17, cloud_tau=1.0):
        self._ck(self._L.rcm_set_forcing(self.h, solar_irr, cloud_layer, cloud_tau))

    def set_column_forcing(self, solar_irr=None, cloud_layer=None, cloud_tau=None):
        a = None if solar_irr is None else _f64(solar_irr, (self.ncol,), "solar_irr")
        b = None if cloud_layer is None else np.ascontiguousarray(cloud_layer, dtype=np.int32)
        c = None if cloud_tau is None else _f64(cloud_tau, (self.ncol,), "cloud_tau")
        if b is not None and b.shape != (self.ncol,):
            raise ValueError("cloud_layer: expected shape (ncol,)")
        self._ck(self._L.rcm_set_column_forcing(self.h, _p(a), None if b is None else b.ctypes.data_as(_ip), _p(c)))

    def solar_setup_columns(self, tau_s, mu_s, albedo, doublings=20, g_asym=0.85, daytime=0.5, E_0=1361.0):
        """main.cpp:214-264 on the device per column; returns [ncol][7] (r_dir, s_dir, t_dir, r, t, r_total,
        solar_irr) and installs solar_irr and cloud_tau = tau_s/2 (main.cpp:267) per column."""
        out = np.zeros((self.ncol, 7))
        self._ck(self._L.rcm_solar_setup_columns(self.h, _p(_f64(tau_s, (self.ncol,))), _p(_f64(mu_s, (self.ncol,))),
                                          _p(_f64(albedo, (self.ncol,))), doublings, g_asym, daytime, E_0, _p(out)))
        return out

    def get_residual(self):
        r = np.zeros(self.ncol)
        self._ck(self._L.rcm_get_residual(self.h, _p(r)))
        return r

    def run_to_equilibrium(self, tol_K, max_steps, check_every=10):
        done, worst = C.c_int(0), C.c_double(0)
        self._ck(self._L.rcm_run_to_equilibrium(self.h, tol_K, max_steps, check_every, C.byref(done), C.byref(worst)))
        return done.value, worst.value

    def checkpoint_save(self, path):
        self._ck(self._L.rcm_checkpoint_save(self.h, os.fsencode(path)))

    def checkpoint_load(self, path):
        self._ck(self._L.rcm_checkpoint_load(self.h, os.fsencode(path)))

    def set_repwvl_tables(self, t):
        arrs = [_f64(t[k]) for k in ("p_grid", "t_ref", "t_pert", "wvl", "weight", "xsec")]
        self._ck(self._L.rcm_set_repwvl_tables(self.h, t["n_tpert"], t["n_species"], t["n_wvl"], t["n_p"],
                                        *[_p(a) for a in arrs]))
        self.ngas = NSPEC_REPWVL

    def set_lbl_tables(self, wvl, tabs, h2o_ref):
        wvl = _f64(wvl)
        tabs = [_f64(x, (wvl.size, self.nlayer), "lbl table") for x in tabs]
        if len(tabs) != NSPEC_LBL:
            raise ValueError("need 5 species tables (CO2, H2O, O3, CH4, N2O)")
        h2o_ref = _f64(h2o_ref, (self.nlayer,), "h2o_ref")
        arr = (_dp * NSPEC_LBL)(*[_p(x) for x in tabs])
        self._ck(self._L.rcm_set_lbl_tables(self.h, wvl.size, _p(wvl), arr, _p(h2o_ref)))
        self.ngas = NSPEC_LBL

    def set_columns(self, Tlayer0, T_surface0, gas, rel_hum):
        if self.ngas is None:
            raise RcmError(-2, "set the spectral tables before the columns")
        T = _f64(Tlayer0, (self.ncol, self.nlayer), "Tlayer0")
        Ts = _f64(T_surface0, (self.ncol,), "T_surface0")
        g = _f64(gas, (self.ncol, self.ngas, self.nlayer), "gas")
        rh = _f64(rel_hum, (self.ncol, self.nlayer), "rel_hum")
        self._ck(self._L.rcm_set_columns(self.h, _p(T), _p(Ts), _p(g), _p(rh)))

    def step(self, n=1):
        self._ck(self._L.rcm_step(self.h, n))

    def sync(self):
        self._ck(self._L.rcm_sync(self.h))

    def get_state(self):
        T = np.zeros((self.ncol, self.nlayer))
        Ts = np.zeros(self.ncol)
        tm = np.zeros(self.ncol, dtype=np.float32)
        dt = np.zeros(self.ncol)
        self._ck(self._L.rcm_get_state(self.h, _p(T), _p(Ts), tm.ctypes.data_as(_fp), _p(dt)))
        return dict(Tlayer=T, T_surface=Ts, time_h=tm, timestep=dt)

    def get_fluxes(self):
        Ed = np.zeros((self.ncol, self.nlayer + 1))
        Eu = np.zeros((self.ncol, self.nlayer + 1))
        dE = np.zeros((self.ncol, self.nlayer))
        self._ck(self._L.rcm_get_fluxes(self.h, _p(Ed), _p(Eu), _p(dE)))
        return dict(E_down=Ed, E_up=Eu, dE=dE)

    def get_profile(self, col=0):
        T = np.zeros(self.nlayer)
        th = np.zeros(self.nlayer)
        tm = C.c_float(0)
        self._ck(self._L.rcm_get_profile(self.h, col, _p(T), _p(th), C.byref(tm)))
        return T, th, tm.value

    def nspectral(self):
        n = self._L.rcm_nspectral(self.h)
        if n < 0:
            self._ck(n)
        return n

    def set_species_merging(self, on):
        """repwvl: False keeps every species' own table rows (optical depths bit-identical under any sharding)."""
        self._ck(self._L.rcm_set_species_merging(self.h, 1 if on else 0))

    def get_tau(self, col=0):
        tau = np.zeros((self.nspectral(), self.nlayer))
        self._ck(self._L.rcm_get_tau(self.h, col, _p(tau)))
        return tau

    def get_tau_production(self, col=0):
        """The optical depths as the production kernels form them (rcm_get_tau_production)."""
        tau = np.zeros((self.nspectral(), self.nlayer))
        self._ck(self._L.rcm_get_tau_production(self.h, col, _p(tau)))
        return tau

    def get_gas(self):
        g = np.zeros((self.ncol, self.ngas, self.nlayer))
        self._ck(self._L.rcm_get_gas(self.h, _p(g)))
        return g

    def pack_state(self, dev_ptr):
        """Writes [ncol][nlayer+1] (Tlayer, T_surface) to a DEVICE pointer (e.g. tensor.data_ptr())."""
        self._ck(self._L.rcm_pack_state(self.h, C.c_void_p(dev_ptr)))

    def gather(self, nccl_comm, dev_recv_ptr, ncol_pad):
        """rcm_gather: pack + ONE ncclAllGather on the handle's stream; `nccl_comm` = the caller's ncclComm_t as
        an integer, `dev_recv_ptr` = DEVICE pointer to [ranks][ncol_pad][nlayer+1] doubles.  Asynchronous."""
        self._ck(self._L.rcm_gather(self.h, C.c_void_p(nccl_comm), C.c_void_p(dev_recv_ptr), ncol_pad))

    def set_spectral_chunks(self, n):
        self._ck(self._L.rcm_set_spectral_chunks(self.h, n))

    def get_spectral_chunks(self):
        n = self._L.rcm_get_spectral_chunks(self.h)
        if n < 0:
            self._ck(n)
        return n

    def set_rt_timing(self, on=True):
        self._ck(self._L.rcm_set_rt_timing(self.h, 1 if on else 0))

    def get_rt_timing(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._ck(self._L.rcm_get_rt_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def configure(e, inp):
    """Applies one input description to an ensemble handle in the order the ABI requires (grid, forcing,
    spectral tables, columns).  `inp`: mode 'repwvl' (key 'table' = read_rtab dict) or 'lbl' (keys 'wvl',
    'tabs'), 'cols' = build_columns(...), 'solar', optional 'cloud_layer' / 'cloud_tau'.  `e` is an Ensemble
    or anything with the same four setters."""
    c = inp["cols"]
    e.set_grid(c["plevel"])
    e.set_forcing(inp["solar"], inp.get("cloud_layer", 17), inp.get("cloud_tau", 1.0))
    if inp["mode"] == "repwvl":
        e.set_repwvl_tables(inp["table"])
    else:
        e.set_lbl_tables(inp["wvl"], inp["tabs"], c["h2o_ref"])
    e.set_columns(c["Tlayer0"], c["T_surface0"], c["gas"], c["rel_hum"])
    return e
