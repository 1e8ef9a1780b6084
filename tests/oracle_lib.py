"""ctypes binding of the CPU oracle (oracle/librcm_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package never does.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librcm_oracle.so")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def load():
    if not os.path.exists(ORACLE_SO):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                       stdout=subprocess.DEVNULL)
    lib = C.CDLL(ORACLE_SO)
    lib.orc_planck_mono.restype = C.c_double
    lib.orc_planck_mono.argtypes = [C.c_double] * 3
    lib.orc_cplkavg.restype = C.c_double
    lib.orc_cplkavg.argtypes = [C.c_double] * 3
    lib.orc_magnus.restype = C.c_double
    lib.orc_magnus.argtypes = [C.c_double]
    lib.orc_calculate_timestep.restype = C.c_double
    lib.orc_calculate_timestep.argtypes = [C.c_double, _dp, C.c_int]
    lib.orc_solar_setup.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                    C.c_double, C.c_double, _dp]
    lib.orc_solar_setup.restype = None
    lib.orc_default_solar.argtypes = [_dp]
    lib.orc_default_solar.restype = None
    lib.orc_lower_pos.argtypes = [_dp, C.c_int, C.c_double]
    lib.orc_atm_load.argtypes = [C.c_char_p, C.c_int, _dp, C.POINTER(C.c_int)]
    lib.orc_column_init.argtypes = [C.c_int, _dp, C.c_double, C.c_double] + [_dp] * 6
    lib.orc_column_init.restype = None
    lib.orc_read_tau.argtypes = [C.c_int] * 4 + [_dp] * 4 + [C.c_int] + [_dp] * 4
    lib.orc_read_tau.restype = None
    lib.orc_radiative_transfer.argtypes = [C.c_int] * 3 + [_dp] * 3 + [C.c_int, _dp, C.c_double,
                                                                      C.c_double] + [_dp] * 3
    lib.orc_radiative_transfer.restype = None
    lib.orc_cloud_into_tau.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.orc_cloud_into_tau.restype = None
    lib.orc_thermodynamics.argtypes = [_dp, C.c_int, C.c_double, _dp, C.c_double, _dp, _dp]
    lib.orc_thermodynamics.restype = None
    lib.orc_convective_adjustment.argtypes = [_dp, _dp, _dp, C.c_int]
    lib.orc_convective_adjustment.restype = None
    lib.orc_water_vapor_feedback.argtypes = [_dp, _dp, _dp, _dp, C.c_int]
    lib.orc_water_vapor_feedback.restype = None
    lib.orc_rtab_load.argtypes = [C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(_dp)]
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_free.restype = None
    lib.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_set_grid.argtypes = [C.c_void_p, _dp]
    lib.orc_set_forcing.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double]
    lib.orc_set_column_forcing.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int), _dp]
    lib.orc_get_residual.argtypes = [C.c_void_p, _dp]
    lib.orc_set_repwvl_tables.argtypes = [C.c_void_p] + [C.c_int] * 4 + [_dp] * 6
    lib.orc_set_lbl_tables.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(_dp), _dp]
    lib.orc_set_columns.argtypes = [C.c_void_p] + [_dp] * 4
    lib.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.orc_step.argtypes = [C.c_void_p, C.c_int]
    lib.orc_get_state.argtypes = [C.c_void_p, _dp, _dp, _fp, _dp]
    lib.orc_get_fluxes.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.orc_get_profile.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _fp]
    lib.orc_get_tau.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.orc_get_gas.argtypes = [C.c_void_p, _dp]
    lib.orc_nspectral.argtypes = [C.c_void_p]
    lib.orc_threads_used.argtypes = [C.c_void_p]
    return lib


def read_rtab(path):
    """Pure-numpy reader of the .rtab blob (tools/flatten_repwvl.py layout)."""
    raw = open(path, "rb").read()
    assert raw[:8] == b"RWVLTAB1", path
    nt, ns, nw, n_p = (int(x) for x in np.frombuffer(raw, "<i8", 4, 8))
    a = np.frombuffer(raw, "<f8", offset=40)
    o = 0

    def take(n, shape=None):
        nonlocal o
        v = np.array(a[o:o + n], dtype=np.float64)
        o += n
        return v if shape is None else v.reshape(shape)

    t = dict(n_tpert=nt, n_species=ns, n_wvl=nw, n_p=n_p)
    t["p_grid"] = take(n_p)
    t["t_ref"] = take(n_p)
    t["t_pert"] = take(nt)
    t["vmrs_ref"] = take(ns * n_p, (ns, n_p))
    t["wvl"] = take(nw)
    t["weight"] = take(nw)
    t["xsec"] = take(nt * ns * nw * n_p, (nt, ns, nw, n_p))
    assert o == a.size
    return t


def atm_load(lib, path, nlevel=21):
    cols = np.zeros((9, nlevel))
    nc = C.c_int(0)
    n = lib.orc_atm_load(path.encode(), nlevel, _p(cols), C.byref(nc))
    if n < 0:
        raise IOError(f"orc_atm_load({path}) = {n}")
    return cols[:, :n].copy(), nc.value


def column_init(lib, atm_cols, co2_factor=1.0, rh_scale=1.0):
    nlevel = atm_cols.shape[1]
    nl = nlevel - 1
    a = np.ascontiguousarray(atm_cols, dtype=np.float64)
    out = dict(plevel=np.zeros(nlevel), player=np.zeros(nl), Tlayer=np.zeros(nl), conv=np.zeros(nl),
               vmr9=np.zeros((9, nl)), rel_hum=np.zeros(nl))
    lib.orc_column_init(nlevel, _p(a), co2_factor, rh_scale, _p(out["plevel"]), _p(out["player"]),
                        _p(out["Tlayer"]), _p(out["conv"]), _p(out["vmr9"]), _p(out["rel_hum"]))
    return out


class OracleEnsemble:
    """Same surface as our_first_climate_model_amd.Ensemble, backed by the CPU oracle."""

    def __init__(self, lib, ncol, nlayer=20, nangle=30):
        self.lib, self.ncol, self.nlayer, self.nangle = lib, ncol, nlayer, nangle
        self.h = C.c_void_p()
        self._ck(lib.orc_create(ncol, nlayer, nangle, C.byref(self.h)))
        self.ngas = None

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"oracle call failed: {rc}")

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_grid(self, plevel):
        p = np.ascontiguousarray(plevel, dtype=np.float64)
        assert p.shape == (self.nlayer + 1,)
        self._ck(self.lib.orc_set_grid(self.h, _p(p)))

    def set_forcing(self, solar_irr, cloud_layer=17, cloud_tau=1.0):
        self._ck(self.lib.orc_set_forcing(self.h, solar_irr, cloud_layer, cloud_tau))

    def set_column_forcing(self, solar_irr=None, cloud_layer=None, cloud_tau=None):
        a = None if solar_irr is None else np.ascontiguousarray(solar_irr, dtype=np.float64)
        b = None if cloud_layer is None else np.ascontiguousarray(cloud_layer, dtype=np.int32)
        c = None if cloud_tau is None else np.ascontiguousarray(cloud_tau, dtype=np.float64)
        self._ck(self.lib.orc_set_column_forcing(self.h, _p(a), None if b is None else b.ctypes.data_as(
            C.POINTER(C.c_int)), _p(c)))

    def get_residual(self):
        r = np.zeros(self.ncol)
        self._ck(self.lib.orc_get_residual(self.h, _p(r)))
        return r

    def run_to_equilibrium(self, tol_K, max_steps, check_every=10):
        # same control flow as rcm_run_to_equilibrium
        done, worst = 0, float("inf")
        while done < max_steps:
            n = min(check_every, max_steps - done)
            self.step(n)
            done += n
            worst = float(self.get_residual().max())
            if worst <= tol_K:
                break
        return done, worst

    def set_repwvl_tables(self, t):
        arrs = [np.ascontiguousarray(t[k], dtype=np.float64)
                for k in ("p_grid", "t_ref", "t_pert", "wvl", "weight", "xsec")]
        self._ck(self.lib.orc_set_repwvl_tables(self.h, t["n_tpert"], t["n_species"], t["n_wvl"],
                                                t["n_p"], *[_p(a) for a in arrs]))
        self.ngas = 9

    def set_lbl_tables(self, wvl, tabs, h2o_ref):
        wvl = np.ascontiguousarray(wvl, dtype=np.float64)
        tabs = [np.ascontiguousarray(x, dtype=np.float64) for x in tabs]
        assert len(tabs) == 5 and all(x.shape == (wvl.size, self.nlayer) for x in tabs)
        h2o_ref = np.ascontiguousarray(h2o_ref, dtype=np.float64)
        arr = (_dp * 5)(*[_p(x) for x in tabs])
        self._ck(self.lib.orc_set_lbl_tables(self.h, wvl.size, _p(wvl), arr, _p(h2o_ref)))
        self.ngas = 5

    def set_columns(self, Tlayer0, T_surface0, gas, rel_hum):
        T = np.ascontiguousarray(Tlayer0, dtype=np.float64)
        Ts = np.ascontiguousarray(T_surface0, dtype=np.float64)
        g = np.ascontiguousarray(gas, dtype=np.float64)
        rh = np.ascontiguousarray(rel_hum, dtype=np.float64)
        assert T.shape == (self.ncol, self.nlayer) and Ts.shape == (self.ncol,)
        assert g.shape == (self.ncol, self.ngas, self.nlayer) and rh.shape == T.shape
        self._ck(self.lib.orc_set_columns(self.h, _p(T), _p(Ts), _p(g), _p(rh)))

    def set_threads(self, n):
        self._ck(self.lib.orc_set_threads(self.h, n))

    def step(self, n=1):
        self._ck(self.lib.orc_step(self.h, n))

    def get_state(self):
        T = np.zeros((self.ncol, self.nlayer))
        Ts = np.zeros(self.ncol)
        tm = np.zeros(self.ncol, dtype=np.float32)
        dt = np.zeros(self.ncol)
        self._ck(self.lib.orc_get_state(self.h, _p(T), _p(Ts), tm.ctypes.data_as(_fp), _p(dt)))
        return dict(Tlayer=T, T_surface=Ts, time_h=tm, timestep=dt)

    def get_fluxes(self):
        Ed = np.zeros((self.ncol, self.nlayer + 1))
        Eu = np.zeros((self.ncol, self.nlayer + 1))
        dE = np.zeros((self.ncol, self.nlayer))
        self._ck(self.lib.orc_get_fluxes(self.h, _p(Ed), _p(Eu), _p(dE)))
        return dict(E_down=Ed, E_up=Eu, dE=dE)

    def get_profile(self, col=0):
        T = np.zeros(self.nlayer)
        th = np.zeros(self.nlayer)
        tm = C.c_float(0)
        self._ck(self.lib.orc_get_profile(self.h, col, _p(T), _p(th), C.byref(tm)))
        return T, th, tm.value

    def nspectral(self):
        return self.lib.orc_nspectral(self.h)

    def get_tau(self, col=0):
        tau = np.zeros((self.nspectral(), self.nlayer))
        self._ck(self.lib.orc_get_tau(self.h, col, _p(tau)))
        return tau

    def get_gas(self):
        g = np.zeros((self.ncol, self.ngas, self.nlayer))
        self._ck(self.lib.orc_get_gas(self.h, _p(g)))
        return g

    def threads_used(self):
        return self.lib.orc_threads_used(self.h)


# ---- the reference's own code, compiled from where it lies into oracle/_ref/ (oracle/build_ref.sh) -------------

REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def load_ref_main():
    """oracle/_ref/libref_main.so = /root/reference/main.cpp:12-21,25-344 (unedited) + oracle/ref_main_shim.inc.
    Returns None when it was not built (no reference tree at build time and no prebuilt copy)."""
    p = os.path.join(REF_DIR, "libref_main.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    d, i = C.c_double, C.c_int
    L.ref_consts.argtypes = [_dp]
    L.ref_cplkavg_layers.argtypes = [d, d, _dp, _dp]
    L.ref_cplkavg_scalar.restype = d
    L.ref_cplkavg_scalar.argtypes = [d, d, d]
    L.ref_emissivity.argtypes = [_dp, d, _dp]
    L.ref_monochromatic_radiative_transfer.argtypes = [_dp, i, _dp, d, i, _dp, _dp, d, _dp, d, _dp, _dp]
    L.ref_radiative_transfer.argtypes = [i, _dp, _dp, _dp, _dp, d, _dp, d, d, _dp, _dp, _dp]
    L.ref_calculate_timestep.restype = d
    L.ref_calculate_timestep.argtypes = [d, _dp]
    L.ref_thermodynamics.restype = d
    L.ref_thermodynamics.argtypes = [_dp, d, _dp, d, _dp]
    L.ref_convective_adjustment.argtypes = [_dp, _dp, _dp]
    L.ref_VMR_level_to_layer.argtypes = [_dp, _dp]
    L.ref_magnus.restype = d
    L.ref_magnus.argtypes = [d]
    L.ref_water_vapor_feedback.argtypes = [_dp, _dp, _dp, _dp]
    L.ref_cloud_into_tau.argtypes = [_dp, i]
    L.ref_solar.argtypes = [_dp]
    for n in ("ref_consts", "ref_cplkavg_layers", "ref_emissivity", "ref_monochromatic_radiative_transfer",
              "ref_radiative_transfer", "ref_convective_adjustment", "ref_VMR_level_to_layer",
              "ref_water_vapor_feedback", "ref_cloud_into_tau", "ref_solar"):
        getattr(L, n).restype = None
    return L


def load_ref_lowerpos():
    p = os.path.join(REF_DIR, "libref_lowerpos.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.ref_lower_pos.argtypes = [_dp, C.c_int, C.c_double]
    L.ref_lower_pos.restype = C.c_int
    return L


def load_ref_readtau():
    """oracle/_ref/libref_readtau.so = every netCDF-free statement of the reference's read_tau
    (repwvl_thermal.cpp:55-111, 124-141, 153-168, 178-206, 210-262, unedited) + oracle/ref_readtau_shim.inc."""
    p = os.path.join(REF_DIR, "libref_readtau.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    i = C.c_int
    L.ref_read_tau.argtypes = [i, i, i, i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, i, _dp, _dp, _dp, i, _dp, _dp, _dp]
    L.ref_read_tau.restype = i
    return L


def ref_read_tau(L, t, plevel_hPa, T, vmr9, prop_at_lev=0):
    """The reference's read_tau on a table dict of read_rtab().  vmr9: [9][nprop] top first (main.cpp's arrays).
    Returns (tau[n_wvl][nlevel-1], wvl, weight) as the reference hands them to main()."""
    pl = np.ascontiguousarray(plevel_hPa, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    v = np.ascontiguousarray(vmr9, dtype=np.float64)
    nlevel = pl.size
    nprop = nlevel - 1 if prop_at_lev == 0 else nlevel
    assert T.shape == (nprop,) and v.shape == (9, nprop)
    xs = np.ascontiguousarray(t["xsec"])
    tau, wvl, weight = np.zeros((t["n_wvl"], nlevel - 1)), np.zeros(t["n_wvl"]), np.zeros(t["n_wvl"])
    n = L.ref_read_tau(t["n_tpert"], t["n_species"], t["n_wvl"], t["n_p"], _p(t["p_grid"]), _p(t["t_ref"]),
                       _p(t["t_pert"]), _p(np.ascontiguousarray(t["vmrs_ref"])), _p(t["wvl"]), _p(t["weight"]), _p(xs),
                       nlevel, _p(pl), _p(T), _p(v), prop_at_lev, _p(tau), _p(wvl), _p(weight))
    assert n == t["n_wvl"]
    return tau, wvl, weight


def ref_mu(nangle=30):
    """main.cpp:355,481-482: dmu = 1/nangle, mu[i] = dmu/2 + dmu*i (the same IEEE operations)."""
    dmu = 1.0 / float(nangle)
    return np.array([dmu / 2.0 + dmu * float(i) for i in range(nangle)]), dmu


def ref_radiative_transfer(L, tau, wvl, weight, Tlayer, T_surface, solar_irr):
    """The reference's radiative_transfer (main.cpp:320-344) on [nwvl][20] optical depths."""
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    wvl = np.ascontiguousarray(wvl, dtype=np.float64)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    T = np.ascontiguousarray(Tlayer, dtype=np.float64)
    assert tau.shape == (wvl.size, 20) and weight.shape == wvl.shape and T.shape == (20,)
    mu, dmu = ref_mu()
    Ed, Eu, dE = np.zeros(21), np.zeros(21), np.zeros(20)
    L.ref_radiative_transfer(wvl.size, _p(tau), _p(wvl), _p(weight), _p(mu), dmu, _p(T), float(T_surface),
                             float(solar_irr), _p(Ed), _p(Eu), _p(dE))
    return Ed, Eu, dE
