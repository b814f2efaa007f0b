"""Known-answer checks of the saturation-pressure seam against PUBLISHED values.

The reference does not contain `wv_saturation`; `oracle/stubs/l0_arithmetic.F90` restates its default, the Goff-Gratch (1946)
formulation over water, and DESIGN.md 2 calls the seam parity-unpinned against a real CAM build.  What CAN be pinned is the
formula itself: Goff & Gratch's saturation pressures over water as tabulated in the Smithsonian Meteorological Tables (6th ed.,
List 1951, Table 94; temperatures on the 1946 scale, 0 degC = 273.16 K, steam point 373.16 K = 1013.246 mb) -- six entries
from -40 to +60 degC and the steam point, each to the digits the table prints.  Checked here for the C restatement (the oracle), for the Fortran stand-in the
reference build links (when oracle/_ref is built), and in tests/test_parity_gpu.py-style for the device function through
`colphys_math_probe_dev` (GPU)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import cases
from cam_nor_physics_amd import synth

# degC on the 1946 scale -> saturation vapour pressure over water, mb (Smithsonian Meteorological Tables, Table 94)
SMT_94 = {-40.0: 0.1891, -20.0: 1.2540, 0.0: 6.1078, 20.0: 23.373, 40.0: 73.777, 60.0: 199.26, 100.0: 1013.246}
T_K = np.array([c + 273.16 for c in SMT_94])
ES_PA = np.array([v * 100.0 for v in SMT_94.values()])
# the table prints four to seven significant digits: an entry is matched when the computed value rounds to it
HALF_ULP_MB = np.array([0.00005, 0.00005, 0.00005, 0.0005, 0.0005, 0.005, 0.0005])


def _check(es_pa, what):
    err_mb = np.abs(es_pa - ES_PA) / 100.0
    assert (err_mb <= HALF_ULP_MB * 1.0001).all(), (what, dict(zip(SMT_94, err_mb)))


def test_c_restatement_reproduces_the_published_goff_gratch_table():
    from c_oracle import COracle
    o = COracle(cases.default_params(32))
    o.lib.oracle_qsat_water.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    es = []
    eps = synth.PHYSCONST["epsilo"]
    for t in T_K:
        e, q = C.c_double(), C.c_double()
        p = 101325.0 if t < 373.0 else 2.0e5
        o.lib.oracle_qsat_water(t, p, C.byref(e), C.byref(q))
        es.append(e.value)
        # qs = eps es / (p - (1 - eps) es): the definition the stand-in documents
        assert abs(q.value - eps * e.value / (p - (1.0 - eps) * e.value)) <= 4e-16 * q.value
    _check(np.array(es), "C restatement")
    # p <= es: saturation mixing ratio 1, es capped at p
    e, q = C.c_double(), C.c_double()
    o.lib.oracle_qsat_water(373.16, 5.0e4, C.byref(e), C.byref(q))
    assert q.value == 1.0 and e.value == 5.0e4


def test_fortran_stand_in_of_the_reference_build_reproduces_the_published_table():
    from ref_oracle import have_ref, ref_lib_path
    if not have_ref(16, 32):
        pytest.skip("oracle/_ref is built only where /root/reference exists")
    lib = C.CDLL(ref_lib_path(16, 32))
    n = len(T_K)
    t = np.ascontiguousarray(T_K); p = np.full(n, 2.0e5); es = np.zeros(n); qs = np.zeros(n)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib.ref_qsat_water(C.c_int(n), dp(t), dp(p), dp(es), dp(qs))
    _check(es, "oracle/stubs/l0_arithmetic.F90 qsat_water")


@pytest.mark.gpu
def test_device_goff_gratch_reproduces_the_published_table():
    from test_parity_gpu import engine
    eng = engine(cases.default_params(32))
    es = eng.math_probe("svp_water", eng.dev(T_K)).cpu().numpy()
    _check(es, "svp_water_gg on the device")
