"""CPU tests of the boundary: libcolphys_hip.so builds, loads and exports every symbol include/colphys.h declares;
without a GPU the engine refuses to compute (there is no CPU path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from cam_nor_physics_amd import abi, synth
from cam_nor_physics_amd.build import LIB, build_hip
from cam_nor_physics_amd.engine import EXPORTED_SYMBOLS, ColphysError, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "colphys.h")


@pytest.fixture(scope="module")
def lib():
    build_hip()
    return load_library()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(colphys_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/colphys.h but not exported by {LIB}"
    assert set(EXPORTED_SYMBOLS) == set(syms)
    assert lib.colphys_abi_version() == 4


def test_gfx950_code_object_is_embedded():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", LIB], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof from a C compiler against the ctypes mirror"""
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "colphys.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(colphys_zm_params), offsetof(colphys_zm_params, limcnv),'
        ' sizeof(colphys_zm_convr_io), offsetof(colphys_zm_convr_io, jt), sizeof(colphys_zm_conv_evap_io), sizeof(colphys_momtran_io),'
        ' sizeof(colphys_aero_tables), offsetof(colphys_aero_tables, nspec), sizeof(colphys_modal_aero_io),'
        ' sizeof(colphys_convtran_io), sizeof(colphys_zm_conv_tend_2_io), sizeof(colphys_zm_conv_tend_io), sizeof(colphys_convect_diag_io), sizeof(colphys_derived_fields_io),'
        ' sizeof(colphys_dry_mass_io), offsetof(colphys_zm_conv_tend_io, cme));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.ZmParams), abi.ZmParams.limcnv.offset, C.sizeof(abi.ZmConvrIO), abi.ZmConvrIO.jt.offset,
            C.sizeof(abi.ZmConvEvapIO), C.sizeof(abi.MomtranIO), C.sizeof(abi.AeroTables), abi.AeroTables.nspec.offset,
            C.sizeof(abi.ModalAeroIO), C.sizeof(abi.ConvtranIO), C.sizeof(abi.ZmConvTend2IO), C.sizeof(abi.ZmConvTendIO), C.sizeof(abi.ConvectDiagIO), C.sizeof(abi.DerivedFieldsIO),
            C.sizeof(abi.DryMassIO), abi.ZmConvTendIO.cme.offset]
    assert got == want


def test_no_cpu_path(lib):
    """On a machine without a GPU init fails loudly, and compute entry points refuse to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, 5)
    rc = lib.colphys_init(0, C.byref(p))
    assert rc == -2, lib.colphys_last_error()          # COLPHYS_ERR_NODEVICE
    assert b"no CPU path" in lib.colphys_last_error()
    rc = lib.colphys_zm_convr_dev(16, 32, 16, C.c_double(900.0), None, None)
    assert rc == -4                                      # COLPHYS_ERR_NOTINIT
    from cam_nor_physics_amd.engine import ColumnPhysics
    with pytest.raises(ColphysError):
        ColumnPhysics(p, 0)


def test_unsupported_configurations_are_rejected(lib):
    """zm_org / zmconv_microp / cam3 are outside the path: init must say so rather than compute something else"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("argument screening happens after the device check")
    for field in ("zm_org", "zmconv_microp", "cam3"):
        p = abi.make_zm_params(synth.PHYSCONST, synth.ZM_NAMELIST, 5)
        setattr(p, field, 1)
        assert lib.colphys_init(0, C.byref(p)) == -5
