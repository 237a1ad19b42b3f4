"""The C-ABI library loads and exports every symbol include/dpomp.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "dpomp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpomp_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    names = header_functions()
    for required in ["dpomp_create", "dpomp_destroy", "dpomp_pf_loglik", "dpomp_pop_alloc", "dpomp_pf_partial",
                     "dpomp_pop_gather", "dpomp_pop_download", "dpomp_last_error"]:
        assert required in names


def test_library_exports_every_header_symbol(dp):
    lib = dp.load_library()
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/dpomp.h but not exported"


def test_python_prototypes_cover_the_header(dp):
    from dpomp_amd import _abi
    assert sorted(_abi.exported_names()) == header_functions()


def test_struct_layout_matches_header(dp):
    """sizeof(dpomp_model_desc) as the C compiler sees it == the ctypes mirror."""
    import subprocess, tempfile
    from dpomp_amd import _abi
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "dpomp.h"\nint main(){printf("%zu %zu %zu %zu %zu",sizeof(dpomp_model_desc),sizeof(dpomp_rate_term),sizeof(dpomp_obs_model),offsetof(dpomp_model_desc,rate),offsetof(dpomp_model_desc,obs_time));return 0;}'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(_abi.ModelDescC), C.sizeof(_abi.RateTermC), C.sizeof(_abi.ObsModelC),
                     _abi.ModelDescC.rate.offset, _abi.ModelDescC.obs_time.offset]


def test_no_cpu_fallback(dp, pooley):
    """Without a GPU the product must fail loudly, not compute on the host."""
    lib = dp.load_library()
    if lib.dpomp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(dp.DpompError) as ei:
        dp.ParticleFilter(dp.generate_model("SIS", [100, 1]), pooley)
    assert "no HIP device" in str(ei.value)


def test_bad_descriptors_are_rejected_before_touching_the_device(dp, pooley):
    from dpomp_amd import _abi
    lib = dp.load_library()
    d = dp.ModelDescriptor(dp.generate_model("SIS", [100, 1]), pooley)
    h = C.c_void_p()
    d.c.abi_version = 99
    assert lib.dpomp_create(d.byref(), 0, C.byref(h)) == _abi.EINVAL
    assert b"abi_version" in lib.dpomp_last_global_error()
    d.c.abi_version = _abi.ABI_VERSION
    d.c.n_states = 0
    assert lib.dpomp_create(d.byref(), 0, C.byref(h)) == _abi.EINVAL
    d.c.n_states = 2
    d.c.obs.sigma = 0.0
    assert lib.dpomp_create(d.byref(), 0, C.byref(h)) == _abi.EINVAL
