"""CPU tests: both C-ABI libraries load and export every symbol that
include/*.h declares; no compute call is made without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*?$", "", src, flags=re.M)          # drop preprocessor lines
    names = re.findall(r"\b((?:perseus|pddc)_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_ddc_header_symbols_exported(pkg):
    names = declared_functions("perseus_ddc.h")
    assert len(names) >= 25
    exp = exported(pkg.DDC_LIB)
    missing = [n for n in names if n not in exp]
    assert not missing, missing


def test_sdr_header_symbols_exported(pkg):
    L = pkg.sdr_lib()
    names = declared_functions("perseus-sdr.h") + declared_functions("perseus-amd-ext.h")
    assert len([n for n in names if not n.startswith("perseus_amd")]) == 20     # the reference's 20 functions
    exp = exported(pkg.SDR_LIB)
    missing = [n for n in names if n not in exp]
    assert not missing, missing
    for g in ("perseus_dbg_level", "perseus_error_str", "perseus_error"):        # perseus-sdr.h:362-364
        assert g in exp
    assert C.c_int.in_dll(L, "perseus_error").value == 0 or True


def test_error_codes_match_reference_values():
    src = open(os.path.join(ROOT, "include", "perseus-sdr.h")).read()
    codes = dict(re.findall(r"#define (PERSEUS_[A-Z]+)\s+(-?\d+)\s*$", src, flags=re.M))
    expect = {"PERSEUS_NOERROR": 0, "PERSEUS_INVALIDDEV": -1, "PERSEUS_NULLDESCR": -2,
              "PERSEUS_ALREADYOPEN": -3, "PERSEUS_DEVNOTOPEN": -5, "PERSEUS_FNNOTAVAIL": -9,
              "PERSEUS_FWNOTLOADED": -16, "PERSEUS_FPGANOTCFGD": -18, "PERSEUS_ASYNCSTARTED": -19,
              "PERSEUS_ERRPARAM": -22, "PERSEUS_BUFFERSIZE": -24, "PERSEUS_ATTERROR": -25,
              "PERSEUS_SNNOTAVAILABLE": -26}
    for k, v in expect.items():
        assert int(codes[k]) == v


def test_no_gpu_means_loud_failure(pkg):
    """Without a device every compute entry point refuses; nothing falls back to a CPU path."""
    L = pkg.ddc_lib()
    assert L.pddc_version() >= 100
    assert L.pddc_nco_freg(7.1e6, 80e6) == 381178347
    if L.pddc_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.PddcError) as e:
        pkg.Pipeline([(8, np.ones(16, np.float32))])
    assert e.value.code == pkg.PDDC_ENODEV
    buf = (C.c_uint8 * 64)()
    assert L.pddc_unpack24_f32(buf, 8, buf, None) == pkg.PDDC_ENODEV
    assert b"no CPU fallback" in L.pddc_last_error()


def test_placement_helpers_check_their_arguments_before_touching_a_device(pkg):
    """pddc_pipeline_arena_place / pddc_pipeline_set_workspace / pddc_malloc_apart: argument errors are reported as such
    (also on a box without a GPU); with good arguments and no GPU they refuse like every compute entry point.  The
    model-stream searches of rounds 2-4 (pddc_arena_search, pddc_arena_place) are no longer exported."""
    L = pkg.ddc_lib()
    sz = C.c_size_t
    o = sz()
    fake = C.c_void_p(1 << 20)                      # never dereferenced: the checks come first
    G = 1 << 30
    assert L.pddc_pipeline_arena_place(None, fake, 64 * G, 8 * G, 1 << 20, 2 * G, C.byref(o), None, None, None, None) == pkg.PDDC_EINVAL
    assert not hasattr(L, "pddc_arena_search") and not hasattr(L, "pddc_arena_place")
    assert L.pddc_pipeline_workspace_size(None, 1 << 20) == 0
    assert L.pddc_pipeline_set_workspace(None, None, 0, 0) == pkg.PDDC_EINVAL
    p = C.c_void_p()
    assert L.pddc_malloc_apart(C.byref(p), 0, None, 0, 4, None, None) == pkg.PDDC_EINVAL
    if L.pddc_device_count() == 0:
        assert L.pddc_malloc_apart(C.byref(p), 1 << 20, None, 0, 4, None, None) == pkg.PDDC_ENODEV


def test_the_int8_kernel_object_passes_the_hazard_check(pkg):
    """csrc/check_hazard_pads.py on the object the library was linked from: no packed fp32 in any k_fir_i8x kernel whose
    loader waves finish tiles beside matrix waves, every result store of a finishing wave padded (the build runs the same
    check and deletes an object that fails it; this test makes the CPU suite say so too)."""
    import subprocess
    obj = os.path.join(pkg.CSRC, "ddc_fir_i8.o")
    if not os.path.exists(obj):
        pkg.build()
    out = subprocess.run([sys.executable, os.path.join(pkg.CSRC, "check_hazard_pads.py"), obj], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "no packed fp32" in out.stdout
    mk = open(os.path.join(pkg.CSRC, "Makefile")).read()
    assert "check_hazard_pads.py $@" in mk and "-fno-slp-vectorize -c ddc_fir_i8.hip" in mk


def test_product_does_not_reference_oracle():
    """The product tree must not import, link or open anything under oracle/."""
    pk = os.path.join(ROOT, "libperseus-sdr_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".c", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in txt and "perseus_oracle" not in txt and "from oracle" not in txt, f
    ldd = subprocess.check_output(["ldd", os.path.join(pk, "libperseus_ddc.so")], text=True)
    assert "oracle" not in ldd


def test_no_getenv_on_the_data_path():
    """Kernel selection is API state (pddc_pipeline_set_option, pddc_set_tunable): the library looks at the environment when
    a pipeline is created and once for the process-wide launcher knobs -- never inside process(), a push or a gang round.
    Source-level check: every getenv in the kernels' host code sits in one of those two places (the GPU suite counts
    the calls of a running stream: tests/test_gpu_api.py::test_a_running_stream_never_calls_getenv)."""
    import re
    csrc = os.path.join(ROOT, "libperseus-sdr_amd", "csrc")
    allowed = {"ddc_pipeline.cpp": ["pddc_pipeline_create"], "ddc_kernels.hip": ["tunables"], "ddc_fir_i8.hip": [],
               "fir8_block.inc": [], "ddc_multi.cpp": None}          # (None: not on the DSP data path -- communicator set-up)
    for name, funcs in allowed.items():
        src = open(os.path.join(csrc, name)).read()
        if funcs is None:
            continue
        # walk the file; remember the last function header seen at column 0
        current = None
        for line in src.split("\n"):
            m = re.match(r"^[A-Za-z_].*?\b([A-Za-z_0-9]+)\s*\([^;]*$", line)
            if m and not line.startswith(("static constexpr", "typedef", "#")):
                current = m.group(1)
            if "getenv(" in line and not line.lstrip().startswith(("*", "/*", "//")):
                assert current in funcs, (name, current, line.strip())
    for forbidden in ("PDDC_ABLATE_", "PDDC_CLOCK_PROBE", "PDDC_EXPERIMENT_NT128", "I8X_PROBE", "I8X_ABL_"):
        for name in ("ddc_kernels.hip", "fir8_block.inc", "ddc_fir_i8.hip", "ddc_pipeline.cpp", "ddc_kernels.h"):
            src = open(os.path.join(csrc, name)).read()
            assert ("#ifdef " + forbidden) not in src and ("defined(" + forbidden) not in src, (name, forbidden)


def test_experiment_patches_apply_to_their_commits(tmp_path):
    """tools/ubench/*.patch are the experiments the notebooks quote (probes, ablations, layouts tried and not adopted).  Every
    one of them is listed in tools/ubench/PATCHES.json with the commit whose tree it applies to, and `git apply --check`
    agrees -- a patch nothing can apply to any more is a dead file (round 5 review)."""
    import glob
    import json
    import shutil
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (the GPU box gets a snapshot without .git)")
    base = json.load(open(os.path.join(ROOT, "tools", "ubench", "PATCHES.json")))
    patches = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tools", "ubench", "*.patch")))
    assert patches == sorted(k for k in base if k.endswith(".patch")), "PATCHES.json and tools/ubench/*.patch differ"
    for name in patches:
        commit = base[name]
        if subprocess.run(["git", "-C", ROOT, "cat-file", "-e", commit + "^{commit}"], capture_output=True).returncode:
            pytest.skip(f"commit {commit} is not in this clone (shallow history)")
        tree = tmp_path / name
        tree.mkdir()
        ar = subprocess.run(["git", "-C", ROOT, "archive", commit, "libperseus-sdr_amd/csrc", "tools", "bench.py"], capture_output=True)
        assert ar.returncode == 0, ar.stderr[-300:]
        subprocess.run(["tar", "-x", "-C", str(tree)], input=ar.stdout, check=True)
        subprocess.run(["git", "init", "-q", str(tree)], check=True)
        r = subprocess.run(["git", "-C", str(tree), "apply", "--check", os.path.join(ROOT, "tools", "ubench", name)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (name, commit, r.stderr[-500:])
