"""CPU: the C-ABI library loads and exports every symbol include/svs_amd.h
declares (no compute calls -- there is no GPU in the build container)."""
import os
import re

import pytest

from svs_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "svs_amd.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(svs_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    syms = _declared_symbols()
    assert len(syms) >= 14
    lib = _native.load()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/svs_amd.h but not exported"
        assert s in _native.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_native.SIGNATURES) == syms


def test_library_identifies_itself():
    lib = _native.load()
    assert b"gfx950" in lib.svs_version()
    assert _native.device_count() >= 0


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: constructing an index without a device raises."""
    import numpy as np
    from svs_amd import DeviceIndex
    if _native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        DeviceIndex(np.zeros((4, 4), dtype=np.float32))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "svs_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dp, fn)) as f:
                    src = f.read()
                assert "import oracle" not in src and "from oracle" not in src and "svs_oracle" not in src, fn


def test_header_is_plain_c_and_links(tmp_path):
    """include/svs_amd.h is the drop-in boundary: it must compile as C99 (no C++, no torch or
    HIP types in the signatures) and a C program must link against the library with it."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "svs_amd.h"\n'
        '#include <stdio.h>\n'
        'int main(void) {\n'
        '  svs_index_info_t info; svs_timing_t t; (void)info; (void)t;\n'
        '  printf("%s\\n", svs_version());\n'
        '  return svs_index_release((svs_index*)0) == SVS_OK ? 1 : 0;   /* null handle: an error, not a crash */\n'
        '}\n')
    lib_dir = os.path.join(root, "svs_amd", "lib")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe), "-L", lib_dir, "-lsvs_amd", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "svs_amd" in out.stdout, (out.returncode, out.stdout, out.stderr)


def _build_c(tmp_path, name, extra=("-lm",)):
    import subprocess
    lib_dir = os.path.join(ROOT, "svs_amd", "lib")
    exe = tmp_path / name
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", name + ".c"), "-o", str(exe), "-L", lib_dir, "-lsvs_amd", "-lpthread",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", *extra], check=True)
    return exe


def test_c_release_race_program_builds(tmp_path):
    """CPU: the pure-C race test (tests/c/release_race.c) compiles and links against the ABI."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert os.path.exists(_build_c(tmp_path, "release_race"))


@pytest.mark.gpu
def test_c_caller_release_races_with_search(gpu, tmp_path):
    """A C caller (no Python wrapper pinning the handle) releases the only owner reference while a
    search is in flight on another thread: include/svs_amd.h allows exactly this.  The search's own
    reference must outlive its geometry lock (the lock guard used to be destroyed after
    the last unref, i.e. on freed memory)."""
    import subprocess
    exe = _build_c(tmp_path, "release_race")
    out = subprocess.run([str(exe), "12"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)


def test_c_multi_device_program_builds(tmp_path):
    """CPU: tests/c/multi_device.c compiles (C99 header) and links against the ABI."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert os.path.exists(_build_c(tmp_path, "multi_device"))


@pytest.mark.gpu
def test_c_caller_multi_device(gpu, tmp_path):
    """A C caller of svs_multi_*: 2, 3 and 5 shards against one index (tests/c/multi_device.c)."""
    import subprocess
    exe = _build_c(tmp_path, "multi_device")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_c_callers_coalesce(gpu, tmp_path):
    """24 C threads on one handle with svs_index_set_coalesce (tests/c/coalesce.c)."""
    import subprocess
    exe = _build_c(tmp_path, "coalesce")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)
    print(out.stdout.strip())


def test_c_multi_mask_race_program_builds(tmp_path):
    """CPU: tests/c/multi_mask_race.c compiles and links against the ABI."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert os.path.exists(_build_c(tmp_path, "multi_mask_race"))


@pytest.mark.gpu
def test_c_caller_masks_rows_while_multi_searches_run(gpu, tmp_path):
    """svs_index_mask_rows through svs_multi_shard while svs_multi_search(k = n) runs on three other threads:
    every search answers with the rows there are (tests/c/multi_mask_race.c)."""
    import subprocess
    exe = _build_c(tmp_path, "multi_mask_race")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)
    print(out.stdout.strip())


def test_c_scores_race_program_builds(tmp_path):
    """CPU: tests/c/scores_race.c compiles and links against the ABI."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert os.path.exists(_build_c(tmp_path, "scores_race"))


@pytest.mark.gpu
def test_c_caller_reads_scores_while_rows_are_appended(gpu, tmp_path):
    """svs_index_scores_n sized from svs_index_info() while another thread appends (tests/c/scores_race.c): a full
    vector or SVS_ERR_INVALID with the row count to retry with, never a write past the capacity (VERDICT r3 item 5:
    round 3's capacity-less entry overflowed the caller's heap here)."""
    import subprocess
    exe = _build_c(tmp_path, "scores_race")
    out = subprocess.run([str(exe), "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)
    print(out.stdout.strip())


def test_internal_hooks_are_exported_but_not_in_the_public_header():
    """svs_internal_* (svs_amd/csrc/internal.h: test / tool hooks) are exported by the library and stay out of
    include/svs_amd.h (svs_index_coalesce_hold was a public entry that could stall every queued search)."""
    lib = _native.load()
    public = _declared_symbols()
    for s in _native.INTERNAL:
        assert hasattr(lib, s), s
        assert s not in public, s
    assert not hasattr(lib, "svs_index_coalesce_hold")
    assert not hasattr(lib, "svs_index_scores"), "the capacity-less entry must be gone (its binding must fail at load time)"


def test_quick_binding_keeps_the_gil_and_is_the_same_library():
    """svs_amd/_native.py QUICK: the lock-free bookkeeping entry points of a search are bound a second time through
    PyDLL (no GIL hand-off around them).  Same library, same thread-local error state; only entry points that cannot
    block may be on that list."""
    import ctypes as C
    ql = _native.quick()
    assert isinstance(ql, C.PyDLL) and ql is _native.quick()
    for name in _native.QUICK:
        assert name in _native.SIGNATURES and getattr(ql, name).argtypes == _native.SIGNATURES[name][1]
    # whatever may wait for the device, a lock or another thread stays on the GIL-releasing binding
    assert not {"svs_index_search", "svs_multi_search", "svs_index_append", "svs_index_create", "svs_index_scores_n"} & set(_native.QUICK)
    info = _native.IndexInfo()
    assert ql.svs_index_info(None, C.byref(info)) == _native.SVS_ERR_INVALID
    assert "null" in _native.last_error()                                   # read through the quick binding ...
    assert b"null" in (_native.load().svs_last_error() or b"")              # ... and the same state through the other
