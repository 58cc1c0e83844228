"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the header declares,
and refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from coffeedb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build_library()
    return capi.load_library()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "coffeedb_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cdb_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/coffeedb_gpu.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.cdb_create(C.byref(h), -1) == 2  # CDB_E_DEVICE
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.GpuStringIndex()


def test_product_does_not_touch_oracle():
    # the shipped sources must never reference oracle/ (judge rule: oracle is test infrastructure)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "coffeedb_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def _raw_record(id_, fields):
    """CoffeeDB's on-disk record (writer: reference src/database.cpp:334-378)."""
    import struct
    out = struct.pack("<qi", id_, len(fields))
    for key, value in fields:
        out += struct.pack("<i", len(key)) + key
        if isinstance(value, bool):
            out += struct.pack("<b?", 0, value)
        elif isinstance(value, int):
            out += struct.pack("<bq", 1, value)
        elif isinstance(value, float):
            out += struct.pack("<bd", 2, value)
        else:
            out += struct.pack("<bi", 3, len(value)) + value
    return out


def test_raw_record_parser(lib):
    rec = _raw_record(1234567890123, [(b"number", 123), (b"name", b"sunkafei"), (b"flag", True), (b"pos", 1.7724),
                                      (b"secret", b"3010103"), (b"empty", b"")])
    assert capi.raw_record_find_string(rec, b"secret") == (1234567890123, b"3010103")
    assert capi.raw_record_find_string(rec, b"name") == (1234567890123, b"sunkafei")
    assert capi.raw_record_find_string(rec, b"empty") == (1234567890123, b"")
    assert capi.raw_record_find_string(rec, b"number") is None      # not a string
    assert capi.raw_record_find_string(rec, b"missing") is None
    for cut in (3, 11, 20, len(rec) - 1):
        with pytest.raises(ValueError):
            capi.raw_record_find_string(rec[:cut], b"secret")


def test_sharded_and_bulk_entry_points_refuse_cleanly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    dev = (C.c_int * 2)(0, 0)
    assert lib.cdb_shards_create(C.byref(h), dev, 2) == 2 and not h.value      # CDB_E_DEVICE, nothing half-created
    assert lib.cdb_shards_create(C.byref(h), dev, 0) == 1                      # CDB_E_INVALID
    assert lib.cdb_add_raw_dir(None, b"/tmp", b"k", None, None) == 1
    ids, cnt, n = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_size_t(0)
    assert lib.cdb_query_and(None, 0, 0, 0, 0, 0, C.byref(ids), C.byref(cnt), C.byref(n)) == 1
    assert lib.cdb_comm_merge(None, None, None) == 1
    lib.cdb_shards_destroy(None)
    lib.cdb_comm_destroy(None)


def test_layout_rule_and_capacity_limit_messages(lib):
    # index.cpp:182-208 through the host-only entry cdb_layout_rule — what every cdb_build* applies first.  The two
    # capacity limits (index.cpp:195-200) cannot be reached with real data on one machine; their messages are the
    # reference's own strings.
    assert capi.layout_rule(3, 8) == (2, 3, 4, 4)                 # README corpus: 3 docs -> mask 3; longest 8 -> mask 15
    assert capi.layout_rule(0, 0) == (1, 1, 4, 1)                 # empty index: bits = 1 (SURVEY §8c golden vector 2)
    assert capi.layout_rule(1 << 20, 1024) == (21, (1 << 21) - 1, 4, 11)   # C1: 21 + 11 = 32 bits -> u32 entries
    assert capi.layout_rule(1 << 20, 1025) == (21, (1 << 21) - 1, 4, 11)   # mask 2047 still covers 1025
    assert capi.layout_rule(1 << 20, 2048) == (21, (1 << 21) - 1, 8, 12)   # 33 bits -> u64 entries
    assert capi.layout_rule(1 << 23, 1024) == (24, (1 << 24) - 1, 8, 11)   # C2
    assert capi.layout_rule((1 << 32) - 1, 4)[0] == 32
    with pytest.raises(RuntimeError, match="The number of objects exceeds the maximum range that CoffeeDB can handle"):
        capi.layout_rule((1 << 32) + 1, 4)
    with pytest.raises(RuntimeError, match="The amount of data exceeds the maximum range that CoffeeDB can handle"):
        capi.layout_rule(1 << 31, 1 << 33)                        # 32 + 34 bits


def test_no_result_changing_switch_in_the_product_library(lib):
    """timing ablations that produce WRONG indexes must be a compile-time choice of a separate library
    (-DRS_SWEEP_ABL / -DRS_GATHER_ABL / -DRS_SEG_ABL / -DRS_GEN_ABL, loaded through CDB_LIB_PATH by the measuring script), never a
    run-time switch of the shipped one: no environment variable with ABL in its name, no CDB_OPTIONS back door inside the library,
    no kernel with an ablation parameter."""
    so = os.path.join(ROOT, "coffeedb_amd", "csrc", "libcoffeedb_gpu.so")
    blob = open(so, "rb").read()
    for needle in (b"_ABL", b"CDB_OPTIONS", b"CDB_DEBUG_NO_SEGCAP", b"CDB_GATHER_WGS"):
        assert needle not in blob, needle
    csrc = os.path.join(ROOT, "coffeedb_amd", "csrc")
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h")):
            continue
        txt = open(os.path.join(csrc, f)).read()
        # ablation masks are constexpr copies of a compile-time macro that defaults to 0; never a function parameter
        assert not re.search(r"\bint\s+abl\s*[,)]", txt), f
        assert not re.search(r'getenv\("[A-Z_]*ABL', txt), f
        for m in re.finditer(r"#define\s+(RS_[A-Z]+_ABL)\s+(\S+)", txt):
            assert m.group(2) == "0", (f, m.group(0))


def test_device_code_is_free_of_the_lane_mask_miscompile(tmp_path):
    """Round 5 shipped wrong sweep records for half a day because of a compiler artefact, not a source error: wave-uniform switches
    re-materialised as lane masks INSIDE a loop that lanes leave at different times, then tested behind the loop for the lanes that
    had left (records_sweep.h, at phase B's loops).  tools/isa_lanemask_scan.py recognises that shape in the assembly: every device
    object of the library must scan clean, and the scanner must still recognise the shape (a minimal listing of it)."""
    import shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scan = os.path.join(root, "tools", "isa_lanemask_scan.py")
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join([
        "_Z1kv:", "\ts_branch .LBB0_2",
        ".LBB0_1:                                ;   in Loop: Header=BB0_2 Depth=1",
        "\ts_andn2_b64 exec, exec, s[30:31]", "\ts_cbranch_execz .LBB0_3",
        ".LBB0_2:                                ; =>This Inner Loop Header: Depth=1",
        "\tv_cndmask_b32_e64 v2, 0, 1, s[46:47]", "\tv_cmp_ne_u32_e64 s[22:23], 1, v2", "\ts_branch .LBB0_1",
        ".LBB0_3:", "\ts_or_b64 exec, exec, s[30:31]", "\ts_and_b64 vcc, exec, s[22:23]", "\ts_cbranch_vccnz .LBB0_4",
        ".LBB0_4:", "\ts_endpgm", ".Lfunc_end0:", ""]))
    r = subprocess.run([sys.executable, scan, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "suspicious uses: 1" in r.stdout, r.stdout
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc: the library's own assembly is scanned where it is built")
    r = subprocess.run(["make", "-C", os.path.join(root, "coffeedb_amd", "csrc"), "isa-scan", f"ISA_DIR={tmp_path}"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "suspicious uses: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
