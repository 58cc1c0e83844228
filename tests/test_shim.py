"""The reference-side C++ binding (coffeedb_amd/csrc/shim/index.{h,cpp}): compiled against the C ABI and
driven the way the reference's database.cpp drives its indexes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    from coffeedb_amd import capi
    capi.build_library()
    subprocess.check_call(["make", "-C", CPP, "test_index_shim"], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "test_index_shim")


def test_shim_compiles_and_numeric_indexes_behave():
    exe = _build()
    out = subprocess.run([exe, "numeric"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_string_index_on_gpu():
    exe = os.path.join(CPP, "test_index_shim")
    if not os.path.exists(exe):
        exe = _build()
    out = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_string_index_spread_over_shards():
    # the same database.cpp-style walk with every string column split over two shards (COFFEEDB_GPUS; both on device
    # 0 here): add / build / query / query_batch through cdb_shards_*, OR / ranking / highlight shard by shard
    exe = os.path.join(CPP, "test_index_shim")
    if not os.path.exists(exe):
        exe = _build()
    env = dict(os.environ, COFFEEDB_GPUS="0,0", COFFEEDB_SHARD_ALL="1")
    out = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_string_index_with_the_resident_query_workgroup():
    # COFFEEDB_RESIDENT_QUERY=1: the same walk with lone query() calls answered by the resident workgroup (one GPU, and
    # two shards with a workgroup each)
    exe = os.path.join(CPP, "test_index_shim")
    if not os.path.exists(exe):
        exe = _build()
    for extra in ({}, {"COFFEEDB_GPUS": "0,0", "COFFEEDB_SHARD_ALL": "1"}):
        env = dict(os.environ, COFFEEDB_RESIDENT_QUERY="1", **extra)
        out = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
