"""Runs the C++ prelude mirror's restatement of the reference's unit tests (tests/cpp/prelude_test.cpp)
— compiled host code above the C ABI, as a Rust caller of graph::prelude would be."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "prelude_test")


@pytest.mark.gpu
def test_cpp_prelude_mirror_passes_reference_unit_tests():
    assert os.path.exists(EXE), "tests/cpp/prelude_test missing: run __graft_entry__.build()"
    r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout


def test_cpp_prelude_header_compiles():
    """CPU check: the header-only mirror compiles against include/graph_mi355x.h and links the library."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "prelude_test"], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)
