"""Code written against pomcpp's C++ API compiled against include/pom_bboard.hpp (the drop-in surface) —
compile+link everywhere, run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "dropin_test")


@pytest.fixture(scope="module")
def dropin_exe(hip_lib):
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp"), "-o", EXE,
                    "-L" + os.path.join(ROOT, "pomcpp_amd"), "-lpom_batch", "-Wl,-rpath," + os.path.join(ROOT, "pomcpp_amd"),
                    "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"], check=True)
    return EXE


def test_reference_style_code_compiles_against_the_drop_in_header(dropin_exe):
    assert os.path.exists(dropin_exe)


@pytest.mark.gpu
def test_reference_style_code_runs_on_the_gpu(dropin_exe):
    out = subprocess.run([dropin_exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "dropin ok" in out.stdout, f"rc={out.returncode}\n{out.stdout}\n{out.stderr}"
