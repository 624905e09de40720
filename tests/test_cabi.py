"""The C-ABI library builds for gfx950, loads, exports every symbol include/pom_batch.h declares, and
fails loudly (no CPU fallback) where there is no HIP device.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pom_batch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pom_[a-z_]+)\s*\(", text)))


def test_exports_match_header(hip_lib):
    syms = _declared_symbols()
    assert "pom_batch_step" in syms and "pom_step" in syms and len(syms) >= 18
    for s in syms:
        assert hasattr(hip_lib, s), f"{s} declared in include/pom_batch.h but not exported"


def test_library_is_a_gfx950_code_object(hip_lib):
    from pomcpp_amd.batch import library_path
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={library_path()}"],
                         capture_output=True, text=True)
    blob = open(library_path(), "rb").read()
    assert b"gfx950" in blob, out.stdout


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "pom_batch.h"\nint main(void){ PomBatchOptions o; o.struct_size = sizeof o; return o.struct_size == 0; }\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        subprocess.run([cc, std, "-x", "c" if cc == "gcc" else "c++", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


def test_state_layout_matches_reference_offsets():
    from pomcpp_amd.state import STATE_DTYPE
    f = STATE_DTYPE.fields
    assert STATE_DTYPE.itemsize == 1004
    assert [f[k][1] for k in ("board", "timeStep", "aliveAgents", "agents", "bombs_queue", "bombs_index", "bombs_count",
                              "flames_queue", "flames_index", "flames_count")] == [0, 484, 488, 492, 588, 668, 672, 676, 996, 1000]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful where no GPU is present")
def test_no_cpu_fallback_without_a_gpu(hip_lib):
    from pomcpp_amd.batch import BatchEnvironment, PomError, step_one
    from pomcpp_amd.state import new_states
    with pytest.raises(PomError) as e:
        BatchEnvironment(64)
    assert e.value.code == 2  # POM_E_HIP
    with pytest.raises(PomError):
        step_one(new_states(1), np.zeros(4, dtype=np.int32))


def test_product_never_touches_the_oracle():
    """No file of the product (package sources, public headers) includes, links, loads or imports the checker."""
    needles = ("pom_oracle", "libpom_oracle", "oracle_lib", "from tests", "import tests", "oracle/", "_ref")
    for d, _, files in os.walk(os.path.join(ROOT, "pomcpp_amd")):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                code = open(os.path.join(d, fn)).read()
                code = re.sub(r"/\*.*?\*/", "", code, flags=re.S)      # comments may mention it
                code = re.sub(r'""".*?"""', "", code, flags=re.S)
                code = re.sub(r"#.*", "", code) if fn.endswith(".py") else code
                for n in needles:
                    assert n not in code, (fn, n)
    for fn in os.listdir(os.path.join(ROOT, "include")):
        assert "pom_oracle" not in open(os.path.join(ROOT, "include", fn)).read(), fn
    syms = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "pomcpp_amd", "libpom_batch.so")],
                          capture_output=True, text=True).stdout
    assert "pom_oracle" not in syms and "ref_step" not in syms
