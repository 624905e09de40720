#!/usr/bin/env python3
"""PMC calibration on a known byte count: pom_step_kernel with ticks=0 reads and writes exactly one packed record per env
(448 B each way) with the same access pattern as the real tick.  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "build", "libpom_batch_diag.so")
if not os.path.exists(lib):
    subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_DIAG", "-I" + ROOT + "/include",
                    "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
n = 65536
env = B.BatchEnvironment(n, mode=B.MODE_ENV, auto_reset=True, max_steps=800)
env.make_game(pa.make_boards(n, seed=1))
L = B.load_library(); L.pom_diag_copy_only.argtypes = [C.c_void_p]
for _ in range(100):
    L.pom_diag_copy_only(env._h)
env.sync()
print("expected bytes per launch each way:", n * 448)
