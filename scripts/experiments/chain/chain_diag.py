#!/usr/bin/env python3
"""Diagnostic (POM_CHAIN_DIAG build): where a chained wavefront spends its time — to the ticket, polling for its tile, in all."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "build", "libpom_batch_chaindiag%s.so" % os.environ.get("POM_TAG", ""))
if not os.path.exists(lib) or "--build" in sys.argv:
    subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_CHAIN_DIAG", *os.environ.get("POM_EXTRA_FLAGS", "").split(), "-I" + ROOT + "/include",
                    "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
if "--build" in sys.argv:
    sys.exit(0)
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = B.BatchEnvironment(N, mode=B.MODE_ENV, auto_reset=True, max_steps=800, issue_mode=B.ISSUE_CHAIN)
env.make_game(pa.make_boards(N, seed=1, kind="ffa"))
env.step_random(1, 1, ticks=300)
L = B.load_library()
tiles = (N + 63) // 64 * 64 // 16
buf = np.zeros((tiles, 68), dtype=np.uint64)
L.pom_chain_diag_read.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
assert L.pom_chain_diag_read(env._h, buf.ctypes.data, tiles) == 0
for steps in (tuple(int(x) for x in os.environ["POM_STEPS"].split(",")) if "POM_STEPS" in os.environ else (20, 500)):
    env.sync()
    t0 = time.perf_counter()
    env.step_random(1, 1, ticks=steps)
    env.sync()
    dt = (time.perf_counter() - t0) / steps * 1e6
    assert L.pom_chain_diag_read(env._h, buf.ctypes.data, tiles) == 0
    b = buf[: N // 16].astype(np.float64) / steps
    print(f"{N} envs, {steps} steps: {dt:.2f} us per step; per wavefront-tick (shader cycles): to the ticket {b[:,0].mean():.0f}, polling {b[:,1].mean():.0f} "
          f"({b[:,3].mean():.2f} polls), in all {b[:,2].mean():.0f}; by tile: in-all p50 {np.percentile(b[:,2],50):.0f} p99 {np.percentile(b[:,2],99):.0f}")
    # the last 32 visits of every tile: when did the wavefronts of a launch start and end (us, relative to the earliest start)?
    st = buf[: N // 16, 4::2].astype(np.int64)
    en = buf[: N // 16, 5::2].astype(np.int64)
    used = st.min(axis=0) > 0
    st, en = st[:, used], en[:, used]
    order = np.argsort(st.min(axis=0))
    t00 = st.min()
    for k in order:
        print(f"    visit slot {k}: starts {(st[:,k].min()-t00)/100:7.2f} .. {(st[:,k].max()-t00)/100:7.2f} (p50 {(np.percentile(st[:,k],50)-t00)/100:7.2f}), "
              f"ends {(en[:,k].min()-t00)/100:7.2f} .. {(en[:,k].max()-t00)/100:7.2f}; lasts mean {(en[:,k]-st[:,k]).mean()/100:.2f} us")
env.close()
