import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in ("", "build/libpom_q3.so", "build/libpom_q2.so"):
    env = dict(os.environ)
    if lib: env["POM_LIB"] = os.path.join(ROOT, lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/quad_test.py")], env=env, capture_output=True, text=True).stdout
    print("== lib", lib or "default (4 waves/SIMD cap)")
    print("\n".join(l for l in out.splitlines() if "QUAD" in l))
