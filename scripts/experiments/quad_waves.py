import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for lib in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if lib != "default": env["POM_LIB"] = os.path.join(ROOT, lib)
    for args in (["--steps", "300", "--warmup", "40"], ["--steps", "150", "--warmup", "40", "--envs", "262144"],
                 ["--steps", "150", "--warmup", "40", "--kind", "stress", "--dist", "stress"], ["--steps", "300", "--warmup", "40", "--envs", "4096"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args, env=env, capture_output=True, text=True).stdout
        import json
        try:
            j = json.loads(out.strip().splitlines()[-1])
            print(f"{lib:24s} {' '.join(args[4:]) or '65536 ffa':28s} {j['ms_per_step']*1e3:8.2f} us/step  {j['value']/1e9:6.3f} G/s")
        except Exception as ex:
            print(lib, args, "FAILED", out[-300:])
