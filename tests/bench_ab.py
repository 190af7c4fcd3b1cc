"""Tuning aid: run one bench script against several builds of the HIP library (bmhrl_amd/csrc/variants/*.so) in ONE
GPU session, interleaved, so that the comparison is not between boxes:  python tests/bench_ab.py script.py [names...]"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
script = sys.argv[1]
names = sys.argv[2:] or sorted(os.path.basename(f)[:-3] for f in glob.glob(os.path.join(root, "bmhrl_amd/csrc/variants/*.so")))
for rep in range(2):
    for n in names:
        env = dict(os.environ, BMHRL_HIP_LIB=os.path.join(root, "bmhrl_amd/csrc/variants", n + ".so"))
        out = subprocess.run([sys.executable, os.path.join(root, script)], env=env, capture_output=True, text=True)
        lines = [l for l in (out.stdout + out.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l]
        print(f"== {n} (pass {rep})", flush=True)
        print("\n".join(lines[-6:]), flush=True)
