"""A/B on one box: run a tool against the product library and against variants built by tools/variant.py, in child processes, R rounds
interleaved.  usage: tools/ab.py <rounds> <tool.py and its arguments, quoted> <variant name | product> ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, cmd, names = int(sys.argv[1]), sys.argv[2].split(), sys.argv[3:]
for r in range(rounds):
    for n in names:
        env = dict(os.environ)
        if n != "product":
            env["VK_LIB"] = os.path.join("tools", "_variants", n, "libvokselis_hip.so")
        p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        print("[%s round %d] rc=%d" % (n, r, p.returncode), (p.stdout.strip().splitlines() or [p.stderr[-500:]])[-1], flush=True)
