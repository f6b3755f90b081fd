"""Per-iteration latency of the search at small batches (the self-play tail), A/B over an environment switch:
    python scripts/tail_ab.py DIEE_CLUSTER_HEADS 1 0 [-- n1 n2 ...]
runs scripts/small_batch_loop.py for each batch size with the variable set to each value (fresh process each)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sizes = [4, 16, 32, 64, 128, 256]
if "--" in args:
    i = args.index("--"); sizes = [int(x) for x in args[i + 1:]]; args = args[:i]
var, vals = args[0], args[1:]
for n in sizes:
    row = [f"n={n:4d}"]
    for rep in range(2):
        for v in vals:
            env = dict(os.environ); env[var] = v
            out = subprocess.run([sys.executable, os.path.join(root, "scripts", "small_batch_loop.py"), str(n), "10"], env=env, cwd=root,
                                 capture_output=True, text=True)
            us = out.stdout.strip().split("=")[-1].split("us")[0].strip() if out.returncode == 0 else "FAILED " + out.stderr[-200:]
            row.append(f"{var}={v}: {us} us/iter")
    print("  ".join(row), flush=True)
