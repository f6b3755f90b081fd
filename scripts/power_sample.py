"""Socket power, shader clock and temperature (rocm-smi) while the fused tower runs back to back at 1024 boards."""
import subprocess, sys, threading, time
sys.path.insert(0, ".")
import diee_amd

e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
rows = []
def smi():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
    return " | ".join(l.split(":", 1)[1].strip() for l in r.splitlines() if any(k in l for k in ("Package Power", "sclk", "junction")))
def poll():
    for _ in range(6):
        time.sleep(0.7); rows.append(smi())
t = threading.Thread(target=poll); t.start()
t0 = time.time()
while time.time() - t0 < 6:
    e.conv_bench(1024, 108, 200)
t.join()
for r in rows: print("under load:", r)
print("right after:", smi())
