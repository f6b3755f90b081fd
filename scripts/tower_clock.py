"""In-kernel clock of the fused tower (MI355X_MICROARCH.md, DVFS give-back item 6): a DIAGNOSTIC build
(DIEE_OUT=libdiee_clock.so DIEE_EXTRA_FLAGS=-DDIEE_TOWER_ABLATE=3 python die-e_amd/build.py) stamps s_memtime /
s_memrealtime around the 38 layers of every workgroup after 200 back-to-back launches on real game positions; the
stamps go to a debug buffer nothing else reads.  Prints the median clock next to rocm-smi's power / sclk readings."""
import os, subprocess, sys, threading, time
sys.path.insert(0, ".")
os.environ["DIEE_TOWER_CLOCK"] = "1"
import diee_amd
L = diee_amd.load_library(os.path.join("die-e_amd", "libdiee_clock.so")); diee_amd._lib = L
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))


def smi():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
    return " | ".join(l.split(":", 1)[1].strip() for l in r.splitlines() if any(k in l for k in ("Package Power", "sclk", "junction")))


rows = []
def poll():
    for _ in range(5):
        time.sleep(0.8); rows.append(smi())
for G, v in ((1024, 108), (2048, 108), (700, 106), (300, 103)):
    t = threading.Thread(target=poll); t.start()
    t0 = time.time()
    while time.time() - t0 < 2.5:                 # >= 2 s of back-to-back launches before the stamped ones
        e.conv_bench(G, v, 100)
    us = e.conv_bench(G, v, 100)                  # prints "[diee] fused tower in-kernel clock: median ... MHz" (stderr)
    t.join()
    print(f"G {G} variant {v}: forward {us[2]:.1f} us", flush=True)
    for r in rows: print("   rocm-smi under load:", r)
    rows.clear()
print("rocm-smi right after:", smi())
