# development tool: sustained shader clock / power while lone proofs (or a register-only Blake2s loop) run.
#   python tools/clock_watch.py [seconds]      -> prints sclk / power samples taken every 0.25 s by rocm-smi's library (amdsmi sysfs)
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0


def sample():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            for line in open(card + "/pp_dpm_sclk").read().splitlines():
                if "*" in line:
                    out["sclk"] = line.strip()
        except Exception as e:
            out["sclk_err"] = str(e)[:60]
        for h in glob.glob(card + "/hwmon/hwmon*"):
            for f, k in (("power1_average", "W"), ("power1_input", "W_in"), ("freq1_input", "freq1"), ("temp1_input", "temp")):
                try:
                    out[k] = int(open(h + "/" + f).read())
                except Exception:
                    pass
    return out


stop = False
samples = []


def watcher():
    while not stop:
        samples.append((time.perf_counter(), sample()))
        time.sleep(0.25)


print("idle:", sample())
be = Backend(0)
dev = be.upload_input(synth_fibonacci(419000))
for _ in range(3):
    be.prove_device(dev).free()
th = threading.Thread(target=watcher)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    be.prove_device(dev).free()
    n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop = True
th.join()
print(f"{n} proofs, {dt / n * 1e3:.3f} ms per proof")
for t, s in samples[::4]:
    print(f"{t - t0:6.2f}s {s}")
try:
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=30).stdout[-1500:])
except Exception as e:
    print("rocm-smi:", e)
