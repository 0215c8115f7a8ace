"""Does the per-process 'slow mode' of the C3 loop go away with time under load (a clock / power-state ramp)?
Back-to-back 20000-step windows with their start time, an idle pause, and more windows.  GPU box only."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "mclk" in l or "fclk" in l or "Power" in l) and "GPU[0]" in l]
        return " | ".join(k.split(":", 1)[1].strip() if ":" in k else k for k in keep)
    except Exception as e:  # pragma: no cover
        return f"(rocm-smi: {e})"


wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
fluid, pot, state, thermo, nbr, integ = bench.make_system(wl, 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
t_start = time.perf_counter()


def sclk():
    import glob
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            cur = [l for l in open(f).read().splitlines() if l.strip().endswith("*")]
            if cur:
                return cur[0].strip()
        except OSError:
            pass
    return "?"


def windows(k, steps=20000):
    global state, nbr
    for _ in range(k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        state, nbr = integ.run(state, thermo, number_of_steps=steps, nbr_list=nbr)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"  t = {t0 - t_start:6.2f} s: {steps / (t1 - t0):8.0f} steps/s   sclk {sclk()}", flush=True)


print("clocks at start:", clocks())
windows(24)
print("clocks after 24 windows:", clocks())
print("idle 4 s"); time.sleep(4.0)
print("clocks after the pause:", clocks())
windows(10)
