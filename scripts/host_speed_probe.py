"""Is the per-process spread of the C3 figure a HOST effect?  Prints, for this process: the CPU it runs on, a pure-Python
loop time, the host time of a list rebuild inside run(), and steps/s.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
import torch
torch.set_num_threads(1)
import bench
from chiron_amd import integrators as I

p = psutil.Process()


def pyloop():
    t0 = time.perf_counter()
    s = 0
    for i in range(2_000_000):
        s += i & 7
    return time.perf_counter() - t0


cpu0 = p.cpu_num()
t_py = min(pyloop() for _ in range(3))
fluid, pot, state, thermo, nbr, integ = bench.make_system("c3", 300.0, 32, "cell")
integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
if os.environ.get("PIN_MAIN") == "1":
    # the calling thread only (the runtime's helper threads exist by now and keep the whole mask)
    os.sched_setaffinity(0, {p.cpu_num()})
if os.environ.get("NICE_MAIN"):
    os.nice(int(os.environ["NICE_MAIN"]))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
    torch.cuda.synchronize()
host = []
orig = I._RunState._rebuild


def reb(self, step=0, speculative=False):
    t = time.perf_counter()
    orig(self, step, speculative)
    host.append(time.perf_counter() - t)


I._RunState._rebuild = reb
def cpu_stat():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except OSError:
            continue
    return out


def ctx_switches():
    c = p.num_ctx_switches()
    return c.voluntary, c.involuntary


# what is the main thread doing when it is not running?  A sampler thread reads /proc/self/task/<tid>/syscall and wchan.
import collections, threading
main_tid = threading.get_native_id()
samples = collections.Counter()
stop_sampling = [False]


def sampler():
    ps, pw = f"/proc/self/task/{main_tid}/syscall", f"/proc/self/task/{main_tid}/wchan"
    while not stop_sampling[0]:
        try:
            sc = open(ps).read().split()
            wc = open(pw).read().strip()
        except OSError:
            break
        samples[(sc[0] if sc else "?", wc)] += 1
        time.sleep(0.0002)


thr = threading.Thread(target=sampler, daemon=True)
if os.environ.get("SAMPLE_SYSCALLS") == "1":
    thr.start()
stat0, cs0 = cpu_stat(), ctx_switches()
thr0 = {t.id: (t.user_time, t.system_time) for t in p.threads()}
res = []
cpus = set()
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, nbr = integ.run(state, thermo, number_of_steps=20000, nbr_list=nbr)
    torch.cuda.synchronize(); res.append(20000 / (time.perf_counter() - t0))
    cpus.add(p.cpu_num())
stop_sampling[0] = True
stat1, cs1 = cpu_stat(), ctx_switches()
thr1 = {t.id: (t.user_time, t.system_time) for t in p.threads()}
busy = sorted(((thr1[k][0] - thr0.get(k, (0, 0))[0], thr1[k][1] - thr0.get(k, (0, 0))[1], k) for k in thr1), reverse=True)[:5]


def tname(tid):
    try:
        return open(f"/proc/self/task/{tid}/comm").read().strip()
    except OSError:
        return "?"


print("   busiest threads over the five windows (user s, system s, name):", [(round(u, 3), round(sy, 3), tname(k) + ("*" if k == threading.get_native_id() else "")) for u, sy, k in busy], flush=True)
if samples:
    print("   main thread samples (syscall nr or 'running', wchan):", samples.most_common(8), flush=True)
print("   cgroup cpu.stat deltas over the five windows:", {k: stat1[k] - stat0.get(k, 0) for k in stat1 if "throttl" in k or k == "nr_periods"},
      " context switches (voluntary, involuntary):", (cs1[0] - cs0[0], cs1[1] - cs0[1]), flush=True)
print(f"cpu {cpu0} -> {sorted(cpus)}  python loop {1e3 * t_py:6.1f} ms  rebuild host time median {1e6 * np.median(host):6.1f} us  "
      f"steps/s median {sorted(res)[2]:.0f}  {[round(r) for r in res]}  affinity {len(os.sched_getaffinity(0))} cpus", flush=True)

# where inside the loop?  GPU time per step between rebuilds (events around every 32-launch C call) ...
I._RunState._rebuild = orig
log = []
orig_enq = I._RunState._enqueue_window


def enq(self, s0, n):
    a = torch.cuda.Event(enable_timing=True); a.record()
    sent = orig_enq(self, s0, n)
    b = torch.cuda.Event(enable_timing=True); b.record()
    log.append((sent, n, a, b))
    return sent


I._RunState._enqueue_window = enq
state, nbr = integ.run(state, thermo, number_of_steps=20000, nbr_list=nbr)
torch.cuda.synchronize()
I._RunState._enqueue_window = orig_enq
per = np.array([a.elapsed_time(b) * 1e3 / 32 for sent, n, a, b in log if sent == n == 32])
# ... and the same launches with the polling loop vs without it, skin check disabled (no rebuild, no halt)
from chiron_amd import ops, unit
from chiron_amd.integrators import _RunState, langevin_constants
from chiron_amd.utils import get_list_of_mass
from chiron_amd._device import as_tensor
nbr.build_from_state(state)
half_dt, a_, b_ = langevin_constants(1.0 * unit.femtoseconds, 1.0 / unit.picoseconds)
mass = as_tensor(get_list_of_mass(fluid.topology).value_in_unit_system(unit.md_unit_system))
st = _RunState(integ, pot, nbr, state.positions, state.velocities, mass, 2.494, half_dt, a_, b_, state.current_PRNG_key, None, "fused")
st.params.skin = 1e6
step = [0]


def call(n, window):
    even, odd = (st.x_a, st.x_b) if (step[0] & 1) == 0 else (st.x_b, st.x_a)
    if window:
        ops.langevin_fused_run(st.params, even, odd, step[0], n, 2, window)
    else:
        ops.langevin_fused_chunk(st.params, even, odd, step[0], n, 2, 1)
    step[0] += n


out = []
sw = []
for window in (12, 0):
    st._gen.host[0] = step[0] - 1
    tt = []
    c0 = ctx_switches()
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(32):
            call(32, window)
        torch.cuda.synchronize(); tt.append((time.perf_counter() - t0) / 1024 * 1e6)
    out.append(np.median(tt))
    sw.append(ctx_switches()[0] - c0[0])
print(f"   voluntary context switches during those 2 x 5120 launches (no rebuild, no sync inside): {sw}")
print(f"   live loop, GPU time per step between rebuilds: median {np.median(per):.2f} us;  1024 launches, skin check off: "
      f"{out[0]:.2f} us per launch through the polling loop, {out[1]:.2f} without it", flush=True)
