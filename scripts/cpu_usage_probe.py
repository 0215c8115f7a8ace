"""Host CPU cost of the Langevin launch loop (GPU box only) -- the question behind "do eight ranks fit one host?".

    python scripts/cpu_usage_probe.py [--workload c3] [--steps 100000] [--procs 1] [--poll sleep|spin] [--pin 1]

One process: pins itself like bench.py, warms the clock, runs `steps` Langevin steps and prints steps/s, the process's
CPU seconds per 100 000 steps split by thread (launching thread / HIP runtime helpers), context switches of the
launching thread and the cgroup's throttling counters (nr_throttled, throttled_usec) before and after.
--procs P > 1: the parent (which never touches the GPU) starts P children of itself at once, LOCAL_RANK = 0..P-1, all
on the one card of a 1-GPU box (the box allows at most 6 processes on its card), and prints their lines plus the
cgroup counters around the whole group: P launch loops under the box's real CPU quota.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cgroup_cpu_stat():
    out = {}
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            for line in fh:
                k, v = line.split()
                out[k] = int(v)
    except OSError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            out["cpu.max"] = fh.read().strip()
    except OSError:
        pass
    return out


def thread_times():
    """{tid: (name, utime+stime seconds, voluntary, involuntary context switches)} of this process."""
    tick = os.sysconf("SC_CLK_TCK")
    res = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as fh:
                raw = fh.read()
            name = raw[raw.index("(") + 1:raw.rindex(")")]
            f = raw[raw.rindex(")") + 2:].split()
            cpu = (int(f[11]) + int(f[12])) / tick
            vol = inv = 0
            with open(f"/proc/self/task/{tid}/status") as fh:
                for line in fh:
                    if line.startswith("voluntary_ctxt_switches"):
                        vol = int(line.split()[1])
                    elif line.startswith("nonvoluntary_ctxt_switches"):
                        inv = int(line.split()[1])
            res[int(tid)] = (name, cpu, vol, inv)
        except (OSError, ValueError):
            continue
    return res


def child(args):
    import threading

    import torch

    torch.set_num_threads(1)
    from chiron_amd._device import pin_process_cpus

    pinned = pin_process_cpus() if args.pin else None
    import bench

    fluid, pot, state, thermo, nbr, integ = bench.make_system(args.workload, 300.0 * 1.05 ** int(os.environ.get("LOCAL_RANK", "0")), 32, "cell")
    integ.run(state, thermo, number_of_steps=0, nbr_list=nbr)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.warm_s:
        state, nbr = integ.run(state, thermo, number_of_steps=2000, nbr_list=nbr)
        torch.cuda.synchronize()
    me = threading.get_native_id()
    th0, cg0 = thread_times(), cgroup_cpu_stat()
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:  # (1000-step segments: the shape of a replica-exchange run)
        seg = min(args.segment, args.steps - done)
        state, nbr = integ.run(state, thermo, number_of_steps=seg, nbr_list=nbr)
        done += seg
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    th1, cg1 = thread_times(), cgroup_cpu_stat()
    per = 1e5 / args.steps
    mine = (th1[me][1] - th0.get(me, (0, 0, 0, 0))[1]) * per
    others = sum((th1[t][1] - th0.get(t, (0, 0, 0, 0))[1]) for t in th1 if t != me) * per
    print(json.dumps({
        "rank": int(os.environ.get("LOCAL_RANK", "0")), "workload": args.workload, "poll": os.environ.get("CHX_POLL", "sleep"),
        "pinned_cpus": pinned, "steps": args.steps, "steps_per_s": round(args.steps / wall, 1),
        "cpu_s_per_100k_steps": {"launching_thread": round(mine, 3), "other_threads": round(others, 3), "total": round(mine + others, 3)},
        "cores_busy": round((mine + others) / per / wall, 3),
        "launching_thread_ctx_switches": {"voluntary": th1[me][2] - th0.get(me, (0, 0, 0, 0))[2],
                                          "involuntary": th1[me][3] - th0.get(me, (0, 0, 0, 0))[3]},
        "cgroup": {"cpu.max": cg1.get("cpu.max"), "nr_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                   "throttled_usec": cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)},
        "rebuilds": integ.n_rebuilds,
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--segment", type=int, default=1000)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--poll", default=None, choices=["sleep", "spin"])
    ap.add_argument("--pin", type=int, default=1)
    ap.add_argument("--warm-s", type=float, default=3.0)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.poll:
        os.environ["CHX_POLL"] = args.poll
    if args.child or args.procs == 1:
        return child(args)
    assert args.procs <= 6, "the GPU box allows at most 6 processes on its card"
    cg0 = cgroup_cpu_stat()
    t0 = time.perf_counter()
    procs = []
    for r in range(args.procs):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(args.procs))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"] + [a for a in sys.argv[1:]],
                                      env=env, stdout=subprocess.PIPE, text=True))
    lines = [p.communicate()[0] for p in procs]
    wall = time.perf_counter() - t0
    cg1 = cgroup_cpu_stat()
    rows = [json.loads(l) for out in lines for l in out.splitlines() if l.startswith("{")]
    for r in rows:
        print(json.dumps(r))
    print(json.dumps({"group": args.procs, "poll": os.environ.get("CHX_POLL", "sleep"), "wall_s": round(wall, 2),
                      "aggregate_steps_per_s": round(sum(r["steps_per_s"] for r in rows), 1),
                      "cpu_s_per_100k_steps_mean": round(sum(r["cpu_s_per_100k_steps"]["total"] for r in rows) / max(1, len(rows)), 3),
                      "cores_busy_total": round(sum(r["cores_busy"] for r in rows), 2),
                      "cgroup": {"cpu.max": cg1.get("cpu.max"), "nr_periods": cg1.get("nr_periods", 0) - cg0.get("nr_periods", 0),
                                 "nr_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                                 "throttled_usec": cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)},
                      "exit_codes": [p.returncode for p in procs]}), flush=True)


if __name__ == "__main__":
    main()
