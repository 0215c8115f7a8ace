#!/bin/bash
# Round 5's evidence in one GPU call (GPU box only):  bash scripts/r05_profiles.sh
#  (1) the driver's own bench command, plain and under rocprofv3 --kernel-trace --stats -> profiles/r05_bench_driver_cmd_*
#  (2) one --stats pass of scripts/kernel_bench.py per quoted system state (equilibrated snapshot on disk) ->
#      profiles/r05_kernel_states.md: every roofline fraction recomputes from one row
#  (3) the PMC passes over the hot kernels (separate --pmc runs) -> profiles/pmc_traffic.json, r05_pmc_hot_kernels.md
#  (4) the default bench line -> profiles/r05_bench_default.json
#  (5) the live C3 loop alone under the profiler -> profiles/r05_c3_live_loop_kernel_stats.md
#  (6) fresh processes alternating between plain HIP launches and the step queue -> gpurun_out/r05_prof/queue_ab_*.log
#  (7) the stand-alone AQL probe (dispatch cost, fence scopes, staleness, hand-over, XCD placement) -> gpurun_out/r05_prof/aql_*.log
set -o pipefail
PARTS=${PARTS:-"1 2 3 4 5 6 7"}   # which of the seven parts to run (two GPU calls: "1 2 3" and "4 5 6 7")
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r05_prof
mkdir -p "$out"
echo "== parts: $PARTS" | tee "$out/progress.log"
if has 1; then
echo "== (1) driver command" | tee -a "$out/progress.log"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_plain.log" 2> "$out/bench_plain.err" || exit 1
rm -rf /tmp/r05_drv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_drv -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$out/bench_prof.log" 2> "$out/bench_prof.err") || exit 1
python3 scripts/stats_to_md.py /tmp/r05_drv r05_driver_cmd "$out/bench_prof.log" "$out/bench_plain.log" \
  "Round 5: rocprofv3 --kernel-trace --stats of the driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5)" || exit 1
fi
if has 2; then
echo "== (2) per-state kernel passes" | tee -a "$out/progress.log"
export KB_SNAPSHOT_DIR=/tmp/kb_snap_r05
rm -rf "$KB_SNAPSHOT_DIR"
{
  echo "# Round 5: one system state per row -- rocprofv3 --kernel-trace --stats of \`scripts/kernel_bench.py <workload>\`"
  echo
  python3 -c "import re; print(re.search(r'^\"\"\"(.*?)\"\"\"', open('scripts/state_roofline.py').read(), re.S).group(1))"
  echo
  echo "| state | kernel | lanes / particle | P_nl | algorithmic bytes | launches | rocprofv3 avg (us) | GB/s | fraction of 8 TB/s | HIP-event time in the same pass (us) |"
  echo "|---|---|---|---|---|---|---|---|---|---|"
} > "$out/r05_kernel_states.md"
for spec in "c3 20000" "c4 20000" "c3_liquid 4000" "c4_liquid 1000"; do
  set -- $spec
  wl=$1; export KB_WARM=$2
  echo "   $wl (equilibrated $KB_WARM steps)" | tee -a "$out/progress.log"
  GS=auto KB_REPS=2 python3 scripts/kernel_bench.py $wl > "$out/kb_${wl}_equil.log" 2>&1 || exit 1
  rm -rf /tmp/r05_kb_$wl
  (cd /tmp && GS=auto KB_SPIN_S=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_kb_$wl -- python3 "$R/scripts/kernel_bench.py" $wl > "$out/kb_${wl}.log" 2>&1) || exit 1
  python3 scripts/state_roofline.py /tmp/r05_kb_$wl "$out/kb_${wl}.log" $wl >> "$out/r05_kernel_states.md" || exit 1
done
cp "$out/r05_kernel_states.md" profiles/r05_kernel_states.md
fi
if has 3; then
echo "== (3) PMC passes" | tee -a "$out/progress.log"
KB_WARM=20000 bash scripts/pmc_collect.sh "$out/pmc" c3 c4 || exit 1
cp "$out/pmc/pmc_traffic.json" profiles/pmc_traffic.json
cp "$out/pmc/pmc_hot_kernels.md" profiles/r05_pmc_hot_kernels.md
fi
if has 4; then
echo "== (4) default bench line" | tee -a "$out/progress.log"
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || exit 1
cp "$out/bench_default.json" profiles/r05_bench_default.json
fi
if has 5; then
echo "== (5) the live C3 loop alone under the profiler" | tee -a "$out/progress.log"
# (the metric kernel on nothing but the production loop: 3 s of warm-up and five 20 000-step windows with their rebuilds
# and refreshes; the driver-command table of part 1 averages the same kernel over every leg of bench.py, two replicas
# sharing the card included)
python3 scripts/steps_rate.py c3 20000 5 > "$out/live_plain.log" 2>&1 || exit 1
rm -rf /tmp/r05_live
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_live -- python3 "$R/scripts/steps_rate.py" c3 20000 5 > "$out/live_prof.log" 2>&1) || exit 1
{
  echo "# Round 5: the live C3 Langevin loop alone under rocprofv3 --kernel-trace --stats (\`scripts/steps_rate.py c3 20000 5\`)"
  echo
  echo "Un-profiled, same box: \`$(grep median "$out/live_plain.log")\`; under the profiler: \`$(grep median "$out/live_prof.log")\`."
  echo "Algorithmic bytes of a step at this state: 96 N + 16 P_nl = 12.96 MB (N = 32 768, P_nl = 613 k): fraction of 8 TB/s = 12.96e6 / (avg us x 1e-6) / 8e12."
  echo
  echo '```'
  python3 scripts/kernel_stats.py /tmp/r05_live k_
  echo '```'
} > profiles/r05_c3_live_loop_kernel_stats.md
fi
if has 6; then
echo "== (6) HIP launches against the step queue, fresh processes alternating" | tee -a "$out/progress.log"
bash scripts/queue_ab.sh ${QUEUE_AB_N:-10} c3 20000 > "$out/queue_ab_c3.log" 2>&1 || exit 1
bash scripts/queue_ab.sh 3 c4 4000 > "$out/queue_ab_c4.log" 2>&1 || exit 1
bash scripts/queue_ab.sh 3 c3_liquid 4000 > "$out/queue_ab_c3_liquid.log" 2>&1 || exit 1
CHX_AQL=0 python3 scripts/carry_probe.py c3 20 100 > "$out/carry_probe_hip.log" 2>&1 || exit 1
python3 scripts/carry_probe.py c3 20 100 > "$out/carry_probe_queue.log" 2>&1 || exit 1
fi
if has 7; then
echo "== (7) stand-alone AQL probe" | tee -a "$out/progress.log"
mkdir -p /tmp/aqlp
F="--offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16"
hipcc $F scripts/ub/aql_probe.hip -o /tmp/aqlp/aql_probe -lhsa-runtime64 > "$out/aql_build.log" 2>&1 || exit 1
hipcc $F --cuda-device-only --no-gpu-bundle-output scripts/ub/aql_probe.hip -o /tmp/aqlp/aql_probe.hsaco >> "$out/aql_build.log" 2>&1 || exit 1
(cd /tmp/aqlp && timeout -k 10 120 ./aql_probe 4000 aql_probe.hsaco > "$out/aql_launch.log" 2>&1) || exit 1
(cd /tmp/aqlp && timeout -k 10 120 ./aql_probe 3000 aql_probe.hsaco stale > "$out/aql_stale.log" 2>&1) || exit 1
(cd /tmp/aqlp && timeout -k 10 120 ./aql_probe 4000 aql_probe.hsaco wait > "$out/aql_wait.log" 2>&1) || exit 1
(cd /tmp/aqlp && timeout -k 10 120 ./aql_probe 200 aql_probe.hsaco xcc > "$out/aql_xcc.log" 2>&1) || exit 1
fi
mkdir -p "$out/profiles"
cp profiles/r05_* profiles/pmc_traffic.json "$out/profiles/"
echo "== done" | tee -a "$out/progress.log"
