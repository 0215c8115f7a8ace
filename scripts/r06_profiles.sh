#!/bin/bash
# Round 6's evidence (GPU box only):  PARTS="1 2 3" bash scripts/r06_profiles.sh   /   PARTS="4 5 6 7" ...
#  (1) the driver's own bench command, plain and under rocprofv3 --kernel-trace --stats -> profiles/r06_driver_cmd_*
#  (2) one --stats pass of scripts/kernel_bench.py per quoted system state -> profiles/r06_kernel_states.md
#  (3) the PMC passes over the hot kernels (separate --pmc runs) -> profiles/pmc_traffic.json, r06_pmc_hot_kernels.md
#  (4) the default bench line -> profiles/r06_bench_default.json
#  (5) the live C3 loop alone under the profiler -> profiles/r06_c3_live_loop_kernel_stats.md
#  (6) eight C3 replicas stepped side by side (the batched launch), plain and under the profiler -> profiles/r06_c5_batched_kernel_stats.md
#  (7) the whole GPU suite under the step queue's debug mode (CHX_QUEUE_DEBUG=250) -> profiles/r06_queue_debug.md
set -o pipefail
PARTS=${PARTS:-"1 2 3 4 5 6 7"}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r06_prof
mkdir -p "$out/profiles"
echo "== parts: $PARTS" | tee "$out/progress.log"
if has 1; then
echo "== (1) driver command" | tee -a "$out/progress.log"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_plain.log" 2> "$out/bench_plain.err" || exit 1
rm -rf /tmp/r06_drv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_drv -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$out/bench_prof.log" 2> "$out/bench_prof.err") || exit 1
python3 scripts/stats_to_md.py /tmp/r06_drv r06_driver_cmd "$out/bench_prof.log" "$out/bench_plain.log" \
  "Round 6: rocprofv3 --kernel-trace --stats of the driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5)" || exit 1
cp profiles/r06_driver_cmd_bench_c3_kernel_stats.* "$out/profiles/"
fi
if has 2; then
echo "== (2) per-state kernel passes" | tee -a "$out/progress.log"
export KB_SNAPSHOT_DIR=/tmp/kb_snap_r06
rm -rf "$KB_SNAPSHOT_DIR"
{
  echo "# Round 6: one system state per row -- rocprofv3 --kernel-trace --stats of \`scripts/kernel_bench.py <workload>\`"
  echo
  python3 -c "import re; print(re.search(r'^\"\"\"(.*?)\"\"\"', open('scripts/state_roofline.py').read(), re.S).group(1))"
  echo
  echo "| state | kernel | lanes / particle | P_nl | algorithmic bytes | launches | rocprofv3 avg (us) | GB/s | fraction of 8 TB/s | HIP-event time in the same pass (us) |"
  echo "|---|---|---|---|---|---|---|---|---|---|"
} > "$out/r06_kernel_states.md"
for spec in "c3 20000" "c4 20000" "c3_liquid 4000" "c4_liquid 1000"; do
  set -- $spec
  wl=$1; export KB_WARM=$2
  echo "   $wl (equilibrated $KB_WARM steps)" | tee -a "$out/progress.log"
  GS=auto KB_REPS=2 python3 scripts/kernel_bench.py $wl > "$out/kb_${wl}_equil.log" 2>&1 || exit 1
  rm -rf /tmp/r06_kb_$wl
  (cd /tmp && GS=auto KB_SPIN_S=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_kb_$wl -- python3 "$R/scripts/kernel_bench.py" $wl > "$out/kb_${wl}.log" 2>&1) || exit 1
  python3 scripts/state_roofline.py /tmp/r06_kb_$wl "$out/kb_${wl}.log" $wl >> "$out/r06_kernel_states.md" || exit 1
done
cp "$out/r06_kernel_states.md" profiles/r06_kernel_states.md
cp profiles/r06_kernel_states.md "$out/profiles/"
fi
if has 3; then
echo "== (3) PMC passes" | tee -a "$out/progress.log"
KB_WARM=20000 bash scripts/pmc_collect.sh "$out/pmc" c3 c4 || exit 1
cp "$out/pmc/pmc_traffic.json" profiles/pmc_traffic.json
sed 's/^# PMC passes/# Round 6: PMC passes/' "$out/pmc/pmc_hot_kernels.md" > profiles/r06_pmc_hot_kernels.md
cp profiles/pmc_traffic.json profiles/r06_pmc_hot_kernels.md "$out/profiles/"
fi
if has 4; then
echo "== (4) default bench line" | tee -a "$out/progress.log"
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || exit 1
cp "$out/bench_default.json" profiles/r06_bench_default.json
cp profiles/r06_bench_default.json "$out/profiles/"
fi
if has 5; then
echo "== (5) the live C3 loop alone under the profiler" | tee -a "$out/progress.log"
python3 scripts/steps_rate.py c3 20000 5 > "$out/live_plain.log" 2>&1 || exit 1
rm -rf /tmp/r06_live
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_live -- python3 "$R/scripts/steps_rate.py" c3 20000 5 > "$out/live_prof.log" 2>&1) || exit 1
{
  echo "# Round 6: the live C3 Langevin loop alone under rocprofv3 --kernel-trace --stats (\`scripts/steps_rate.py c3 20000 5\`)"
  echo
  echo "Un-profiled, same box: \`$(grep median "$out/live_plain.log")\`; under the profiler: \`$(grep median "$out/live_prof.log")\`."
  echo "Algorithmic bytes of a step at this state: 96 N + 16 P_nl = 12.96 MB (N = 32 768, P_nl = 613 k): fraction of 8 TB/s = 12.96e6 / (avg us x 1e-6) / 8e12."
  echo
  echo '```'
  python3 scripts/kernel_stats.py /tmp/r06_live k_
  echo '```'
} > profiles/r06_c3_live_loop_kernel_stats.md
cp profiles/r06_c3_live_loop_kernel_stats.md "$out/profiles/"
fi
if has 6; then
echo "== (6) eight replicas side by side" | tee -a "$out/progress.log"
python3 scripts/batch_rate.py 8 4000 c3 > "$out/batch_plain.log" 2>&1 || exit 1
rm -rf /tmp/r06_batch
(cd /tmp && BATCH_MODES=batched rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_batch -- python3 "$R/scripts/batch_rate.py" 8 4000 c3 > "$out/batch_prof.log" 2>&1) || exit 1
{
  echo "# Round 6: eight N = 32 768 replicas (300 K x 1.05^k) stepped side by side on one MI355X -- \`scripts/batch_rate.py 8 4000 c3\`"
  echo
  echo "One launch per MD step for all replicas (\`k_lj_step_batch\`: blockIdx.y = replica; \`integrators.run_batch\`), each replica with"
  echo "its own box, temperature, key, lists, control block and rebuild schedule; states bit-identical to the serial loop's"
  echo "(\`tests/test_gpu_batch.py\`).  Un-profiled, aggregate replica-steps/s over three windows of 4000 steps:"
  echo
  echo '```'
  grep "R=8" "$out/batch_plain.log"
  echo '```'
  echo
  echo "Under rocprofv3 --kernel-trace --stats (batched loop only; the profiler's interceptor makes the loop dispatch-bound):"
  echo
  echo '```'
  grep "R=8" "$out/batch_prof.log"
  python3 scripts/kernel_stats.py /tmp/r06_batch k_
  echo '```'
  echo
  echo "Algorithmic bytes of one batched launch (SURVEY 8d per replica: 96 N + 16 P_nl): 8 x 12.95 MB = 103.6 MB; fraction of 8 TB/s ="
  echo "103.6e6 / (avg us x 1e-6) / 8e12 (the average above includes the launches in which some replicas sit out behind a halt)."
} > profiles/r06_c5_batched_kernel_stats.md
cp profiles/r06_c5_batched_kernel_stats.md "$out/profiles/"
fi
if has 7; then
echo "== (7) the GPU suite under the step queue's debug mode" | tee -a "$out/progress.log"
python3 -m pytest tests -q -m gpu > "$out/suite_plain.log" 2>&1; echo "rc=$?" >> "$out/suite_plain.log"
CHX_QUEUE_DEBUG=250 python3 -m pytest tests -q -m gpu -rA > "$out/suite_queue_debug.log" 2>&1; echo "rc=$?" >> "$out/suite_queue_debug.log"
{
  echo "# Round 6: the GPU suite under the step queue's debug mode (\`CHX_QUEUE_DEBUG=250 python -m pytest tests -m gpu\`)"
  echo
  echo "Debug mode (include/chx.h: chx_queue_debug; chiron_amd/ops.py: fresh): a 250 us idle packet in front of the first packet"
  echo "behind every gate, of every launch that is not a fused step / Monte Carlo trial (unpack, the list build chain, pack,"
  echo "prologues) and of a run's final launch -- the launches whose outputs are read from outside the queue -- and NaN / -1"
  echo "in every such output before the packet that writes it is dispatched.  A consumer that skips chx_steps_join then reads"
  echo "poison every time (tests/test_gpu_step_queue.py::test_a_missing_join_reads_poison_every_time: 10 of 10, both the"
  echo "Verlet-list and the all-pairs path, with a deliberately skipped join)."
  echo
  echo "Plain run of the same suite on the same box: \`$(tail -2 "$out/suite_plain.log" | head -1)\`"
  echo
  echo "Under the debug mode: \`$(tail -2 "$out/suite_queue_debug.log" | head -1)\`"
  echo
  echo '```'
  grep -E "^(PASSED|FAILED|ERROR)" "$out/suite_queue_debug.log" | sed 's/^PASSED //' | head -400
  echo '```'
} > profiles/r06_queue_debug.md
cp profiles/r06_queue_debug.md "$out/profiles/"
fi
echo "== done" | tee -a "$out/progress.log"
