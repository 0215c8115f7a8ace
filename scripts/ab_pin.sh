# alternating fresh bench.py processes with and without pin_process_cpus (GPU box only): value + steady windows
for i in $(seq 1 ${1:-5}); do
  for p in 0 4; do
    CHX_PIN_CPUS=$p timeout -k 10 150 python bench.py --steps 20000 --cpu-steps 0 --extra-c4 0 --extra-overlap 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pin=$p', d['value'], [w['steps_per_s'] for w in d['steady_state']['windows']], d['roofline']['kernel_us'], d['config']['pinned_cpus'])" | tee -a gpurun_out/ab_pin.log
  done
done
