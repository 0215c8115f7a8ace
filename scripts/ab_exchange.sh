# what does the replica-exchange path cost?  alternating fresh bench.py processes, plain loop vs a one-rank RCCL group
# with an exchange attempt every 1000 steps (GPU box only)
for i in 1 2 3 4 5; do
  timeout -k 10 120 python bench.py --steps 20000 --cpu-steps 0 --extra-c4 0 --extra-overlap 0 --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plain', d['value'])"
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2951$i CHX_BENCH_EXCHANGE=1 timeout -k 10 120 python bench.py --steps 20000 --cpu-steps 0 --extra-c4 0 --extra-overlap 0 --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('exch ', d['value'])"
done
