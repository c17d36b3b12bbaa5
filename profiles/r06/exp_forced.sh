F="--gpus 1 --steps 8 --warmup 2 --traffic none --no-cpu-baseline --no-unpruned --no-ceiling --no-e04 --no-layouts --no-dropin"
P='import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(j["value"], j["ms_per_step"], j["stage_ms_per_step"]["total"], j["stage_ms_per_step"]["syncmers"], j["stage_ms_per_step"]["query"])'
for q in 8 12 16 24; do for b in 2 8; do
  echo "HW queues $q, batches $b: forced, plain"; GPU_MAX_HW_QUEUES=$q TAXOR_BENCH_FORCE_DIST=1 python3 bench.py $F --batches $b 2>/dev/null | python3 -c "$P"
  GPU_MAX_HW_QUEUES=$q python3 bench.py $F --batches $b 2>/dev/null | python3 -c "$P"
done; done
