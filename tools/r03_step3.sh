#!/bin/bash
tag=${1:-r03d}
{
  for pin in none spread pack; do
    echo "== gather probe, 4 passes, pin=$pin =="
    PROBE_PASSES=4 PROBE_PIN=$pin tools/ubench/gather_probe 6000000 "memcpy nt512+pf4"
  done
} > gpurun_out/${tag}_gather_pin.txt 2>&1
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
echo "bench rc=$?"; tail -c 600 gpurun_out/${tag}_bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device 0 --steps 5 --warmup 2 --rows 4000000 > gpurun_out/${tag}_bench_2ranks.json 2> gpurun_out/${tag}_bench_2ranks.err
echo "2ranks rc=$?"; tail -c 400 gpurun_out/${tag}_bench_2ranks.err
