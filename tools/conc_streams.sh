#!/bin/bash
F="--warmup 1 --no-cpu-baseline --no-alt --no-extra-legs --no-roofline"
python bench.py --batch 8 --steps 3 $F 2>/dev/null | tail -1 > gpurun_out/conc_b8.json
python bench.py --batch 4 --steps 3 $F 2>/dev/null | tail -1 > gpurun_out/conc_b4_alone.json
python bench.py --batch 4 --steps 6 $F 2>/dev/null | tail -1 > gpurun_out/conc_b4_a.json &
P=$!
python bench.py --batch 4 --steps 6 $F 2>/dev/null | tail -1 > gpurun_out/conc_b4_b.json
wait $P
for f in b8 b4_alone b4_a b4_b; do python -c "
import json;b=json.load(open('gpurun_out/conc_$f.json'));print('$f',b['value'],b['ms_per_step'])"; done
