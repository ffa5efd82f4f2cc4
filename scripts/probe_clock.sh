#!/bin/bash
# Samples the GPU's shader clock and power while the headline step runs (context for roofline.frac: the matrix peak in the
# guide assumes the 2.4 GHz boost clock).  Usage (GPU box, repo root): bash scripts/probe_clock.sh
python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-extras > /tmp/clock_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 60); do
  c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | sed 's/.*(\([0-9]*\)Mhz).*/\1/' | head -1)
  p=$(rocm-smi --showpower 2>/dev/null | grep -i "Power (W)" | sed 's/.*: //' | head -1)
  echo "t=$i sclk=${c}MHz power=${p}W"
  kill -0 $BP 2>/dev/null || break
  sleep 0.3
done
wait $BP
python3 -c "import json; d=json.load(open('/tmp/clock_bench.json')); print('value', d['value'], 'frac', d['roofline']['frac'])"
