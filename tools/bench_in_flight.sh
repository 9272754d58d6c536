#!/bin/bash
# Throughput of ONE MI355X with P independent edits in flight (P ranks on the device, bench.py --edits-in-flight P).  Usage (GPU box): tools/bench_in_flight.sh "1 2 4 8"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "| edits in flight | s per edit and rank | edits/min (whole GPU) | vs one edit at a time | device memory reserved per rank (GiB) |" > gpurun_out/r04_in_flight.md
echo "|---|---|---|---|---|" >> gpurun_out/r04_in_flight.md
for P in ${1:-1 2 3 4 6 8}; do
  timeout 900 python3 bench.py --edits-in-flight $P --steps 10 --warmup 4 --no-cpu-baseline 2>gpurun_out/in_flight_$P.err | tail -1 > gpurun_out/r04_bench_in_flight_$P.json
  python3 - <<PY >> gpurun_out/r04_in_flight.md
import json
l = json.loads(open("gpurun_out/r04_bench_in_flight_$P.json").read())
base = json.loads(open("gpurun_out/r04_bench_in_flight_1.json").read())["value"] if $P > 1 else l["value"]
print(f"| $P | {l['ms_per_step'] / 1e3:.3f} | {60 * l['value']:.1f} | {l['value'] / base:.2f} x | {l['config']['device_allocator_in_timed_region']['reserved_GiB']} |")
PY
done
cat gpurun_out/r04_in_flight.md
