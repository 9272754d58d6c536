#!/bin/bash
# Records MIOpen's find-db for the benchmark's convolution shapes on an MI355X into gpurun_out/miopen_db/ (copy it to
# geodiffuser_amd/miopen_db/ and commit).  Runs the bench twice in fresh processes: the second run shows what a warm db saves.
#   gpurun --timeout 1500 -- 'tools/record_miopen_db.sh'
cd "$(dirname "$0")/.."
export GD_MIOPEN_DB=$PWD/gpurun_out/miopen_db
export GD_MIOPEN_DB_RECORD=1      # work IN that directory (the default is a scratch copy of the seed, removed at exit)
rm -rf "$GD_MIOPEN_DB"; mkdir -p "$GD_MIOPEN_DB"
cp -r geodiffuser_amd/miopen_db/. "$GD_MIOPEN_DB"/ 2>/dev/null
for run in cold warm; do
  python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline > gpurun_out/miopen_db_$run.json 2> gpurun_out/miopen_db_$run.err
  python3 - <<PY
import json
d = json.load(open("gpurun_out/miopen_db_$run.json"))
print("$run", "first warm-up edit:", d["config"]["first_warmup_edit_s"], "s; timed edit:", d["ms_per_step"], "ms")
PY
done
# the other benchmarked configurations (fp16; 768^2 = BASELINE configs[3]'s size; the SDXL-shaped 1024^2 harness = configs[4]'s shape)
for extra in "--dtype fp16" "--size 768" "--model sdxl --size 1024"; do
  tag=$(echo "$extra" | tr -d ' -')
  for run in cold warm; do
    python3 bench.py $extra --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/miopen_db_${tag}_$run.json 2> gpurun_out/miopen_db_${tag}_$run.err
    python3 - <<PY
import json
d = json.load(open("gpurun_out/miopen_db_${tag}_$run.json"))
print("$extra", "$run", "first warm-up edit:", d["config"]["first_warmup_edit_s"], "s; timed edit:", d["ms_per_step"], "ms")
PY
  done
done
du -sh "$GD_MIOPEN_DB"; find "$GD_MIOPEN_DB" -type f | head -20
