#!/bin/bash
# Device gzip after the boundary-first loop (20 decode waves per CU = 5120 wave slots): batch size, first batch, symbol slots on configs[3].
mkdir -p gpurun_out/r03
V='[{"name":"default (4096 segments per batch)","env":{}},
{"name":"5120 per batch","env":{"SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"5120"}},
{"name":"5120, first batch a quarter","env":{"SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"5120","SCFQ_GZ_DEVICE_FIRST_BATCH_DIV":"4"}},
{"name":"2560 per batch","env":{"SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"2560"}},
{"name":"5120, three symbol slots","env":{"SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"5120","SCFQ_GZ_DEVICE_SLOTS":"3"}},
{"name":"default again","env":{}}]'
SCFQ_MEASURE_VARIANTS="$V" python scripts/measure_gz_device.py ${1:-10e9} /tmp > gpurun_out/r03/gz_variants_b.jsonl 2> gpurun_out/r03/gz_variants_b.err
python - <<'PY'
import json
for l in open("gpurun_out/r03/gz_variants_b.jsonl"):
    j = json.loads(l); p = j.get("phases_ms", {})
    print(j["inflate"], "| wall", j["wall_s"], "first", j["first_call_wall_s"], {k: p[k] for k in p if "wall" in k or "decode" in k}, [x.split("high water")[1][:9] for x in j.get("summary", []) if "high water" in x])
PY
