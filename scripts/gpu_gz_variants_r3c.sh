#!/bin/bash
# Device gzip after the boundary-first loop: the pipeline's knobs again (first batch, ring piece, priorities) on configs[3].
mkdir -p gpurun_out/r03
V='[{"name":"default","env":{}},
{"name":"first batch a quarter","env":{"SCFQ_GZ_DEVICE_FIRST_BATCH_DIV":"4"}},
{"name":"first batch half","env":{"SCFQ_GZ_DEVICE_FIRST_BATCH_DIV":"2"}},
{"name":"ring 64 MiB","env":{"SCFQ_GZ_DEVICE_RING_MB":"64"}},
{"name":"ring 32 MiB","env":{"SCFQ_GZ_DEVICE_RING_MB":"32"}},
{"name":"decode streams at low priority","env":{"SCFQ_GZ_DEVICE_DECODE_LOW_PRIORITY":"1"}},
{"name":"3584 per batch","env":{"SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"3584"}},
{"name":"default again","env":{}}]'
SCFQ_MEASURE_VARIANTS="$V" python scripts/measure_gz_device.py ${1:-10e9} /tmp > gpurun_out/r03/gz_variants_c.jsonl 2> gpurun_out/r03/gz_variants_c.err
python - <<'PY'
import json
for l in open("gpurun_out/r03/gz_variants_c.jsonl"):
    j = json.loads(l); p = j.get("phases_ms", {})
    print(j["inflate"], "| wall", j["wall_s"], "first", j["first_call_wall_s"], {k: p[k] for k in p if "wall" in k or "decode" in k or "copy" in k})
PY
