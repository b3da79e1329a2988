#!/bin/bash
# Device gzip after the boundary-first loop: segment size x segments per batch on configs[3] (a decode kernel lasts as long as its slowest
# wave; smaller segments = more, shorter waves).
mkdir -p gpurun_out/r03
V='[{"name":"64 KiB x 4096 (default)","env":{}},
{"name":"32 KiB x 8192","env":{"SCFQ_GZ_DEVICE_SEGMENT_KB":"32","SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"8192"}},
{"name":"32 KiB x 5120","env":{"SCFQ_GZ_DEVICE_SEGMENT_KB":"32","SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"5120"}},
{"name":"48 KiB x 5120","env":{"SCFQ_GZ_DEVICE_SEGMENT_KB":"48","SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"5120"}},
{"name":"48 KiB x 10240","env":{"SCFQ_GZ_DEVICE_SEGMENT_KB":"48","SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"10240"}},
{"name":"96 KiB x 4096","env":{"SCFQ_GZ_DEVICE_SEGMENT_KB":"96"}},
{"name":"64 KiB x 4096 again","env":{}}]'
SCFQ_MEASURE_VARIANTS="$V" python scripts/measure_gz_device.py ${1:-10e9} /tmp > gpurun_out/r03/gz_variants_d.jsonl 2> gpurun_out/r03/gz_variants_d.err
python - <<'PY'
import json
for l in open("gpurun_out/r03/gz_variants_d.jsonl"):
    j = json.loads(l); p = j.get("phases_ms", {})
    print(j["inflate"], "| wall", j["wall_s"], "first", j["first_call_wall_s"], {k: p[k] for k in p if "wall" in k or "decode" in k or "search" in k}, [x.split("high water")[1][:9] for x in j.get("summary", []) if "high water" in x])
PY
