#!/bin/bash
# A/B of gz_segment_decode's waves per workgroup (1 = the build's default; ablate/libsc_fqcount_hip_gzw{2,4}.so = builds with
# -DSCFQ_GZ_WAVES_PER_WG=2|4) crossed with the number of symbol slots, on configs[3] (one 10 GB member).
mkdir -p gpurun_out/r03
A=$PWD/seq-collection_amd/ablate
V='[{"name":"w1 slots2","env":{}},{"name":"w1 slots3","env":{"SCFQ_GZ_DEVICE_SLOTS":"3"}},
{"name":"w4 slots2","env":{"SCFQ_LIB_OVERRIDE":"'$A'/libsc_fqcount_hip_gzw4.so"}},
{"name":"w4 slots3","env":{"SCFQ_LIB_OVERRIDE":"'$A'/libsc_fqcount_hip_gzw4.so","SCFQ_GZ_DEVICE_SLOTS":"3"}},
{"name":"w2 slots2","env":{"SCFQ_LIB_OVERRIDE":"'$A'/libsc_fqcount_hip_gzw2.so"}},
{"name":"w1 slots2 again","env":{}}]'
SCFQ_MEASURE_VARIANTS="$V" python scripts/measure_gz_device.py 10e9 /tmp > gpurun_out/r03/gz_wavewg.jsonl 2> gpurun_out/r03/gz_wavewg.err
python - <<'PY'
import json
for l in open("gpurun_out/r03/gz_wavewg.jsonl"):
    j = json.loads(l); p = j.get("phases_ms", {})
    print(j["inflate"], "wall", j["wall_s"], "first", j["first_call_wall_s"], {k: p[k] for k in p if "wall" in k or "decode" in k}, [s for s in j.get("summary", []) if "high water" in s])
PY
