#!/bin/bash
# A/B of the symbol loops (SCFQ_INFLATE_LOOP=lanes | dense) on configs[3] (one 10 GB member) and on a 4 GB BGZF file.
mkdir -p gpurun_out/r03
V='[{"name":"lanes","env":{"SCFQ_INFLATE_LOOP":"lanes"}},{"name":"dense","env":{"SCFQ_INFLATE_LOOP":"dense"}},{"name":"lanes again","env":{"SCFQ_INFLATE_LOOP":"lanes"}},{"name":"dense again","env":{"SCFQ_INFLATE_LOOP":"dense"}}]'
SCFQ_MEASURE_VARIANTS="$V" python scripts/measure_gz_device.py ${1:-10e9} /tmp > gpurun_out/r03/gz_dense_ab.jsonl 2> gpurun_out/r03/gz_dense_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r03/gz_dense_ab.jsonl"):
    j = json.loads(l); p = j.get("phases_ms", {})
    print(j["inflate"], "wall", j["wall_s"], "first", j["first_call_wall_s"], {k: p[k] for k in p if "wall" in k or "decode" in k})
PY
for loop in lanes dense; do SCFQ_INFLATE_LOOP=$loop python scripts/measure_bgzf_device.py 4e9 > gpurun_out/r03/bgzf_$loop.jsonl 2> gpurun_out/r03/bgzf_$loop.err; echo $loop; cut -c1-600 gpurun_out/r03/bgzf_$loop.jsonl; done
