#!/bin/bash
# A/B of the symbol loops (SCFQ_INFLATE_LOOP=lanes | dense) on configs[3] (one member, default 10 GB) and on a 4 GB BGZF file;
# with ablate/libsc_fqcount_hip_noalias.so present (a -DSCFQ_GZ_CLTAB_ALIAS=0 build of the library — a make target of rounds 3 and 4; round 5 made the alias unconditional, in bgzf_inflate too —:
# 16 decode waves per CU instead of 20), that build as well.
mkdir -p gpurun_out/r03
A=$PWD/seq-collection_amd/ablate/libsc_fqcount_hip_noalias.so
V='[{"name":"lanes","env":{"SCFQ_INFLATE_LOOP":"lanes"}},{"name":"dense","env":{"SCFQ_INFLATE_LOOP":"dense"}}'
[ -f $A ] && V=$V',{"name":"lanes, 16 waves per CU","env":{"SCFQ_INFLATE_LOOP":"lanes","SCFQ_LIB_OVERRIDE":"'$A'"}},{"name":"dense, 16 waves per CU","env":{"SCFQ_INFLATE_LOOP":"dense","SCFQ_LIB_OVERRIDE":"'$A'"}}'
V=$V',{"name":"lanes again","env":{"SCFQ_INFLATE_LOOP":"lanes"}},{"name":"dense again","env":{"SCFQ_INFLATE_LOOP":"dense"}}]'
SCFQ_MEASURE_VARIANTS="$V" python scripts/measure_gz_device.py ${1:-10e9} /tmp > gpurun_out/r03/gz_dense_ab.jsonl 2> gpurun_out/r03/gz_dense_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r03/gz_dense_ab.jsonl"):
    j = json.loads(l); p = j.get("phases_ms", {})
    print(j["inflate"], "wall", j["wall_s"], "first", j["first_call_wall_s"], {k: p[k] for k in p if "wall" in k or "decode" in k})
PY
[ -n "$SKIP_BGZF" ] || for loop in lanes dense; do SCFQ_INFLATE_LOOP=$loop python scripts/measure_bgzf_device.py 4e9 > gpurun_out/r03/bgzf_$loop.jsonl 2> gpurun_out/r03/bgzf_$loop.err; echo $loop; head -1 gpurun_out/r03/bgzf_$loop.jsonl | cut -c1-300; done
