# A/B runs of the device gzip path on one file: [inflated bytes] ; variants in $SCFQ_MEASURE_VARIANTS (JSON) or the default set
N=${1:-10e9}
OUT=${2:-gpurun_out/gzv}
mkdir -p $OUT
export SCFQ_MEASURE_VARIANTS=${SCFQ_MEASURE_VARIANTS:-'[
 {"name":"plain streams","env":{"SCFQ_GZ_DEVICE_RESERVE_CUS":"0"}},
 {"name":"decode low priority","env":{"SCFQ_GZ_DEVICE_RESERVE_CUS":"0","SCFQ_GZ_DEVICE_DECODE_LOW_PRIORITY":"1"}},
 {"name":"16 CUs reserved","env":{"SCFQ_GZ_DEVICE_RESERVE_CUS":"16"}},
 {"name":"plain streams, 8 hw queues","env":{"SCFQ_GZ_DEVICE_RESERVE_CUS":"0","GPU_MAX_HW_QUEUES":"8"}},
 {"name":"low priority, 8 hw queues","env":{"SCFQ_GZ_DEVICE_RESERVE_CUS":"0","SCFQ_GZ_DEVICE_DECODE_LOW_PRIORITY":"1","GPU_MAX_HW_QUEUES":"8"}},
 {"name":"one batch","env":{"SCFQ_GZ_DEVICE_RESERVE_CUS":"0","SCFQ_GZ_DEVICE_BATCH_SEGMENTS":"65536"}}
]'}
SCFQ_MEASURE_LOG=$OUT/log.txt timeout -k 10 900 python scripts/measure_gz_device.py $N > $OUT/variants.jsonl 2> $OUT/err.txt
python - <<PY
import json
for l in open("$OUT/variants.jsonl"):
    j = json.loads(l)
    print("%-34s %7.1f ms %6.2f GB/s  %s" % (j["inflate"], j["wall_s"] * 1e3, j["inflated_GBps"], {k: round(v) for k, v in j.get("phases_ms", {}).items()}))
PY
