mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_gz_device.py -q -m gpu 2>&1 | tail -3
python - <<'PY'
import os, sys, subprocess, time
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
plan = scfq.synth_plan(0, 20260101, int(2e9))
data, info = scfq.synth_host(0, 20260101, plan.records)
data.tofile("/tmp/g.fq")
t0 = time.time()
# 16 pieces compressed in parallel, then the members' deflate streams cannot simply be joined: use one member made by pigz-like trick is unavailable -> plain gzip
subprocess.check_call(["gzip", "-6", "-k", "-f", "/tmp/g.fq"])
print("gzip took %.0f s" % (time.time() - t0), flush=True)
PY
for cfg in "128 8192" "64 16384" "256 8192"; do set -- $cfg; echo "== SEGMENT_KB=$1 MAX_SEGMENTS=$2"; SCFQ_GZ_DEVICE_SEGMENT_KB=$1 SCFQ_GZ_DEVICE_MAX_SEGMENTS=$2 SCFQ_VERBOSE=1 python - <<'PY' 2>&1 | grep -v "^scfq pgz" | tail -14
import sys, time
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
scfq.count_file("tests/golden/dup.fq.gz")
for rep in range(2):
    t = time.time(); c = scfq.count_file("/tmp/g.fq.gz"); dt = time.time() - t
    sys.stderr.write("rep %d: %.1f ms  %.2f GB/s  reads %d\n" % (rep, dt * 1e3, c.input_bytes / dt / 1e9, c.reads))
PY
done
