# quick look at the device gzip path's phases: [inflated bytes] (default 5e8)
N=${1:-5e8}
python - <<PY
import sys, subprocess
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
plan = scfq.synth_plan(0, 20260101, int($N))
data, info = scfq.synth_host(0, 20260101, plan.records)
data.tofile("/tmp/g.fq")
subprocess.check_call(["gzip", "-6", "-k", "-f", "/tmp/g.fq"])
PY
for cfg in ${CFGS:-"128 8192"}; do set -- $cfg; echo "== SEGMENT_KB=$1 MAX_SEGMENTS=$2"; SCFQ_GZ_DEVICE_SEGMENT_KB=$1 SCFQ_GZ_DEVICE_MAX_SEGMENTS=$2 SCFQ_VERBOSE=1 python - <<'PY' 2>&1 | grep -v "^scfq pgz" | tail -${TAILN:-15}
import sys, time
sys.path.insert(0, "seq-collection_amd/pyhost")
import scfq
scfq.count_file("tests/golden/dup.fq.gz")
for rep in range(2):
    t = time.time(); c = scfq.count_file("/tmp/g.fq.gz"); dt = time.time() - t
    sys.stderr.write("rep %d: %.1f ms  %.2f GB/s  reads %d\n" % (rep, dt * 1e3, c.input_bytes / dt / 1e9, c.reads))
PY
done
