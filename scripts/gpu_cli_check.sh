set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, gzip
sys.path.insert(0, "seq-collection_amd/pyhost"); sys.path.insert(0, "tests")
import scfq, numpy as np
plan = scfq.synth_plan(0, 5, 60_000_000)
a, info = scfq.synth_host(0, 5, plan.records)
a.tofile("/tmp/c.fq")
open("/tmp/c.fq.gz", "wb").write(gzip.compress(a.tobytes(), 6))
from test_ingest_sources import bgzf_file
open("/tmp/cb.fq.gz", "wb").write(bgzf_file(a.tobytes()))
print(plan.records, info.gc_bases, info.n_bases, info.bases)
PY
S=seq-collection_amd/sc
$S fq-count -t -b /tmp/c.fq /tmp/c.fq.gz /tmp/cb.fq.gz
$S fq-count --devices=0,0 /tmp/c.fq
$S fq-count --devices=0,0,0 --qual-hist --struct-check /tmp/c.fq 2>&1 | head -3 | cut -c1-150
$S fq-count --jobs=3 -b /tmp/c.fq /tmp/c.fq.gz /tmp/cb.fq.gz
$S fq-count --stats /tmp/cb.fq.gz 2>&1 | cut -c1-300
$S fq-meta -t /tmp/c.fq.gz | cut -c1-200
$S fq-meta --whole-file /tmp/cb.fq.gz | cut -c1-200
cat /tmp/c.fq /tmp/c.fq > /tmp/cc.fq
$S fq-dedup /tmp/cc.fq 2>/tmp/dd.err | cmp - /tmp/c.fq && cat /tmp/dd.err
gzip -c /tmp/cc.fq > /tmp/cc.fq.gz; $S fq-dedup /tmp/cc.fq.gz 2>/dev/null | cmp - /tmp/c.fq && echo "dedup gz ok"
