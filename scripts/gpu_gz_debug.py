#!/usr/bin/env python3
"""small files through the device gzip path with SCFQ_VERBOSE: where does a decision to leave the device come from"""
import os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_ingest_sources import fastq_bytes
data = fastq_bytes(6_000_000, seed=31)
co = zlib.compressobj(6, zlib.DEFLATED, 31)
open("/tmp/dbg.fq.gz", "wb").write(co.compress(data) + co.flush())
sc = os.path.join(ROOT, "seq-collection_amd", "sc")
for env in ({"SCFQ_GZ_DEVICE_MIN_MB": "0", "SCFQ_GZ_DEVICE_SEGMENT_KB": "32"}, {"SCFQ_GZ_DEVICE_MIN_MB": "0", "SCFQ_GZ_DEVICE_SEGMENT_KB": "32", "SCFQ_GZ_DEVICE_BATCH_SEGMENTS": "8", "SCFQ_GZ_DEVICE_CHAIN_GROUP": "3"},
            {"SCFQ_GZ_DEVICE_MIN_MB": "0", "SCFQ_VMM": "0"}):
    r = subprocess.run([sc, "fq-count", "/tmp/dbg.fq.gz"], capture_output=True, text=True, env=dict(os.environ, SCFQ_VERBOSE="1", **env))
    print(env, r.returncode, r.stdout.strip()); print(r.stderr)
