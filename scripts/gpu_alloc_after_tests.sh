#!/bin/bash
# Does what ran before on the box change what an allocation costs?  (round 3: cold `sc fq-count x.fq.gz` 0.6 s on a fresh box, 1.5 s after pytest)
A=./scripts/ubench/alloc_cost
echo "== fresh box"; $A pattern 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 6 | tail -3
echo "== after a 200 GB process"; $A pattern 10 10 10 10 10 10 10 10 10 10 10 10 10 10 10 10 10 10 10 10 | tail -1
$A pattern 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 6 | tail -3
echo "== after 40 small processes"; for i in $(seq 40); do $A pattern 0.5 0.2 0.1 > /dev/null; done
$A pattern 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 6 | tail -3
echo "== after the 4.6 GB gzip test"; python -m pytest tests/test_gpu_gz_rooms.py -q -m gpu -k "beyond_4_gib or 1024" 2>&1 | tail -1
$A pattern 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 6 | tail -3
sleep 20; echo "== 20 s later"
$A pattern 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 6 | tail -3
echo "== after the gz device tests"; python -m pytest tests/test_gpu_gz_device.py -q -m gpu 2>&1 | tail -1
$A pattern 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 2 6 | tail -3
