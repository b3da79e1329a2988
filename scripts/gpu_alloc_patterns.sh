#!/bin/bash
# What a process pays for device memory when it follows another process that used the same memory (round 3: the cold
# `sc fq-count x.fq.gz`).  Each line of a block is one process; processes run back to back.
A=./scripts/ubench/alloc_cost
echo "== r2 footprint of a 10 GB .gz (3 x 1.3 + 2 x 28 + 11 GB), three processes back to back"
for i in 1 2 3; do $A pattern 1.3 1.3 1.3 28 28 11; done
sleep 5
echo "== the same bytes in 4 GB pieces"
for i in 1 2; do $A pattern 4 4 4 4 4 4 4 4 4 4 4 4 4 4 4 4 4 4; done
sleep 5
echo "== 12 GB (lean footprint), three processes back to back"
for i in 1 2 3; do $A pattern 0.6 0.6 0.6 4.5 4.5 2.2; done
echo "== 25 GB, three processes back to back"
for i in 1 2 3; do $A pattern 1.3 1.3 1.3 10 10 5; done
echo "== 25 GB, 3 s apart"
for i in 1 2 3; do $A pattern 1.3 1.3 1.3 10 10 5; sleep 3; done
