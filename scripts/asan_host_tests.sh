#!/bin/bash
# CPU-only sanitizer pass (GPU ASan is not available on the pool): the oracle and the HOST side of the product library
# (partial monoid, formatting, synthetic generator, plain / gzip / BGZF byte sources incl. the library's own DEFLATE decoder on
# valid, truncated and bit-flipped streams, fq-meta parsing) under ASan + UBSan.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/oracle asan >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared \
  -o /tmp/libsc_fqcount_hip_asan.so $R/seq-collection_amd/csrc/scfq_api.hip $R/seq-collection_amd/csrc/scfq_host.cpp \
  $R/seq-collection_amd/csrc/scfq_synth.hip $R/seq-collection_amd/csrc/scfq_dedup.hip $R/seq-collection_amd/csrc/scfq_meta.cpp $R/seq-collection_amd/csrc/scfq_comm.cpp -lz -lpthread -ldl
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 SCFQ_LIB_OVERRIDE=/tmp/libsc_fqcount_hip_asan.so \
  python -m pytest tests/test_ingest_sources.py tests/test_inflate_host.py tests/test_meta_host.py tests/test_dedup_oracle.py tests/test_property_host.py tests/test_abi.py tests/test_gz_shard_rules_host.py tests/test_comm_host.py -q -m "not gpu"
