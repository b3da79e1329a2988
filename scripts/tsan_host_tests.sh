#!/bin/bash
# CPU-only ThreadSanitizer pass over the threaded host readers (serial reader's decoder/consumer hand-off, the parallel
# single-member reader's batches and phases, BGZF block threads) and, from round 3, the communicator's worker thread over the TCP transport (tickets, deadlines, the refcounted
# stdout redirect).  Clean as of round 3.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-gpu-sanitize -shared \
  -o /tmp/libsc_fqcount_hip_tsan.so $R/seq-collection_amd/csrc/scfq_api.hip $R/seq-collection_amd/csrc/scfq_host.cpp \
  $R/seq-collection_amd/csrc/scfq_synth.hip $R/seq-collection_amd/csrc/scfq_dedup.hip $R/seq-collection_amd/csrc/scfq_meta.cpp $R/seq-collection_amd/csrc/scfq_comm.cpp -lz -lpthread -ldl
TSAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
cd $R
LD_PRELOAD=$TSAN TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0" SCFQ_LIB_OVERRIDE=/tmp/libsc_fqcount_hip_tsan.so \
  python -m pytest tests/test_inflate_host.py tests/test_ingest_sources.py tests/test_comm_host.py -q -m "not gpu" -k "parallel or hands_many or member_framing or bgzf or tcp_transport or late_answer or stuck_exchange or missing_rank" 2>&1 | tee /tmp/tsan.log | tail -3
if grep -q "WARNING: ThreadSanitizer" /tmp/tsan.log; then echo "TSAN REPORTS"; exit 1; fi
echo "tsan clean"
