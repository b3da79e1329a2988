# fq_count.nim — drop-in replacement of the reference's src/fq_count.nim for hosts that have a Nim
# toolchain (the build image has none: this file is shipped as the reference-side binding, untested here;
# the tested host above the same C ABI is seq-collection_amd/cli/sc_main.cpp).
#
# Keeps the reference's exported surface exactly:
#   const fq_count_header*                                   (src/fq_count.nim:7-11)
#   proc fq_count*(fastq: string, basename: bool, absolute: bool)   (src/fq_count.nim:14)
# and moves lines :30-51 (open stream, line loop, counters, number formatting) behind libsc_fqcount_hip.so.
# Compile inside the reference tree:  nim c -d:release --passL:"-L<dir> -lsc_fqcount_hip" sc.nim
import utils/helpers
import strutils

const fq_count_header* = ["reads", "gc_content", "gc_bases", "n_bases", "bases"].join("\t")

type
  ScfqCounts {.bycopy.} = object
    struct_size, abi_version: uint64
    reads, gc_bases, n_bases, bases: uint64
    lines, newlines, input_bytes: uint64
    bad_at, bad_plus: uint64
    qual_hist: array[256, uint64]
  ScfqOpts {.bycopy.} = object
    struct_size: uint64
    n_devices: int32
    device_ids: ptr int32
    flags, reserved: uint32
    chunk_bytes: uint64
    wait_stream: pointer        # hipStream_t of the caller, read when SCFQ_WAIT_STREAM is set in flags (unused here)

const
  SCFQ_OK = 0
  SCFQ_EOPEN = -1

proc scfq_count_file(path: cstring, opts: ptr ScfqOpts, outp: ptr ScfqCounts): cint
  {.importc, cdecl, dynlib: "libsc_fqcount_hip.so".}
proc scfq_format_tsv(c: ptr ScfqCounts, buf: cstring, cap: uint64): cint
  {.importc, cdecl, dynlib: "libsc_fqcount_hip.so".}
proc scfq_strerror(rc: cint): cstring {.importc, cdecl, dynlib: "libsc_fqcount_hip.so".}
proc scfq_last_error_detail(): cstring {.importc, cdecl, dynlib: "libsc_fqcount_hip.so".}

proc fq_count*(fastq: string, basename: bool, absolute: bool) =
  discard fastq[^3 .. ^1]                      # keeps the reference's IndexError on paths shorter than 3 chars (:31)
  var c: ScfqCounts
  c.struct_size = uint64(sizeof(ScfqCounts))
  var o: ScfqOpts
  o.struct_size = uint64(sizeof(ScfqOpts))
  let rc = scfq_count_file(fastq.cstring, addr o, addr c)
  if rc == SCFQ_EOPEN:
    if fastq[^3 .. ^1] == ".gz":
      raise newException(OSError, "Unable to open file: " & fastq)   # zip/gzipfiles raises -> sc.nim:299-305, exit 1
    quit_error("Unable to open file: " & fastq, 2)                    # src/fq_count.nim:35-36
  if rc != SCFQ_OK:
    quit_error($scfq_strerror(rc) & ": " & $scfq_last_error_detail(), 1)
  var row = newString(256)
  let n = scfq_format_tsv(addr c, row.cstring, 256)
  row.setLen(n)
  echo output_w_fnames(row, fastq, basename, absolute)                # src/fq_count.nim:53
