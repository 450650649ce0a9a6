// kernels.cuh - the gfx950 kernels of libflowagg.
//
// Hot path (per batch):  probe_kernel -> wtile_kernel<KEYSETS> -> deferred_kernel -> agg_kernel
//   wire bytes in HBM --global_load_lds_dwordx4 nt (async DMA)--> the wave's LDS tile (<= 64 records)
//   -> one record per lane parsed out of LDS (wire.cuh, parse_canon)
//   -> key = (TimeReceived/granule, SrcAS, DstAS, EType)   [create.sh:92-110]
//   -> per-workgroup LDS hash table absorbs hot keys (mocker.go:61-62 has 9 groups)
//   -> everything else leaves the workgroup as a 16-byte tuple in the workgroup's PRIVATE segment of the
//      key's hash partition, 8 tuples = one aligned 128-byte line at a time (LDS bins; positions from LDS
//      counters - no global atomics: MI355X retires only ~23.7 G global-atomic line requests/s,
//      tools/sink_bench.hip);
//   (tile_kernel = the 256-thread workgroup-tile form: decode path, direct sink, FA_TILE=wg)
//   agg_kernel: one 1024-thread workgroup per key partition streams the partition's segments back,
//      aggregates them in an LDS hash table (two packed 64-bit LDS atomics per tuple) and adds each
//      group to the device-wide table once.
//   Records parse_canon is not sure about go to deferred_kernel (parse_fast, any field order, then
//   parse_generic, complete semantics); values that do not fit a tuple take
//   the direct device-wide-table path (64-bit atomics).
//
// Roofline: HBM-bound integer/byte work; algorithmic bytes = wire bytes, read
// once (DESIGN.md "Roofline").  No MFMA anywhere - nothing here is a contraction.
#pragma once
#include "sinks.cuh"
#include "ingest.cuh"
#include "agg.cuh"
#include "wagg.cuh"
#include "maintenance.cuh"
