// kernels.cuh - the gfx950 kernels of libflowagg.
//
// Hot path (per batch):  wtile_kernel<KEYSETS, T8> -> deferred_kernel -> agg8_kernel | agg_kernel [-> cms_agg_kernel] [-> wagg_kernel]
//   wire bytes in HBM --global_load_lds_dwordx4 nt (async DMA)--> the wave's LDS tile (<= 64 records)
//   -> one record per lane parsed out of LDS (wire.cuh: parse_canon lean / FULL, parse_fast - three tiers in place)
//   -> key = (TimeReceived/granule, SrcAS, DstAS, EType)   [create.sh:92-110]
//   -> per-workgroup LDS hash table absorbs hot keys (mocker.go:61-62 has 9 groups)
//   -> everything else leaves the workgroup as a tuple (compact 8 bytes / wide 16, table.cuh) in the workgroup's PRIVATE
//      segment of the key's partition, whole store units at a time (LDS bins; positions from LDS counters - no global
//      atomics: MI355X retires only ~24 G memory-side atomics/s);
//   (tile_kernel = the 256-thread workgroup-tile form: decode path, direct sink, FA_TILE=wg)
//   agg8_kernel / agg_kernel: one 1024-thread workgroup per key partition streams the partition's segments back,
//      aggregates them in an LDS hash table (two packed 64-bit LDS atomics per tuple) and adds each group to the
//      device-wide table once - agg8_kernel with plain loads and stores: the partition owns the key's table region.
//   cms_agg_kernel: Count-Min tuples per (sketch, slice) folded in a dense LDS array, plain read-modify-write.
//   wagg_kernel: (SrcAddr,DstPort,Proto) tuples per region of the wide table, LDS dedup, plain loads and stores.
//   Records the in-place parsers are not sure about go to deferred_kernel (parse_generic, complete semantics); values
//   that do not fit a tuple take the direct device-wide-table path (64-bit atomics).
// Around the hot path:
//   framing.cuh   offsets == NULL: the framed chain cut into records on the device (guess, fixed-point proof, emit);
//   merge.cuh     window close in HBM: rows collected, sorted over the key bits that differ (rowplan.cuh), equal keys summed,
//                 emit order; hash partition of a row set for the multi-GPU close (row_partition_kernel);
//   maintenance.cuh  table scans, rebuilds, the wide log's reads / folds / drops (wdrop_kernel: a close zeroes sums in place).
//
// Roofline: HBM-bound integer/byte work; algorithmic bytes = wire bytes, read
// once (DESIGN.md "Roofline").  No MFMA anywhere - nothing here is a contraction.
#pragma once
#include "sinks.cuh"
#include "ingest.cuh"
#include "agg.cuh"
#include "wagg.cuh"
#include "maintenance.cuh"
#include "merge.cuh"
#include "framing.cuh"
